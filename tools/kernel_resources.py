#!/usr/bin/env python3
"""kernel_resources.py -- registers, spills, occupancy and static LDS of every kernel of the in-tree library, from
`hipcc -Rpass-analysis=kernel-resource-usage` over csrc/rbf_api.hip (cross-compiles: no GPU needed).
Usage: python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "new_bloom_filter_repo_amd", "csrc", "rbf_api.hip")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"),
                      "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src], capture_output=True, text=True).stderr
rows, cur = {}, None
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["/usr/bin/c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"^void ", "", cur).split("(")[0].replace("rbf::", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
print("# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage over rbf_api.hip (the in-tree library), %s" % (sys.argv[1] if len(sys.argv) > 1 else "HEAD"))
print("%-84s %6s %6s %8s %7s %10s %12s" % ("kernel", "VGPRs", "SGPRs", "scratch", "spills", "waves/SIMD", "LDS (static)"))
for k, v in rows.items():
    print("%-84s %6s %6s %8s %7s %10s %12s" % (k[:84], v.get("VGPRs"), v.get("TotalSGPRs"), v.get("ScratchSize [bytes/lane]"), v.get("VGPRs Spill"), v.get("Occupancy [waves/SIMD]"),
                                               v.get("LDS Size [bytes/block]")))
