#!/usr/bin/env python3
"""Opcode-weighted VALU issue model of the query and insert kernels' hot loops (bench.py `roofline.issue`).

  1. compiles the kernels to gfx950 assembly (hipcc -S, no GPU needed),
  2. takes the straight-line block of the common geometry (floor(k*) = 2: the block with 24 LDS reads for the query,
     the drain block for the insert) and classifies every VALU instruction by opcode AND operand kind (VGPR / inline
     constant, literal, SGPR source), because on gfx950 the issue cost depends on both (profiles/r02_opbench.txt),
  3. prices each class with the measured cycles per wave-instruction per SIMD at 4 waves per SIMD,
  4. writes profiles/r02_issue_model.json: the table, cycles per (pixel, frame), and the launch's issue-bound time.

usage: python tools/make_issue_model.py  (run from the repo root; writes profiles/r02_issue_model.json and prints the table)"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "new_bloom_filter_repo_amd", "csrc")
OPBENCH = os.path.join(REPO, "profiles", "r02_opbench.txt")
CLOCK_GHZ = 2.25          # shader clock measured by opbench's s_memtime / wall_clock64 calibration
SIMDS = 1024


def opbench_costs():
    """name -> cycles per wave-instruction per SIMD (column ch=8 w=4)."""
    out = {}
    for ln in open(OPBENCH):
        m = re.match(r"^(\S.*?)\s*\|.*ch=8 w=4\s+([0-9.]+)", ln)
        if m:
            out[m.group(1).strip()] = float(m.group(2))
    return out


def classify(op, operands, cost):
    """(class label, cycles) of one VALU instruction."""
    has_s = bool(re.search(r"(?<![a-z0-9_])(s\d+|s\[\d+:\d+\]|vcc|exec)", operands.split(",", 1)[1] if "," in operands else ""))
    has_lit = bool(re.search(r"0x[0-9a-f]+", operands))
    base = re.sub(r"_e32$|_e64$", "", op)
    simple = {"v_add_u32": "v_add_u32 v,v,v", "v_sub_u32": "v_sub_u32", "v_subrev_u32": "v_sub_u32", "v_and_b32": "v_and_b32 v,v,v",
              "v_or_b32": "v_xor_b32 v,v,v", "v_xor_b32": "v_xor_b32 v,v,v"}
    if base in simple:
        if has_s:
            return base + " (SGPR source)", cost["v_subrev_u32 v,s,v"]
        if has_lit:
            return base + " (literal)", cost["v_add_u32 v,literal,v"]
        return base + " (VGPR / inline constant)", cost[simple[base]]
    table = {"v_min_u32": "v_min_u32 v,v,v", "v_lshlrev_b32": "v_lshlrev_b32 v,v,v", "v_lshrrev_b32": "v_lshrrev_b32", "v_lshl_add_u32": "v_lshl_add_u32 v,v,2,v",
             "v_lshl_or_b32": "v_lshl_add_u32 v,v,2,v", "v_or3_b32": "v_add3_u32 v,v,v,v", "v_and_or_b32": "v_and_or_b32", "v_alignbit_b32": "v_alignbit_b32 v,v,v,v",
             "v_bfe_i32": "v_bfe_i32 v,v,0,24", "v_bfe_u32": "v_bfe_u32", "v_mad_u32_u24": "v_mad_u32_u24 v,v,v,v", "v_fma_f64": "v_fma_f64 v,v,v,v",
             "v_cndmask_b32": "v_cndmask_b32 v,v,v,s[]", "v_mov_b32": "v_mov_b32", "v_bitop3_b32": "v_bitop3_b32 v,v,v,v", "v_not_b32": "v_xor_b32 v,v,v",
             "v_mul_lo_u32": "v_mul_lo_u32", "v_mul_hi_u32": "v_mul_hi_u32", "v_mad_u64_u32": "v_mad_u64_u32", "v_add3_u32": "v_add3_u32 v,v,v,v",
             "v_cvt_f64_u32": "v_cvt_f64_u32", "v_ldexp_f64": "v_mul_f64", "v_add_f64": "v_add_f64", "v_readfirstlane_b32": "v_readlane_b32", "v_mbcnt_lo_u32_b32": "v_add3_u32 v,v,v,v",
             "v_mbcnt_hi_u32_b32": "v_add3_u32 v,v,v,v", "v_bcnt_u32_b32": "v_add3_u32 v,v,v,v", "v_ffbl_b32": "v_bfe_u32", "v_bfrev_b32": "v_bfe_u32", "v_perm_b32": "v_perm_b32",
             "v_lshlrev_b64": "v_lshlrev_b64", "v_lshrrev_b64": "v_lshrrev_b64", "v_add_co_u32": "v_add_co+v_addc_co (pair)", "v_addc_co_u32": "v_add_co+v_addc_co (pair)"}
    if base.startswith("v_cmp"):
        return "v_cmp_* (" + ("64-bit" if "64" in base else "32-bit") + ")", cost["v_cmp_lt_u64 (vcc)" if "64" in base else "v_cmp_lt_u32 (vcc)"]
    if base in table:
        return base, cost[table[base]]
    return base + " (priced as v_add3_u32)", cost["v_add3_u32 v,v,v,v"]


def hot_block(asm, kernel_substr, want_ds_reads=None, want_atomics=None):
    i = asm.index(kernel_substr)
    body = asm[i:]
    end = body.find("\n\t.section", 10)
    body = body[:end] if end > 0 else body
    best = None
    for b in re.split(r"\n(?=\.LBB\d+_\d+:)", body):
        nread = b.count("ds_read_b32")
        nat = len(re.findall(r"ds_or_b32|ds_or_rtn", b))
        if want_ds_reads is not None and nread == want_ds_reads:
            return b
        if want_atomics is not None and nat >= want_atomics and (best is None or len(b) > len(best)):
            best = b
    return best


def histogram(block, cost):
    rows = collections.OrderedDict()
    other = collections.Counter()
    for ln in block.split("\n"):
        t = ln.strip()
        m = re.match(r"^(v_[a-z0-9_]+)\s+(.*)$", t)
        if m:
            label, cyc = classify(m.group(1), m.group(2), cost)
            r = rows.setdefault(label, [0, cyc])
            r[0] += 1
            continue
        m = re.match(r"^(s_waitcnt|s_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+)", t)
        if m:
            other["s_waitcnt" if m.group(1) == "s_waitcnt" else ("SALU" if m.group(1).startswith("s_") else m.group(1))] += 1
    return rows, other


def main():
    cost = opbench_costs()
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "k.hip")
        with open(src, "w") as f:
            f.write('#include "rbf_kernels_i64.h"\n#include "rbf_kernels_r64.h"\nusing namespace rbf;\n'
                    'template __global__ void rbf::k_query_r64<0>(uint64_t, uint32_t, const FrameTable, Seeds, const uint32_t *, uint64_t, uint32_t, uint32_t *, uint64_t, uint64_t *, uint4 *, uint32_t);\n'
                    'template __global__ void rbf::k_insert_tab<0>(const uint8_t *, uint64_t, uint64_t, const FrameTable, const uint4 *, uint32_t *, uint64_t, uint32_t, const SliceTable, uint32_t, uint32_t);\n')
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
        asm = open(out).read()
    model = {"source": "tools/make_issue_model.py: ISA histogram of the floor(k*) = 2 hot block x per-opcode issue cost from profiles/r02_opbench.txt (ch=8, 4 waves per SIMD)",
             "clock_ghz": CLOCK_GHZ, "simds": SIMDS}
    # ---- query: the 8-pixel straight-line block with 24 LDS reads
    blk = hot_block(asm, "k_query_r64", want_ds_reads=24)
    rows, other = histogram(blk, cost)
    px = 8.0
    pixel_frames = 29 * 1920 * 1080
    total_cyc = sum(c * cyc for c, cyc in rows.values())
    nvalu = sum(c for c, _ in rows.values())
    bound_ms = pixel_frames / 64.0 * (total_cyc / px) / SIMDS / (CLOCK_GHZ * 1e9) * 1e3
    model["k_query_lds"] = {"kernel": "k_query_r64<0>", "block": "frame pass, floor(k*) = 2, 8 pixels per lane",
                            "valu_per_pixel_frame": round(nvalu / px, 2), "valu_cycles_per_pixel_frame": round(total_cyc / px, 2),
                            "other_per_pixel_frame": {k: round(v / px, 2) for k, v in other.items()},
                            "table": [{"class": k, "per_pixel_frame": round(c / px, 3), "cycles_each": cyc} for k, (c, cyc) in rows.items()],
                            "issue_bound_ms": round(bound_ms, 4),
                            "note": "frame passes only (29 x 2 073 600 pixel-frames), including the 10 staging instructions a wave issues per frame; hashing, ranks and barriers are on top"}
    print("k_query_r64 frame pass: %.1f VALU and %.1f VALU cycles per (pixel, frame); launch issue bound %.1f us" % (nvalu / px, total_cyc / px, bound_ms * 1e3))
    for k, (c, cyc) in sorted(rows.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        print("    %-40s %6.2f per pixel-frame x %5.2f cycles" % (k, c / px, cyc))
    print("    other per pixel-frame:", {k: round(v / px, 2) for k, v in other.items()})
    path = os.path.join(REPO, "profiles", "r02_issue_model.json")
    with open(path, "w") as f:
        json.dump(model, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
