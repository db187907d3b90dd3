#!/usr/bin/env python3
"""The integer-issue roofline of the whole step ("expect the kernel to sit on the integer-ALU ceiling first; report both"),
regenerable by ONE command from files under profiles/ and the in-tree sources -- no GPU needed:

    python tools/make_issue_model.py            ->  profiles/r06_issue_model.json  (+ a table on stdout)

What goes in
  * profiles/r06_counters.json   rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS / SQ_INSTS_VMEM_* per launch of every kernel of
                                 a step, one pipeline alone, for the launch shapes g1 (1080p, one 30-frame GOP per call), g4 (four GOPs per
                                 call) and u4 (16-bit, four GOPs) -- tools/r06_profile.sh + tools/make_traffic.py write it; the launch
                                 times beside them are the --stats pass of the same shape.
  * profiles/r04_opbench2.txt    what ONE wave-instruction costs a SIMD in issue cycles, per opcode and operand kind, measured on this chip as
                                 launch throughput at 4 waves per SIMD (tools/opbench2.hip), and the shader clock under that load.
  * the library's ISA            `hipcc -S` of csrc/rbf_api.hip (cross-compiles here): per kernel the opcode mix of the instructions that sit
                                 inside loops gives the MEAN price of a vector instruction of that kernel (the counters say how many were
                                 issued, not which).  Straight-line code (the query kernel's hashing prologue, 16 % of its instructions at
                                 one GOP per call, 5 % at four) is priced at the loops' mean.

What comes out, per shape: for every kernel the issue bound  VALU x mean price / (1024 SIMDs x clock)  -- and a second bound that also
charges every scalar instruction the ~3 cycles it costs a SIMD in the micro-benchmark (not confirmed by the kernels, HISTORY.md round 4) --
next to its launch time alone, and the same summed over the step: all kernels of all pipelines share the same 1024 SIMDs, so the
overlapped step cannot be shorter than the sum of its kernels' issue times, whatever overlaps.  bench.py divides these bounds by the times
it measures live (`roofline.issue`).
"""
import collections
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "new_bloom_filter_repo_amd", "csrc")
PROF = os.path.join(REPO, "profiles")
SIMDS = 1024                                      # 256 CUs x 4
ROUND = "r06"

ROLE = [("k_residual_mask", "mask"), ("k_insert", "insert"), ("k_filter_reduce", "reduce"), ("k_query", "query"), ("k_chunk_offsets", "scan"),
        ("k_compact_witness", "stitch"), ("k_hash_table", "hashtab"), ("k_pack_records", "pack"), ("k_finish_ones", "mask"), ("k_expand_mask", "expand")]
REMAINDER = {"query": "one workgroup barrier per frame, the LDS-DMA staging of the next probe image, dependent latency that four waves per SIMD do not cover, ~15 us of prologue per launch at one wave-slot per pixel",
             "insert": "the 16-byte table gather (texture addresser walks 64 scattered lanes per load), LDS atomics, the FP64 reductions' dependent chains",
             "mask": "HBM: the kernel is bandwidth-bound (1 byte per pixel and frame read, 1/8 written), not issue-bound",
             "stitch": "latency of ~3 700 small workgroups per GOP: launch ramp, one LDS round trip per word",
             "reduce": "HBM / L2: streams the partial filters", "scan": "one small workgroup per frame: launch latency"}


def opbench_costs(path):
    """label -> cycles per wave-instruction per SIMD at 4 waves per SIMD; and the shader clock (GHz) measured in the same run."""
    cost, clock = {}, None
    for ln in open(path):
        m = re.search(r"shader clock measured in-kernel.*?: ([0-9.]+) MHz", ln)
        if m:
            clock = float(m.group(1)) / 1e3
        m = re.match(r"^(\S.*?)\s*\|.*?w=4\s+([0-9.]+)", ln)
        if m:
            cost[m.group(1).strip()] = float(m.group(2))
    if clock is None or "v_add_u32 v,v,v" not in cost:
        raise SystemExit("%s: no clock / no rows" % path)
    return cost, clock


CHEAP = {"v_add_u32": "v_add_u32 v,v,v", "v_sub_u32": "v_sub_u32 v,v,v", "v_subrev_u32": "v_sub_u32 v,v,v", "v_and_b32": "v_and_b32 v,v,v", "v_or_b32": "v_xor_b32 v,v,v",
         "v_xor_b32": "v_xor_b32 v,v,v", "v_not_b32": "v_xor_b32 v,v,v", "v_mov_b32": "v_mov_b32 v,v", "v_bitop3_b32": "v_bitop3_b32 v,v,v,v"}
PRICED = {"v_min_u32": "v_min_u32 v,v,v", "v_lshl_add_u32": "v_lshl_add_u32 v,v,2,v", "v_lshl_or_b32": "v_lshl_or_b32 v,v,v,v", "v_add3_u32": "v_add3_u32 v,v,v,v",
          "v_alignbit_b32": "v_alignbit_b32 v,v,v,31", "v_bfe_u32": "v_bfe_u32 v,v,5,27", "v_bfe_i32": "v_bfe_u32 v,v,5,27", "v_and_or_b32": "v_and_or_b32 v,v,s,v",
          "v_perm_b32": "v_perm_b32 v,v,v,s", "v_mul_lo_u32": "v_mul_lo_u32 v,v,s", "v_mul_hi_u32": "v_mul_hi_u32 v,v,s", "v_mad_u32_u24": "v_mad_u32_u24 v,v,v,v",
          "v_mad_u64_u32": "v_mad_u64_u32 v[2],vcc,v,s,v[2]", "v_lshl_add_u64": "v_mad_u64_u32 v[2],vcc,v,s,v[2]", "v_fma_f64": "v_fma_f64 v,v,s,v", "v_add_f64": "v_add_f64 v,v,s",
          "v_cvt_f64_u32": "v_cvt_f64_u32", "v_readfirstlane_b32": "v_readfirstlane_b32 s,v", "v_readlane_b32": "v_readlane_b32 s,v,3",
          "v_mbcnt_lo_u32_b32": "v_mbcnt_lo_u32_b32 v,s,v", "v_mbcnt_hi_u32_b32": "v_mbcnt_lo_u32_b32 v,s,v"}


def classify(op, operands, cost):
    """(class label, issue cycles) of one VALU instruction by opcode and operand kind; None for anything that is not VALU."""
    if not op.startswith("v_"):
        return None
    base = re.sub(r"_e32$|_e64$|_dpp$|_sdwa$", "", op)
    srcs = operands.split(",", 1)[1] if "," in operands else ""
    scalar_src = bool(re.search(r"(?<![a-z0-9_])(s\d+|s\[\d+:\d+\]|vcc|exec|m0)", srcs)) or bool(re.search(r"0x[0-9a-f]{3,}", srcs))     # SGPR or literal
    if op.endswith("_dpp"):
        return "dpp", cost["v_mov_b32_dpp row_shr:1"]
    if base.startswith("v_cmp"):
        return "compare", cost["v_cmp_le_u32_sdwa s[2],v.b0,v"]
    if base == "v_cndmask_b32":
        return ("cndmask through VCC", cost["v_cndmask_b32 v,v,v,vcc"]) if re.search(r"vcc\s*$", operands.strip()) else ("cndmask, SGPR pair", cost["v_cndmask_b32 v,v,v,s[2]"])
    if base in CHEAP:
        return ("simple op with an SGPR / literal source", cost["v_add_u32 v,s,v"]) if scalar_src else ("simple op (add, sub, and, or, xor, mov, bitop3), VGPR / inline sources", cost[CHEAP[base]])
    if base in ("v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32"):
        if re.match(r"^\s*v\d+,\s*\d+,", operands):
            return "shift by a constant", cost["v_lshrrev_b32 v,5,v"]
        return "shift by a register", cost["v_lshlrev_b32 v,v,v"]
    if base in PRICED:
        return base, cost[PRICED[base]]
    return "other VALU (priced as v_add3_u32)", cost["v_add3_u32 v,v,v,v"]


def disassemble():
    """`hipcc -S` of the library's one translation unit -> {demangled kernel name: asm text}."""
    out = os.path.join(REPO, "build", "rbf_api.s")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = os.path.join(CSRC, "rbf_api.hip")
    newest = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC))
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(REPO, "include"),
                               "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
    asm = open(out).read()
    kernels = {}
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)s_endpgm", asm, flags=re.S | re.M):
        kernels[m.group(1)] = m.group(2)
    names = subprocess.run(["c++filt"] + list(kernels), capture_output=True, text=True).stdout.split("\n")
    out = {}
    for sym, dem in zip(kernels, names):
        dem = re.sub(r"^void ", "", dem).split("(")[0].replace("rbf::", "")
        out[dem] = kernels[sym]
    return out


# The query kernel compiles one frame loop per floor(k*) class; a launch runs the one its frames need.  BASELINE's k* = 2.3 is floor(k*) = 2:
# the loop whose rows pass issues 8 pixels x 3 probes = 24 ds_read_b32, without the partial-wave path (which reduces with v_mad_u32_u24).
HOT_LOOP = {"k_query_u64": {"ds_read_b32": 24, "skip_blocks_with": "v_mad_u32_u24"}}


def outer_loop_of(block):
    """Label of the depth-1 loop a basic block belongs to (None: straight-line code)."""
    head = "\n".join(block.split("\n")[:6])
    m = re.match(r"\.L(BB\d+_\d+):.*Loop Header: Depth=1", head)
    if m:
        return m.group(1)
    m = re.search(r"(?:Header=|Parent Loop )(BB\d+_\d+) Depth=1", head)
    return m.group(1) if m else None


def loop_mix(body, cost, hot=None):
    """Mean issue price of the VALU instructions that sit inside loops (basic blocks the compiler annotates as loop members); the whole
    kernel when its loops hold fewer than 24 of them.  hot: restrict to ONE depth-1 loop, picked by an instruction count (HOT_LOOP).
    Returns (mean cycles, {class: share}, in-loop VALU count, whole-kernel VALU count)."""
    blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
    chosen = None
    if hot:
        groups = collections.defaultdict(list)
        for b in blocks:
            key = outer_loop_of(b)
            if key and hot["skip_blocks_with"] not in b:
                groups[key].append(b)
        for key, bl in groups.items():
            if sum(b.count("ds_read_b32") for b in bl) == hot["ds_read_b32"]:
                chosen = set(id(b) for b in bl)
                break
    rows_loop, rows_all = collections.Counter(), collections.Counter()
    price = {}
    for b in blocks:
        head = b.split("\n", 1)[0]
        in_loop = ("Loop Header" in head or "in Loop:" in head or "Parent Loop" in head) if chosen is None else id(b) in chosen
        for ln in b.split("\n"):
            m = re.match(r"^\s+([a-z][a-z0-9_]+)\s*(.*)$", ln)
            if not m:
                continue
            c = classify(m.group(1), m.group(2).split(";")[0], cost)
            if c is None:
                continue
            price[c[0]] = c[1]
            rows_all[c[0]] += 1
            if in_loop:
                rows_loop[c[0]] += 1
    rows = rows_loop if sum(rows_loop.values()) >= 24 else rows_all
    total = sum(rows.values())
    if not total:
        return cost["v_add3_u32 v,v,v,v"], {}, 0, 0
    mean = sum(n * price[k] for k, n in rows.items()) / total
    mix = {k: {"share": round(n / total, 4), "cycles": price[k]} for k, n in rows.most_common()}
    return mean, mix, sum(rows_loop.values()), sum(rows_all.values())


def role_of(name):
    for prefix, role in ROLE:
        if name.startswith(prefix):
            return role
    return None


def main():
    counters_path = os.path.join(PROF, "%s_counters.json" % ROUND)
    counters = json.load(open(counters_path))
    cost, clock = opbench_costs(os.path.join(PROF, "r04_opbench2.txt"))
    salu_cycles = round(2 * cost["MIX v_add_u32 + s_add_u32 (count: both)"] - cost["v_add_u32 v,v,v"], 2)      # the pair minus the v_add alone
    asm = disassemble()
    model = {"round": ROUND, "generator": "python tools/make_issue_model.py", "simds": SIMDS, "shader_clock_ghz": round(clock, 4),
             "prices": "profiles/r04_opbench2.txt (tools/opbench2.hip: launch throughput at 4 waves per SIMD on this chip; two classes of vector instruction: ~2.4-2.8 cycles for add / sub / "
                       "and / or / xor / mov / bitop3 / constant shifts on VGPR or inline sources, ~4.0-4.3 for everything else, 4.75 for v_mad_u64_u32, 21.6 for v_cndmask through VCC)",
             "counters": "profiles/%s_counters.json (%s)" % (ROUND, counters.get("source", "")),
             "salu_cycles_next_to_valu": salu_cycles,
             "how": "per kernel: bound = SQ_INSTS_VALU x mean price of a vector instruction of that kernel / (1024 SIMDs x clock); the mean price is the opcode mix of the instructions "
                    "inside the kernel's loops (ISA of the in-tree library, hipcc -S) priced per opcode and operand kind; `valu_salu` adds SQ_INSTS_SALU x the cycles a scalar instruction "
                    "costs a SIMD in the micro-benchmark.  step = sum over the kernels of one launch sequence: every pipeline's kernels share the same SIMDs.",
             "shapes": {}}
    print("# integer-issue roofline (%s): %d SIMDs at %.3f GHz; prices profiles/r04_opbench2.txt; counters profiles/%s_counters.json" % (ROUND, SIMDS, clock, ROUND))
    for shape, kernels in counters["shapes"].items():
        rows, step = {}, collections.Counter()
        print("\n== %s" % shape)
        print("  %-58s %10s %9s %6s %9s %9s %9s %6s" % ("kernel", "VALU", "SALU", "cyc/V", "bound us", "+SALU us", "alone us", "frac"))
        for name, c in kernels.items():
            role = role_of(name)
            if role is None or "SQ_INSTS_VALU" not in c:
                continue
            body = asm.get(name)
            if body is None:                      # rocprofv3 prints template arguments a little differently now and then: match on the stem
                stem = name.split("<")[0]
                cands = [k for k in asm if k.split("<")[0] == stem]
                body = asm[cands[0]] if len(cands) >= 1 else None
            if body is None:
                raise SystemExit("no ISA for kernel %s" % name)
            mean, mix, nloop, nall = loop_mix(body, cost, HOT_LOOP.get(name.split("<")[0]))
            valu, salu = c["SQ_INSTS_VALU"], c.get("SQ_INSTS_SALU", 0.0)
            b_valu = valu * mean / (SIMDS * clock * 1e3)                    # us
            b_both = b_valu + salu * salu_cycles / (SIMDS * clock * 1e3)
            alone = c.get("avg_launch_us")
            row = {"role": role, "valu": int(valu), "salu": int(salu), "lds": int(c.get("SQ_INSTS_LDS", 0)), "vmem": int(c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)),
                   "mean_valu_cycles": round(mean, 3), "isa_valu_in_loops": nloop, "isa_valu_total": nall, "mix": mix,
                   "valu_bound_us": round(b_valu, 2), "valu_salu_bound_us": round(b_both, 2), "alone_us": alone,
                   "frac_of_alone": round(b_valu / alone, 3) if alone else None,
                   "remainder_us": round(alone - b_valu, 2) if alone else None, "remainder_is": REMAINDER.get(role)}
            rows[name] = row
            print("  %-58s %10d %9d %6.2f %9.2f %9.2f %9s %6s" % (name[:58], valu, salu, mean, b_valu, b_both, "%.2f" % alone if alone else "-", "%.2f" % (b_valu / alone) if alone else "-"))
            if role not in ("hashtab", "pack", "expand"):      # (the table is built once per process; packing belongs to the gather)
                n = 2 if name.startswith("k_insert_records") else 1      # (two tile launches per step at 2160p)
                step["valu"] += valu * n
                step["salu"] += salu * n
                step["valu_bound_us"] += b_valu * n
                step["valu_salu_bound_us"] += b_both * n
                step["alone_us"] += (alone or 0.0) * n
        m = re.match(r"(\d+)x(\d+)x(\d+)_(\d+)bit_gpc(\d+)", shape)
        W, H, F, bits, gpc = [int(x) for x in m.groups()]
        pixels = W * H * (F - 1) * gpc
        model["shapes"][shape] = {"kernels": rows, "coded_frames_per_step": (F - 1) * gpc, "pixels_per_step": pixels,
                                  "step": {"valu": int(step["valu"]), "salu": int(step["salu"]), "valu_bound_ms": round(step["valu_bound_us"] / 1e3, 5),
                                           "valu_salu_bound_ms": round(step["valu_salu_bound_us"] / 1e3, 5), "kernels_alone_sum_ms": round(step["alone_us"] / 1e3, 5),
                                           "mpixels_per_s_at_the_valu_bound": round(pixels / step["valu_bound_us"], 1) if step["valu_bound_us"] else None,
                                           "valu_per_pixel": round(step["valu"] * 64 / pixels, 2)}}
        s = model["shapes"][shape]["step"]
        print("  step: %d VALU + %d SALU wave-instructions (%.1f vector lane-instructions per pixel) -> issue bound %.4f ms (VALU) / %.4f ms (VALU + SALU); kernels alone sum %.4f ms; %.0f Mpixel/s at the VALU bound"
              % (s["valu"], s["salu"], s["valu_per_pixel"], s["valu_bound_ms"], s["valu_salu_bound_ms"], s["kernels_alone_sum_ms"], s["mpixels_per_s_at_the_valu_bound"] or 0))
    # measured steps of the round's evidence run, when it is there (bench.py recomputes the fractions from what it measures live)
    bench = os.path.join(PROF, "%s_bench_default.json" % ROUND)
    if os.path.exists(bench):
        d = json.loads([ln for ln in open(bench).read().splitlines() if ln.startswith("{")][-1])
        meas = {"1920x1080x30_8bit_gpc1": (d.get("steady_state") or {}).get("ms_per_step"), "1920x1080x30_8bit_gpc4": (d.get("batched_gops") or {}).get("ms_per_step")}
        for shape, ms in meas.items():
            if ms and shape in model["shapes"]:
                s = model["shapes"][shape]["step"]
                s["measured_ms"] = ms
                s["measured_source"] = "profiles/%s_bench_default.json (%s)" % (ROUND, "steady_state" if shape.endswith("gpc1") else "batched_gops leg")
                s["frac_of_measured"] = [round(s["valu_bound_ms"] / ms, 3), round(s["valu_salu_bound_ms"] / ms, 3)]
                print("%s: measured %.4f ms per step with four pipelines -> the step runs at %.0f %% (VALU) / %.0f %% (VALU + SALU) of its issue bound"
                      % (shape, ms, 100 * s["frac_of_measured"][0], 100 * s["frac_of_measured"][1]))
    path = os.path.join(PROF, "%s_issue_model.json" % ROUND)
    with open(path, "w") as f:
        json.dump(model, f, indent=1)
    print("\nwrote", os.path.relpath(path, REPO))


if __name__ == "__main__":
    sys.exit(main())
