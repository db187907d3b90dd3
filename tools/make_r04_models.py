#!/usr/bin/env python3
"""Round-4 replayed constants of bench.py, from files under profiles/ and the compiled library -- no GPU needed to re-derive them.

  profiles/r04_query_traffic.json   HBM bytes per launch of k_query_u64 (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate --pmc passes of
                                    tools/profile.sh; reads doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane reads,
                                    calibrated on the mask kernel of the same run)
  profiles/r04_issue_model.json     what bounds k_query_u64 instead of HBM: the instruction mix of its floor(k*) = 2 frame loop (ISA of
                                    the in-tree library, hipcc -S) priced with the per-opcode issue costs of profiles/r04_opbench2.txt
                                    (launch-time throughput at 4 waves per SIMD), next to the counters (SQ_INSTS_*) and the measured
                                    per-frame time of the kernel (tools/bench_query5.hip: time against frame count)

usage: python tools/make_r04_models.py [profiles/r04_rocprofv3_summary.txt]      (run from the repo root)"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "new_bloom_filter_repo_amd", "csrc")
PROF = os.path.join(REPO, "profiles")
SIMDS, WAVES_PER_SIMD = 1024, 4
FRAMES, PIXELS = 29, 1920 * 1080


def opbench_costs(path):
    """label -> cycles per wave-instruction per SIMD at 4 waves per SIMD; also the measured shader clock."""
    cost, clock = {}, None
    for ln in open(path):
        m = re.search(r"shader clock measured in-kernel.*?: ([0-9.]+) MHz", ln)
        if m:
            clock = float(m.group(1)) / 1e3
        m = re.match(r"^(\S.*?)\s*\|.*?w=4\s+([0-9.]+)", ln)
        if m:
            cost[m.group(1).strip()] = float(m.group(2))
    return cost, clock


def classify(op, operands, cost):
    """(label, cycles) of one instruction: VALU by opcode and operand kind, SALU / LDS / VMEM by what a scalar instruction costs
    a SIMD next to a VALU stream (the v_add + s_add row: pair 2 x 2.87, v_add alone 2.72 -> ~3 cycles)."""
    base = re.sub(r"_e32$|_e64$|_dpp$|_sdwa$", "", op)
    ops = operands
    srcs = ops.split(",", 1)[1] if "," in ops else ""
    has_s = bool(re.search(r"(?<![a-z0-9_])(s\d+|s\[\d+:\d+\]|vcc|exec)", srcs))
    if base.startswith("s_") or base.startswith("ds_") or base.startswith("global_") or base.startswith("buffer_"):
        if base in ("s_waitcnt", "s_nop", "s_setprio", "s_barrier"):
            return base + " (no ALU)", 0.0
        kind = "SALU" if base.startswith("s_") else "LDS" if base.startswith("ds_") else "VMEM"
        return kind, 2.87 * 2 - cost["v_add_u32 v,v,v"] if kind == "SALU" else 0.0
    fast = {"v_add_u32": "v_add_u32 v,v,v", "v_sub_u32": "v_sub_u32 v,v,v", "v_subrev_u32": "v_sub_u32 v,v,v", "v_and_b32": "v_and_b32 v,v,v", "v_or_b32": "v_xor_b32 v,v,v",
            "v_xor_b32": "v_xor_b32 v,v,v", "v_not_b32": "v_xor_b32 v,v,v", "v_mov_b32": "v_mov_b32 v,v", "v_bitop3_b32": "v_bitop3_b32 v,v,v,v"}
    if base in fast:
        if has_s:
            return base + " (SGPR source)", cost["v_add_u32 v,s,v"]
        return base + " (VGPR / inline constant)", cost[fast[base]]
    if base in ("v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32"):
        if has_s:
            return base + " (SGPR shift)", cost["v_lshrrev_b32 v,s,v"]
        if re.match(r"^\s*v\d+,\s*\d+,", ops):
            return base + " (constant shift)", cost["v_lshrrev_b32 v,5,v"]
        return base + " (VGPR shift)", cost["v_lshlrev_b32 v,v,v"]
    table = {"v_min_u32": "v_min_u32 v,v,v", "v_lshl_add_u32": "v_lshl_add_u32 v,v,2,v", "v_lshl_or_b32": "v_lshl_or_b32 v,v,v,v", "v_add3_u32": "v_add3_u32 v,v,v,v",
             "v_alignbit_b32": "v_alignbit_b32 v,v,v,31", "v_bfe_u32": "v_bfe_u32 v,v,5,27", "v_bfe_i32": "v_bfe_u32 v,v,5,27", "v_and_or_b32": "v_and_or_b32 v,v,s,v",
             "v_perm_b32": "v_perm_b32 v,v,v,s", "v_mul_lo_u32": "v_mul_lo_u32 v,v,s", "v_mul_hi_u32": "v_mul_hi_u32 v,v,s", "v_mad_u32_u24": "v_mad_u32_u24 v,v,v,v",
             "v_mad_u64_u32": "v_mad_u64_u32 v[2],vcc,v,s,v[2]", "v_fma_f64": "v_fma_f64 v,v,s,v", "v_add_f64": "v_add_f64 v,v,s", "v_cvt_f64_u32": "v_cvt_f64_u32",
             "v_cndmask_b32": "v_cndmask_b32 v,v,v,s[2]", "v_readfirstlane_b32": "v_readfirstlane_b32 s,v", "v_readlane_b32": "v_readlane_b32 s,v,3",
             "v_mbcnt_lo_u32_b32": "v_mbcnt_lo_u32_b32 v,s,v", "v_mbcnt_hi_u32_b32": "v_mbcnt_lo_u32_b32 v,s,v", "v_bcnt_u32_b32": "v_add3_u32 v,v,v,v",
             "v_lshl_add_u64": "v_mad_u64_u32 v[2],vcc,v,s,v[2]", "v_mov_b64": "v_mov_b32 v,v"}
    if base.startswith("v_cmp"):
        return "v_cmp_*", cost["v_cmp_le_u32_sdwa s[2],v.b0,v"]
    if op.endswith("_dpp"):
        return base + " (dpp)", cost["v_mov_b32_dpp row_shr:1"]
    if base in table:
        return base, cost[table[base]]
    return base + " (priced as v_add3_u32)", cost["v_add3_u32 v,v,v,v"]


def frame_loop_blocks(asm, kernel_substr, nreads):
    """The basic blocks of the frame loop whose rows pass issues `nreads` LDS probe reads (floor(k*) = 2: 24): every block of that
    loop but the plain pass (the partial-wave path: it reduces with v_mad_u32_u24) -- the path a whole wave takes."""
    i = asm.index(kernel_substr)
    body = asm[i:]
    body = body[:body.index("s_endpgm")]
    blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
    loops = collections.defaultdict(list)
    for b in blocks:
        m = re.search(r"(?:Loop Header|in Loop: Header=|Parent Loop )\s*(BB\d+_\d+)?", b.split("\n", 1)[0])
        hdr = re.match(r"\.L(BB\d+_\d+):.*This (?:Inner )?Loop Header", b)
        key = hdr.group(1) if hdr else (re.search(r"Header=(BB\d+_\d+)", b.split("\n", 1)[0]).group(1) if "Header=" in b.split("\n", 1)[0] else None)
        if key:
            loops[key].append(b)
    for key, bl in loops.items():
        reads = sum(b.count("ds_read_b32") for b in bl if "v_mad_u32_u24" not in b)
        if reads == nreads:
            return [b for b in bl if "v_mad_u32_u24" not in b]
    raise SystemExit("no frame loop with %d probe reads in %s" % (nreads, kernel_substr))


def price(blocks, cost):
    rows = collections.OrderedDict()
    seen = set()
    for b in blocks:
        # the deferred outputs (flush) are compiled twice -- inside the rows pass and in front of the plain pass -- and a frame runs one
        # of the two: blocks with the same opcode sequence count once
        key = tuple(re.findall(r"^\s+((?:v|s|ds|global)_[a-z0-9_]+)", b, flags=re.M))
        if len(key) >= 2 and key in seen:
            continue
        seen.add(key)
        # the image's LDS-DMA is compiled twice (RowDmaC::issue): five loads under one M0 for a wave whose five KiB lie inside the row,
        # five loads with an exec mask each for the one wave that holds the row's end -- the second form is not what 15 of 16 waves run
        ragged_dma = b.count("global_load_lds_dwordx4") >= 5 and b.count("s_mov_b64 exec") >= 5
        w = 0.5 if ("_dpp" in b or "s_and_saveexec_b64" in b or "global_store_dword " in b) else 1.0      # the packed pass counts are reduced and stored every second frame
        for ln in b.split("\n"):
            t = ln.strip()
            if ragged_dma and re.match(r"^(global_load_lds_dwordx4|s_mov_b64 exec|s_mov_b64 s\[\d+:\d+\], exec|s_mov_b32 m0|s_mov_b32 s\d+, m0)", t):
                continue
            m = re.match(r"^((?:v|s|ds|global|buffer)_[a-z0-9_]+)\s*(.*)$", t)
            if not m:
                continue
            label, cyc = classify(m.group(1), m.group(2).split(";")[0], cost)
            r = rows.setdefault(label, [0.0, cyc])
            r[0] += w
    return rows


def counters(summary, kernel, section=None):
    """mean per launch of every counter the summary holds for `kernel` (first table that has it per counter; `section`: only tables
    whose path contains it, e.g. "pmc5"); "_launches" = launches behind the last counter taken."""
    out, cur, sec_ok = {}, None, True
    for ln in open(summary):
        if ln.startswith("=="):
            cur = None
            sec_ok = section is None or section in ln
            continue
        m = re.match(r"^\s{2}(\S.*?)\s+launches (\d+)", ln)
        if m:
            cur, nl = m.group(1).strip(), int(m.group(2))
            continue
        m = re.match(r"^\s{6}(\S+)\s+([0-9.]+)", ln)
        if m and cur and sec_ok and kernel in cur and m.group(1) not in out:
            out[m.group(1)] = float(m.group(2))
            out["_launches"] = nl
    return out


def query_traffic(summary):
    """profiles/r04_query_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh (pmc5 / pmc6: force bit 15, no table
    rewrite -- the condition of the default four-pipeline run; pmc3 / pmc4: one pipeline alone, sole holder of the hash table)."""
    q5, q6 = counters(summary, "k_query_u64", "pmc5"), counters(summary, "k_query_u64", "pmc6")
    q3, q4 = counters(summary, "k_query_u64", "pmc3"), counters(summary, "k_query_u64", "pmc4")
    m5, m6 = counters(summary, "k_residual_mask_gop", "pmc5"), counters(summary, "k_residual_mask_gop", "pmc6")
    if "FETCH_SIZE" not in q5 or "WRITE_SIZE" not in q6:
        return None
    alg = None
    try:
        b = json.load(open(os.path.join(PROF, "r04_bench_default.json")))
        alg = b["roofline"]["algorithmic_bytes_per_launch"]
    except Exception:
        pass
    t = {"kernel": "k_query_u64", "workload": "1920x1080 YUV444 8-bit, 30-frame GOP (29 inter-frames per launch), k*=2.3, planar Y resident",
         "source": "tools/profile.sh r04final (tools/collect_r04.sh): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace only "
                   "(profiles/r04_rocprofv3_summary.txt); passes pmc5/pmc6 (force bit 15: no table rewrite, as in the default four-pipeline run), mean of %d / %d launches"
                   % (q5.get("_launches", 0), q6.get("_launches", 0)),
         "fetch_size_kb": q5["FETCH_SIZE"], "write_size_kb": q6["WRITE_SIZE"],
         "correction": "MI355X_MICROARCH.md 'HBM': on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced reads -- the probe-image LDS-DMA reads 16 bytes per "
                       "lane -- so reads are doubled (calibration in the same run: the planar mask kernel reads 62.2 MB of Y planes + 3 re-read frames = 68.4 MB; see mask_kernel); "
                       "WRITE_SIZE used as reported",
         "hbm_bytes_per_launch": int(round((2 * q5["FETCH_SIZE"] + q6["WRITE_SIZE"]) * 1024)),
         "algorithmic_bytes_per_launch": alg}
    if "FETCH_SIZE" in q3 and "WRITE_SIZE" in q4:
        t["with_table_rewrite_bytes_per_launch"] = int(round((2 * q3["FETCH_SIZE"] + q4["WRITE_SIZE"]) * 1024))
    if "FETCH_SIZE" in m5 and "WRITE_SIZE" in m6:
        t["mask_kernel"] = {"kernel": "k_residual_mask_gop<unsigned char, 1, false, true>", "fetch_size_kb": m5["FETCH_SIZE"], "write_size_kb": m6["WRITE_SIZE"],
                            "algorithmic_read_bytes": 30 * 1920 * 1080,
                            "read_bytes_x2": int(round(2 * m5["FETCH_SIZE"] * 1024)),
                            "note": "reads (x2 correction) against 62.2 MB of luma: three frames are read twice (four temporal chunks); writes = 7.5 MB of masks + 7.5 MB of cleared witness rows"}
    json.dump(t, open(os.path.join(PROF, "r04_query_traffic.json"), "w"), indent=1)
    return t


def main():
    summary = sys.argv[1] if len(sys.argv) > 1 else os.path.join(PROF, "r04_rocprofv3_summary.txt")
    cost, clock = opbench_costs(os.path.join(PROF, "r04_opbench2.txt"))
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, "rbf_api.hip")],
                              stderr=subprocess.DEVNULL)
        asm = open(out).read()
    rows = price(frame_loop_blocks(asm, "_ZN3rbf11k_query_u64E", 24), cost)
    kinds = collections.Counter()
    cyc = collections.Counter()
    for label, (cnt, c) in rows.items():
        kind = "SALU" if label == "SALU" else "LDS" if label == "LDS" else "VMEM" if label == "VMEM" else "wait / nop / prio / barrier" if "(no ALU)" in label else "VALU"
        kinds[kind] += cnt
        cyc[kind] += cnt * c
    valu_cycles, salu_cycles = cyc["VALU"], cyc["SALU"]
    per_frame_simd = (valu_cycles + salu_cycles) * WAVES_PER_SIMD
    model = {
        "kernel": "k_query_u64", "workload": "1920x1080, 29 coded frames, floor(k*) = 2",
        "source": "tools/make_r04_models.py: ISA of the in-tree library's frame loop (the whole-wave path) x per-opcode issue cost of profiles/r04_opbench2.txt at 4 waves per SIMD",
        "shader_clock_ghz": clock, "simds": SIMDS, "waves_per_simd": WAVES_PER_SIMD,
        "instructions_per_wave_and_frame": {k: round(v, 1) for k, v in kinds.items()},
        "issue_cycles_per_wave_and_frame": {"VALU": round(valu_cycles, 1), "SALU (what a scalar instruction costs the SIMD next to a VALU stream)": round(salu_cycles, 1)},
        "table": [{"class": k, "per_wave_and_frame": round(c, 1), "cycles_each": round(x, 2)} for k, (c, x) in sorted(rows.items(), key=lambda kv: -kv[1][0] * kv[1][1]) if x],
        "frame_loop_issue_bound_ms": round(FRAMES * per_frame_simd / (clock * 1e9) * 1e3, 4),
        "frame_loop_issue_bound_valu_only_ms": round(FRAMES * valu_cycles * WAVES_PER_SIMD / (clock * 1e9) * 1e3, 4),
        "note": "a bound on the 29 frame passes only: the prologue (launch, 3 XXH64 per pixel index, rank search, frame records, first image: ~14.7 us measured as the "
                "one-frame launch minus one frame in profiles/r04_query_u64.txt) is on top.  Opcode costs differ by class on gfx950: v_add / v_sub / v_and / v_xor / v_mov / "
                "constant shifts issue in ~2.4-2.7 cycles per wave-instruction, everything else (min, variable shifts, lshl_add, lshl_or, alignbit, fma_f64, compares, "
                "cndmask, readfirstlane) in ~4.0-4.3, v_mad_u64_u32 in 4.75: round 3's 'every VALU instruction occupies its SIMD for four cycles' holds for the second "
                "class only, round 2's 1.2-cycle rows were an artefact of dividing a per-wave median by an assumed residency (tools/opbench2.hip measures launch throughput).",
    }
    harness = os.path.join(PROF, "r04_query_u64_harness.txt")      # ./build/bench_query5: launches of 1, 2, 8, 15, 29 frames
    if os.path.exists(harness):
        pts = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r"frames\s+(\d+): k_query_u64 full\s+([0-9.]+)", open(harness).read())}
        if 1 in pts and FRAMES in pts:
            per_frame = (pts[FRAMES] - pts[1]) / (FRAMES - 1)
            model["measured"] = {"source": "profiles/r04_query_u64_harness.txt (tools/bench_query5.hip: the same kernel launched on 1 and on %d frames)" % FRAMES,
                                 "launch_1_frame_us": pts[1], "launch_%d_frames_us" % FRAMES: pts[FRAMES], "per_frame_us": round(per_frame, 3)}
            model["prologue_ms"] = round((pts[1] - per_frame) * 1e-3, 4)
            model["frame_loop_measured_ms"] = round(per_frame * FRAMES * 1e-3, 4)
    if os.path.exists(summary):
        c = counters(summary, "k_query_u64")
        if c:
            model["counters_per_launch"] = {k: c[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE") if k in c}
            model["counters_source"] = os.path.relpath(summary, REPO)
            wf = FRAMES * (PIXELS // 512)
            model["counter_check"] = {"wave_frames_per_launch": wf, "modelled_valu_in_frame_loop": int(kinds["VALU"] * wf), "modelled_salu_in_frame_loop": int(kinds["SALU"] * wf),
                                      "note": "SQ_INSTS_VALU - modelled = the prologue's VALU instructions; the model's SALU count includes the loop's branches (s_cbranch*), "
                                              "which SQ_INSTS_SALU does not count, and the deferred-output blocks that only every second frame runs"}
            if "SQ_INSTS_VALU" in c:
                avg = sum(x * n for x, n in ((v[1], v[0]) for k, v in rows.items() if k not in ("SALU", "LDS", "VMEM") and "(no ALU)" not in k)) / max(1.0, kinds["VALU"])
                model["whole_kernel_valu_bound_ms"] = round(c["SQ_INSTS_VALU"] * avg / SIMDS / (clock * 1e9) * 1e3, 4)
                model["whole_kernel_valu_bound_note"] = "SQ_INSTS_VALU of the whole launch (prologue included) x the frame loop's mean cost per VALU instruction (%.2f cycles)" % avg
        t = query_traffic(summary)
        if t:
            print("traffic:", t["hbm_bytes_per_launch"], "bytes per launch (algorithmic", t["algorithmic_bytes_per_launch"], ")")
    json.dump(model, open(os.path.join(PROF, "r04_issue_model.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in model.items() if k != "table"}, indent=1))
    for r in model["table"][:14]:
        print("   %-44s %6.1f x %5.2f" % (r["class"], r["per_wave_and_frame"], r["cycles_each"]))


if __name__ == "__main__":
    main()
