"""The kernels of the headline step and of BASELINE config 4 must not spill: a spill in a kernel that runs for the whole step turns into
scratch traffic on every launch (round 4: the generalised k_query_s64t spilled 146 bytes per lane and 2160p fell from 206 to 143
Gpixel/s without a single test failing).  Cross-compiles with hipcc: no GPU needed."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HOT = (                                   # demangled name prefixes (tools/kernel_resources.py prints them)
    "k_query_u64", "k_query_u64w", "k_query_s64t<0>", "k_query_s64t<1>", "k_query_s64t<2>", "k_insert_tab<false, true>", "k_insert_tab<false, false>", "k_insert_positions<true>",
    "k_insert_records", "k_filter_reduce", "k_chunk_offsets", "k_compact_witness", "k_residual_mask_gop<unsigned char, 1, true, true>", "k_expand_mask", "k_hash_table",
)


def test_hot_kernels_do_not_spill_and_keep_their_occupancy():
    out = subprocess.run(["python", os.path.join(REPO, "tools", "kernel_resources.py")], capture_output=True, text=True, check=True).stdout
    rows = {}
    for ln in out.splitlines():
        if ln.startswith("#") or ln.startswith("kernel "):
            continue
        name, rest = ln[:84].strip(), ln[84:].split()
        rows[name] = rest                  # VGPRs, SGPRs, scratch, spills, waves/SIMD, LDS
    for k in HOT:
        hits = [n for n in rows if n.startswith(k)]
        assert hits, (k, sorted(rows)[:60])
        for n in hits:
            vgprs, _, scratch, spills, waves = rows[n][:5]
            assert scratch == "0" and spills == "0", (n, rows[n])
    # the 1024-thread kernels need four waves per SIMD: at most 128 registers
    for k in ("k_query_u64", "k_query_u64w", "k_query_s64t<0>", "k_query_s64t<1>", "k_query_s64t<2>", "k_insert_tab<false, true>"):
        for n in (n for n in rows if n.startswith(k)):
            assert int(rows[n][0]) <= 128, (n, rows[n])
