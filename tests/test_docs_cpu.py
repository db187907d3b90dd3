"""The documents cite evidence by path: every `profiles/...`, `tools/...`, `tests/...`, `oracle/...`, `include/...` and
`new_bloom_filter_repo_amd/...` path that DESIGN.md, README.md and INTEGRATION.md name must exist in the tree (wildcards must
match something), and bench.py's replayed constants must be the files it names."""
import glob
import json
import os
import re

from conftest import REPO

DOCS = ("DESIGN.md", "README.md", "INTEGRATION.md")
ROOTS = ("profiles/", "tools/", "tests/", "oracle/", "include/", "new_bloom_filter_repo_amd/")


def cited_paths(text):
    for m in re.finditer(r"`([^`\s]+)`", text):
        tok = m.group(1).split("::")[0].rstrip(".,;:)")
        if tok.startswith(ROOTS) and "…" not in tok and "<" not in tok and tok != "oracle/_ref":     # (a build directory the docs say does not exist)
            yield tok


def test_every_cited_path_exists():
    missing = []
    for doc in DOCS:
        text = open(os.path.join(REPO, doc), encoding="utf-8").read()
        for tok in sorted(set(cited_paths(text))):
            pattern = os.path.join(REPO, tok)
            if "*" in tok:
                ok = bool(glob.glob(pattern))
            else:
                ok = os.path.exists(pattern) or os.path.exists(pattern.rstrip("/"))
            if not ok:
                missing.append((doc, tok))
    assert not missing, missing


def test_bench_replays_the_constants_it_names():
    src = open(os.path.join(REPO, "bench.py"), encoding="utf-8").read()
    for name in ("r05_traffic.json", "r04_issue_model.json"):
        assert name in src, name
        d = json.load(open(os.path.join(REPO, "profiles", name)))
        assert isinstance(d, dict) and d
    for key, t in json.load(open(os.path.join(REPO, "profiles/r05_traffic.json"))).items():
        # FETCH_SIZE doubled (16-byte-per-lane reads on gfx950, MI355X_MICROARCH.md) + WRITE_SIZE, in KiB
        assert abs(t["hbm_bytes_per_launch"] - (2 * t["fetch_size_kb"] + t["write_size_kb"]) * 1024) < 2048, key
    m = json.load(open(os.path.join(REPO, "profiles/r04_issue_model.json")))
    # the bound is the table it is printed next to: sum(count x cycles) x waves per SIMD x frames / clock
    cyc = sum(r["per_wave_and_frame"] * r["cycles_each"] for r in m["table"])
    bound = 29 * cyc * m["waves_per_simd"] / (m["shader_clock_ghz"] * 1e9) * 1e3
    assert abs(bound - m["frame_loop_issue_bound_ms"]) < 1e-3, (bound, m["frame_loop_issue_bound_ms"])
    assert m["frame_loop_issue_bound_valu_only_ms"] < m["frame_loop_issue_bound_ms"]
    assert m["kernel"].startswith("k_query_u64") and "k_query_u64" in src



def test_integration_names_every_entry_point():
    """INTEGRATION.md is the map from the C ABI to the reference: every function include/rbf.h declares appears in it."""
    hdr = open(os.path.join(REPO, "include", "rbf.h"), encoding="utf-8").read()
    doc = open(os.path.join(REPO, "INTEGRATION.md"), encoding="utf-8").read()
    declared = set(re.findall(r"\b(rbf_[a-z0-9_]+)\s*\(", hdr))
    missing = sorted(n for n in declared if n not in doc)
    assert not missing, missing
