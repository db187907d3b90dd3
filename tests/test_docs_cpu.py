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
    for name in ("r03_query_traffic.json", "r03_issue_model.json"):
        assert name in src, name
        d = json.load(open(os.path.join(REPO, "profiles", name)))
        assert isinstance(d, dict) and d
    t = json.load(open(os.path.join(REPO, "profiles/r03_query_traffic.json")))
    assert abs(t["hbm_bytes_per_launch"] - (2 * t["fetch_size_kb"] + t["write_size_kb"]) * 1024) < 2048
    m = json.load(open(os.path.join(REPO, "profiles/r03_issue_model.json")))
    q = m["k_query_s64"]
    assert abs(q["valu_bound_ms"] - q["valu_wave_insts_per_launch"] * m["cycles_per_valu"] / (m["simds"] * m["clock_ghz"] * 1e6)) < 1e-4
