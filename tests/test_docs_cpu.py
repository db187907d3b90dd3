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


def test_paths_cited_inside_bench_and_tools_exist():
    """bench.py, HISTORY.md and the tools cite evidence and generators by path as well (bench.py named a tool the prune had
    deleted): every `profiles/...` / `tools/...` / `tests/...` path in their text must exist.  HISTORY.md may name files that earlier
    rounds pruned -- only in sentences that say so (they contain 'pruned', 'git history' or 'removed')."""
    missing = []
    files = ["bench.py", "__graft_entry__.py"] + [os.path.join("tools", f) for f in sorted(os.listdir(os.path.join(REPO, "tools"))) if f.endswith((".py", ".sh"))]
    for rel in files:
        text = open(os.path.join(REPO, rel), encoding="utf-8").read()
        for m in re.finditer(r"(?<![\w/$])((?:profiles|tools|tests)/[A-Za-z0-9_./*-]+)", text):
            tok = m.group(1).rstrip(".,;:)")
            if tok.endswith("/") or "<" in tok or "$" in tok or "NN" in tok:
                continue
            if tok.startswith("profiles/") and (rel.startswith("tools/") or "%" in text[m.end():m.end() + 1]):
                continue                       # tools WRITE profiles/ files (and bench.py formats some names)
            pattern = os.path.join(REPO, tok)
            if not (glob.glob(pattern) if "*" in tok else os.path.exists(pattern)):
                missing.append((rel, tok))
    assert not missing, missing


def test_bench_replays_the_constants_it_names():
    src = open(os.path.join(REPO, "bench.py"), encoding="utf-8").read()
    for name in ("r06_traffic.json", "r06_issue_model.json"):
        assert name in src, name
        d = json.load(open(os.path.join(REPO, "profiles", name)))
        assert isinstance(d, dict) and d
    for key, t in json.load(open(os.path.join(REPO, "profiles/r06_traffic.json"))).items():
        # FETCH_SIZE doubled (16-byte-per-lane reads on gfx950, MI355X_MICROARCH.md) + WRITE_SIZE, in KiB
        assert abs(t["hbm_bytes_per_launch"] - (2 * t["fetch_size_kb"] + t["write_size_kb"]) * 1024) < 2048, key
    m = json.load(open(os.path.join(REPO, "profiles/r06_issue_model.json")))
    assert m["generator"] == "python tools/make_issue_model.py" and os.path.exists(os.path.join(REPO, "tools", "make_issue_model.py"))
    counters = json.load(open(os.path.join(REPO, "profiles/r06_counters.json")))["shapes"]
    for shape, sh in m["shapes"].items():
        # every kernel's bound is its counter x its mean price / (SIMDs x clock); the step is the sum of its kernels (the table once, packing never)
        total = 0.0
        for name, k in sh["kernels"].items():
            assert k["valu"] == int(counters[shape][name]["SQ_INSTS_VALU"]), (shape, name)
            bound = k["valu"] * k["mean_valu_cycles"] / (m["simds"] * m["shader_clock_ghz"] * 1e3)
            assert abs(bound - k["valu_bound_us"]) < 0.02 + 0.002 * bound, (shape, name, bound, k["valu_bound_us"])
            assert 2.3 < k["mean_valu_cycles"] < 6.0 and abs(sum(c["share"] for c in k["mix"].values()) - 1.0) < 0.01
            if k["role"] not in ("hashtab", "pack", "expand"):
                total += k["valu_bound_us"] * (2 if name.startswith("k_insert_records") else 1)
        assert abs(total / 1e3 - sh["step"]["valu_bound_ms"]) < 1e-4, shape
        assert sh["step"]["valu_bound_ms"] < sh["step"]["valu_salu_bound_ms"] < sh["step"]["kernels_alone_sum_ms"]
    assert {"1920x1080x30_8bit_gpc1", "1920x1080x30_8bit_gpc4"} <= set(m["shapes"])


def test_issue_model_is_regenerable_by_one_command(tmp_path):
    """`python tools/make_issue_model.py` re-derives profiles/r06_issue_model.json from the committed counters, the committed opcode prices and
    the in-tree sources (hipcc cross-compiles the ISA listing here): same bounds as the committed file."""
    import shutil
    import subprocess
    import sys
    committed = json.load(open(os.path.join(REPO, "profiles/r06_issue_model.json")))
    keep = tmp_path / "model.json"
    shutil.copy(os.path.join(REPO, "profiles/r06_issue_model.json"), keep)
    try:
        subprocess.run([sys.executable, os.path.join(REPO, "tools", "make_issue_model.py")], check=True, capture_output=True, timeout=600, cwd=REPO)
        again = json.load(open(os.path.join(REPO, "profiles/r06_issue_model.json")))
    finally:
        shutil.copy(keep, os.path.join(REPO, "profiles/r06_issue_model.json"))
    for shape, sh in committed["shapes"].items():
        assert abs(again["shapes"][shape]["step"]["valu_bound_ms"] - sh["step"]["valu_bound_ms"]) <= 0.01 * sh["step"]["valu_bound_ms"], shape



def test_integration_names_every_entry_point():
    """INTEGRATION.md is the map from the C ABI to the reference: every function include/rbf.h declares appears in it."""
    hdr = open(os.path.join(REPO, "include", "rbf.h"), encoding="utf-8").read()
    doc = open(os.path.join(REPO, "INTEGRATION.md"), encoding="utf-8").read()
    declared = set(re.findall(r"\b(rbf_[a-z0-9_]+)\s*\(", hdr))
    missing = sorted(n for n in declared if n not in doc)
    assert not missing, missing
