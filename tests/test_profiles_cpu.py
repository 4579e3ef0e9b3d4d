"""Profile hygiene: a committed counter file is quoted by bench.py only when it was taken on the running tree's
kernel sources; every r06 profile under profiles/ records the source hash it was taken on."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_bench_refuses_counters_of_another_tree(tmp_path, monkeypatch):
    import bench
    from ov_plane_amd.build import source_tree_hash

    here = source_tree_hash()
    assert len(here) == 16 and here == source_tree_hash()            # deterministic
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "_ROOT", str(tmp_path))
    kern = {"void ovp::k_chol2<15, false>(...)": {"FETCH_SIZE_KB_avg_per_launch": 100.0, "WRITE_SIZE_KB_avg_per_launch": 50.0, "launches": 400}}
    # no file
    b, why = bench.pmc_traffic("k_chol2", "config3", 1)
    assert b is None and "not found" in why
    # a file of another tree: not quoted, and the line says why
    (prof / bench.PMC_TRAFFIC_FILE).write_text(json.dumps({"note": "x", "source_hash": "0123456789abcdef", "kernels": kern}))
    b, why = bench.pmc_traffic("k_chol2", "config3", 1)
    assert b is None and "stale" in why and "0123456789abcdef" in why and here in why
    # a file without a hash (the pre-r06 format) is not quoted either
    (prof / bench.PMC_TRAFFIC_FILE).write_text(json.dumps({"note": "x", "kernels": kern}))
    assert bench.pmc_traffic("k_chol2", "config3", 1)[0] is None
    # the running tree's own file
    (prof / bench.PMC_TRAFFIC_FILE).write_text(json.dumps({"note": "x", "source_hash": here, "kernels": kern}))
    b, why = bench.pmc_traffic("k_chol2", "config3", 1)
    assert b == (2 * 100.0 + 50.0) * 1024.0 and here in why
    # other workloads / rank counts: the passes were taken on the config-3 step of one GPU
    assert bench.pmc_traffic("k_chol2", "config4", 1)[0] is None and bench.pmc_traffic("k_chol2", "config3", 8)[0] is None


def test_source_hash_follows_the_kernel_sources(tmp_path, monkeypatch):
    from ov_plane_amd import build

    h0 = build.source_tree_hash()
    src = os.path.join(build.CSRC, "k_gram.hip")
    txt = open(src).read()
    fake = tmp_path / "csrc"
    fake.mkdir()
    for f in os.listdir(build.CSRC):
        if f.endswith((".hip", ".h")):
            (fake / f).write_text(open(os.path.join(build.CSRC, f)).read())
    (tmp_path / "x").mkdir()
    monkeypatch.setattr(build, "CSRC", str(fake))
    assert build.source_tree_hash() == h0                               # same content elsewhere: same hash
    (fake / "k_gram.hip").write_text(txt + "\n// edit\n")
    assert build.source_tree_hash() != h0


def test_committed_r06_final_profiles_carry_a_source_hash():
    """Whatever r06_final_* set is committed names the tree it was taken on (the file the bench quotes must, the CSVs should)."""
    prof = os.path.join(ROOT, "profiles")
    finals = [f for f in os.listdir(prof) if f.startswith("r06_final_")]
    for f in finals:
        p = os.path.join(prof, f)
        if f.endswith("_pmc.json"):
            assert "source_hash" in json.load(open(p)), f
        elif f.endswith("kernel_stats.csv"):
            assert open(p).readline().startswith("# source_hash "), f
