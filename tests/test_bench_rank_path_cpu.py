"""bench.py's OWN multi-rank code path on two gloo ranks (no GPU): `python bench.py --gpus 2 --standin tests.bench_standin`.

The multi-GPU bench has never run on more than one GPU (the driver's 8-GPU node was not available in any round), so everything in
it that is not a kernel is executed here exactly as the driver would start it: main() re-executes itself under
torch.distributed.run, the ranks form a process group, rank 0 draws the communicator id and the others receive it, all ranks agree
that each of them got a communicator, settle() takes its go-on decision collectively, the timed windows are bracketed by barriers,
the elapsed time is the max over the ranks, the stage pass and the point-only side run execute, and rank 0 alone prints ONE JSON
line.  Device work is played by tests/bench_standin.py (test infrastructure on the oracle); the split of the leftovers is the
library's own ovp_shard_range_of_mask."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None, timeout=420):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env["OMP_NUM_THREADS"] = "1"
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--standin", "tests.bench_standin"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout          # rank 0 alone prints, exactly one line, nothing else on stdout
    return json.loads(lines[0]), p.stderr


def test_two_ranks_through_bench_main_native_collective(tmp_path):
    dump = str(tmp_path / "rank%d.npz")
    line, _ = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"OVP_STANDIN_DUMP": dump})
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["value"] is None and line["data"].startswith("standin")          # nothing measured this way is a result
    assert line["config"]["baseline_config"] == "config4" and line["config"]["planes"] == 3
    mg = line["multi_gpu"]
    assert mg["rccl_ranks"] == 2 and mg["backend"] == "gloo"
    assert mg["collective"].startswith("rccl-native") and mg["collective_note"] is None
    assert set(mg["rank0_stage_ms"]) == {"plane_loop_ms", "points_build_ms", "allreduce_ms", "update_ms"}
    s = mg["serial_fraction"]
    assert 0.0 < s < 1.0
    assert abs(mg["amdahl_bound_speedup"] - 1.0 / (s + (1.0 - s) / 2.0)) < 1e-12
    assert abs(mg["amdahl_bound_speedup_at_8_ranks"] - 1.0 / (s + (1.0 - s) / 8.0)) < 1e-12
    assert mg["one_gpu_same_workload"]["ms_per_step"] > 0 and mg["speedup_vs_one_gpu_same_workload"] > 0   # the line's own strong-scaling reference
    po = mg["point_only_scaling"]
    assert po is not None and po["ms_per_step"] > 0 and po["rank0_point_shard"] == 12      # 24 point features over two ranks
    assert po["one_gpu_ms_per_step"] > 0 and po["speedup_vs_one_gpu"] > 0
    assert mg["rank0_point_shard"] > 0
    assert line["prewarm"]["steps"] >= 4 and line["prewarm"]["steps"] % 2 == 0             # settle(): the ranks stopped together
    # both replicas ended the last config-4 step with the same covariance and correction, bit for bit
    a, b = np.load(dump % 0), np.load(dump % 1)
    assert np.array_equal(a["P"], b["P"]) and np.array_equal(a["dx"], b["dx"])


@pytest.mark.parametrize("fail", ["uid", "pre:1", "comm:1", "comm:0"])
def test_a_rank_that_cannot_create_its_communicator_does_not_hang_the_others(fail):
    """a rank that throws before / inside ncclCommInitRank used to leave its peers blocked.  The id-or-None broadcast and
    the MIN agreement make the failure collective: every rank falls back to torch.distributed and the run completes."""
    line, err = _bench(["--gpus", "2", "--steps", "2", "--warmup", "0", "--workload", "config2"], {"OVP_STANDIN_FAIL": fail}, timeout=300)
    mg = line["multi_gpu"]
    assert mg["collective"] == "torch.distributed" and mg["collective_note"]
    assert "falling back to torch.distributed" in err
    assert mg["rccl_ranks"] == 2 and mg["rank0_point_shard"] == 12


def test_one_rank_sharded_path_through_bench_main():
    """--gpus 1 --sharded-path: the same code path on a process group of one rank (what a one-GPU box can run)."""
    line, _ = _bench(["--gpus", "1", "--steps", "2", "--warmup", "0", "--sharded-path"])
    assert line["n_gpus"] == 1 and line["multi_gpu"]["rccl_ranks"] == 1
    assert line["multi_gpu"]["collective"].startswith("rccl-native")
