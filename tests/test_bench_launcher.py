"""`python bench.py --gpus N` must start N ranks by itself: without WORLD_SIZE in the
environment the script re-executes under torch.distributed.run, one rank per GPU.  Here on CPU: tests/bench_stub.py — bench.py's
own `main` with a workload that sleeps — over gloo, world size 2: launcher, rendezvous, barriers, max-over-ranks timing, the JSON
line, the per-rank rows and the corpus checksum (independent of the number of shards)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=240, script="tests/bench_stub.py"):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, script)] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_plain_invocation_starts_two_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--settle", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                 # rank 0 prints, nobody else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["rccl_world_size"] == 2 and len(d["config"]["per_rank_kernel_ms"]) == 2
    assert all(ms >= 1.0 for ms in d["config"]["per_rank_kernel_ms"])   # both ranks really stepped
    assert d["data"] == "stub" and d["scaling"] == "weak" and d["higher_is_better"] is True
    # whole-job value: both ranks' bytes over the max-over-ranks time
    assert abs(d["value"] - 2 * (1 << 30) / (d["ms_per_step"] * 1e-3) / 1e9) / d["value"] < 0.01
    # per-rank achieved bandwidth on algorithmic bytes (round 6: an N-GPU line names every rank's GB/s and roofline fraction)
    ach = d["config"]["per_rank_achieved_GBps"]
    assert len(ach) == 2 and len(d["config"]["per_rank_roofline_frac"]) == 2
    for a, ms, rows in zip(ach, d["config"]["per_rank_kernel_ms"], d["config"]["per_rank_rows"]):
        assert abs(a - ((1 << 30) + 16 * rows) / (ms * 1e-3) / 1e9) / a < 0.01
    # per-rank rows and the corpus checksum: the same table split over one rank gives the same line
    assert d["config"]["per_rank_rows"] == [500, 500] and d["config"]["matches_total"] == 1000
    r1 = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--settle", "0"])
    assert r1.returncode == 0, r1.stderr[-2000:]
    d1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    assert d1["config"]["per_rank_rows"] == [1000] and d1["config"]["corpus_checksum"] == d["config"]["corpus_checksum"]
    import numpy as np
    sys.path.insert(0, ROOT)
    from coregex_amd import sharding
    k = np.arange(1000, dtype=np.int64)
    assert d["config"]["corpus_checksum"] == "%016x" % sharding.row_checksum(np.stack([100 * k, 100 * k + 7], axis=1), 0)


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr


def test_more_ranks_than_gpus_is_refused():
    # no GPU in the CPU tier: the real (non-stub) launcher must refuse instead of stacking ranks on one device
    import torch
    if torch.cuda.device_count() >= 2:
        return
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], script="bench.py")
    assert r.returncode != 0 and "one rank per GPU" in r.stderr


def test_pmc_children_are_pinned_to_rank_zeros_device():
    """Round 6: with N > 1 ranks the two rocprofv3 --pmc child passes run rank 0's shard alone on rank 0's device."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.pmc_child_device_env(0, {}) == {"HIP_VISIBLE_DEVICES": "0"}
    assert bench.pmc_child_device_env(3, {}) == {"HIP_VISIBLE_DEVICES": "3"}
    assert bench.pmc_child_device_env(1, {"HIP_VISIBLE_DEVICES": "4,6,7"}) == {"HIP_VISIBLE_DEVICES": "6"}
    assert bench.pmc_child_device_env(0, {"HIP_VISIBLE_DEVICES": "5"}) == {"HIP_VISIBLE_DEVICES": "5"}
