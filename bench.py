#!/usr/bin/env python3
"""bench.py — FindAllIndex throughput of the MI355X path on BASELINE.json's headline workload.

    python bench.py --gpus N --steps K --warmup W          (N>1: one rank per GPU via torch.distributed.run)

Workload (BASELINE.json configs[1]): FindAllIndex of `\\d+\\.\\d+\\.\\d+\\.\\d+` over 1 GiB of synthetic
log lines ("synthlog-v1" config 2) per GPU, corpus resident in HBM before the timed region,
match spans written to HBM as int64 pairs inside it.  A step = one full pass over the rank's shard.
The path shards by byte range with no data-path collective (page-aligned shards are independent,
DESIGN.md "Multi-GPU"), so scaling is weak: value = all ranks' bytes / max-over-ranks time.

Extra objects on the JSON line:
  roofline     — HBM-bound; achieved = algorithmic bytes (N + 16*M) per launch / mean kernel time,
                 the kernel time measured with HIP events on the launch stream inside the library.
  cpu_baseline — rank 0, N=1 only: C++ port of the reference's CPU algorithm for this strategy
                 (oracle/cpu_baseline.cpp), 1 thread, on the same corpus; its spans double as a
                 full-size parity check of the GPU result.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PATTERN = r"\d+\.\d+\.\d+\.\d+"
SYNTH_CONFIG = 2
SEED = 0xC0FFEE02
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # The kernel time of this workload drifts for the first ~30 launches after idle (0.31 -> 0.34 -> 0.30 ms, clock /
    # power management, scripts/time_dist.py): --settle passes run before the W warm-up steps.
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle", type=int, default=40, help="untimed passes before the warm-up steps (clock settling)")
    ap.add_argument("--gib-per-gpu", type=float, default=1.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pattern", default=PATTERN)
    ap.add_argument("--synth-config", type=int, default=SYNTH_CONFIG)
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import coregex_amd as cx
    cx.set_device(local_rank)
    assert cx.device_count() > local_rank, "no MI355X visible to the HIP library"

    rx = cx.compile(args.pattern)
    if not rx.supported:
        raise SystemExit(f"pattern not supported by the device path: {rx.why_unsupported}")
    npages = int(args.gib_per_gpu * (1 << 30)) // 4096
    nbytes = npages * 4096
    buf = cx.DeviceBuffer(nbytes)
    buf.fill_synth(args.synth_config, SEED, rank * npages)          # shard = pages [rank*npages, (rank+1)*npages)
    base = rank * nbytes
    nmatch = rx.find_all_device(buf.ptr, nbytes)                     # sizes the output array
    out = torch.empty((nmatch + 16, 2), dtype=torch.int64, device="cuda")
    stream = 0                                                        # the library's own stream; events are recorded on it

    def step(timing=None):
        n = rx.find_all_device(buf.ptr, nbytes, out.data_ptr(), nmatch + 16, base=base, stream=stream, timing=timing)
        assert n == nmatch or os.environ.get("CXG_DEBUG"), (n, nmatch)
        return n

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Clock settling: the first ~30 launches after idle drift (scripts/time_dist.py); these untimed passes come
    # before the W warm-up steps of the contract so that a short --warmup still times the steady state.
    for _ in range(args.settle):
        step()
    for _ in range(args.warmup):
        step()
    t = cx.Timing()
    kernel_ms = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(t)
        kernel_ms.append(t.kernel_ms)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
        tm = torch.tensor([float(nmatch)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tm, op=dist.ReduceOp.SUM)
        total_matches = int(tm.item())
    else:
        total_matches = nmatch

    ms_per_step = elapsed * 1e3 / args.steps
    total_bytes = nbytes * world
    value = total_bytes / (elapsed / args.steps) / 1e9
    k_ms = float(np.mean(kernel_ms))
    alg_bytes = nbytes + 16 * nmatch                                  # per launch, this rank (DESIGN.md "Roofline")
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9

    result = {
        "metric": "GB/s haystack scanned, FindAllIndex IP-regex",
        "value": round(value, 3),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": f"FindAllIndex `{args.pattern}` over {args.gib_per_gpu:g} GiB/GPU synthlog-v1 config {args.synth_config} "
                        f"(BASELINE.json configs[1]), corpus resident in HBM, int64 span pairs written to HBM",
            "strategy": rx.strategy,
            "bytes_per_gpu": nbytes,
            "matches_total": total_matches,
            "sharding": f"byte-range x{world}, page-aligned, no collective on the data path",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": _pmc_traffic(args, nbytes),
            "kernel": {"1": "k_scan_dfa<digit>", "2": "k_scan_digit_flat", "3": "k_scan_digit_list", "4": "k_scan_digit_chain", "5": "k_scan_digit_wave"}.get(os.environ.get("CXG_DIGIT_KERNEL", "6"), "k_scan_chain_wave<2,false,false,false,4>") if rx.strategy == "UseDigitPrefilter" else rx.strategy,
            "kernel_ms_avg": round(k_ms, 4),
            "algorithmic_bytes_per_launch": alg_bytes,
            "read_only_GBps": round(nbytes / (k_ms * 1e-3) / 1e9, 2),
        },
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline and rx.strategy == "UseDigitPrefilter" and not os.environ.get("CXG_DEBUG"):
        from oracle import oracle as O
        L = O.lib()
        L.orc_baseline_digit_find_all.restype = C.c_int64
        L.orc_baseline_digit_find_all.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        sample_bytes = min(nbytes, 1 << 30)
        host = buf.download(0, sample_bytes)                          # the very bytes the GPU scanned
        orx = O.Regex(args.pattern)
        spans = np.empty(2 * (nmatch + 16), dtype=np.int64)
        c0 = time.perf_counter()
        nv = L.orc_baseline_digit_find_all(orx._h, host.ctypes.data, host.size, spans.ctypes.data, spans.size)
        cpu_s = time.perf_counter() - c0
        cpu_spans = spans[:nv].reshape(-1, 2)
        gpu_spans = out[:nmatch].cpu().numpy() - base
        if sample_bytes == nbytes:
            same = cpu_spans.shape == gpu_spans.shape and bool(np.array_equal(cpu_spans, gpu_spans))
        else:
            k = len(cpu_spans)
            same = bool(np.array_equal(cpu_spans[: k - 1], gpu_spans[: k - 1]))
        if not same:
            raise SystemExit("PARITY FAILURE: GPU spans differ from the CPU port on the benchmark corpus")
        result["cpu_baseline"] = {
            "value": round(sample_bytes / cpu_s / 1e9, 4),
            "unit": "GB/s",
            "cores": 1,
            "kind": "port",
            "sample": f"first {sample_bytes >> 20} MiB of the same corpus (downloaded from HBM), {cpu_s:.1f} s, "
                      f"AVX2 digit scan + flat-table anchored DFA + run skip, g++ -O2 -mavx2; spans equal the GPU's",
            "host_cpu": _cpu_model(),
            "host_threads_available": os.cpu_count(),
        }
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


def _pmc_traffic(args, nbytes):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of this very command
    (profiles/*_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled
    per MI355X_MICROARCH.md "HBM" for wide coalesced reads on gfx950).  None when no profile matches the workload:
    counters cannot be read from inside the timed process."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("pattern") == args.pattern and d.get("synth_config") == args.synth_config and d.get("bytes_per_gpu") == nbytes \
                and d.get("digit_kernel", "6") == os.environ.get("CXG_DIGIT_KERNEL", "6"):
            return d.get("traffic_bytes_per_launch")
    return None


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
