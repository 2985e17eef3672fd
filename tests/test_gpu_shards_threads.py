"""Multi-GPU readiness on the 1-GPU box: the node-level sharding of bench.py --gpus 8 — page-aligned byte
ranges, `base` = the shard's offset, rows concatenated in shard order, no collective — driven by 8 host threads on ONE device
(cxg_set_device(0) each, as 8 ranks would call cxg_set_device(rank)).  The 8-shard corpus checksum and row count must equal the
1-shard ones and the oracle's, whatever the interleaving of the 8 callers."""
import threading

import numpy as np
import pytest

import coregex_amd as cx
from coregex_amd import sharding

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,pat", [(2, r"\d+\.\d+\.\d+\.\d+"), (4, r"[\w]+")])
def test_eight_shards_on_one_device_equal_one_shard(oracle, cfg, pat):
    import torch
    total = (2 << 30) if cfg == 2 else (256 << 20)
    world = 8
    seed = 0xC0FFEE00 + cfg
    rx = cx.compile(pat)
    assert sharding.shardable(rx)
    shards = sharding.plan_shards(total, world)
    res, errs = [None] * world, []

    def rank(r):
        try:
            cx.set_device(0)
            lo, hi = shards[r]
            buf = cx.DeviceBuffer(hi - lo)
            buf.fill_synth(cfg, seed, lo // 4096)
            n = rx.find_all_device(buf.ptr, hi - lo)
            out = torch.empty((n + 8, 2), dtype=torch.int64, device="cuda")
            assert rx.find_all_device(buf.ptr, hi - lo, out.data_ptr(), n + 8, base=lo) == n
            res[r] = out[:n]
        except Exception as e:                                      # noqa: BLE001
            errs.append((r, repr(e)))

    th = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    counts = [int(x.shape[0]) for x in res]
    firsts = np.concatenate([[0], np.cumsum(counts)[:-1]])

    def dev_checksum(rows, first):
        k = torch.arange(first + 1, first + rows.shape[0] + 1, dtype=torch.int64, device=rows.device)
        return sum(int((rows[:, j] * (k + 7 * j)).sum().item()) for j in range(2)) & ((1 << 64) - 1)

    sharded = sum(dev_checksum(res[r], int(firsts[r])) for r in range(world)) & ((1 << 64) - 1)
    # one shard: the whole corpus in one call
    buf = cx.DeviceBuffer(total)
    buf.fill_synth(cfg, seed, 0)
    n1 = rx.find_all_device(buf.ptr, total)
    out = torch.empty((n1 + 8, 2), dtype=torch.int64, device="cuda")
    assert rx.find_all_device(buf.ptr, total, out.data_ptr(), n1 + 8) == n1
    assert sum(counts) == n1
    assert sharded == dev_checksum(out[:n1], 0)
    ref = oracle.scan_synth(pat, cfg, seed, 0, total // 4096, width=2)
    assert n1 == ref["rows"]
    k = torch.arange(1, n1 + 1, dtype=torch.int64, device="cuda")
    assert [int((out[:n1, j] * (k + 7 * j)).sum().item()) & ((1 << 64) - 1) for j in range(2)] == ref["sums"]
    # the shards' rows in shard order ARE the whole call's rows
    assert torch.equal(torch.cat(res), out[:n1])
