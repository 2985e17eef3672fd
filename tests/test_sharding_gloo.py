"""N>1 path on CPU: world-size-2 gloo.  Each rank takes its page-aligned shard of one synthlog corpus,
produces its rows (the device walks compiled for the host stand in for the GPU here), rebases them with
`base`, and rank 0 gathers the concatenation — which must equal the oracle's FindAll over the whole
corpus.  Also checks the cut-safety rule that makes shards independent."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, pat, cfg, npages, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import coregex_amd as cx
    from coregex_amd import sharding
    import emu
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rx = cx.compile(pat)
        nbytes = npages * 4096
        lo, hi = sharding.plan_shards(nbytes, world)[rank]
        shard = cx.synth_pages(cfg, 0xC0FFEE00 + cfg, lo // 4096, (hi - lo) // 4096)
        if lo > 0:
            prev = cx.synth_pages(cfg, 0xC0FFEE00 + cfg, lo // 4096 - 1, 1)[-1]
            assert sharding.cut_is_safe(sharding.sync_table_of(rx), int(prev))
        rows = emu.find_all(rx.blob(), shard) + lo          # rebase: what `base` does on the device
        allrows = sharding.gather_rows(rows, dist)
        # the corpus checksum of bench.py's line: every rank's part with the rows of the ranks in front as its first row index
        import torch
        cnt = torch.zeros(world, dtype=torch.int64)
        cnt[rank] = len(rows)
        dist.all_reduce(cnt)
        part = sharding.row_checksum(rows, int(cnt[:rank].sum()))
        halves = torch.zeros(2 * world, dtype=torch.int64)
        halves[2 * rank], halves[2 * rank + 1] = part & 0xFFFFFFFF, part >> 32
        dist.all_reduce(halves)
        if rank == 0:
            q.put((allrows, sum(int(halves[2 * r]) | (int(halves[2 * r + 1]) << 32) for r in range(world)) & ((1 << 64) - 1)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pat,cfg", [(r"\d+\.\d+\.\d+\.\d+", 2), (r"error", 1)])
def test_two_rank_sharding_equals_whole(oracle, pat, cfg):
    import torch.multiprocessing as mp
    import coregex_amd as cx
    from coregex_amd import sharding
    npages, world = 37, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, pat, cfg, npages, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, checksum = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    whole = cx.synth_pages(cfg, 0xC0FFEE00 + cfg, 0, npages)
    exp = oracle.Regex(pat).find_all_index(whole)
    assert np.array_equal(got, exp)
    assert checksum == sharding.row_checksum(exp, 0)        # sum of the shards' parts == the single-shard checksum of the same corpus
    assert np.array_equal(sharding.apply_limit(got, 5), oracle.Regex(pat).find_all_index(whole, 5))


def test_plan_shards_is_page_aligned_and_complete():
    from coregex_amd import sharding
    for nbytes, world in [(4096 * 10, 3), (4096 * 8, 8), (4096 * 7 + 100, 2), (1 << 30, 8)]:
        sh = sharding.plan_shards(nbytes, world)
        assert sh[0][0] == 0 and sh[-1][1] == nbytes
        for (a, b), (c, d) in zip(sh[:-1], sh[1:]):
            assert b == c and b % 4096 == 0


def test_newline_is_a_sync_byte_for_all_benchmark_patterns():
    import coregex_amd as cx
    from coregex_amd import sharding
    for pat in (r"\d+\.\d+\.\d+\.\d+", r"[\w]+", r"error",
                "error|warning|fatal|critical|panic|timeout|refused|denied|googlebot|bingbot|yandexbot|crawler|spider|failure|exception|overflow"):
        assert sharding.cut_is_safe(sharding.sync_table_of(cx.compile(pat)), ord("\n")), pat


def test_programs_that_must_not_be_sharded_say_so():
    """nullable programs (an empty match at the end of every shard and at position 0 of the next) and quote-pair programs
    (pairing by parity from the haystack's first byte) are whole-haystack programs; everything the benchmarks shard is shardable."""
    import coregex_amd as cx
    from coregex_amd import sharding
    for pat in (r"\d+\.\d+\.\d+\.\d+", r"[\w]+", r"error", r"(\w+)@(\w+)\.(\w+)", r"\[[^\]]+\]"):
        assert sharding.shardable(cx.compile(pat)), pat
    for pat in (r"a*", r"x?y*", r'"[^"]*"'):
        rx = cx.compile(pat)
        assert rx.supported and not sharding.shardable(rx), pat
    # round 6: a text anchor holds at the TEXT's first / last position only — a shard's own ends are not the text's
    for pat in (r"(?:^|,)\d+", r"foo|\Abar", r"a$|z", r"x\z|foo", r"^\s+|\s+$"):
        rx = cx.compile(pat)
        assert rx.supported and not sharding.shardable(rx), pat
    for pat in (r"\berror\b", r"(?m)^\d+", r"(?m)[a-z]+$"):                      # line anchors and word boundaries shard at '\n' like everything else
        assert sharding.shardable(cx.compile(pat)), pat
    # the spurious row the rule prevents: `a*` over `xb|ay` cut behind the b
    import emu
    from twins import rows_on_twin
    rx = cx.compile(r"a*")
    whole = rows_on_twin(rx, b"xbay").tolist()
    parts = rows_on_twin(rx, b"xb").tolist() + (rows_on_twin(rx, b"ay") + 2).tolist()
    assert [2, 2] in parts and [2, 2] not in whole and [2, 3] in whole
