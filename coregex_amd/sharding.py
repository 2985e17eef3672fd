"""Byte-range sharding of one corpus over the GPUs of a node.

FindAll over a buffer equals the concatenation of FindAll over pieces cut right after a byte outside
the pattern's alphabet, with offsets rebased — so each rank scans its own shard with the single-GPU
kernel and there is **no collective on the data path**.  The only communication is optional result
collection: `gather_rows` moves the (small) index arrays to rank 0 with `torch.distributed`
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
from __future__ import annotations

import numpy as np

PAGE = 4096  # synthlog pages end in '\n'; shards are cut at page boundaries


def plan_shards(nbytes: int, world: int, page: int = PAGE):
    """Contiguous [lo, hi) byte ranges, page aligned, sizes differing by at most one page."""
    npages = (nbytes + page - 1) // page
    base, extra = divmod(npages, world)
    out, lo = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        hi = min(nbytes, lo + n * page)
        out.append((lo, hi))
        lo = hi
    return out


def shardable(rx) -> bool:
    """May this program be scanned shard by shard and its rows concatenated?  Not when a match depends on the haystack as a WHOLE:
    nullable programs (`a*`: FindAll emits an empty match at the end of every shard and at position 0 of the next, and skips an empty
    match only at the end of a match of the SAME call, meta/findall.go:251-257) and quote-pair programs (`"[^"]*"`: which quote opens
    is the parity of the quotes from the haystack's first byte), and programs with a text anchor (`(^|,)\\d+`, `a$|z`: a shard's first or last
    position is not the text's).  Such programs are scanned as one haystack (64 GiB fit one MI355X)."""
    import struct
    if rx.nullable:
        return False
    img = rx.fsm_image()
    if img is not None and len(img) >= 37 * 4:
        # FsmHeader::rev_text_col / end_col (device/fsm.hpp): the pattern holds \A / ^ resp. \z / $ — only the haystack's own first
        # and last position are the text's
        if struct.unpack_from("<I", img, 29 * 4)[0] or struct.unpack_from("<I", img, 36 * 4)[0]:
            return False
    blob = rx.blob()
    kind, flags = struct.unpack_from("<II", blob, 4)
    if kind == 3 and flags & 64:                                        # kKindCharClass with ranges: CharClassAux.pairs
        aux_off = struct.unpack_from("<I", blob, 14 * 4)[0]
        nr, lo, hi, neg, pairs = struct.unpack_from("<I4s4sII", blob, aux_off)
        if pairs:
            return False
    return True


def cut_is_safe(sync_table: np.ndarray, byte_before_cut: int) -> bool:
    """A shard may start at `cut` iff hay[cut-1] is a sync byte of the program (info table bit 0)."""
    return bool(sync_table[byte_before_cut] & 1)


def sync_table_of(rx) -> np.ndarray:
    """The 256-entry byte-info table of a compiled program (bit 0 = outside the pattern alphabet)."""
    import struct
    blob = rx.blob()
    info_off = struct.unpack_from("<I", blob, 12 * 4)[0]
    return np.frombuffer(blob, dtype=np.uint8, count=256, offset=info_off)


def apply_limit(rows: np.ndarray, n: int) -> np.ndarray:
    """`n > 0` limits are applied after concatenation (global prefix), as FindAllIndex(b, n) would."""
    return rows if n <= 0 else rows[:n]


def gather_rows(local_rows: np.ndarray, dist, device="cpu"):
    """All ranks call; rank 0 receives the concatenation in rank order (already sorted by start)."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    width = local_rows.shape[1]
    cnt = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    mx = max(counts + [1])
    pad = torch.zeros((mx, width), dtype=torch.int64, device=device)
    pad[: local_rows.shape[0]] = torch.from_numpy(np.ascontiguousarray(local_rows)).to(device)
    bufs = [torch.zeros((mx, width), dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(bufs, pad)
    if rank != 0:
        return None
    return np.concatenate([b[:c].cpu().numpy() for b, c in zip(bufs, counts)], axis=0)


def row_checksum(rows: np.ndarray, first_row: int = 0) -> int:
    """Order-sensitive 64-bit checksum of a block of rows that are rows first_row, first_row + 1, ... of the WHOLE corpus
    (0-based; offsets absolute): sum over rows K and columns j of offset x (K + 1 + 7 j), mod 2^64, unset capture slots (-1)
    counting as 0.  Additive over shards: the sum of the shards' checksums, each taken with the number of rows in the shards in
    front of it as first_row, equals the checksum of the concatenation — however the corpus was split (bench.py prints it as
    config.corpus_checksum so that the 1-, 2-, 4- and 8-GPU lines of one corpus can be compared)."""
    r = np.asarray(rows, dtype=np.int64).reshape(len(rows), -1)
    k = (np.arange(first_row + 1, first_row + 1 + len(r), dtype=np.uint64))[:, None] + np.uint64(7) * np.arange(r.shape[1], dtype=np.uint64)[None, :]
    v = np.where(r < 0, 0, r).astype(np.uint64)
    with np.errstate(over="ignore"):
        return int((v * k).sum(dtype=np.uint64))
