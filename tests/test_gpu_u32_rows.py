"""cxg_find_all_device_u32 (round 4): compact rows — two uint32 relative to the haystack — from the kernels
that have the epilogue: char-class programs (incl. the `\\S+`-style class runs of the DFA strategies) and fields programs on the
persistent kernel.  Rows must equal the int64 entry point's, and the oracle's."""
import numpy as np
import pytest

import coregex_amd as cx

pytestmark = pytest.mark.gpu


def _both(rx, hay):
    import torch
    d = torch.from_numpy(np.ascontiguousarray(hay)).cuda()
    n = rx.find_all_device(d.data_ptr(), hay.size)
    o64 = torch.empty((n + 8, 2), dtype=torch.int64, device="cuda")
    assert rx.find_all_device(d.data_ptr(), hay.size, o64.data_ptr(), n + 8) == n
    assert rx.find_all_device_u32(d.data_ptr(), hay.size) == n
    o32 = torch.full((n + 8, 2), -1, dtype=torch.int32, device="cuda")
    t = cx.Timing()
    assert rx.find_all_device_u32(d.data_ptr(), hay.size, o32.data_ptr(), n + 8, timing=t) == n
    return o64[:n].cpu().numpy(), o32[:n].cpu().numpy().astype(np.int64) & 0xFFFFFFFF, o32[n:].cpu().numpy(), t


@pytest.mark.parametrize("pat,cfg", [(r"[\w]+", 4), (r"\S+", 2), (r"[^,]+", 2), (r"\d+\.\d+\.\d+\.\d+", 2), (r"\d+:\d+:\d+", 2)])
def test_compact_rows_equal_the_int64_rows(pat, cfg, oracle):
    rx = cx.compile(pat)
    hay = cx.synth_pages(cfg, 0xC0FFEE00 + cfg, 3, 5000)            # 20 MB: runs and super-runs cross tiles, groups, units
    r64, r32, tail, t = _both(rx, hay)
    assert np.array_equal(r32, r64) and (tail == -1).all() and t.n_launches == 1
    assert np.array_equal(r64, oracle.Regex(pat).find_all_index(hay))


def test_long_runs_cross_many_tiles(oracle):
    rx = cx.compile(r"[^,]+")
    hay = np.frombuffer((b"x" * 70000 + b"," + b"y" * 5 + b",,") * 40, dtype=np.uint8)
    r64, r32, tail, _ = _both(rx, hay)
    assert np.array_equal(r32, r64) and np.array_equal(r64, oracle.Regex(r"[^,]+").find_all_index(hay))


def test_programs_without_the_compact_epilogue_say_so():
    import torch
    hay = cx.synth_pages(1, 0xC0FFEE01, 0, 64)
    d = torch.from_numpy(hay).cuda()
    out = torch.empty((1 << 16, 2), dtype=torch.int32, device="cuda")
    for pat in (r"rr", r"(\w+)@(\w+)\.(\w+)", r"a*"):            # `rr`: a literal with a border stays on the chain kernel (int64 rows only)
        rx = cx.compile(pat)
        with pytest.raises(cx.CoregexError):
            rx.find_all_device_u32(d.data_ptr(), hay.size, out.data_ptr(), out.shape[0])
        if pat == r"rr":                                             # counting needs no rows (nullable and UseBoth programs are refused outright)
            assert rx.find_all_device_u32(d.data_ptr(), hay.size) == rx.find_all_device(d.data_ptr(), hay.size)
    # `error` runs in the persistent kernel's literal mode since round 5, which has the compact epilogue
    rx = cx.compile(r"error")
    n = rx.find_all_device_u32(d.data_ptr(), hay.size, out.data_ptr(), out.shape[0])
    o64 = torch.empty((n + 8, 2), dtype=torch.int64, device="cuda")
    assert rx.find_all_device(d.data_ptr(), hay.size, o64.data_ptr(), n + 8) == n
    assert torch.equal(out[:n].to(torch.int64) & 0xFFFFFFFF, o64[:n])
