"""`Q[^Q]*Q` programs on the device (`"[^"]*"`; round 4): the char-class wave kernel's pairs mode against the oracle — rows, counts,
FindAll with an n, compact rows, pairs that span tiles, groups and megabytes, an unpaired last quote."""
import random

import numpy as np
import pytest

import coregex_amd as cx
from routing import routed

pytestmark = pytest.mark.gpu
K_CC = 8


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8)


def _dev(rx, hay, n=-1):
    import torch
    d = torch.from_numpy(np.ascontiguousarray(hay)).cuda() if hay.size else torch.zeros(16, dtype=torch.uint8, device="cuda")
    t = cx.Timing()
    cnt = rx.find_all_device(d.data_ptr(), hay.size, n=n, timing=t)
    out = torch.full((cnt + 4, 2), -7, dtype=torch.int64, device="cuda")
    got = rx.find_all_device(d.data_ptr(), hay.size, out.data_ptr(), cnt, n=n, timing=t)     # cap == count: an unpaired last quote must not write
    assert got == cnt and (out[cnt:] == -7).all()
    return out[:cnt].cpu().numpy(), t


@pytest.mark.parametrize("pat", [r'"[^"]*"', r"'[^']*'", r"\|[^|]*\|"])
def test_rows(pat, oracle):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    q = pat[1] if pat[0] == "\\" else pat[0]
    rng = random.Random(ord(q) + 1)
    alpha = (q + "ab \n,").encode() + "é".encode() + b"\xff"
    hays = [b"", q.encode(), (q + q).encode(), (q + "a" + q + q).encode(), ("x" + q) .encode() * 5]
    for n in [100, 3839, 3840, 3841, 61440, 61441, 200000, 3_000_000]:
        for w in (1, 8):
            hays.append(bytes(rng.choices(alpha, weights=[w] + [12] * (len(alpha) - 1), k=n)))
    hays.append(q.encode() + b"z" * 500_000 + q.encode() + b"  " + q.encode() + b"y" * 70_000)      # a pair over 130 tiles, an unpaired last quote
    for hay in hays:
        a = _u8(hay)
        exp = o.find_all_index(a)
        got, t = _dev(rx, a)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay), hay[:40], got[:4].tolist(), exp[:4].tolist())
        if a.size:
            assert routed(t.kernel == K_CC and t.n_launches == 1, t.kernel, t.n_launches)
        for n in (1, 3):
            gotn, _ = _dev(rx, a, n=n)
            assert np.array_equal(gotn, exp[:n]), (pat, len(hay), n)
        assert np.array_equal(rx.find_all_index(a), exp) and rx.count(a) == len(exp)


def test_compact_rows_and_synthlog(oracle):
    import torch
    pat = r'"[^"]*"'
    rx = cx.compile(pat)
    hay = cx.synth_pages(2, 0xC0FFEE02, 0, 8192)                     # 32 MB of the config-2 corpus: quoted request lines and user agents
    exp = oracle.Regex(pat).find_all_index(hay)
    got, t = _dev(rx, hay)
    assert np.array_equal(got, exp) and len(exp) > 100000
    d = torch.from_numpy(hay).cuda()
    o32 = torch.empty((len(exp) + 4, 2), dtype=torch.int32, device="cuda")
    assert rx.find_all_device_u32(d.data_ptr(), hay.size, o32.data_ptr(), len(exp) + 4) == len(exp)
    assert np.array_equal(o32[:len(exp)].cpu().numpy().astype(np.int64) & 0xFFFFFFFF, exp)


def test_too_many_quotes_in_a_tile_is_an_input_refusal():
    rx = cx.compile(r'"[^"]*"')
    with pytest.raises(cx.UnsupportedInput):
        rx.find_all_index(_u8(b'""' * 4000))
