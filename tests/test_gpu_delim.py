"""`O [^E]+ E` / `O [^E]* E` programs on the device (`\\[[^\\]]+\\]`, `<[^>]+>`; round 4): scan_delim_wave.hip against the oracle — rows,
counts, compact rows, rows that span tiles, groups and megabytes, openings without their E, FindAll with an n (the transducer)."""
import random

import numpy as np
import pytest

import coregex_amd as cx
from routing import routed
from test_delim_cpu import DELIM

pytestmark = pytest.mark.gpu
K_DELIM = 16


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8)


def _dev(rx, hay, n=-1):
    import torch
    d = torch.from_numpy(np.ascontiguousarray(hay)).cuda() if hay.size else torch.zeros(16, dtype=torch.uint8, device="cuda")
    t = cx.Timing()
    cnt = rx.find_all_device(d.data_ptr(), hay.size, n=n, timing=t)
    out = torch.full((cnt + 4, 2), -7, dtype=torch.int64, device="cuda")
    got = rx.find_all_device(d.data_ptr(), hay.size, out.data_ptr(), cnt, n=n, timing=t)     # cap == count: an opening without its E must not write
    assert got == cnt and (out[cnt:] == -7).all()
    return out[:cnt].cpu().numpy(), t


@pytest.mark.parametrize("pat,o,c,plus", DELIM)
def test_rows(pat, o, c, plus, oracle):
    rx, orc = cx.compile(pat), oracle.Regex(pat)
    rng = random.Random(ord(o) * 3 + plus)
    alpha = (o + c + "xy \n,").encode() + "é".encode() + b"\xff"
    hays = [b"", o.encode(), c.encode(), (o + c).encode(), (o + "x" + c).encode(), (o + o + c + c).encode(), (o + c + "x" + c).encode()]
    for n in [100, 3839, 3840, 3841, 61440, 61441, 200000, 3_000_000]:
        for wo, wc in ((1, 1), (5, 1), (1, 5), (1, 0)):
            hays.append(bytes(rng.choices(alpha, weights=[wo, wc] + [14] * (len(alpha) - 2), k=n)))
    hays.append(o.encode() + b"z" * 500_000 + c.encode() + b"  " + o.encode() + b"y" * 70_000)      # a row over 130 tiles and 9 groups, an opening without its E
    hays.append(b"q" * 200_000 + c.encode() + o.encode() + b"r" * 130_000 + c.encode())             # groups without an O or E in front of a closing E: PASS kinds
    for hay in hays:
        a = _u8(hay)
        exp = orc.find_all_index(a)
        got, t = _dev(rx, a)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay), hay[:40], got[:4].tolist(), exp[:4].tolist())
        if a.size:
            assert routed(t.kernel == K_DELIM and t.n_launches == 1, t.kernel, t.n_launches)
        try:
            gotn, tn = _dev(rx, a, n=2)                                                            # FindAll with an n: the transducer (early stop)
            assert np.array_equal(gotn, exp[:2]) and (not a.size or tn.kernel != K_DELIM)
        except cx.UnsupportedInput:                                                                # ... which walks a start back over at most 128 KiB
            assert len(exp) and int((exp[:, 1] - exp[:, 0]).max()) > 128 * 1024
        assert np.array_equal(rx.find_all_index(a), exp) and rx.count(a) == len(exp)


def test_synthlog_and_compact_rows(oracle):
    import torch
    pat = r"\[[^\]]+\]"
    rx = cx.compile(pat)
    hay = cx.synth_pages(2, 0xC0FFEE02, 0, 8192)                     # 32 MB: a bracketed timestamp per line
    exp = oracle.Regex(pat).find_all_index(hay)
    got, t = _dev(rx, hay)
    assert np.array_equal(got, exp) and len(exp) > 100000 and t.kernel == K_DELIM
    d = torch.from_numpy(hay).cuda()
    o32 = torch.empty((len(exp) + 4, 2), dtype=torch.int32, device="cuda")
    assert rx.find_all_device_u32(d.data_ptr(), hay.size, o32.data_ptr(), len(exp) + 4) == len(exp)
    assert np.array_equal(o32[:len(exp)].cpu().numpy().astype(np.int64) & 0xFFFFFFFF, exp)


def test_dense_brackets_fall_back_to_the_transducer(oracle):
    pat = r"<[^>]+>"
    rx, orc = cx.compile(pat), oracle.Regex(pat)
    a = _u8(b"<a>" * 40000)                                          # 1280 rows per tile: beyond the 1024-row staging
    got, t = _dev(rx, a)
    assert np.array_equal(got, orc.find_all_index(a)) and t.n_launches >= 2
