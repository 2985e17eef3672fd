"""Alphabet-run programs on the device (round 4): scan_runs_wave.hip against the oracle — rows, counts, compact rows, FindAll with an
n, runs across tiles and groups, the give-ups (long runs, crowded tiles, many rows in one run) that hand the haystack to the
transducer, and the log corpus."""
import random

import numpy as np
import pytest

import coregex_amd as cx
import emu
from routing import routed
from test_runs_cpu import README_IP, RUNS

pytestmark = pytest.mark.gpu
K_RUNS, K_FSM = 17, 10


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8)


def _dev(rx, hay, n=-1):
    import torch
    d = torch.from_numpy(np.ascontiguousarray(hay)).cuda() if hay.size else torch.zeros(16, dtype=torch.uint8, device="cuda")
    cnt = rx.find_all_device(d.data_ptr(), hay.size, n=n)
    t = cx.Timing()
    out = torch.full((cnt + 4, 2), -7, dtype=torch.int64, device="cuda")
    got = rx.find_all_device(d.data_ptr(), hay.size, out.data_ptr(), cnt + 4, n=n, timing=t)
    assert got == cnt and (out[cnt:] == -7).all()
    return out[:cnt].cpu().numpy(), t


@pytest.mark.parametrize("pat,alpha", RUNS)
def test_rows(pat, alpha, oracle):
    rx, orc = cx.compile(pat), oracle.Regex(pat)
    assert rx.runs_image() is not None
    rng = random.Random(len(pat) * 5)
    hays = [b"1", b"1.1.1.1", b"1.1.1.1.1.1.1.1", b"256.1.1.1", b"1234.1.1.1", b"999", b"1.5x", b"ab", b"12:30\n1:2", b"0x1f"]
    for n in [100, 3839, 3840, 3841, 4096, 61440, 61441, 200000, 2_000_000]:
        for heavy in (1, 3):
            w = [heavy] * (len(alpha) - 3) + [2, 2, 2]
            hays.append(bytes(rng.choices(alpha, weights=w, k=n)))
    img = rx.runs_image()
    for hay in hays:
        a = _u8(hay)
        exp = orc.find_all_index(a)
        rx = cx.compile(pat)                                          # (a give-up is remembered per program)
        got, t = _dev(rx, a)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay), hay[:40], got[:4].tolist(), exp[:4].tolist())
        if not isinstance(emu.find_all_runs(img, a), int):            # the twin knows the kernel's give-ups
            routed(t.kernels == [K_RUNS], pat, len(hay), t.kernels)
        gotn, _ = _dev(rx, a, n=3)
        assert np.array_equal(gotn, exp[:3]), (pat, len(hay), "n=3")
        assert rx.count(a) == len(exp)


def test_runs_at_tile_and_group_edges(oracle):
    rx, orc = cx.compile(README_IP), oracle.Regex(README_IP)
    for edge in (3840, 7680, 61440, 122880):                          # tile, tile, group, group
        for back in range(0, 16):
            hay = bytearray(b" " * (edge + 400))
            ip = b"192.168.100.200"
            hay[edge - back:edge - back + len(ip)] = ip               # starts `back` bytes in front of the edge
            hay[edge + 200:edge + 207] = b"1.2.3.4"
            a = _u8(hay)
            got, t = _dev(rx, a)
            assert np.array_equal(got, orc.find_all_index(a)) and len(got) == 2, (edge, back, got.tolist())
    tail = _u8(b"x" * 5000 + b"10.0.0.1")                            # a match that ends with the haystack
    assert np.array_equal(_dev(rx, tail)[0], orc.find_all_index(tail))
    head = _u8(b"10.0.0.1" + b"x" * 5000)
    assert np.array_equal(_dev(rx, head)[0], orc.find_all_index(head))


def test_text_start_and_word_context(oracle):
    for pat in (r"\b[0-9]{3}\b", r"(?m)^\d+:\d+", r"x\d+\b"):
        rx, orc = cx.compile(pat), oracle.Regex(pat)
        for hay in (b"123 a123 123", b"12:34\n56:78 9:9\n1:1", b"x12 x3_ x45", b"123", b"1:2"):
            for shift in (0, 3839, 3840, 61440):
                a = _u8(b" " * shift + hay) if shift else _u8(hay)
                got, t = _dev(rx, a)
                assert np.array_equal(got, orc.find_all_index(a)), (pat, hay, shift, got.tolist())


def test_give_ups_take_the_transducer(oracle):
    rx, orc = cx.compile(README_IP), oracle.Regex(README_IP)
    cases = [b" " * 100 + b"1" * 300 + b".1.1.1 " + b"7.7.7.7" + b" " * 5000,       # a run of 300 digits
             b"1.1.1.1 " * 6000,                                                      # 480 addresses per tile
             (b" " + b"1.1.1.1." * 8 + b" ") * 50]                                   # eight rows in one run
    for hay in cases:
        rx = cx.compile(README_IP)                                                    # (a give-up is remembered per program)
        a = _u8(hay)
        got, t = _dev(rx, a)
        assert np.array_equal(got, orc.find_all_index(a)), (hay[:40], got[:4].tolist())
        routed(t.kernels[-1] == K_FSM, hay[:20], t.kernels)
        got2, t2 = _dev(rx, a)                                                        # remembered: the transducer at once
        assert np.array_equal(got2, got)
        routed(K_RUNS not in t2.kernels, t2.kernels)


def test_log_corpus_and_compact_rows(oracle):
    import torch
    rx, orc = cx.compile(README_IP), oracle.Regex(README_IP)
    hay = cx.synth_pages(2, 0xC0FFEE02, 0, 8192)                     # 32 MB of log text: an address in two lines of three
    exp = orc.find_all_index(hay)
    got, t = _dev(rx, hay)
    assert np.array_equal(got, exp) and len(exp) > 100000
    routed(t.kernels[0] == K_RUNS and len(t.kernels) == 1, t.kernels)
    d = torch.from_numpy(hay).cuda()
    o32 = torch.empty((len(exp) + 4, 2), dtype=torch.int32, device="cuda")
    assert rx.find_all_device_u32(d.data_ptr(), hay.size, o32.data_ptr(), len(exp) + 4) == len(exp)
    assert np.array_equal(o32[:len(exp)].cpu().numpy().astype(np.int64) & 0xFFFFFFFF, exp)
