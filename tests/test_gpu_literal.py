"""GPU tier: the persistent kernel's literal mode (scan_fields_wave.hip lit_core, round 5; BASELINE configs[0] `error`) against the
oracle through the C ABI — the kernel that ran is recorded (cxg_timing.kernel == CXG_K_LITERAL_PERS), occurrences on every word,
lane, tile and unit border, literals longer than 32 bytes, count-only calls, FindAll with an n (the chain kernel: its look-back has
the early stop), literals with a border (left to the chain kernel), and the 1 GiB / 8 GiB corpus rows of config 1."""
import random

import numpy as np
import pytest

import coregex_amd as cx
from routing import routed

pytestmark = pytest.mark.gpu
K_LIT, K_CHAIN = 17, 6
WT = 3840


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b


def _dev_rows(rx, hay, n_limit=-1):
    import torch
    a = _u8(hay)
    d = torch.from_numpy(np.concatenate([a, np.zeros(64, dtype=np.uint8)])).cuda()
    t = cx.Timing()
    n = rx.find_all_device(d.data_ptr(), a.size, n=n_limit, timing=t)
    tc = int(t.kernel)
    out = torch.empty((n + 8, 2), dtype=torch.int64, device="cuda")
    assert rx.find_all_device(d.data_ptr(), a.size, out.data_ptr(), n + 8, n=n_limit, timing=t) == n
    return out[:n].cpu().numpy(), t, tc


def test_literals_on_every_border(oracle):
    for lit in ("error", "GET", "ab", "abcd", "xyyyyyyyyyyyyyyyyyyyyyyyyyyyyyyyyyz", "ab" + "b" * 40):
        rx, o = cx.compile(lit), oracle.Regex(lit)
        b = lit.encode()
        for pos in [0, 1, 63, 64 - len(b), 64, 127, 128, WT - len(b), WT - 1, WT, WT + 1, WT + 128 - len(b) // 2, 9 * WT - 2, 9 * WT, 9 * WT + 127, 20 * WT - 3]:
            hay = b"." * pos + b + b"-" * 70 + b + b[:-1] + b"." * 500 + b
            rows, t, tc = _dev_rows(rx, hay)
            exp = o.find_all_index(_u8(hay))
            assert np.array_equal(rows, exp), (lit, pos, rows.tolist(), exp.tolist())
            assert routed(t.kernel == K_LIT and tc == K_LIT and t.n_launches == 1, lit, t.kernel, tc, t.n_launches)


def test_random_text(oracle):
    rng = random.Random(5)
    for lit in ("error", "abc", "aab", "HTTP/"):
        rx, o = cx.compile(lit), oracle.Regex(lit)
        alpha = sorted(set(lit)) + list("x \n")
        for _ in range(12):
            n = rng.choice([5, 700, 4100, 40000, 300000])
            parts = []
            while sum(map(len, parts)) < n:
                parts.append(lit if rng.random() < 0.2 else lit[: rng.randrange(1, len(lit) + 1)] if rng.random() < 0.4 else "".join(rng.choices(alpha, k=rng.randrange(1, 30))))
            hay = "".join(parts)[:n].encode()
            rows, t, _ = _dev_rows(rx, hay)
            assert np.array_equal(rows, o.find_all_index(_u8(hay))), (lit, n)


def test_match_dense_input_takes_the_ladder(oracle):
    hay = b"ab" * 50000                                             # 64 rows per tile are the mode's budget: handed over, rows still the oracle's
    rx = cx.compile("ab")
    rows, t, _ = _dev_rows(rx, hay)
    assert np.array_equal(rows, oracle.Regex("ab").find_all_index(_u8(hay)))
    assert t.kernel != K_LIT or t.n_launches == 1


def test_limit_and_bordered_literals_stay_on_the_chain_kernel(oracle):
    hay = cx.synth_pages(1, 0xC0FFEE01, 0, 256)
    rx, o = cx.compile("error"), oracle.Regex("error")
    rows, t, _ = _dev_rows(rx, hay, n_limit=5)
    assert np.array_equal(rows, o.find_all_index(hay, 5)) and routed(t.kernel == K_CHAIN, t.kernel)
    for lit in ("abab", "aa", "denied"):
        h2 = (lit * 3 + " x " + lit + lit[:2]).encode() * 40
        rows, t, _ = _dev_rows(cx.compile(lit), h2)
        assert np.array_equal(rows, oracle.Regex(lit).find_all_index(_u8(h2))), lit
        assert routed(t.kernel != K_LIT, lit, t.kernel)


def test_config1_corpus_rows_and_count(oracle):
    import torch
    n = 256 << 20
    buf = cx.DeviceBuffer(n)
    buf.fill_synth(1, 0xC0FFEE01, 0)
    rx = cx.compile("error")
    t = cx.Timing()
    cnt = rx.find_all_device(buf.ptr, n, timing=t)
    assert routed(t.kernel == K_LIT, t.kernel)
    out = torch.empty((cnt + 8, 2), dtype=torch.int64, device="cuda")
    assert rx.find_all_device(buf.ptr, n, out.data_ptr(), cnt + 8, base=1 << 40, timing=t) == cnt
    assert routed(t.kernel == K_LIT and t.n_launches == 1, t.kernel, t.n_launches)
    ref = oracle.scan_synth("error", 1, 0xC0FFEE01, 0, n // 4096, width=2)
    k = torch.arange(1, cnt + 1, dtype=torch.int64, device="cuda")
    sums = [int(((out[:cnt, j] - (1 << 40)) * (k + 7 * j)).sum().item()) & ((1 << 64) - 1) for j in range(2)]
    assert cnt == ref["rows"] and sums == ref["sums"]
