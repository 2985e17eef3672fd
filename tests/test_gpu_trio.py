"""GPU tier: k_scan_trio_wave (scan_fields_wave.hip, round 3) against the oracle, through the C ABI.

Programs of the shape run(F) byte(a) run(F) byte(b) run(F) — `(\\w+)@(\\w+)\\.(\\w+)` (BASELINE configs[4]), `\\d+-\\d+:\\d+`,
`[a-c]+x[a-c]+y[a-c]+` — spans and capture rows, bit-exact; the kernel that ran is asserted where the input lies inside its
budgets, anything it hands over must still give the oracle's rows through the fallback ladder."""
import random

import numpy as np
import pytest

import coregex_amd as cx

pytestmark = pytest.mark.gpu

K_TRIO = 14         # CXG_K_TRIO_WAVE
WT = 3840
EMAIL = r"(\w+)@(\w+)\.(\w+)"


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b


def _dev(rx, hay, sub):
    import torch
    a = _u8(hay)
    d = torch.from_numpy(np.concatenate([a, np.zeros(64, dtype=np.uint8)])).cuda()
    t = cx.Timing()
    scan = rx.find_all_submatch_device if sub else rx.find_all_device
    w = 2 * rx.num_groups if sub else 2
    n = scan(d.data_ptr(), a.size, timing=t)
    out = torch.full((n + 8, w), -7, dtype=torch.int64, device="cuda")
    assert scan(d.data_ptr(), a.size, out.data_ptr(), n + 8, timing=t) == n
    return out[:n].cpu().numpy(), t


def _check(oracle, pat, hay, want_kernel=K_TRIO):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    a = _u8(hay)
    exp = o.find_all_index(a)
    rows, t = _dev(rx, a, False)
    assert rows.shape == exp.shape and np.array_equal(rows, exp), (pat, bytes(a[:60]), rows[:4].tolist(), exp[:4].tolist())
    if want_kernel is not None and a.size:
        assert t.kernel == want_kernel and t.n_launches == 1, (pat, t.kernel, t.n_launches, t.fallback_reason)
    if rx.num_groups > 1:
        exps = o.find_all_submatch_index(a)
        subs, ts = _dev(rx, a, True)
        assert subs.shape == exps.shape and np.array_equal(subs, exps), (pat, bytes(a[:60]), subs[:3].tolist(), exps[:3].tolist())
        if want_kernel is not None and a.size:
            assert ts.kernel == want_kernel and ts.n_launches == 1, (pat, ts.kernel, ts.n_launches, ts.fallback_reason)
    return t


def test_edges(oracle):
    for hay in [b"a@b.c", b"a", b"", b"a@b", b"a@b.", b"@b.c", b"a@@b.c", b"a@b..c", b"a@b@c.d", b"a@b.c@d.e", b"a@b.c@d.e@f.g", b"a@b.c@d.e@f.g@h.i x@y.z",
                b"x a@b.c.d y", b"aa@bb.cc dd@ee.ff\ngg@hh.ii", b"a.b@c.d", b"a.b.c@d", b"_9@Z_.q0 ", b"\xe9a@b.c\xff", bytes(range(256)) * 2,
                b"user@example.com, other.user@mail.example.org; x@y", b"a@b.c" * 50, b"ab@cd.ef@" * 40 + b"gh.ij"]:
        _check(oracle, EMAIL, hay)


def test_every_border(oracle):
    tok = b"someone@example.com"
    offs = list(range(40, 70)) + list(range(WT - 24, WT + 70)) + list(range(WT + 170, WT + 200)) + list(range(4 * WT - 22, 4 * WT + 4)) + list(range(32 * WT - 22, 32 * WT + 4))
    hay = np.full(33 * WT + 300, ord(" "), dtype=np.uint8)
    for off in offs:
        h = hay.copy()
        h[off:off + len(tok)] = np.frombuffer(tok, dtype=np.uint8)
        _check(oracle, EMAIL, h)
    line = b"mail from bob@site.org to a_1@b2.c3 (ok) size=12 ................................................................ pad pad pad..\n"   # 128 bytes, 2 rows: 60 rows per wave-tile, inside the row buffers
    assert len(line) == 128
    text = line * 1100
    for n in [1, 63, 64, 65, WT - 1, WT, WT + 1, WT + 63, WT + 64, WT + 65, WT + 191, WT + 192, WT + 193, 4096, 4097, 2 * WT, 32 * WT - 1, 32 * WT, 32 * WT + 1, 32 * WT + 4095]:
        _check(oracle, EMAIL, text[:n])


@pytest.mark.parametrize("pat,alpha", [(EMAIL, "ab_9@@..  x\n"), (r"\d+-\d+:\d+", "0123--:: \n"), (r"([a-c]+)x([a-c]+)y([a-c]+)", "abcxy z"), (r"(\w+)=(\w+);(\w+)", "ab_1==;; \n"), (r"(\d+)/(\d+) (\d+)", "0189// x"),
                                       (r"(\w+)=(\w+)", "ab_1== \n"), (r"(\d+):(\d+)", "xyz   \n,;w01:"), (r"(\d+)-(\d+):(\d+)/(\d+)", "019--::// x"), (r"(\w+)@(\w+)", "ab_9@@ x")])
def test_random_text(oracle, pat, alpha):
    rng = random.Random(len(pat) * 7)
    served = 0
    for it in range(24):
        n = rng.choice([700, 4100, 9000, 40000, 130000, 500000])
        kind = it % 3
        w = ([3, 3, 1] + [1] * len(alpha) if kind == 0 else [1] * len(alpha) if kind == 1 else [1] * (len(alpha) - 3) + [6, 6, 6])[: len(alpha)]
        hay = "".join(rng.choices(alpha, weights=w, k=n)).encode()
        t = _check(oracle, pat, hay, want_kernel=None)
        served += t.kernel in (K_TRIO, 13) and t.n_launches == 1      # (13: spans of a two-field program with one separator class are the fields kernel's)
    assert served >= 4, served


def test_synthlog_16mib(oracle):
    import torch
    rx, o = cx.compile(EMAIL), oracle.Regex(EMAIL)
    npages = 4096
    buf = cx.DeviceBuffer(npages * 4096)
    buf.fill_synth(5, 0xC0FFEE05, 0)
    host = cx.synth_pages(5, 0xC0FFEE05, 0, npages)
    exp = o.find_all_submatch_index(host)
    out = torch.empty((len(exp) + 8, 8), dtype=torch.int64, device="cuda")
    t = cx.Timing()
    n = rx.find_all_submatch_device(buf.ptr, npages * 4096, out.data_ptr(), len(exp) + 8, timing=t)
    assert n == len(exp) and np.array_equal(out[:n].cpu().numpy(), exp)
    assert t.kernel == K_TRIO and t.n_launches == 1
    n = rx.find_all_submatch_device(buf.ptr, npages * 4096, out.data_ptr(), len(exp) + 8, base=1 << 40, timing=t)
    assert np.array_equal(out[:n].cpu().numpy(), exp + (1 << 40))
    assert rx.find_all_submatch_device(buf.ptr, npages * 4096) == len(exp) and rx.find_all_device(buf.ptr, npages * 4096) == len(exp)
    spans = torch.empty((len(exp) + 8, 2), dtype=torch.int64, device="cuda")
    assert rx.find_all_device(buf.ptr, npages * 4096, spans.data_ptr(), len(exp) + 8, timing=t) == len(exp)
    assert np.array_equal(spans[:len(exp)].cpu().numpy(), exp[:, :2]) and t.kernel == K_TRIO


def test_long_tokens_and_handover(oracle):
    for n1 in (5, 40, 63, 64, 65, 90, 130):
        for n2 in (1, 30, 64, 70):
            hay = b"x " + b"u" * n1 + b"@" + b"h" * n2 + b".org y bob@site.org"
            _check(oracle, EMAIL, hay, want_kernel=None)
    _check(oracle, EMAIL, b"y" * 3800 + b"@".join([b"ab"] * 300) + b".c", want_kernel=None)          # a super-run past its window: handed over
    _check(oracle, EMAIL, b"a@b." + b"c" * 5000, want_kernel=None)
    _check(oracle, EMAIL, (b"p@q.r " * 30000), want_kernel=None)                     # match-dense: row buffers overflow, dense modes answer
