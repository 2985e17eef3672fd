"""GPU tier: k_scan_trio_wave (scan_fields_wave.hip, round 3) against the oracle, through the C ABI.

Programs of the shape run(F) byte(a) run(F) byte(b) run(F) — `(\\w+)@(\\w+)\\.(\\w+)` (BASELINE configs[4]), `\\d+-\\d+:\\d+`,
`[a-c]+x[a-c]+y[a-c]+` — spans and capture rows, bit-exact; the kernel that ran is asserted where the input lies inside its
budgets, anything it hands over must still give the oracle's rows through the fallback ladder."""
import random

import numpy as np
import pytest

import coregex_amd as cx
from routing import routed

pytestmark = pytest.mark.gpu

class _TrioIds:
    """CXG_K_TRIO_WAVE (14: the grouped kernel — FindAll with an n, demoted mode) or CXG_K_TRIO_PERS (18: the same tile mathematics on the
    persistent grid, what a plain call gets since round 5)."""
    def __eq__(self, k):
        return int(k) in (14, 18)
    __hash__ = None


K_TRIO = _TrioIds()
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
        assert routed(want_kernel == t.kernel and t.n_launches == 1, t.kernel, t.n_launches, t.fallback_reason), (pat, t.kernel, t.n_launches, t.fallback_reason)
    if rx.num_groups > 1:
        exps = o.find_all_submatch_index(a)
        subs, ts = _dev(rx, a, True)
        assert subs.shape == exps.shape and np.array_equal(subs, exps), (pat, bytes(a[:60]), subs[:3].tolist(), exps[:3].tolist())
        if want_kernel is not None and a.size:
            assert routed(want_kernel == ts.kernel and ts.n_launches == 1, ts.kernel, ts.n_launches, ts.fallback_reason), (pat, ts.kernel, ts.n_launches, ts.fallback_reason)
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
                                       (r"(\w+)=(\w+)", "ab_1== \n"), (r"(\d+):(\d+)", "xyz   \n,;w01:"), (r"(\d+)-(\d+):(\d+)/(\d+)", "019--::// x"), (r"(\w+)@(\w+)", "ab_9@@ x"),
                                       (r"(\d+)\.(\d+)\.(\d+)\.(\d+)", "0189.. x"), (r"(\w+)@(\w+)@(\w+)", "ab_9@@ x\n"), (r"([a-c]+)-([a-c]+)-([a-c]+)", "abc-- x")])
def test_random_text(oracle, pat, alpha):
    rng = random.Random(len(pat) * 7)
    served = 0
    for it in range(24):
        n = rng.choice([700, 4100, 9000, 40000, 130000, 500000])
        kind = it % 3
        w = ([3, 3, 1] + [1] * len(alpha) if kind == 0 else [1] * len(alpha) if kind == 1 else [1] * (len(alpha) - 3) + [6, 6, 6])[: len(alpha)]
        hay = "".join(rng.choices(alpha, weights=w, k=n)).encode()
        t = _check(oracle, pat, hay, want_kernel=None)
        served += int(t.kernel) in (14, 18, 13, 15) and t.n_launches == 1      # (13: spans of a two-field program with one separator class are the fields kernel's)
    assert routed(served >= 4, served)


def test_one_separator_for_every_link(oracle):
    """`(\\d+)\\.(\\d+)\\.(\\d+)\\.(\\d+)`: the capture rows of the headline pattern.  Two candidates may share up to three runs; the
    matches of a stretch of fields are its fields four at a time from its start.  Spans of this shape are the fields kernel's."""
    pat = r"(\d+)\.(\d+)\.(\d+)\.(\d+)"
    rx, o = cx.compile(pat), oracle.Regex(pat)
    for hay in [b"1.2.3.4", b"1.2.3.4.5.6.7.8.9", b"1.2.3.4.5.6.7.8", b".1.2.3.4.", b"1..2.3.4.5", b"1.2.3 4.5.6.7", b"10.20.30.40.50 1.2.3.4", b"1.2.3", b"",
                b"1.2.3." + b"4.5.6.7." * 50, b"x" * (WT - 9) + b"192.168.100.200 and 10.0.0.1", b"GET / 172.16.254.1 - 8.8.8.8.8\n" * 400]:
        a = _u8(hay)
        exp = o.find_all_submatch_index(a)
        subs, t = _dev(rx, a, True)
        assert subs.shape == exp.shape and np.array_equal(subs, exp), (hay[:40], subs[:3].tolist(), exp[:3].tolist())
        if a.size:
            assert routed(K_TRIO == t.kernel and t.n_launches == 1, t.kernel, t.n_launches, t.fallback_reason), (hay[:40], t.kernel, t.n_launches, t.fallback_reason)
        rows, t = _dev(rx, a, False)
        assert np.array_equal(rows, exp[:, :2]) and (not a.size or routed(t.kernel in (13, 15), t.kernel))
    tok = b"192.168.100.200"
    for off in list(range(50, 70)) + list(range(WT - 20, WT + 70)) + list(range(32 * WT - 20, 32 * WT + 4)):
        h = np.full(33 * WT + 300, ord(" "), dtype=np.uint8)
        h[off:off + len(tok)] = np.frombuffer(tok, dtype=np.uint8)
        subs, t = _dev(rx, h, True)
        assert np.array_equal(subs, o.find_all_submatch_index(h)) and routed(K_TRIO == t.kernel and t.n_launches == 1, t.kernel, t.n_launches, t.fallback_reason), off
    # the headline log, 16 MiB
    import torch
    npages = 4096
    buf = cx.DeviceBuffer(npages * 4096)
    buf.fill_synth(2, 0xC0FFEE02, 0)
    host = cx.synth_pages(2, 0xC0FFEE02, 0, npages)
    exp = o.find_all_submatch_index(host)
    out = torch.empty((len(exp) + 8, 10), dtype=torch.int64, device="cuda")
    t = cx.Timing()
    n = rx.find_all_submatch_device(buf.ptr, npages * 4096, out.data_ptr(), len(exp) + 8, timing=t)
    assert n == len(exp) and n > 100000 and np.array_equal(out[:n].cpu().numpy(), exp)
    assert routed(K_TRIO == t.kernel and t.n_launches == 1, t.kernel, t.n_launches, t.fallback_reason)
    for pat3 in (r"(\d+)\.(\d+)\.(\d+)", r"((\d+)\.(\d+))\.(\d+)\.(\d+)"):      # three fields; a group around two of them (12 slots: 6 pairs, 8 lanes per row)
        rx3, o3 = cx.compile(pat3), oracle.Regex(pat3)
        h = host[:1 << 20]
        subs, t3 = _dev(rx3, h, True)
        assert np.array_equal(subs, o3.find_all_submatch_index(h)) and routed(K_TRIO == t3.kernel and t3.n_launches == 1, t3.kernel, t3.n_launches, t3.fallback_reason), pat3


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
    assert routed(K_TRIO == t.kernel and t.n_launches == 1, t.kernel, t.n_launches, t.fallback_reason)
    n = rx.find_all_submatch_device(buf.ptr, npages * 4096, out.data_ptr(), len(exp) + 8, base=1 << 40, timing=t)
    assert np.array_equal(out[:n].cpu().numpy(), exp + (1 << 40))
    assert rx.find_all_submatch_device(buf.ptr, npages * 4096) == len(exp) and rx.find_all_device(buf.ptr, npages * 4096) == len(exp)
    spans = torch.empty((len(exp) + 8, 2), dtype=torch.int64, device="cuda")
    assert rx.find_all_device(buf.ptr, npages * 4096, spans.data_ptr(), len(exp) + 8, timing=t) == len(exp)
    assert np.array_equal(spans[:len(exp)].cpu().numpy(), exp[:, :2]) and routed(K_TRIO == t.kernel, t.kernel)


def test_long_tokens_and_handover(oracle):
    for n1 in (5, 40, 63, 64, 65, 90, 130):
        for n2 in (1, 30, 64, 70):
            hay = b"x " + b"u" * n1 + b"@" + b"h" * n2 + b".org y bob@site.org"
            _check(oracle, EMAIL, hay, want_kernel=None)
    _check(oracle, EMAIL, b"y" * 3800 + b"@".join([b"ab"] * 300) + b".c", want_kernel=None)          # a super-run past its window: handed over
    _check(oracle, EMAIL, b"a@b." + b"c" * 5000, want_kernel=None)
    _check(oracle, EMAIL, (b"p@q.r " * 30000), want_kernel=None)                     # match-dense: row buffers overflow, dense modes answer
