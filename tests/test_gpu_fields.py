"""GPU tier: scan_fields_wave.hip (round 3 headline kernel) against the oracle, through the C ABI.

Bit-exact rows for the fields programs (`\\d+\\.\\d+\\.\\d+\\.\\d+`, `\\d+:\\d+:\\d+`, `\\d+\\.\\d+`, `a+ba+`) on inputs that put matches
on every lane, word and wave-tile border; the kernel that ran is asserted (cxg_timing.kernel), so a silent fallback to
another kernel cannot pass as coverage.  Inputs the kernel hands over (super-runs past their window, matches longer than
its start search, match-dense tiles) must still give the oracle's rows — through the fallback ladder."""
import random

import numpy as np
import pytest

import coregex_amd as cx
from routing import routed

pytestmark = pytest.mark.gpu

K_FIELDS = 13       # CXG_K_FIELDS_WAVE (the grouped kernel: FindAll with an n, tickets, CXG_NO_PERSIST)
K_PERS = 15         # CXG_K_FIELDS_PERS (round 4: the same mathematics on a persistent grid — what a plain call gets)


def _is(k, want):
    return k == want or (want == K_FIELDS and k == K_PERS)
WT = 3840           # bytes per wave-tile
IP = r"\d+\.\d+\.\d+\.\d+"


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b


def _dev_rows(rx, hay):
    """Rows of one device call over a device-resident copy of `hay` + the timing record of that call."""
    import torch
    a = _u8(hay)
    d = torch.from_numpy(np.concatenate([a, np.zeros(64, dtype=np.uint8)])).cuda()
    t = cx.Timing()
    n = rx.find_all_device(d.data_ptr(), a.size, timing=t)
    out = torch.empty((n + 8, 2), dtype=torch.int64, device="cuda")
    n2 = rx.find_all_device(d.data_ptr(), a.size, out.data_ptr(), n + 8, timing=t)
    assert n2 == n
    return out[:n].cpu().numpy(), t


def _check(oracle, pat, hay, want_kernel=K_FIELDS, launches=1):
    rx = cx.compile(pat)
    exp = oracle.Regex(pat).find_all_index(_u8(hay))
    rows, t = _dev_rows(rx, hay)
    assert rows.shape == exp.shape and np.array_equal(rows, exp), (pat, bytes(_u8(hay)[:60]), rows[:4].tolist(), exp[:4].tolist())
    if want_kernel is not None:
        assert routed(_is(t.kernel, want_kernel) and t.n_launches == launches, t.kernel, t.n_launches, t.fallback_reason), (pat, t.kernel, t.n_launches, t.fallback_reason)
    return t


def test_edges_run_on_the_fields_kernel(oracle):
    for hay in [b"1", b"1.2.3.4", b".1.2.3.4.", b"1.2.3.", b"1..2.3.4", b"a1.2.3.4\n5.6.7.8", b"11..2.3.4.5 999.1.1.1.", b"1.2.3.4.5.6.7.8.9",
                b"x 1.2.3.4.5 y", b"\xb1.\xb2.\xb3.\xb4 1.2.3.4", bytes(range(256)) * 3, b"12.34.56.78." * 30 + b" end", b"9." * 40 + b"9 tail 1.1.1.1"]:
        _check(oracle, IP, hay)


def test_every_border(oracle):
    """One address placed across every offset around the lane-word, wave-tile, window and workgroup borders; haystack ends
    on and around the same borders."""
    ip = b"192.168.100.200"
    offs = list(range(40, 70)) + list(range(WT - 20, WT + 70)) + list(range(WT + 180, WT + 200)) + list(range(4 * WT - 18, 4 * WT + 4)) \
        + list(range(32 * WT - 18, 32 * WT + 4))
    hay = np.full(33 * WT + 300, ord("x"), dtype=np.uint8)
    for off in offs:
        h = hay.copy()
        h[off:off + len(ip)] = np.frombuffer(ip, dtype=np.uint8)
        _check(oracle, IP, h)
    line = b"10.0.0.1 - - [27/Sep/2026] GET /index.html 200 1234 pad pad pad\n"    # 64 bytes: 60 rows per wave-tile, inside the row buffers
    text = line * 2100
    for n in [1, 63, 64, 65, WT - 1, WT, WT + 1, WT + 63, WT + 64, WT + 65, WT + 191, WT + 192, WT + 193, 4096, 4097, 2 * WT, 32 * WT - 1, 32 * WT, 32 * WT + 1, 32 * WT + 4095]:
        _check(oracle, IP, text[:n])


@pytest.mark.parametrize("pat,alpha", [(IP, "0123456789..  x\n"), (r"\d+:\d+:\d+", "0123:: \n"), (r"\d+\.\d+", "01..x"), (r"a+ba+", "aab c"), (r"\d+-\d+-\d+", "0189--/ ")])
def test_random_text(oracle, pat, alpha):
    rng = random.Random(len(pat))
    served = 0
    for it in range(24):
        n = rng.choice([700, 4100, 9000, 40000, 130000, 500000])
        kind = it % 3
        w = ([3, 3, 1] + [1] * len(alpha) if kind == 0 else [1] * len(alpha) if kind == 1 else [5] + [1] * len(alpha))[: len(alpha)]
        hay = "".join(rng.choices(alpha, weights=w, k=n)).encode()
        t = _check(oracle, pat, hay, want_kernel=None)
        served += _is(t.kernel, K_FIELDS) and t.n_launches == 1
    assert routed(served >= 4, served)                      # sparse mixes stay on the kernel; dense ones overflow its row buffers (64 rows per wave-tile)


def test_synthlog_16mib(oracle):
    pat = IP
    rx = cx.compile(pat)
    npages = 4096
    buf = cx.DeviceBuffer(npages * 4096)
    buf.fill_synth(2, 0xC0FFEE02, 0)
    host = cx.synth_pages(2, 0xC0FFEE02, 0, npages)
    exp = oracle.Regex(pat).find_all_index(host)
    import torch
    out = torch.empty((len(exp) + 8, 2), dtype=torch.int64, device="cuda")
    t = cx.Timing()
    n = rx.find_all_device(buf.ptr, npages * 4096, out.data_ptr(), len(exp) + 8, timing=t)
    assert n == len(exp) and np.array_equal(out[:n].cpu().numpy(), exp)
    assert routed(_is(t.kernel, K_FIELDS) and t.n_launches == 1, t.kernel, t.n_launches, t.fallback_reason)
    # shard origin: rows move by `base`
    n = rx.find_all_device(buf.ptr, npages * 4096, out.data_ptr(), len(exp) + 8, base=1 << 40, timing=t)
    assert np.array_equal(out[:n].cpu().numpy(), exp + (1 << 40))
    # count only
    assert rx.find_all_device(buf.ptr, npages * 4096) == len(exp)


def test_long_fields_and_handover(oracle):
    """Fields longer than a lane word (propagate lanes, start search in the previous lane) stay on the kernel while the match is
    at most a word long; longer ones, super-runs past the window and `1.1.1.1...` come back right through the ladder."""
    pat = r"\d+\.\d+"
    for pre in (0, 63, 64, 100):
        for n1 in (30, 63, 64, 65, 100, 128, 200):
            for n2 in (1, 64, 130):
                hay = b"x" * pre + b"5" * n1 + b"." + b"6" * n2 + b" 1.5 y"
                _check(oracle, pat, hay, want_kernel=None)
    _check(oracle, pat, b"x" * 10 + b"5" * 20 + b"." + b"6" * 30 + b" 1.5 y")                 # 51 bytes: on the kernel
    t = _check(oracle, IP, b"y" * 3800 + b"1." * 300 + b"1", want_kernel=None)                  # super-run past its window
    assert t.kernel not in (K_FIELDS, K_PERS) and t.n_launches >= 2
    t = _check(oracle, IP, b"1." * (1 << 16), want_kernel=None)                                  # no synchronising structure at all
    assert t.n_launches >= 2
    _check(oracle, IP, b"y" * 100 + b"1." * 300 + b"1 z")                                       # 600-byte super-run inside one window: 75 groups of four


def test_dense_input(oracle):
    """Rows are buffered per wave and group of eight tiles (512): 96 rows per wave-tile overflow that buffer (reason 0x10) and
    the call reruns on the chain kernel's dense mode, as do 480 — with the oracle's rows either way."""
    _check(oracle, IP, b"10.0.0.1 - some words of padding here..\n" * 12000, want_kernel=None)      # 96 rows per wave-tile
    t = _check(oracle, IP, b"1.2.3.4 " * 60000, want_kernel=None)                    # 480 rows per wave-tile
    # (round 4: on a haystack this short the persistent kernel's units are ONE wave-tile with the whole 512-row buffer — no rerun)
    assert t.n_launches >= 1
    hay = np.frombuffer(b"1.2.3.4 " * (26 << 20), dtype=np.uint8)                    # 208 MiB: full rounds of eight tiles per unit — they overflow
    rx = cx.compile(IP)
    import torch
    d = torch.from_numpy(hay.copy()).cuda()
    t = cx.Timing()
    n = rx.find_all_device(d.data_ptr(), hay.size, timing=t)
    assert n == (26 << 20) and t.n_launches >= 2
    # the rungs of the call (cxg_timing.ladder): the persistent fields kernel first, then whatever took the dense input
    assert len(t.kernels) >= 2 and t.kernels[0] == K_PERS and t.kernels[-1] == t.kernel, t.kernels


@pytest.mark.parametrize("env,kernel", [({"CXG_NO_FIELDS_KERNEL": "1"}, 6), ({"CXG_TICKETS": "1"}, K_FIELDS), ({"CXG_NO_EPOCH": "1"}, K_PERS), ({"CXG_NO_PERSIST": "1"}, K_FIELDS),
                                        ({"CXG_NO_PERSIST": "1", "CXG_NO_EPOCH": "1"}, K_FIELDS), ({}, K_PERS)])
def test_ab_switches(oracle, env, kernel):
    """The A/B switches of the scripts: the chain kernel instead of the fields kernel; tickets instead of static group
    assignment; zeroed look-back words instead of launch epochs.  Rows == oracle in a fresh process for each."""
    import os
    import subprocess
    import sys
    code = ("import numpy as np, torch, zlib, coregex_amd as cx\n"
            "rx = cx.compile(r'\\d+\\.\\d+\\.\\d+\\.\\d+'); t = cx.Timing()\n"
            "h = cx.synth_pages(2, 0xC0FFEE02, 0, 2048)\n"
            "d = torch.from_numpy(h).cuda(); n = rx.find_all_device(d.data_ptr(), h.size, timing=t)\n"
            "out = torch.empty((n + 8, 2), dtype=torch.int64, device='cuda'); n2 = rx.find_all_device(d.data_ptr(), h.size, out.data_ptr(), n + 8, timing=t)\n"
            "print(t.fallback_reason, n, n2, t.kernel, t.n_launches, zlib.crc32(out[:n].cpu().numpy().tobytes()))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, PYTHONPATH=root, CXG_VERBOSE="1", **env), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    print(r.stdout[-300:], r.stderr[-600:])
    n, n2, k, nl, crc = (int(v) for v in r.stdout.split()[-5:])
    import zlib
    exp = oracle.Regex(IP).find_all_index(cx.synth_pages(2, 0xC0FFEE02, 0, 2048))
    assert n == len(exp) and n2 == n and k == kernel and nl == 1 and crc == zlib.crc32(exp.tobytes())


def test_limit_stops_the_scan_early(oracle):
    """FindAllIndex(b, n) with n > 0 (meta/findall.go:196): the first n rows, and — on the fields kernel — without scanning the
    whole haystack: groups dispatched after the n-th row was counted publish nothing (block_common.hpp tile_lookback)."""
    import torch
    npages = (2 << 30) // 4096
    buf = cx.DeviceBuffer(npages * 4096)
    buf.fill_synth(2, 0xC0FFEE02, 0)
    rx = cx.compile(IP)
    head = cx.synth_pages(2, 0xC0FFEE02, 0, 64)
    exp = oracle.Regex(IP).find_all_index(head)
    out = torch.empty((1 << 20, 2), dtype=torch.int64, device="cuda")
    t_full, t_lim = cx.Timing(), cx.Timing()
    full = rx.find_all_device(buf.ptr, npages * 4096, timing=t_full)
    for n in (1, 10, 1000):
        got = rx.find_all_device(buf.ptr, npages * 4096, out.data_ptr(), out.shape[0], n=n, timing=t_lim)
        assert got == n and np.array_equal(out[:n].cpu().numpy(), exp[:n]), n
        assert routed(t_lim.kernel == K_FIELDS, t_lim.kernel)
    import os
    assert full > 1000
    if not os.environ.get("CXG_TICKETS"):             # (with ticket atomics every skipping group still draws its ticket: 73 ns each)
        assert t_lim.kernel_ms < 0.5 * t_full.kernel_ms, (t_lim.kernel_ms, t_full.kernel_ms)
    assert rx.find_all_device(buf.ptr, npages * 4096, n=7) == 7          # Count(b, 7)
