"""Literals between assertions on the device (round 4): `k_scan_teddy_wave` / `k_scan_teddy_pair` (round 6) with the assertions in their verification, against the
oracle; the transducer as its fallback (dense haystacks)."""
import random

import numpy as np
import pytest

import coregex_amd as cx
from routing import routed
from test_wrapped_cpu import FOLDED, FTOKS, TOKS, WRAPPED

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pat", WRAPPED)
def test_rows(pat, oracle):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    rng = random.Random(len(pat) * 3 + 1)
    served = 0
    for n in [0, 1, 7, 3839, 3840, 3841, 61441, 500000, 3_000_000]:
        for sparse in (False, True):
            toks = TOKS + ([b" pad pad pad pad pad pad pad pad "] * 12 if sparse else [])
            hay = np.frombuffer(b"".join(rng.choice(toks) for _ in range(max(1, n // 3)))[:n], dtype=np.uint8)
            exp = o.find_all_index(hay)
            t = cx.Timing()
            if hay.size:
                import torch
                d = torch.from_numpy(hay.copy()).cuda()
                cnt = rx.find_all_device(d.data_ptr(), hay.size, timing=t)
                assert cnt == len(exp), (pat, n, sparse)
                served += t.kernels[0] in (7, 21)                   # the literal kernels: scan_teddy_wave.hip, scan_teddy_pair.hip (round 6)
            got = rx.find_all_index(hay)
            if got.shape != exp.shape or not np.array_equal(got, exp):                # (what differs, for the log)
                miss = sorted(set(map(tuple, exp.tolist())) - set(map(tuple, got.tolist())))[:4]
                extra = sorted(set(map(tuple, got.tolist())) - set(map(tuple, exp.tolist())))[:4]
                raise AssertionError((pat, n, sparse, len(got), len(exp), "missing", miss, "extra", extra, [bytes(hay[max(0, m[0] - 6):m[1] + 6]) for m in miss]))
            assert np.array_equal(rx.find_all_index(hay, 2), exp[:2])
    assert routed(served >= 8, served)


def test_error_on_the_config_1_corpus(oracle):
    import torch
    hay = cx.synth_pages(1, 0xC0FFEE01, 0, 8192)
    d = torch.from_numpy(hay).cuda()
    for pat in (r"\berror\b", r"(?m)^(GET|POST|PUT|DELETE|PATCH)"):
        rx = cx.compile(pat)
        exp = oracle.Regex(pat).find_all_index(hay)
        out = torch.empty((len(exp) + 4, 2), dtype=torch.int64, device="cuda")
        t = cx.Timing()
        assert rx.find_all_device(d.data_ptr(), hay.size, out.data_ptr(), len(exp) + 4, timing=t) == len(exp)
        assert np.array_equal(out[:len(exp)].cpu().numpy(), exp) and routed(t.kernels in ([7], [21]), t.kernels)


@pytest.mark.parametrize("pat", FOLDED)
def test_case_insensitive_alternations(pat, oracle):
    """`(?i)(error|fail|exception|panic|fatal)`: UseNFA in the reference (its PikeVM), one folded literal set on the literal kernel
    (walk.hpp kTeddyFold); 6.9 ms per GiB on the transducer before."""
    import torch
    rx, o = cx.compile(pat), oracle.Regex(pat)
    rng = random.Random(len(pat) * 7 + 2)
    served = 0
    for n in [1, 7, 3839, 3840, 3841, 61441, 500000, 1_500_000]:
        for dense in (False, True):
            parts, have = [], 0
            while have < n:
                parts.append(rng.choice(FTOKS) if rng.random() < (0.5 if dense else 0.05) else bytes(rng.choices(b"abcdefghijklmnopqrstuvwxyzEORFJS  \n:_0", k=rng.randint(1, 12))))
                have += len(parts[-1])
            hay = np.frombuffer(b"".join(parts)[:n], dtype=np.uint8)
            exp = o.find_all_index(hay)
            d = torch.from_numpy(hay.copy()).cuda()
            t = cx.Timing()
            try:
                cnt = rx.find_all_device(d.data_ptr(), hay.size, timing=t)
            except cx.UnsupportedInput:
                assert dense and rx.fsm_image() is None                # the literal kernel gave up and the pattern has no transducer
                continue
            assert cnt == len(exp), (pat, n, dense)
            served += t.kernels[0] in (7, 21)                   # the literal kernels: scan_teddy_wave.hip, scan_teddy_pair.hip (round 6)
            got = rx.find_all_index(hay)
            assert got.shape == exp.shape and np.array_equal(got, exp), (pat, n, dense, bytes(hay[:60]), got[:4].tolist(), exp[:4].tolist())
            assert np.array_equal(rx.find_all_index(hay, 3), exp[:3])
            if rx.submatch_supported:
                es = o.find_all_submatch_index(hay)
                gs = rx.find_all_submatch_index(hay)
                assert gs.shape == es.shape and np.array_equal(gs, es), (pat, n, dense, "submatch")
    assert routed(served >= 8, pat, served)


def test_case_insensitive_keywords_on_the_log_corpus(oracle):
    import torch
    hay = cx.synth_pages(2, 0xC0FFEE02, 0, 8192)
    d = torch.from_numpy(hay).cuda()
    pat = r"(?i)(error|fail|exception|panic|fatal)"
    rx = cx.compile(pat)
    exp = oracle.Regex(pat).find_all_index(hay)
    out = torch.empty((len(exp) + 4, 2), dtype=torch.int64, device="cuda")
    t = cx.Timing()
    assert rx.find_all_device(d.data_ptr(), hay.size, out.data_ptr(), len(exp) + 4, timing=t) == len(exp) and len(exp) > 1000
    assert np.array_equal(out[:len(exp)].cpu().numpy(), exp) and routed(t.kernels in ([7], [21]), t.kernels)
