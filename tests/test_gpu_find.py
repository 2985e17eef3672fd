"""Round 6: Engine.Find / Engine.IsMatch of a whole haystack on the device (meta/find.go:29, meta/ismatch.go:27 — SURVEY §2.1 "next") through
cxg_find / cxg_is_match and their device-pointer forms: the first row of the oracle's FindAll, for every kernel family; the early stop of
FindAll's n keeps a haystack whose match comes early far below the cost of its length."""
import time

import numpy as np
import pytest

import coregex_amd as cx

pytestmark = pytest.mark.gpu

PATS = [r"\d+\.\d+\.\d+\.\d+", r"error", r"[\w]+", r"fatal|panic|error|warning", r"(?:25[0-5]|2[0-4][0-9]|[01]?[0-9][0-9]?)\.\d+", r"\berror\b", r"a*", r"\d+\.\d+x?",
        r"(?m)^\d+", r"zzzzq"]


def test_first_match_and_is_match_equal_the_oracle(oracle):
    rng = np.random.default_rng(6)
    hays = [cx.synth_pages(1, 0xC0FFEE01, 3, 64).tobytes(), cx.synth_pages(2, 0xC0FFEE02, 9, 64).tobytes(), b"", b"x", b"no digits here " * 5000,
            b"q" * 300000 + b" error 10.20.30.40 zzzzq", bytes(rng.integers(97, 123, size=70000, dtype=np.uint8)) + b" 1.2.3.4"]
    for pat in PATS:
        rx, o = cx.compile(pat), oracle.Regex(pat)
        assert rx.supported, (pat, rx.why_unsupported)
        for hay in hays:
            exp = o.find_all_index(hay)
            want = tuple(int(v) for v in exp[0]) if len(exp) else None
            assert rx.find_index(hay) == want, (pat, len(hay), want)
            assert rx.is_match(hay) == (want is not None), (pat, len(hay))


def test_device_forms_and_the_early_stop(oracle):
    import torch
    n = 1 << 32                                                              # 4 GiB: 35 000 groups, ~1 500 resident at a time
    buf = cx.DeviceBuffer(n)
    buf.fill_synth(2, 0xC0FFEE02, 0)
    head = cx.synth_pages(2, 0xC0FFEE02, 0, 256)
    pat = r"\d+\.\d+\.\d+\.\d+"
    rx = cx.compile(pat)
    first = oracle.Regex(pat).find_all_index(head)[0]
    assert rx.find_device(buf.ptr, n, base=1000) == (int(first[0]) + 1000, int(first[1]) + 1000)
    assert rx.is_match_device(buf.ptr, n)
    none = cx.compile(r"zzzzq")
    assert none.find_device(buf.ptr, n) is None and not none.is_match_device(buf.ptr, n)
    # early stop: the match sits in the first KiB — the call must not cost a scan of 4 GiB (a scan of 4 GiB takes ~0.9 ms of kernel
    # time; the stop word lets every workgroup that starts after the first counted row leave)
    for _ in range(3):
        rx.is_match_device(buf.ptr, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        assert rx.is_match_device(buf.ptr, n)
    early = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        assert not none.is_match_device(buf.ptr, n)
    full = (time.perf_counter() - t0) / 20
    print(f"is_match over 4 GiB: match in the first KiB {early * 1e3:.3f} ms per call, no match (the whole haystack is scanned) {full * 1e3:.3f} ms per call")
    assert early < 0.5 * full                                                # groups that start after the first counted row leave at once
