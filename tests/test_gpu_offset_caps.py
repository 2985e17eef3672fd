"""Offset captures on the device (round 4): FindAllSubmatch = FindAll (whatever kernel serves the spans: class runs, quote pairs,
delimiters, literal + DFA, transducer) + the expansion kernel of capi_nullable.hip scanOffsetCaps, against the oracle's capture rows."""
import random

import numpy as np
import pytest

import coregex_amd as cx

pytestmark = pytest.mark.gpu


def _u8(b):
    return np.frombuffer(bytes(b), dtype=np.uint8)


@pytest.mark.parametrize("pat", [r"user=(\S+)", r'"([^"]*)"', r"\[([^\]]+)\]", r"<(\w+)>", r"(\S+)", r"k=(\d+);", r"(ab)(cd)e+", r"e+(ab)(cd)", r"((a)b)"])
def test_rows(pat, oracle):
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.offset_captures is not None and rx.submatch_supported
    rng = random.Random(len(pat) + 5)
    alpha = b'user= "ab[]<>xy@.k1;cde\n'
    hays = [b"", b"user=x", b'""', b"[a]"]
    for n in [50, 3900, 70000, 1_000_000]:
        hays.append(bytes(rng.choices(alpha, k=n)))
    for hay in hays:
        a = _u8(hay)
        exp = o.find_all_submatch_index(a)
        got = rx.find_all_submatch_index(a)
        assert got.shape == exp.shape and np.array_equal(got, exp), (pat, len(hay), got[:3].tolist(), exp[:3].tolist())
        assert np.array_equal(rx.find_all_submatch_index(a, 3), exp[:3])


def test_synthlog_user_and_quotes(oracle):
    import torch
    hay = cx.synth_pages(2, 0xC0FFEE02, 0, 4096)
    d = torch.from_numpy(hay).cuda()
    for pat in (r"user=(\S+)", r'"([^"]*)"', r"\[([^\]]+)\]"):
        rx = cx.compile(pat)
        exp = oracle.Regex(pat).find_all_submatch_index(hay)
        n = rx.find_all_submatch_device(d.data_ptr(), hay.size)
        assert n == len(exp)
        out = torch.empty((n + 4, 4), dtype=torch.int64, device="cuda")
        t = cx.Timing()
        assert rx.find_all_submatch_device(d.data_ptr(), hay.size, out.data_ptr(), n + 4, base=7 << 20, timing=t) == n
        assert np.array_equal(out[:n].cpu().numpy(), exp + (7 << 20)), pat
        small = torch.empty((10, 4), dtype=torch.int64, device="cuda")
        with pytest.raises(cx.CoregexError):
            rx.find_all_submatch_device(d.data_ptr(), hay.size, small.data_ptr(), 10)


def test_use_both_programs_keep_the_pikevm_spans(oracle):
    """Found by the device fuzz (round 4): FindAllIndex of a UseBoth program restarts inside matches longer than 100 bytes
    (find_indices.go:425-431), FindAllSubmatch is the PikeVM over the whole haystack — the spans differ, so such a program gets no
    offset captures although its group sits at a fixed distance from the end."""
    pat = r"[\d.]+[x-z]+.*(a|b)"
    rx, o = cx.compile(pat), oracle.Regex(pat)
    assert rx.strategy == "UseBoth" and rx.offset_captures is None
    hay = _u8((b"12.5xyz" + b"q" * 300 + b"a \n" + b"7.z" + b"c" * 40 + b"b\n") * 200)
    exp = o.find_all_submatch_index(hay)
    if rx.submatch_supported:
        try:
            got = rx.find_all_submatch_index(hay)
        except cx.UnsupportedInput:
            return
        assert np.array_equal(got, exp)
