"""GPU tier: cxg_find_all_device_async / cxg_wait (round 5).  Several launches of one thread in flight on its stream, each with its own
output array: every one returns exactly what the synchronous call returns; a haystack that needs another rung of the ladder (match-dense
input) is finished synchronously inside cxg_wait; programs without an async-capable launch (nullable) complete inside the async call."""
import threading

import numpy as np
import pytest

import coregex_amd as cx

pytestmark = pytest.mark.gpu
IP = r"\d+\.\d+\.\d+\.\d+"


def test_pending_launches_return_the_synchronous_rows(oracle):
    import torch
    n = 64 << 20
    bufs = []
    for i in range(4):
        b = cx.DeviceBuffer(n)
        b.fill_synth(2, 0xC0FFEE02, i * (n // 4096))
        bufs.append(b)
    rx = cx.compile(IP)
    counts = [rx.find_all_device(b.ptr, n) for b in bufs]
    sync = []
    for b, c in zip(bufs, counts):
        o = torch.empty((c + 8, 2), dtype=torch.int64, device="cuda")
        assert rx.find_all_device(b.ptr, n, o.data_ptr(), c + 8, base=7) == c
        sync.append(o[:c].clone())
    outs = [torch.zeros((c + 8, 2), dtype=torch.int64, device="cuda") for c in counts]
    torch.cuda.synchronize()
    pend = [rx.find_all_device_async(b.ptr, n, o.data_ptr(), c + 8, base=7) for b, o, c in zip(bufs, outs, counts)]
    assert cx.path_state(0)["order_waiters"] == 0
    for p, o, c, s in zip(pend, outs, counts, sync):
        t = cx.Timing()
        assert p.wait(t) == c
        assert int(t.kernel) == 15 and t.kernel_ms >= 0                # (kernel_ms: only with CXG_ASYNC_TIMING, a start event per pending launch)
        assert torch.equal(o[:c], s)
    # count-only and a too small output array
    assert rx.find_all_device_async(bufs[0].ptr, n).wait() == counts[0]
    small = torch.empty((16, 2), dtype=torch.int64, device="cuda")
    with pytest.raises(cx.CoregexError) as ei:
        rx.find_all_device_async(bufs[0].ptr, n, small.data_ptr(), 16).wait()
    assert ei.value.code == -3                                       # CXG_E_CAPACITY
    # afterwards the synchronous entry still runs on the fast path (the order slot was given back)
    t = cx.Timing()
    assert rx.find_all_device(bufs[1].ptr, n, timing=t) == counts[1] and int(t.kernel) == 15


def test_a_launch_that_needs_the_ladder_is_finished_in_wait(oracle):
    import torch
    hay = np.frombuffer(b"1.2.3.4 " * 40000 + b"x" * 64, dtype=np.uint8)     # match-dense: the fields kernel's row buffers overflow
    d = torch.from_numpy(hay.copy()).cuda()
    rx = cx.compile(IP)
    exp = oracle.Regex(IP).find_all_index(hay[:-64])
    out = torch.empty((len(exp) + 8, 2), dtype=torch.int64, device="cuda")
    p = rx.find_all_device_async(d.data_ptr(), hay.size - 64, out.data_ptr(), len(exp) + 8)
    assert p.wait() == len(exp)
    assert np.array_equal(out[:len(exp)].cpu().numpy(), exp)


def test_programs_without_an_async_launch_and_the_wrong_thread(oracle):
    import torch
    hay = np.frombuffer(b"xaab aaa b" * 1000 + b"\x00" * 64, dtype=np.uint8)
    d = torch.from_numpy(hay.copy()).cuda()
    for pat in (r"a*", r"[\w]+", r"\berror\b|a+b"):
        rx = cx.compile(pat)
        exp = oracle.Regex(pat).find_all_index(hay[:-64])
        out = torch.empty((len(exp) + 8, 2), dtype=torch.int64, device="cuda")
        assert rx.find_all_device_async(d.data_ptr(), hay.size - 64, out.data_ptr(), len(exp) + 8).wait() == len(exp), pat
        assert np.array_equal(out[:len(exp)].cpu().numpy(), exp), pat
    rx = cx.compile(IP)
    p = rx.find_all_device_async(d.data_ptr(), hay.size - 64)
    res = []
    th = threading.Thread(target=lambda: res.append(pytest.raises(cx.CoregexError, p.wait)))
    th.start(); th.join()
    assert res and res[0].value.code == -9                           # CXG_E_THREAD: the handle belongs to the launching thread (and is left untouched)
    assert p.wait() == len(oracle.Regex(IP).find_all_index(hay[:-64]))   # ... where it still completes


def test_a_pending_call_blocks_nobody_and_a_dropped_handle_is_collected(oracle):
    """round 5 kept the device's launch mutex until cxg_wait — a handle that was never waited for hung every other
    thread.  Round 6 orders launch sections on the device (an event chain): with a call pending on this thread another thread scans
    and gets its rows; a handle that is dropped is waited for by its finaliser; `with` waits on exit."""
    import gc
    import torch
    n = 1 << 22
    buf = cx.DeviceBuffer(n)
    buf.fill_synth(2, 0xC0FFEE02, 0)
    host = cx.synth_pages(2, 0xC0FFEE02, 0, n // 4096)
    rx = cx.compile(IP)
    exp = oracle.Regex(IP).find_all_index(host)
    out = torch.empty((len(exp) + 8, 2), dtype=torch.int64, device="cuda")
    p = rx.find_all_device_async(buf.ptr, n, out.data_ptr(), len(exp) + 8)        # pending, not waited for
    res = []

    def other():
        o2 = torch.empty((len(exp) + 8, 2), dtype=torch.int64, device="cuda")
        res.append((rx.find_all_device(buf.ptr, n, o2.data_ptr(), len(exp) + 8), o2[:len(exp)].cpu().numpy()))

    th = threading.Thread(target=other)
    th.start(); th.join(timeout=60)
    assert not th.is_alive() and res and res[0][0] == len(exp) and np.array_equal(res[0][1], exp)
    del p                                                                          # dropped: the finaliser waits, the slot comes back
    gc.collect()
    for _ in range(40):                                                            # more calls than the thread has slots: none of them leaked
        with rx.find_all_device_async(buf.ptr, n, out.data_ptr(), len(exp) + 8) as q:
            pass
        assert q.rows == len(exp)
    assert np.array_equal(out[:len(exp)].cpu().numpy(), exp)
