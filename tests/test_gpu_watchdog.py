"""Watchdog hygiene (capi_internal.hpp PathState): a spin-watchdog hit demotes a launch mode for a TERM of calls and
the mode is tried again afterwards; two threads that each scan a large haystack on the same device do not starve each other's
launches (order-dependent launches take turns per device) and both get the reference's rows."""
import threading

import numpy as np
import pytest

import coregex_amd as cx

pytestmark = pytest.mark.gpu
PAT = r"\d+\.\d+\.\d+\.\d+"
K_FIELDS_WAVE, K_FIELDS_PERS, K_DELIM, K_FSM = 13, 15, 16, 10


def _kernel_of(rx, buf, n):
    t = cx.Timing()
    rx.find_all_device(buf.ptr, n, timing=t)
    return int(t.kernel), t.kernels


def test_persistent_mode_is_demoted_for_a_term_and_comes_back():
    n = 64 << 20
    buf = cx.DeviceBuffer(n)
    buf.fill_synth(2, 0xC0FFEE02, 0)
    rx = cx.compile(PAT)
    want = rx.find_all_device(buf.ptr, n)
    assert _kernel_of(rx, buf, n)[0] == K_FIELDS_PERS
    assert cx._lib.lib().cxg_debug_demote(0, 1) == 0              # as if the persistent kernel's watchdog had fired
    st = cx.path_state(0)
    term = st["persistent_penalty"]
    assert term >= 8 and st["persistent_hits"] >= 1
    for i in range(term):                                           # the term: the grouped kernel, same rows
        t = cx.Timing()
        assert rx.find_all_device(buf.ptr, n, timing=t) == want
        assert int(t.kernel) == K_FIELDS_WAVE, (i, int(t.kernel))
    assert cx.path_state(0)["persistent_penalty"] == 0
    assert _kernel_of(rx, buf, n)[0] == K_FIELDS_PERS               # ... and the fast mode is back


def test_static_groups_are_demoted_for_a_term_and_come_back(oracle):
    hay = cx.synth_pages(2, 0xC0FFEE02, 7, 256)
    rx = cx.compile(PAT)
    exp = oracle.Regex(PAT).find_all_index(hay)
    assert cx._lib.lib().cxg_debug_demote(0, 0) == 0
    term = cx.path_state(0)["static_penalty"]
    assert term >= 8
    for _ in range(term):                                           # ticket mode: still the oracle's rows
        assert np.array_equal(rx.find_all_index(hay), exp)
    assert cx.path_state(0)["static_penalty"] == 0
    assert np.array_equal(rx.find_all_index(hay), exp)


def test_two_threads_scan_a_gib_each_on_one_device(oracle):
    import torch
    n = 1 << 30
    rx = cx.compile(PAT)
    hits_before = cx.path_state(0)
    res, errs = {}, []
    gate = threading.Barrier(2)

    def work(tid):
        try:
            cx.set_device(0)
            buf = cx.DeviceBuffer(n)
            buf.fill_synth(2, 0xC0FFEE02, tid * (n // 4096))
            cnt = rx.find_all_device(buf.ptr, n)
            out = torch.empty((cnt + 8, 2), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            gate.wait(timeout=300)                                  # only the library's own launches run side by side below (torch's kernels are foreign to it)
            kernels = set()
            for _ in range(12):
                t = cx.Timing()
                assert rx.find_all_device(buf.ptr, n, out.data_ptr(), cnt + 8, timing=t) == cnt
                kernels.add(int(t.kernel))
            gate.wait(timeout=600)
            k = torch.arange(1, cnt + 1, dtype=torch.int64, device="cuda")
            sums = [int((out[:cnt, j] * (k + 7 * j)).sum().item()) & ((1 << 64) - 1) for j in range(2)]
            res[tid] = (cnt, sums, kernels)
        except Exception as e:                                      # noqa: BLE001 (reported by the main thread)
            errs.append((tid, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for tid in range(2):
        ref = oracle.scan_synth(PAT, 2, 0xC0FFEE02, tid * (n // 4096), n // 4096, width=2)
        cnt, sums, kernels = res[tid]
        assert cnt == ref["rows"] and sums == ref["sums"], (tid, cnt, ref["rows"])
        assert kernels <= {K_FIELDS_WAVE, K_FIELDS_PERS}, kernels
    st = cx.path_state(0)
    assert st["persistent_hits"] == hits_before["persistent_hits"] and st["static_hits"] == hits_before["static_hits"], st   # nobody waited for a watchdog
    buf = cx.DeviceBuffer(64 << 20)
    buf.fill_synth(2, 0xC0FFEE02, 0)
    assert _kernel_of(rx, buf, 64 << 20)[0] == K_FIELDS_PERS       # the fast mode is still selected afterwards
