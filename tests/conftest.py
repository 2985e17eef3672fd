import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The pair kernel for literal sets (scan_teddy_pair.hip) serves haystacks of every length by default; CXG_PAIR_MIN_BYTES is a border for A/B
# runs.  The test tier pins it to 0 whatever the caller's environment says less explicitly; the default routing is asserted in a subprocess
# (tests/test_gpu_teddy_pair.py::test_default_routing).
os.environ.setdefault("CXG_PAIR_MIN_BYTES", "0")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _fresh_path_state(request):
    """GPU tier: launch-mode demotions are process-wide (a watchdog hit of an earlier test — one that ran scans beside foreign kernels or many
    threads — keeps a mode off for a term of calls).  Tests assert kernels and the compact-row entry needs the persistent kernel, so every
    GPU test starts from a clean slate (cxg_path_reset); the demotion logic itself is tested in tests/test_gpu_watchdog.py."""
    if request.node.get_closest_marker("gpu") is not None:
        import coregex_amd as cx
        cx.path_reset(0)
    yield


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with g++."""
    from oracle import oracle as O
    O.build()
    return O


def pytest_sessionfinish(session, exitstatus):
    """Kernel-routing expectations recorded by tests/routing.py are judged when the session ENDS, whatever subset of the tier ran
    (`-k`, a single file or `-x` never reached the last-collected test that used to assert them)."""
    try:
        import routing
    except ImportError:
        return
    if routing.MISSES and exitstatus == 0:
        tr = session.config.pluginmanager.get_plugin("terminalreporter")
        lines = ["ROUTING REGRESSION: a program ran on another kernel (or more launches) than the routing table names:"] + [repr(m) for m in routing.MISSES[:40]]
        if tr is not None:
            for ln in lines:
                tr.write_line(ln, red=True)
        else:
            print("\n".join(lines))
        session.exitstatus = 1
