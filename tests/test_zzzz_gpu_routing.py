"""Last test of the GPU tier: every kernel-id / launch-count expectation the parity tests recorded (tests/routing.py).
A miss here is a ROUTING regression (a program served by a slower kernel or by a relaunch ladder), not a wrong row."""
import pytest

import routing

pytestmark = pytest.mark.gpu


def test_every_program_ran_on_the_kernel_the_routing_table_names():
    assert not routing.MISSES, "\n".join(repr(m) for m in routing.MISSES[:40])
