"""Device-side self checks of arithmetic shortcuts the kernels take (each must agree with the plain form on its whole operand set)."""
import ctypes as C

import pytest

from impact_amd import capi

pytestmark = pytest.mark.gpu


def test_mesher_division_is_the_ieee_division_on_its_operand_set(ctx):
    """role_sn_emit divides decoded distances (t = d1 / (d1 - d2), surface_nets.rs:396-404) and edge counts by a short correctly-rounding
    sequence (sn_roles.hpp div_ranged); the device compares it with the `/` operator over every operand pair it can meet."""
    bad = C.c_uint32(123)
    capi.check(capi.lib().ivx_selftest_mesher_division(ctx.h, C.byref(bad)))
    assert bad.value == 0
