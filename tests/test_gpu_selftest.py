"""The wave-level primitives under every search loop of the path (csrc/qn_device.cuh: DPP prefix sums and running maxima, DPP reductions + v_readlane, the slot -> segment map of a
candidate chunk, the f64 wave sum whose bits the reproducibility tests rely on) against plain restatements on the device: qn_debug_selftest counts the lanes that disagree."""
import ctypes as C
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_wave_primitives_agree_with_their_restatements(seed):
    from qn_amd import engine
    ctx = engine.Context(4096)
    bad = C.c_uint32(0xffffffff)
    ctx._l.qn_debug_selftest.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    ctx.check(ctx._l.qn_debug_selftest(ctx.h, C.c_uint32(2048), C.c_uint32(seed), C.byref(bad)))
    assert bad.value == 0
    ctx.close()
