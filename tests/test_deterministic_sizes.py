"""Host-side contracts of deterministic mode (include/dgmr_hip.h, ABI 11): the scratch the caller must provide is sized by pure host
functions of the library - they must grow when the mode is on, shrink back when it is off, and the weight-gradient plan must announce
enough bias rows for every kernel it may dispatch.  No kernel runs: this is the part of the contract a host without a GPU can check."""
import ctypes

import pytest


@pytest.fixture()
def lib():
    import __graft_entry__ as ge

    ge.build()
    from skillful_nowcasting_amd import _lib

    lib = _lib.load()
    was = lib.dgmr_get_deterministic()
    yield lib
    lib.dgmr_set_deterministic(was)


def test_scratch_sizes_follow_the_mode(lib):
    lib.dgmr_set_deterministic(0)
    assert lib.dgmr_get_deterministic() == 0
    assert lib.dgmr_reduce_doubles(108, 100000, 768) == 108 * 2 * 768
    assert lib.dgmr_wgrad_dot_floats(108) == 108
    assert lib.dgmr_grid_cell_acc_doubles(10**8) == 1
    lib.dgmr_set_deterministic(1)
    assert lib.dgmr_get_deterministic() == 1
    n = 108 * 2 * 768
    got = lib.dgmr_reduce_doubles(108, 100000, 768)
    assert got % n == 0 and 2 * n <= got <= (1 + 512) * n
    assert got * 8 <= (32 << 20) + 2 * n * 8, "the per-workgroup rows must stay below 32 MB"
    # few rows: as many row blocks as the launch has, never fewer than one
    assert lib.dgmr_reduce_doubles(1, 10, 4) == 2 * 2 * 4
    assert lib.dgmr_reduce_doubles(1, 64 * 512 * 4, 48) == (1 + 512) * 2 * 48
    assert lib.dgmr_wgrad_dot_floats(108) == 108 * 1025 and lib.dgmr_wgrad_dot_floats(1) == 1025
    assert 2 <= lib.dgmr_grid_cell_acc_doubles(10**8) <= 1 + 4096
    assert lib.dgmr_grid_cell_acc_doubles(100) == 2


@pytest.mark.parametrize("geom", [(1728, 1, 64, 64, 96, 96, 1, 3, 3, 0, 108), (288, 1, 128, 128, 96, 48, 1, 3, 3, 1, 18), (32, 22, 64, 64, 48, 48, 3, 3, 3, 0, 1),
                                  (96, 1, 8, 8, 384, 384, 1, 3, 3, 0, 6), (1728, 1, 64, 64, 48, 96, 1, 1, 1, 0, 108)])
@pytest.mark.parametrize("precision", [0, 1, 2, 3])
def test_weight_gradient_plan_announces_the_bias_rows(lib, geom, precision):
    from skillful_nowcasting_amd._lib import WgradArgs

    n, d, h, w, cin, cout, kd, kh, kw, up, groups = geom
    was = lib.dgmr_get_precision()
    lib.dgmr_set_precision(precision)
    try:
        wa = WgradArgs()
        wa.N, wa.D, wa.H, wa.W, wa.Cin, wa.Cout, wa.KD, wa.KH, wa.KW = n, d, h, w, cin, cout, kd, kh, kw
        wa.upsample, wa.pre_group, wa.groups = up, 1, groups
        assert lib.dgmr_conv_wgrad_plan(ctypes.byref(wa)) == 0
        assert wa.nsplit >= groups and wa.nsplit % groups == 0
        m = n * d * h * w
        assert wa.bias_rows >= wa.nsplit, "a row per slab for the kernels that write their own"
        assert wa.bias_rows >= max(1, min(1024, m // 512)), "rows of the column-sum pass for the kernels that do not"
    finally:
        lib.dgmr_set_precision(was)
