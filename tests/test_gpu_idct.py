"""K4 (dequantise + IDCT + level shift + colour + tiling) through the C ABI vs the oracle."""
import numpy as np
import pytest

import kpeg_testlib as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import libkpeg_amd
    c = libkpeg_amd.Context(0)
    yield c
    c.close()


def _case(ctx, data, mode):
    p = T.oracle_parse(data)
    assert p.status == T.DECODE_DONE
    rc, coef = T.oracle_entropy(p)
    assert rc == 0
    want = T.oracle_idct_colour(coef, p.qt, p.width, p.height)
    ctx.set_idct_mode(mode)
    got = ctx.idct_colour(T.make_frame(p), T.zz_to_natural(coef))
    bad = np.argwhere(got != want)
    assert bad.size == 0, "first mismatches (y,x,c): %s of %d" % (bad[:8].tolist(), len(bad))
    return ctx


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("w,h,q,sigma,smode", [(64, 64, 75, 6.0, 0), (256, 128, 75, 6.0, 0), (264, 72, 50, 12.0, 0),
                                                (128, 64, 95, 0.0, 1), (1920, 1080, 75, 6.0, 0)])
def test_idct_colour_matches_oracle(ctx, w, h, q, sigma, smode, mode):
    _case(ctx, T.synth_jpeg(w, h, seed=7, quality=q, sigma=sigma, mode=smode), mode)


def _coef_case(ctx, coef_nat, qt_zz, w, h, mode=0):
    """coef_nat [nmcu,3,8,8] int16 natural order; qt_zz [2,64] zig-zag quantisers."""
    import libkpeg_amd
    f = libkpeg_amd.Frame()
    f.width, f.height = w, h
    for t in range(2):
        for k in range(64):
            f.qt[t][k] = int(qt_zz[t][k])
    zz = T.zz_table()
    coef_zz = coef_nat.reshape(-1, 3, 64)[..., zz]  # natural -> zig-zag for the oracle
    want = T.oracle_idct_colour(coef_zz, np.vstack([qt_zz, qt_zz]), w, h)
    ctx.set_idct_mode(mode)
    got = ctx.idct_colour(f, coef_nat)
    bad = np.argwhere(got != want)
    assert bad.size == 0, "first mismatches (y,x,c): %s of %d" % (bad[:8].tolist(), len(bad))
    return ctx.timings()


def test_extreme_coefficients_take_the_whole_tile_exact_path(ctx):
    """Huge dequantised values: every error bound exceeds 0.5, every pixel is unsafe, the queue
    overflows and K4 must fall back to evaluating whole tiles in reference order."""
    rng = np.random.default_rng(5)
    w, h = 136, 24
    nmcu = (w // 8) * (h // 8)
    coef = rng.integers(-2000, 2001, size=(nmcu, 3, 8, 8), dtype=np.int16)
    qt = np.full((2, 64), 255, np.uint16)
    t = _coef_case(ctx, coef, qt, w, h)
    assert t["exact_pixels"] >= w * h


def test_structural_ties(ctx):
    """DC + equal and opposite (0,1)/(1,0) terms: the block's diagonal is an exact tie n + 0.5 in real
    arithmetic; only the reference's own rounding order decides it."""
    w, h = 256, 64
    nmcu = (w // 8) * (h // 8)
    rng = np.random.default_rng(6)
    coef = np.zeros((nmcu, 3, 8, 8), np.int16)
    coef[:, :, 0, 0] = rng.integers(-60, 61, size=(nmcu, 3)) * 2 + 1   # odd: F00 = 4 * odd -> F00 / 8 = k + 0.5
    s = rng.integers(1, 4, size=(nmcu, 3))
    coef[:, :, 0, 1] = s
    coef[:, :, 1, 0] = -s
    qt = np.full((2, 64), 9, np.uint16)
    qt[:, 0] = 4
    t = _coef_case(ctx, coef, qt, w, h)
    assert t["exact_pixels"] > 0


def test_dc_only_and_zero_blocks(ctx):
    w, h = 64, 16
    nmcu = (w // 8) * (h // 8)
    coef = np.zeros((nmcu, 3, 8, 8), np.int16)
    coef[:, :, 0, 0] = np.arange(nmcu * 3).reshape(nmcu, 3) * 7 - 150
    coef[3] = 0
    qt = np.full((2, 64), 16, np.uint16)
    qt[1, 0] = 4
    _coef_case(ctx, coef, qt, w, h)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_sparse_blocks(ctx, seed):
    rng = np.random.default_rng(seed)
    w, h = 200, 40
    nmcu = (w // 8) * (h // 8)
    coef = np.zeros((nmcu, 3, 64), np.int16)
    for b in range(nmcu * 3):
        n = rng.integers(0, 12)
        pos = rng.choice(64, size=n, replace=False)
        coef.reshape(-1, 64)[b, pos] = rng.integers(-40, 41, size=n)
    coef[:, :, 0] = rng.integers(-120, 121, size=(nmcu, 3))
    qt = rng.integers(1, 64, size=(2, 64)).astype(np.uint16)
    _coef_case(ctx, coef.reshape(nmcu, 3, 8, 8), qt, w, h)
