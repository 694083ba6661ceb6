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
