"""On-device entropy decode (K0..K2) and the whole seam (scan bytes -> RGB) vs the oracle."""
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


CASES = [(64, 64, 75, 6.0, 0), (8, 8, 75, 6.0, 0), (16, 8, 90, 6.0, 0), (256, 128, 75, 6.0, 0), (264, 72, 50, 12.0, 0),
         (128, 64, 95, 0.0, 1), (512, 512, 30, 3.0, 0), (1920, 1080, 75, 6.0, 0)]


@pytest.mark.parametrize("w,h,q,sigma,smode", CASES)
def test_entropy_decode_matches_oracle(ctx, w, h, q, sigma, smode):
    import torch
    data = T.synth_jpeg(w, h, seed=11, quality=q, sigma=sigma, mode=smode)
    p = T.oracle_parse(data)
    rc, coef = T.oracle_entropy(p)
    assert rc == 0
    want = T.zz_to_natural(coef)
    d_scan = torch.frombuffer(bytearray(p.scan), dtype=torch.uint8).cuda()
    d_coef = torch.full((want.size,), 0x5555, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx.entropy_decode_dev(T.make_frame(p), d_scan.data_ptr(), len(p.scan), d_coef.data_ptr())
    ctx.sync()
    got = d_coef.cpu().numpy().reshape(want.shape)
    bad = np.argwhere(got != want)
    assert bad.size == 0, "first mismatches (mcu,c,u,v): %s of %d; rounds=%s" % (bad[:8].tolist(), len(bad), ctx.timings())


@pytest.mark.parametrize("w,h,q,sigma,smode", CASES)
def test_decode_scan_matches_oracle(ctx, w, h, q, sigma, smode):
    data = T.synth_jpeg(w, h, seed=5, quality=q, sigma=sigma, mode=smode)
    st, want = T.oracle_decode(data)
    assert st == T.DECODE_DONE
    p = T.oracle_parse(data)
    ctx.set_idct_mode(0)
    got = ctx.decode_scan(T.make_frame(p), p.scan)
    bad = np.argwhere(got != want)
    assert bad.size == 0, "first mismatches (y,x,c): %s of %d" % (bad[:8].tolist(), len(bad))


@pytest.mark.parametrize("w,h,interval", [(64, 64, 8), (256, 64, 32), (128, 128, 5), (1920, 1080, 240)])
def test_decode_restart_intervals(ctx, w, h, interval):
    data = T.synth_jpeg(w, h, seed=3, quality=75, restart_interval=interval)
    want, p, _ = T.oracle_decode_rst(data, interval)
    got = ctx.decode_scan(T.make_frame(p, interval), p.scan)
    bad = np.argwhere(got != want)
    assert bad.size == 0, "first mismatches (y,x,c): %s of %d" % (bad[:8].tolist(), len(bad))
