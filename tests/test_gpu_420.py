"""The 4:2:0 extension on the GPU: kpeg_frame.components = KPEG_FRAME_420 through the whole-image entry points against
the oracle's restatement (parity unpinned, see tests/test_420.py), bit for bit."""
import io
import os

import numpy as np
import pytest

import kpeg_testlib as T
from test_420 import _photo, encode420

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import libkpeg_amd
    c = libkpeg_amd.Context(0)
    yield c
    c.close()


def _check(ctx, data, dri=False):
    import libkpeg_amd as K
    st, want = T.oracle_decode_420(data)
    assert st == T.DECODE_DONE
    rc, frame, scan = K.host_parse(data, allow_dri=dri, allow_420=True)
    assert rc == K.DECODE_DONE and frame.components == K.FRAME_420
    got = ctx.decode_scan(frame, scan)
    assert got.shape == want.shape
    bad = np.argwhere(got != want)
    assert bad.size == 0, "first mismatches (y,x,c) %s of %d" % (bad[:8].tolist(), len(bad))
    return frame, scan, want


@pytest.mark.parametrize("w,h,q,kw", [(16, 16, 75, {}), (1, 1, 80, {}), (17, 9, 90, {}), (640, 424, 90, {}), (320, 208, 85, {}), (333, 201, 50, {"optimize": True}),
                                      (640, 424, 97, {}), (320, 200, 85, {"restart_marker_blocks": 3}), (640, 400, 75, {"restart_marker_rows": 1})])
def test_photograph_crops_match_the_oracle(ctx, w, h, q, kw):
    """One MCU to 0.27 Mpixel, sparse and dense streams, optimised tables,
    restart intervals (in 16x16 MCUs), sizes that are not multiples of 16."""
    pytest.importorskip("PIL.Image")
    data = encode420(_photo()[:h, :w], quality=q, **kw)
    _check(ctx, data, bool(kw.get("restart_marker_blocks") or kw.get("restart_marker_rows")))


@pytest.mark.parametrize("w,h", [(1920, 1080), (2048, 1536), (4000, 3000)])   # (2048 x 1536: whole MCUs, the pixel kernel writes the caller's buffer, no crop)
def test_large_synthetic_pictures_match_the_oracle(ctx, w, h):
    """Millions of pixels: many K1/K2 workgroups, blocks and MCUs split across sub-sequences and workgroups, flat parts
    (quirk Q1 blocks), saturated colour edges."""
    pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(w + h)
    y, x = np.mgrid[0:h, 0:w]
    px = np.stack([x * 255.0 / (w - 1), y * 255.0 / (h - 1), ((x // 64 + y // 64) % 2) * 255.0], -1)
    px[: h // 2, : w // 2] += rng.normal(0, 12, (h // 2, w // 2, 3))
    px[h // 2:, w // 2:] = (200, 30, 90)       # a flat quarter: DC differences of zero
    _check(ctx, encode420(np.clip(px, 0, 255).astype(np.uint8), quality=80))


@pytest.mark.parametrize("case", ["photo_q92", "noise_q12", "ramps_q60", "flat"])
def test_fast_kernel_against_the_reference_order_kernel(ctx, case):
    """k_idct_colour_fast_420 (idct mode 0, the default) against k_idct_colour_exact_420 (mode 1: every sample in the reference's
    order) and the oracle: photographs (ties in the chroma blocks), black-and-white noise at quality 12 (large coefficients under coarse quantisers: chroma outside the f32
    colour arithmetic's range, blocks whose bound is infinite), ramps (structural ties), a flat picture (no AC term anywhere)."""
    pytest.importorskip("PIL.Image")
    import libkpeg_amd as K
    rng = np.random.default_rng(5)
    if case == "photo_q92":
        data = encode420(_photo()[:424, :640], quality=92)
    elif case == "noise_q12":
        data = encode420((rng.integers(0, 2, (232, 328, 3)) * 255).astype(np.uint8), quality=12)   # black and white: the largest coefficients there are, coarse quantisers
    elif case == "ramps_q60":
        y, x = np.mgrid[0:200, 0:333]
        data = encode420(np.stack([(x * 2) % 256, (y * 3) % 256, (x + y) % 256], -1).astype(np.uint8), quality=60)
    else:
        data = encode420(np.full((64, 80, 3), (200, 30, 90), np.uint8), quality=75)
    st, want = T.oracle_decode_420(data)
    assert st == T.DECODE_DONE
    rc, frame, scan = K.host_parse(data, allow_420=True)
    assert rc == K.DECODE_DONE
    try:
        ctx.set_idct_mode(0)
        fast = ctx.decode_scan(frame, scan)
        settled = ctx.timings()["exact_pixels"]
        ctx.set_idct_mode(1)
        exact = ctx.decode_scan(frame, scan)
    finally:
        ctx.set_idct_mode(0)
    assert np.array_equal(exact, want)
    bad = np.argwhere(fast != want)
    assert bad.size == 0, "first mismatches (y,x,c) %s of %d" % (bad[:8].tolist(), len(bad))
    if case != "flat":
        assert 0 < settled < want.shape[0] * want.shape[1] * (1.01 if case == "noise_q12" else 0.2), settled   # some pixels took the reference-order path, not all of them
    else:
        assert settled == 0


def test_both_sub_sequence_sizes(ctx):
    """4:2:0 streams take the 384-bit sub-sequences by default; the 96-bit kernels are the same code and must agree."""
    pytest.importorskip("PIL.Image")
    data = encode420(_photo()[:424, :640], quality=88)
    try:
        for ss in (96, 384):
            assert ctx.lib.kpeg_hip_debug_set(ctx._h, 4, ss) == 0
            _check(ctx, data)
    finally:
        ctx.lib.kpeg_hip_debug_set(ctx._h, 4, 0)


def test_the_real_references_reject_fixture_decodes(ctx):
    _check(ctx, open(os.path.join(T.GOLDEN, "rej_420.jpg"), "rb").read())


def test_420_then_444_then_gray_on_one_context(ctx):
    import libkpeg_amd as K
    pytest.importorskip("PIL.Image")
    d420 = encode420(_photo()[:100, :150], quality=85)
    d444 = T.synth_jpeg(152, 104, seed=3, sigma=20.0)
    st, w444 = T.oracle_decode(d444)
    p = T.oracle_parse(d444)
    g = open(os.path.join(T.GOLDEN, "gray_ramp_64x48_q75.jpg"), "rb").read()
    st, wg = T.oracle_decode_gray(g)
    rcg, fg, sg = K.host_parse(g, allow_gray=True)
    for _ in range(2):
        _check(ctx, d420)
        assert np.array_equal(ctx.decode_scan(T.make_frame(p), p.scan), w444)
        assert np.array_equal(ctx.decode_scan(fg, sg), wg)


def test_corrupt_420_streams_report_and_recover(ctx):
    import libkpeg_amd as K
    pytest.importorskip("PIL.Image")
    data = encode420(_photo()[:208, :320], quality=85)
    frame, scan, want = _check(ctx, data)
    rng = np.random.default_rng(9)
    failed = 0
    for case in range(40):
        s = scan.copy()
        if case % 2:
            s = s[:int(rng.integers(1, s.size))].copy()
        else:
            for _ in range(4):
                s[int(rng.integers(0, s.size))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        try:
            ctx.decode_scan(frame, s)
        except K.KpegError as e:
            assert e.code == K.E_STREAM, e
            failed += 1
    assert failed > 5
    assert np.array_equal(ctx.decode_scan(frame, scan), want)


def test_batch_entry_points_take_420_and_the_stripe_entry_rejects_it(ctx):
    """Batches decode 4:2:0 pictures one by one (round-robin over the context's lanes; host buffers: picture by picture); a stripe of
    a 4:2:0 picture is not offered."""
    import torch
    import libkpeg_amd as K
    pytest.importorskip("PIL.Image")
    files = [encode420(_photo()[k * 16:k * 16 + 75, k * 8:k * 8 + 101], quality=85) for k in range(4)]
    wants = [T.oracle_decode_420(d)[1] for d in files]
    parsed = [K.host_parse(d, allow_420=True) for d in files]
    frame = parsed[0][1]
    scans = [np.ascontiguousarray(sc) for _, _, sc in parsed]
    d_scans = [torch.from_numpy(sc).cuda() for sc in scans]
    d_rgbs = [torch.zeros((75, 101, 3), dtype=torch.uint8, device="cuda") for _ in files]
    torch.cuda.synchronize()
    ctx.decode_batch_dev(frame, [t.data_ptr() for t in d_scans], [t.numel() for t in d_scans], [t.data_ptr() for t in d_rgbs])
    ctx.sync()
    for k in range(len(files)):
        assert np.array_equal(d_rgbs[k].cpu().numpy(), wants[k]), k
    outs = ctx.decode_batch(frame, scans)
    for k in range(len(files)):
        assert np.array_equal(outs[k], wants[k]), k
    with pytest.raises(K.KpegError) as e:
        ctx.decode_stripe_dev(frame, d_scans[0].data_ptr(), d_scans[0].numel(), 0, 8, d_rgbs[0].data_ptr())
    assert e.value.code == K.E_UNSUPPORTED


def test_cli_allow_420(tmp_path):
    import subprocess
    import libkpeg_amd as K
    pytest.importorskip("PIL.Image")
    data = encode420(_photo()[:201, :333], quality=85)
    dst = tmp_path / "p420.jpg"
    dst.write_bytes(data)
    subprocess.run([K.CLI, str(dst)], cwd=tmp_path, capture_output=True, timeout=120)
    assert not os.path.exists(tmp_path / "p420.ppm")          # the reference's answer: TERMINATE, nothing written
    out = subprocess.run([K.CLI, "--allow-420", str(dst)], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout[-500:] + out.stderr[-500:]
    st, want = T.oracle_decode_420(data)
    assert open(tmp_path / "p420.ppm", "rb").read() == T.ppm_header(333, 201) + want.tobytes()
