"""The one-component (grayscale) extension on the GPU: kpeg_frame.components = 1 through the C ABI against the
oracle's restatement (parity unpinned, see tests/test_gray.py), bit for bit."""
import io
import json
import os

import numpy as np
import pytest

import kpeg_testlib as T

pytestmark = pytest.mark.gpu
GOLD = T.GOLDEN
MAN = json.load(open(os.path.join(GOLD, "manifest_gray.json")))


@pytest.fixture(scope="module")
def ctx():
    import libkpeg_amd
    c = libkpeg_amd.Context(0)
    yield c
    c.lib.kpeg_hip_debug_set(c._h, 7, 0)
    c.close()


def _decode(ctx, data, dri=False):
    import libkpeg_amd as K
    rc, frame, scan = K.host_parse(data, allow_dri=dri, allow_gray=True)
    assert rc == K.DECODE_DONE and frame.components == 1
    return ctx.decode_scan(frame, scan), frame, scan


@pytest.mark.parametrize("name", sorted(MAN))
def test_fixtures_match_the_oracle(ctx, name):
    data = open(os.path.join(GOLD, name), "rb").read()
    st, want = T.oracle_decode_gray(data)
    assert st == T.DECODE_DONE and T.sha256(want.tobytes()) == MAN[name]["oracle_rgb_sha256"]
    for layout in (0, 1, 2):        # 2 = the compact stream wherever the width allows it (one-component frames too, since the end of round 3)
        assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, layout) == 0
        got, _, _ = _decode(ctx, data, "rst" in name)
        bad = np.argwhere(got != want)
        assert bad.size == 0, "%s layout %d: %s of %d" % (name, layout, bad[:8].tolist(), len(bad))
    ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 0)


@pytest.mark.parametrize("w,h,q,kw", [
    (8, 8, 75, {}), (16, 8, 30, {}), (1920, 1080, 75, {}), (1920, 1080, 97, {"optimize": True}),
    (2048, 256, 60, {"restart_marker_rows": 1}), (4096, 2048, 85, {"restart_marker_blocks": 7}), (520, 8, 90, {})])
def test_pillow_encodings_match_the_oracle(ctx, w, h, q, kw):
    """Sizes from one block to 8 Mpx, noise over a gradient (long and short codes, many zero DC differences in the flat
    parts -> quirk Q1 blocks), optimised tables, restart intervals that do and do not divide the row."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(w * 31 + h)
    y, x = np.mgrid[0:h, 0:w]
    px = (x * 255.0 / max(w - 1, 1) * 0.6 + y * 255.0 / max(h - 1, 1) * 0.4)
    px[:, : w // 2] += rng.normal(0, 20, (h, w // 2))
    px[h // 2:, w // 2:] = 97       # a flat quarter: DC differences of zero
    buf = io.BytesIO()
    Image.fromarray(np.clip(px, 0, 255).astype(np.uint8), "L").save(buf, "JPEG", quality=q, **kw)
    data = buf.getvalue()
    st, want = T.oracle_decode_gray(data)
    assert st == T.DECODE_DONE
    for layout in (0, 1, 2):        # the library's choice, the dense coefficients, the compact stream wherever the width allows it
        assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, layout) == 0
        try:
            got, _, _ = _decode(ctx, data, bool(kw.get("restart_marker_rows") or kw.get("restart_marker_blocks")))
        finally:
            ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 0)
        bad = np.argwhere(got != want)
        assert bad.size == 0, "layout %d: first mismatches (y,x,c) %s of %d" % (layout, bad[:8].tolist(), len(bad))
        assert np.array_equal(got[..., 0], got[..., 1]) and np.array_equal(got[..., 0], got[..., 2])


def test_gray_then_colour_then_gray_on_one_context(ctx):
    """The chroma blocks' bound words are preset per call: a colour decode between two one-component decodes of the
    same geometry must leave nothing behind."""
    g = open(os.path.join(GOLD, "gray_ramp_64x48_q75.jpg"), "rb").read()
    st, want_g = T.oracle_decode_gray(g)
    c = T.synth_jpeg(64, 48, seed=9, sigma=30.0)
    st, want_c = T.oracle_decode(c)
    p = T.oracle_parse(c)
    for _ in range(2):
        assert np.array_equal(_decode(ctx, g)[0], want_g)
        assert np.array_equal(ctx.decode_scan(T.make_frame(p), p.scan), want_c)
    assert np.array_equal(_decode(ctx, g)[0], want_g)


def test_batch_of_gray_frames(ctx):
    import torch
    Image = pytest.importorskip("PIL.Image")
    import libkpeg_amd as K
    rng = np.random.default_rng(3)
    frame, scans, wants = None, [], []
    for i in range(6):
        px = np.clip(rng.normal(128, 10 + 8 * i, (72, 192)), 0, 255).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(px, "L").save(buf, "JPEG", quality=80)        # standard tables: identical across the batch
        st, want = T.oracle_decode_gray(buf.getvalue())
        rc, frame, scan = K.host_parse(buf.getvalue(), allow_gray=True)
        assert st == T.DECODE_DONE and rc == K.DECODE_DONE
        scans.append(torch.from_numpy(np.ascontiguousarray(scan)).cuda())
        wants.append(want)
    outs = [torch.zeros((72, 192, 3), dtype=torch.uint8, device="cuda") for _ in scans]
    torch.cuda.synchronize()
    ctx.decode_batch_dev(frame, [t.data_ptr() for t in scans], [t.numel() for t in scans], [t.data_ptr() for t in outs])
    ctx.sync()
    for i, o in enumerate(outs):
        assert np.array_equal(o.cpu().numpy(), wants[i]), i


def test_corrupt_gray_streams_report_and_recover(ctx):
    import libkpeg_amd as K
    data = open(os.path.join(GOLD, "gray_flower_640x424_q80.jpg"), "rb").read()
    st, want = T.oracle_decode_gray(data)
    rc, frame, scan = K.host_parse(data, allow_gray=True)
    rng = np.random.default_rng(5)
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


def test_cli_allow_gray(tmp_path):
    """`kpeg file.jpg` rejects a one-component file as the reference does (no PPM); `kpeg --allow-gray file.jpg`
    writes the P6 file with R = G = B."""
    import shutil, subprocess
    import libkpeg_amd as K
    name = "gray_flower_320x208_q60_opt"
    dst = tmp_path / (name + ".jpg")
    shutil.copy(os.path.join(GOLD, name + ".jpg"), dst)
    subprocess.run([K.CLI, str(dst)], cwd=tmp_path, capture_output=True, timeout=120)
    assert not os.path.exists(tmp_path / (name + ".ppm"))
    out = subprocess.run([K.CLI, "--allow-gray", str(dst)], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout[-500:] + out.stderr[-500:]
    ppm = open(tmp_path / (name + ".ppm"), "rb").read()
    st, want = T.oracle_decode_gray(open(dst, "rb").read())
    assert ppm == T.ppm_header(320, 208) + want.tobytes()
