"""The any-size extension on the GPU: widths / heights that are not multiples of 8 through the whole-image, stripe and batch
entry points (padded decode + crop kernel), against the reference's own committed outputs cropped (see tests/test_any_size.py for
why that pins everything but the crop) and against the oracle on fresh encodings."""
import io
import os

import numpy as np
import pytest

import kpeg_testlib as T
from test_any_size import CASES, patched

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import libkpeg_amd
    c = libkpeg_amd.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name,w,h", CASES)
def test_patched_fixtures_equal_the_cropped_reference_output(ctx, name, w, h):
    import libkpeg_amd as K
    data, want = patched(name, w, h)
    rc, frame, scan = K.host_parse(data, allow_any_size=True)
    assert rc == K.DECODE_DONE
    got = ctx.decode_scan(frame, scan)
    assert got.shape == (h, w, 3)
    bad = np.argwhere(got != want)
    assert bad.size == 0, "first mismatches (y,x,c) %s of %d" % (bad[:8].tolist(), len(bad))


@pytest.mark.parametrize("w,h,q,kw", [(1, 1, 75, {}), (7, 9, 90, {}), (100, 75, 85, {}), (1919, 1079, 80, {}), (1921, 1081, 80, {}),
                                      (4097, 13, 60, {"optimize": True}), (13, 4097, 60, {}), (1000, 600, 85, {"restart_marker_blocks": 11})])
def test_pillow_encodings_match_the_oracle(ctx, w, h, q, kw):
    Image = pytest.importorskip("PIL.Image")
    import libkpeg_amd as K
    rng = np.random.default_rng(w * 7 + h)
    y, x = np.mgrid[0:h, 0:w]
    px = np.stack([x * 255.0 / max(w - 1, 1), y * 255.0 / max(h - 1, 1), (x + y) % 256], -1) + rng.normal(0, 5, (h, w, 3))
    buf = io.BytesIO()
    Image.fromarray(np.clip(px, 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=q, subsampling=0, **kw)
    data = buf.getvalue()
    dri = bool(kw.get("restart_marker_blocks"))
    if dri:
        # the oracle's restart path takes the stripped file and the interval (kpeg_testlib.oracle_decode_rst); here: padded geometry
        i = data.find(b"\xff\xdd\x00\x04")
        stripped = data[:i] + data[i + 6:]
        p = T.oracle_parse(stripped)
        pw, ph = (w + 7) & ~7, (h + 7) & ~7
        p.width, p.height = pw, ph
        rc, coef = T.oracle_entropy(p, kw["restart_marker_blocks"])
        assert rc == 0
        want = T.oracle_idct_colour(coef, p.qt, pw, ph)[:h, :w]
    else:
        st, want = T.oracle_decode_any_size(data)
        assert st == T.DECODE_DONE
    rc, frame, scan = K.host_parse(data, allow_dri=dri, allow_any_size=True)
    assert rc == K.DECODE_DONE and (frame.width, frame.height) == (w, h)
    got = ctx.decode_scan(frame, scan)
    bad = np.argwhere(got != want)
    assert bad.size == 0, "first mismatches (y,x,c) %s of %d" % (bad[:8].tolist(), len(bad))


def test_device_destination_and_both_layouts(ctx):
    """kpeg_hip_decode_scan_dev into a caller's device buffer (rows packed at 3 * width bytes), dense and compact stream."""
    import torch
    import libkpeg_amd as K
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(4)
    w, h = 2047, 1031          # padded width 2048: a multiple of 64, so the compact stream can be forced
    px = np.clip(rng.normal(128, 30, (h, w, 3)), 0, 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(px).save(buf, "JPEG", quality=50, subsampling=0)
    st, want = T.oracle_decode_any_size(buf.getvalue())
    rc, frame, scan = K.host_parse(buf.getvalue(), allow_any_size=True)
    d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda()
    for layout in (1, 2, 0):
        assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, layout) == 0
        d_rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.decode_scan_dev(frame, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
        ctx.sync()
        assert np.array_equal(d_rgb.cpu().numpy(), want), layout
    ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 0)


def _odd_pictures(n, w, h, q=80, seed=11):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    out = []
    for k in range(n):
        px = np.stack([(x * (k + 2)) % 256, (y * 3 + k * 17) % 256, (x + y + k) % 256], -1) * 0.6 + rng.normal(50, 9, (h, w, 3))
        buf = io.BytesIO()
        Image.fromarray(np.clip(px, 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=q, subsampling=0)
        out.append(buf.getvalue())
    return out


def test_batch_entry_points_take_any_size(ctx):
    """kpeg_hip_decode_batch_dev: the padded pictures through the fused batch path into scratch, one crop per picture into the
    callers' buffers; kpeg_hip_decode_batch (host buffers): picture by picture."""
    import torch
    import libkpeg_amd as K
    w, h, n = 333, 201, 5
    files = _odd_pictures(n, w, h)
    wants = [T.oracle_decode_any_size(d)[1] for d in files]
    parsed = [K.host_parse(d, allow_any_size=True) for d in files]
    frame = parsed[0][1]
    assert all(rc == K.DECODE_DONE for rc, _, _ in parsed)
    scans = [np.ascontiguousarray(sc) for _, _, sc in parsed]
    d_scans = [torch.from_numpy(sc).cuda() for sc in scans]
    d_rgbs = [torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda") for _ in range(n)]
    torch.cuda.synchronize()
    ctx.decode_batch_dev(frame, [t.data_ptr() for t in d_scans], [t.numel() for t in d_scans], [t.data_ptr() for t in d_rgbs])
    ctx.sync()
    for k in range(n):
        assert np.array_equal(d_rgbs[k].cpu().numpy(), wants[k]), k
    outs = ctx.decode_batch(frame, scans)
    for k in range(n):
        assert np.array_equal(outs[k], wants[k]), k


def test_stripe_entry_point_takes_any_size(ctx):
    """kpeg_hip_decode_stripe_dev on a picture whose width and height are no multiples of 8: the whole picture as one stripe, and -- a
    restart interval per MCU row -- as stripes of MCU rows, the last of which is cut short by the picture's height."""
    import torch
    import libkpeg_amd as K
    Image = pytest.importorskip("PIL.Image")
    w, h = 205, 123      # 26 x 16 MCUs, the last MCU row has 3 pixel rows
    rng = np.random.default_rng(3)
    px = np.clip(rng.normal(120, 40, (h, w, 3)), 0, 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(px).save(buf, "JPEG", quality=70, subsampling=0, restart_marker_rows=1)
    data = buf.getvalue()
    rc, frame, scan = K.host_parse(data, allow_dri=True, allow_any_size=True)
    assert rc == K.DECODE_DONE and frame.restart_interval == 26
    i = data.find(b"\xff\xdd\x00\x04")
    p = T.oracle_parse(data[:i] + data[i + 6:])
    p.width, p.height = 208, 128
    rc, coef = T.oracle_entropy(p, 26)
    assert rc == 0
    want = T.oracle_idct_colour(coef, p.qt, 208, 128)[:h, :w]
    d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda()
    d_rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.decode_stripe_dev(frame, d_scan.data_ptr(), d_scan.numel(), 0, 16, d_rgb.data_ptr())
    ctx.sync()
    assert np.array_equal(d_rgb.cpu().numpy(), want)
    # stripes of 5 MCU rows from the bytes of their own intervals (the markers RST0..7 in between are the stripe's to skip)
    marks = [k for k in range(len(scan) - 1) if scan[k] == 0xFF and 0xD0 <= scan[k + 1] <= 0xD7]
    starts = [0] + [m + 2 for m in marks]
    assert len(starts) == 16
    d_rgb.zero_()
    for r0 in range(0, 16, 5):
        rows = min(5, 16 - r0)
        lo, hi = starts[r0], (marks[r0 + rows - 1] if r0 + rows < 16 else len(scan))
        part = torch.from_numpy(np.ascontiguousarray(scan[lo:hi])).cuda()
        torch.cuda.synchronize()
        ctx.decode_stripe_dev(frame, part.data_ptr(), part.numel(), r0, rows, d_rgb[r0 * 8:].data_ptr())
        ctx.sync()
    assert np.array_equal(d_rgb.cpu().numpy(), want)
    with pytest.raises(K.KpegError) as e:     # past the padded picture's last MCU row
        ctx.decode_stripe_dev(frame, d_scan.data_ptr(), d_scan.numel(), 0, 17, d_rgb.data_ptr())
    assert e.value.code == K.E_ARG


def test_the_kernels_own_entry_points_keep_the_contract(ctx):
    """kpeg_hip_entropy_decode_dev and kpeg_hip_idct_colour_dev work on whole blocks: multiples of 8 only."""
    import torch
    import libkpeg_amd as K
    data, _ = patched("pil_96x64_q85", 93, 59)
    rc, frame, scan = K.host_parse(data, allow_any_size=True)
    d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda()
    d_coef = torch.zeros(12 * 8 * 192, dtype=torch.int16, device="cuda")
    d_rgb = torch.zeros((64, 96, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(K.KpegError) as e:
        ctx.entropy_decode_dev(frame, d_scan.data_ptr(), d_scan.numel(), d_coef.data_ptr())
    assert e.value.code == K.E_ARG
    with pytest.raises(K.KpegError) as e:
        ctx.idct_colour_dev(frame, d_coef.data_ptr(), d_rgb.data_ptr())
    assert e.value.code == K.E_ARG


def test_cli_allow_any_size(tmp_path):
    import subprocess
    import libkpeg_amd as K
    data, want = patched("pil_96x64_q85", 93, 59)
    dst = tmp_path / "odd.jpg"
    dst.write_bytes(data)
    subprocess.run([K.CLI, str(dst)], cwd=tmp_path, capture_output=True, timeout=120)
    assert not os.path.exists(tmp_path / "odd.ppm")          # without the flag: ERROR, nothing written
    out = subprocess.run([K.CLI, "--allow-any-size", str(dst)], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout[-500:] + out.stderr[-500:]
    assert open(tmp_path / "odd.ppm", "rb").read() == T.ppm_header(93, 59) + want.tobytes()
