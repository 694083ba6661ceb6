"""On-device entropy decode (K0..K2) and the whole seam (scan bytes -> RGB) vs the oracle."""
import os
import sys

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


@pytest.mark.parametrize("w,h", [(65528, 8), (16, 65528), (8000, 16), (24, 24), (4104, 40)])
def test_extreme_geometries(ctx, w, h):
    """One MCU row of the maximum width, one MCU column of the maximum height, widths that leave a partial K4 tile."""
    data = T.synth_jpeg(w, h, seed=77)
    st, want = T.oracle_decode(data)
    assert st == T.DECODE_DONE
    p = T.oracle_parse(data)
    got = ctx.decode_scan(T.make_frame(p), p.scan)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("subseq", [0, 64, 96, 384])
def test_random_sweep_against_the_oracle(ctx, subseq):
    """Seeded sweep over sizes, qualities, noise levels, dense-noise mode and restart intervals (48 cases), with
    K1/K2's sub-sequence size chosen from the bit rate and the picture's size (0) and forced to the small pictures', the sparse and
    the dense size."""
    assert ctx.lib.kpeg_hip_debug_set(ctx._h, 4, subseq) == 0
    try:
        _random_sweep(ctx)
    finally:
        ctx.lib.kpeg_hip_debug_set(ctx._h, 4, 0)


def _random_sweep(ctx):
    rng = np.random.default_rng(20240607)
    for case in range(48):
        w, h = int(rng.integers(1, 40)) * 8, int(rng.integers(1, 24)) * 8
        q = int(rng.choice([10, 30, 50, 75, 90, 95, 100]))
        mode = int(rng.random() < 0.2)
        sigma = 0.0 if mode else float(rng.choice([0.0, 2.0, 6.0, 20.0, 60.0]))
        seed = int(rng.integers(1, 1 << 30))
        interval = 0 if rng.random() < 0.6 else int(rng.integers(1, max(2, (w // 8) * (h // 8))))
        data = T.synth_jpeg(w, h, seed=seed, quality=q, sigma=sigma, mode=mode, restart_interval=interval)
        if interval:
            want, p, _ = T.oracle_decode_rst(data, interval)
            frame = T.make_frame(p, interval)
        else:
            st, want = T.oracle_decode(data)
            assert st == T.DECODE_DONE
            p = T.oracle_parse(data)
            frame = T.make_frame(p)
        got = ctx.decode_scan(frame, p.scan)
        assert np.array_equal(got, want), (case, w, h, q, sigma, mode, seed, interval)


def test_more_entropy_data_than_the_frame_needs(ctx):
    """SOF0 says 64x32 but the scan holds the blocks of 64x64: the reference decodes (W*H)/64 MCUs and ignores the
    rest of the bits (Decoder.cpp:670); so does K2 -- and it must not write the surplus blocks anywhere."""
    data = bytearray(T.synth_jpeg(64, 64, seed=31))
    i = data.find(b"\xff\xc0")
    assert i > 0 and data[i + 5:i + 7] == bytes([0, 64])
    data[i + 5:i + 7] = bytes([0, 32])
    data = bytes(data)
    st, want = T.oracle_decode(data)
    assert st == T.DECODE_DONE and want.shape == (32, 64, 3)
    p = T.oracle_parse(data)
    got = ctx.decode_scan(T.make_frame(p), p.scan)
    assert np.array_equal(got, want)


def test_corrupted_scans_fail_cleanly(ctx):
    """120 seeded corruptions of an entropy-coded segment (bit flips, byte runs overwritten, truncations, surplus
    bytes): every decode either succeeds or reports KPEG_HIP_E_STREAM -- no fault, no hang -- and the context
    decodes the clean stream correctly afterwards."""
    import libkpeg_amd as K
    data = T.synth_jpeg(256, 128, seed=77, sigma=12.0)
    st, want = T.oracle_decode(data)
    p = T.oracle_parse(data)
    frame = T.make_frame(p)
    clean = np.frombuffer(p.scan, dtype=np.uint8)
    rng = np.random.default_rng(99)
    failed = 0
    for case in range(120):
        s = clean.copy()
        kind = case % 4
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                s[int(rng.integers(0, s.size))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        elif kind == 1:
            a = int(rng.integers(0, s.size - 40))
            s[a:a + int(rng.integers(1, 40))] = rng.integers(0, 256, dtype=np.uint8)
        elif kind == 2:
            s = s[:int(rng.integers(1, s.size))].copy()
        else:
            s = np.concatenate([s, rng.integers(0, 256, int(rng.integers(1, 300)), dtype=np.uint8)])
        try:
            out = ctx.decode_scan(frame, s)
            assert out.shape == want.shape
        except K.KpegError as e:
            assert e.code == -4, e   # KPEG_HIP_E_STREAM
            failed += 1
    assert failed > 10   # most truncations must be detected
    assert np.array_equal(ctx.decode_scan(frame, p.scan), want)


def test_flat_image(ctx):
    """A constant image is a periodic bit string (14 bits per MCU: DC diff 0 + EOB, three times)."""
    rgb = np.empty((1024, 2048, 3), np.uint8)
    rgb[:] = (200, 30, 77)
    data = T.encode_rgb(rgb, quality=75)
    st, want = T.oracle_decode(data)
    assert st == T.DECODE_DONE
    p = T.oracle_parse(data)
    got = ctx.decode_scan(T.make_frame(p), p.scan)
    bad = np.argwhere(got != want)
    assert bad.size == 0, "first mismatches (y,x,c): %s of %d" % (bad[:8].tolist(), len(bad))


_STRESS = r"""
import sys
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import numpy as np, kpeg_testlib as T, libkpeg_amd
ctx = libkpeg_amd.Context(0)
assert ctx.lib.kpeg_hip_debug_set(ctx._h, 4, 64) == 0   # the stress build's small sub-sequences, whatever the bit rate
worst = 0
for (w, h, q, sigma, mode, warm) in [(1024, 512, 95, 0.0, 1, -1), (1024, 512, 95, 0.0, 1, 0), (1920, 1080, 75, 6.0, 0, 0), (512, 512, 30, 3.0, 0, -1)]:
    data = T.synth_jpeg(w, h, seed=21, quality=q, sigma=sigma, mode=mode)
    st, want = T.oracle_decode(data)
    p = T.oracle_parse(data)
    assert ctx.lib.kpeg_hip_debug_set(ctx._h, 2, warm) == 0
    # layout 1: the dense coefficients and with them the separate launches (verifying passes, chained pass), which small pictures took
    # until the end of round 3; layout 0: the library's choice -- the compact stream and the one kernel with its strict second launch
    for layout in (1, 0):
        assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, layout) == 0
        got = ctx.decode_scan(T.make_frame(p), p.scan)
        passes = int(ctx.timings()["sync_rounds"])
        if layout == 1:
            worst = max(worst, passes)
        bad = np.argwhere(got != want)
        assert bad.size == 0, (w, h, q, mode, warm, layout, passes, bad[:8].tolist(), len(bad))
print("PASSES", worst)
"""


def test_boundary_passes_and_chained_pass_on_tiny_workgroups():
    """libkpeg_hip_stress.so = the same sources with 127 x 64-bit sub-sequences per workgroup and a 64-bit
    warm-up (libkpeg_amd/build.py): dense noise re-synchronises over many of those workgroups, so the
    verifying passes leave work and the chained last pass has to end the ripple.  Product geometry
    (48-Kbit workgroups) practically never gets there.  The same build has a pool of only two second-level
    Huffman tables: the long codes of the Annex-K tables take the canonical-search fallback."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "libkpeg_amd", "libkpeg_hip_stress.so")
    assert os.path.exists(lib), "run libkpeg_amd.build.build_all()"
    env = dict(os.environ, KPEG_HIP_LIB=lib)
    out = subprocess.run([sys.executable, "-c", _STRESS % {"tests": os.path.join(root, "tests"), "root": root}], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    passes = int(out.stdout.strip().split("PASSES")[-1])
    assert passes >= 3, "the stress geometry no longer reaches the chained pass (passes=%d)" % passes


def test_dense_noise_stream(ctx):
    """Dense noise at q95 over ~190 workgroups: long codes, slow re-synchronisation."""
    data = T.synth_jpeg(1024, 512, seed=21, quality=95, sigma=0.0, mode=1)
    st, want = T.oracle_decode(data)
    assert st == T.DECODE_DONE
    p = T.oracle_parse(data)
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


@pytest.mark.parametrize("w,h,nstripes", [(128, 64, 2), (256, 96, 3), (1920, 1080, 4)])
def test_stripes_decode_independently(ctx, w, h, nstripes):
    """Row-stripe sharding: each stripe from the bytes of its own restart intervals (what one rank
    per GPU does), concatenated, equals the whole image."""
    import torch
    import libkpeg_amd as K
    mw = w // 8
    data = T.synth_jpeg(w, h, seed=13, restart_interval=mw)
    want, p, _ = T.oracle_decode_rst(data, mw)
    rc, frame, scan = K.host_parse(data, allow_dri=True)
    assert rc == K.DECODE_DONE
    parts = []
    for first, rows, b0, b1 in K.stripe_ranges(scan, h // 8, mw, mw, nstripes):
        d_scan = torch.from_numpy(np.ascontiguousarray(scan[b0:b1])).cuda()
        d_rgb = torch.empty((rows * 8, w, 3), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.decode_stripe_dev(frame, d_scan.data_ptr(), d_scan.numel(), first, rows, d_rgb.data_ptr())
        ctx.sync()
        parts.append(d_rgb.cpu().numpy())
    got = np.concatenate(parts, 0)
    assert np.array_equal(got, want)


def test_truncated_and_corrupt_streams_are_reported(ctx):
    import libkpeg_amd as K
    data = T.synth_jpeg(128, 64, seed=17)
    p = T.oracle_parse(data)
    f = T.make_frame(p)
    with pytest.raises(K.KpegError) as e:
        ctx.decode_scan(f, p.scan[: len(p.scan) // 2])
    assert e.value.code == K.E_STREAM
    # and the context stays usable
    st, want = T.oracle_decode(data)
    assert np.array_equal(ctx.decode_scan(f, p.scan), want)


def test_bad_arguments(ctx):
    import libkpeg_amd as K
    data = T.synth_jpeg(64, 64, seed=1)
    p = T.oracle_parse(data)
    f = T.make_frame(p)
    f.width = 60  # not a multiple of 8: the reference reads out of bounds here; the whole-image, stripe and batch entry points take it
    # as the any-size extension (tests/test_gpu_any_size.py) -- the kernels' own entry points work on whole blocks and keep the contract
    import torch
    d_scan = torch.from_numpy(np.frombuffer(p.scan, np.uint8).copy()).cuda()
    d_coef = torch.zeros(64 * 192, dtype=torch.int16, device="cuda")
    d_rgb = torch.zeros((64, 64, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(K.KpegError) as e:
        ctx.entropy_decode_dev(f, d_scan.data_ptr(), d_scan.numel(), d_coef.data_ptr())
    assert e.value.code == K.E_ARG
    with pytest.raises(K.KpegError) as e:
        ctx.decode_stripe_dev(f, d_scan.data_ptr(), d_scan.numel(), 0, 9, d_rgb.data_ptr())   # past the picture's last MCU row
    assert e.value.code == K.E_ARG
    f.width = 0
    with pytest.raises(K.KpegError) as e:
        ctx.decode_scan(f, p.scan)
    assert e.value.code == K.E_ARG
    g = T.make_frame(p)
    for k in range(16):
        g.dht[1][0].counts[k] = 255  # not a prefix code
    with pytest.raises(K.KpegError) as e:
        ctx.decode_scan(g, p.scan)
    assert e.value.code == K.E_TABLES


def test_golden_fixtures_through_the_product_path(ctx):
    """tests/golden: JPEG in, the reference's own PPM bytes out -- C++ marker parser + C ABI."""
    import json, os
    import libkpeg_amd as K
    man = json.load(open(os.path.join(T.GOLDEN, "manifest.json")))
    for name, g in sorted(man["decode"].items()):
        data = open(os.path.join(T.GOLDEN, name + ".jpg"), "rb").read()
        rc, frame, scan = K.host_parse(data)
        assert rc == K.DECODE_DONE
        rgb = ctx.decode_scan(frame, scan)
        assert T.sha256(T.ppm_bytes(rgb)) == g["ppm_sha256"], name


def test_lena_through_the_cli(tmp_path):
    """The reference's own sample image (misc/images/lena.jpg, committed as a data fixture): `kpeg lena.jpg` must
    write the PPM whose SHA-256 SURVEY.md section 4 records for the reference decoder."""
    import json, os, shutil, subprocess
    import libkpeg_amd as K
    man = json.load(open(os.path.join(T.GOLDEN, "manifest.json")))["lena"]
    src = os.path.join(T.GOLDEN, man["fixture"])
    assert T.sha256(open(src, "rb").read()) == man["jpg_sha256"]
    dst = tmp_path / "lena.jpg"
    shutil.copy(src, dst)
    out = subprocess.run([K.CLI, str(dst)], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout[-500:] + out.stderr[-500:]
    assert T.sha256(open(tmp_path / "lena.ppm", "rb").read()) == man["ppm_sha256"] == "064dace1c86b7d2d887ae5d2b76fc53444136e155cc45cf2bad867fcdb13078d"


def test_cli_writes_the_reference_ppm(tmp_path):
    """`kpeg <file.jpg>` -> <file>.ppm, byte-identical to the reference CLI's output."""
    import json, os, shutil, subprocess
    import libkpeg_amd as K
    man = json.load(open(os.path.join(T.GOLDEN, "manifest.json")))
    for name in ("synth_64x64_q75", "pil_96x64_q60_opt", "pil_32x32_saturated"):
        src = os.path.join(T.GOLDEN, name + ".jpg")
        dst = tmp_path / (name + ".jpg")
        shutil.copy(src, dst)
        out = subprocess.run([K.CLI, str(dst)], cwd=tmp_path, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout[-500:] + out.stderr[-500:]
        ppm = open(tmp_path / (name + ".ppm"), "rb").read()
        assert T.sha256(ppm) == man["decode"][name]["ppm_sha256"]
        assert ppm == open(os.path.join(T.GOLDEN, name + ".ppm"), "rb").read()
    # rejected inputs produce no PPM (reference: DRI -> ERROR)
    rej = tmp_path / "rej_dri.jpg"
    shutil.copy(os.path.join(T.GOLDEN, "rej_dri.jpg"), rej)
    subprocess.run([K.CLI, str(rej)], cwd=tmp_path, capture_output=True, timeout=120)
    assert not os.path.exists(tmp_path / "rej_dri.ppm")
    # ... unless the restart-marker extension is asked for
    subprocess.run([K.CLI, "--allow-dri", str(rej)], cwd=tmp_path, capture_output=True, timeout=120)
    assert os.path.exists(tmp_path / "rej_dri.ppm")


def test_cli_batch_front_end(tmp_path):
    """`kpeg --batch <dir> <file>...` (extension): files of identical geometry and tables go through the
    fused batch path together, the others alone; every PPM is what the single-file front end / the reference writes; a
    corrupt stream inside a group costs only its own PPM; names and streams the single-file front end rejects are skipped."""
    import os, shutil, subprocess
    import libkpeg_amd as K
    d = tmp_path / "in"
    d.mkdir()
    want = {}
    # five images of one geometry and quality (one group), two others
    for i in range(5):
        data = T.synth_jpeg(160, 96, seed=100 + i, quality=75)
        (d / ("a%d.jpg" % i)).write_bytes(data)
        want["a%d" % i] = T.ppm_bytes(T.oracle_decode(data)[1])
    for name, (w, h, q) in {"b0": (64, 64, 50), "b1": (200, 40, 90)}.items():
        data = T.synth_jpeg(w, h, seed=7, quality=q)
        (d / (name + ".jpg")).write_bytes(data)
        want[name] = T.ppm_bytes(T.oracle_decode(data)[1])
    # same geometry and tables as the group, but the entropy-coded data is garbage: fails alone
    good = T.synth_jpeg(160, 96, seed=100, quality=75)
    p = T.oracle_parse(good)
    cut = good.find(p.scan[:16])
    (d / "a9_corrupt.jpg").write_bytes(good[:cut] + bytes([0xFF, 0x00] * 40) + b"\xff\xd9")
    # rejected by the parser (restart markers without --allow-dri), a name that does not end in .jpg, a golden file given by name
    shutil.copy(os.path.join(T.GOLDEN, "rej_dri.jpg"), d / "rej_dri.jpg")
    (d / "note.txt").write_text("not an image")
    extra = tmp_path / "synth_64x64_q75.jpg"
    shutil.copy(os.path.join(T.GOLDEN, "synth_64x64_q75.jpg"), extra)
    out = subprocess.run([K.CLI, "--batch", str(d), str(extra)], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert "8 PPM written" in out.stdout and "1 rejected" in out.stdout and "1 failed" in out.stdout, out.stdout[-800:] + out.stderr[-400:]
    assert out.returncode != 0   # a failure is reported
    for name, ppm in want.items():
        assert open(d / (name + ".ppm"), "rb").read() == ppm, name
    assert open(tmp_path / "synth_64x64_q75.ppm", "rb").read() == open(os.path.join(T.GOLDEN, "synth_64x64_q75.ppm"), "rb").read()
    assert not os.path.exists(d / "a9_corrupt.ppm") and not os.path.exists(d / "rej_dri.ppm") and not os.path.exists(d / "note.ppm")


def test_cli_batch_front_end_with_the_extensions(tmp_path):
    """`kpeg --batch --allow-any-size --allow-420 --allow-gray <dir>`: pictures whose sizes are no multiples of 8 (a group of three of one
    geometry: the fused batch path into scratch, a crop per picture), 4:2:0 pictures (a group of two: picture by picture) and a grayscale
    one; every PPM is the oracle's; without the flags the same call writes nothing."""
    import io, subprocess
    Image = pytest.importorskip("PIL.Image")
    import libkpeg_amd as K
    d = tmp_path / "in"
    d.mkdir()
    rng = np.random.default_rng(12)
    want = {}
    y, x = np.mgrid[0:75, 0:101]
    for k in range(3):
        px = np.clip(np.stack([(x * (k + 2)) % 256, (y * 3) % 256, (x + y + 9 * k) % 256], -1) * 0.7 + rng.normal(40, 6, (75, 101, 3)), 0, 255).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(px).save(b, "JPEG", quality=80, subsampling=0)
        (d / ("odd%d.jpg" % k)).write_bytes(b.getvalue())
        want["odd%d" % k] = T.ppm_header(101, 75) + T.oracle_decode_any_size(b.getvalue())[1].tobytes()
    for k in range(2):
        px = np.clip(rng.normal(120, 35, (50, 70, 3)), 0, 255).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(px).save(b, "JPEG", quality=85, subsampling=2)
        (d / ("sub%d.jpg" % k)).write_bytes(b.getvalue())
        want["sub%d" % k] = T.ppm_header(70, 50) + T.oracle_decode_420(b.getvalue())[1].tobytes()
    b = io.BytesIO()
    Image.fromarray(np.clip(rng.normal(128, 30, (48, 64)), 0, 255).astype(np.uint8), "L").save(b, "JPEG", quality=75)
    (d / "gray.jpg").write_bytes(b.getvalue())
    want["gray"] = T.ppm_header(64, 48) + T.oracle_decode_gray(b.getvalue())[1].tobytes()
    out = subprocess.run([K.CLI, "--batch", str(d)], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert "0 PPM written" in out.stdout and "6 rejected" in out.stdout, out.stdout[-800:] + out.stderr[-400:]
    out = subprocess.run([K.CLI, "--batch", "--allow-any-size", "--allow-420", "--allow-gray", str(d)], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert "6 PPM written" in out.stdout and "0 rejected" in out.stdout and "0 failed" in out.stdout, out.stdout[-800:] + out.stderr[-400:]
    assert out.returncode == 0
    for name, ppm in want.items():
        assert open(d / (name + ".ppm"), "rb").read() == ppm, name


def test_full_size_8k_properties(ctx):
    """BASELINE's full size (7680x4320): too slow for the scalar oracle in a unit test budget beyond one
    pass, so check it once against the multi-threaded oracle and by a checksum of row checksums."""
    import hashlib
    data = T.synth_jpeg(7680, 4320, seed=1234)
    p = T.oracle_parse(data)
    got = ctx.decode_scan(T.make_frame(p), p.scan)
    st, want = T.oracle_decode(data, nthreads=16)
    assert st == T.DECODE_DONE
    assert hashlib.sha256(got.tobytes()).hexdigest() == hashlib.sha256(want.tobytes()).hexdigest()


@pytest.mark.parametrize("q,subseq", [(90, 0), (90, 96), (75, 96)])
def test_photographic_content_whose_workgroups_guess_wrong(ctx, q, subseq):
    """Synthetic fields never have a workgroup whose assumed entry state (from its warm-up) fails; photographs do -- smooth
    regions repeat, and a decoder that is off by a component stays off until the content changes.  A golden photograph tiled
    to 2560 x 1696 and re-encoded: K1's verifying launch has real work (tools/photo_k1.py: a few per cent of the
    workgroups), with the bit rate's own sub-sequence size and with the sparse one forced."""
    from PIL import Image
    im = np.asarray(Image.open(os.path.join(T.GOLDEN, "nat_china_640x424_q90.jpg")).convert("RGB"))
    big = np.ascontiguousarray(np.tile(im, (4, 4, 1)))
    data = T.encode_rgb(big, quality=q)
    st, want = T.oracle_decode(data)
    assert st == T.DECODE_DONE
    p = T.oracle_parse(data)
    assert ctx.lib.kpeg_hip_debug_set(ctx._h, 4, subseq) == 0
    try:
        got = ctx.decode_scan(T.make_frame(p), p.scan)
    finally:
        ctx.lib.kpeg_hip_debug_set(ctx._h, 4, 0)
    bad = np.argwhere(got != want)
    assert bad.size == 0, "first mismatches (y,x,c): %s of %d" % (bad[:8].tolist(), len(bad))


@pytest.mark.parametrize("case", ["synthetic", "photograph", "small", "dense", "corrupt", "unaligned"])
def test_one_kernel_for_sync_and_write(ctx, case):
    """k_sync_write (K1's pass 0 and K2 in one kernel; debug key 9) on the calls it takes -- one image, no restart markers,
    compact stream, sparse sub-sequences -- and on those it must hand on: a photograph (some workgroup's assumed entry
    state fails: the verifying launch and k_write behind it finish the call), a dense stream and a corrupt one."""
    from PIL import Image
    assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 2) == 0     # the compact stream wherever it is possible
    assert ctx.lib.kpeg_hip_debug_set(ctx._h, 9, 1) == 0
    try:
        if case == "photograph":
            im = np.asarray(Image.open(os.path.join(T.GOLDEN, "nat_china_640x424_q90.jpg")).convert("RGB"))
            data = T.encode_rgb(np.ascontiguousarray(np.tile(im, (4, 4, 1))), quality=85)
        elif case == "small":
            data = T.synth_jpeg(64, 64, seed=5)
        elif case == "dense":
            data = T.synth_jpeg(1024, 512, seed=21, quality=95, sigma=0.0, mode=1)
        else:
            data = T.synth_jpeg(2560, 1440, seed=77, quality=75, sigma=6.0)
        st, want = T.oracle_decode(data)
        p = T.oracle_parse(data)
        frame = T.make_frame(p)
        if case == "corrupt":
            import libkpeg_amd as K
            rng = np.random.default_rng(5)
            for _ in range(6):
                bad = bytearray(p.scan)
                for pos in rng.integers(0, len(bad), 4):
                    bad[pos] ^= 1 << int(rng.integers(0, 8))
                try:
                    ctx.decode_scan(frame, bytes(bad))
                except K.KpegError:
                    pass
            got = ctx.decode_scan(frame, p.scan)      # and the context is as good as new
        elif case == "unaligned":
            import torch
            buf = torch.zeros(len(p.scan) + 16, dtype=torch.uint8, device="cuda")
            buf[3:3 + len(p.scan)] = torch.frombuffer(bytearray(p.scan), dtype=torch.uint8).cuda()
            out = torch.empty(want.shape, dtype=torch.uint8, device="cuda")
            for _ in range(3):
                ctx.decode_scan_dev(frame, buf.data_ptr() + 3, len(p.scan), out.data_ptr())
            ctx.sync()
            got = out.cpu().numpy()
        else:
            got = None
            for _ in range(3):                         # (the call's number changes; the flags of the call before must not count)
                got = ctx.decode_scan(frame, p.scan)
        bad = np.argwhere(got != want)
        assert bad.size == 0, "first mismatches (y,x,c): %s of %d" % (bad[:8].tolist(), len(bad))
        # the same call through the separate launches (k_sync_write is the default where it applies)
        assert ctx.lib.kpeg_hip_debug_set(ctx._h, 9, 0) == 0
        got = ctx.decode_scan(frame, p.scan)
        assert int(ctx.timings()["sync_rounds"]) >= 2
        assert np.array_equal(got, want)
    finally:
        ctx.lib.kpeg_hip_debug_set(ctx._h, 9, 1)
        ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 0)


def test_one_kernel_for_sync_and_write_leaves_nothing_from_the_call_before(ctx):
    """k_sync_write makes no presets (every first-record entry and error bound is written by the one workgroup that owns it):
    two different pictures of one size, decoded in turns, must not see each other's."""
    assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 2) == 0
    assert ctx.lib.kpeg_hip_debug_set(ctx._h, 9, 1) == 0
    try:
        pics = []
        for seed, q, sigma in ((11, 75, 6.0), (12, 50, 1.0), (13, 88, 9.0)):
            data = T.synth_jpeg(1920, 1088, seed=seed, quality=q, sigma=sigma)
            st, want = T.oracle_decode(data)
            p = T.oracle_parse(data)
            pics.append((T.make_frame(p), p.scan, want))
        for turn in range(6):
            frame, scan, want = pics[turn % 3]
            got = ctx.decode_scan(frame, scan)
            assert int(ctx.timings()["sync_rounds"]) == 1, "k_sync_write did not finish this call itself"
            bad = np.argwhere(got != want)
            assert bad.size == 0, (turn, bad[:8].tolist(), len(bad))
    finally:
        ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 0)


@pytest.mark.parametrize("offset", [1, 7])
def test_unaligned_scan_pointer(ctx, offset):
    """The scan pointer has any alignment.  K0 (kept for restart segments and batches; forced here with debug key 8) loads
    16 bytes per thread when the pointer allows it and falls back to bytes when not; without K0 the workgroups of K1 and
    K2 un-stuff the chunks they stage and load them as aligned words around the chunk."""
    import torch
    data = T.synth_jpeg(512, 256, seed=91, restart_interval=0)
    st, want = T.oracle_decode(data)
    p = T.oracle_parse(data)
    buf = torch.zeros(len(p.scan) + 64, dtype=torch.uint8, device="cuda")
    buf[offset:offset + len(p.scan)] = torch.frombuffer(bytearray(p.scan), dtype=torch.uint8).cuda()
    try:
        for k0 in (0, 1):
            assert ctx.lib.kpeg_hip_debug_set(ctx._h, 8, k0) == 0
            d_rgb = torch.zeros((256, 512, 3), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            ctx.decode_scan_dev(T.make_frame(p), buf.data_ptr() + offset, len(p.scan), d_rgb.data_ptr())
            ctx.sync()
            assert np.array_equal(d_rgb.cpu().numpy(), want), k0
    finally:
        ctx.lib.kpeg_hip_debug_set(ctx._h, 8, 0)


@pytest.mark.parametrize("w,h,q,sigma,smode", [(8, 8, 75, 6.0, 0), (16, 8, 90, 6.0, 0), (64, 64, 95, 0.0, 1), (712, 472, 95, 0.0, 1),
                                               (1920, 1080, 98, 40.0, 0), (3840, 2160, 75, 6.0, 0)])
def test_with_and_without_k0(ctx, w, h, q, sigma, smode):
    """One image without restart markers is decoded without K0: K1's and K2's workgroups un-stuff the 12-byte (48-byte)
    chunks of the scan they stage, positions that leave a workgroup are chunk << 7 | bit.  Same pixels as with K0 (debug
    key 8) and as the oracle: scans of one chunk, dense noise (an FF 00 every few chunks: chunks of every length from 6 to
    12 bytes, stuffing on chunk and workgroup boundaries), both sub-sequence sizes, both coefficient layouts."""
    data = T.synth_jpeg(w, h, seed=5 + w, quality=q, sigma=sigma, mode=smode)
    st, want = T.oracle_decode(data, 16)
    assert st == T.DECODE_DONE
    p = T.oracle_parse(data)
    assert w < 64 or p.scan.count(b"\xff\x00") > 3
    frame = T.make_frame(p)
    try:
        for k0 in (0, 1):
            for layout in (1, 2):
                assert ctx.lib.kpeg_hip_debug_set(ctx._h, 8, k0) == 0 and ctx.lib.kpeg_hip_debug_set(ctx._h, 7, layout) == 0
                got = ctx.decode_scan(frame, p.scan)
                bad = np.argwhere(got != want)
                assert bad.size == 0, "K0 %d layout %d: first mismatches (y,x,c) %s of %d" % (k0, layout, bad[:8].tolist(), len(bad))
    finally:
        ctx.lib.kpeg_hip_debug_set(ctx._h, 8, 0)
        ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 0)


def test_stuffing_at_the_very_end_without_k0(ctx):
    """byteStuffScanData's tail rule (Decoder.cpp:631-650: a 00 after an FF stays if it is the scan's last byte) in the
    last chunk of the last workgroup: the committed fixture the real reference decoded, and scans cut so that they end in
    FF 00 / FF / 00."""
    import os
    import libkpeg_amd as K
    for name in ("ok_trailing_ff", "ok_no_eoi", "ok_comment"):
        data = open(os.path.join(T.GOLDEN, name + ".jpg"), "rb").read()
        st, want = T.oracle_decode(data)
        assert st == T.DECODE_DONE
        rc, frame, scan = K.host_parse(data)
        for k0 in (0, 1):
            ctx.lib.kpeg_hip_debug_set(ctx._h, 8, k0)
            assert np.array_equal(ctx.decode_scan(frame, scan), want), (name, k0)
    ctx.lib.kpeg_hip_debug_set(ctx._h, 8, 0)
    # the same stream with 1..3 bytes of FF / 00 appended (bits after the last block are ignored, as the reference ignores them)
    data = T.synth_jpeg(128, 64, seed=12, sigma=20.0)
    st, want = T.oracle_decode(data)
    p = T.oracle_parse(data)
    frame = T.make_frame(p)
    for tail in (b"\xff\x00", b"\xff", b"\x00", b"\xff\x00\xff\x00", b"\xff\x00\x00"):
        for k0 in (0, 1):
            ctx.lib.kpeg_hip_debug_set(ctx._h, 8, k0)
            assert np.array_equal(ctx.decode_scan(frame, p.scan + tail), want), (tail, k0)
    ctx.lib.kpeg_hip_debug_set(ctx._h, 8, 0)


def test_decode_batch(ctx):
    """kpeg_hip_decode_batch: independent images of one geometry and one set of tables, pipelined over the
    context's lanes (more images than lanes, scans of different lengths)."""
    import libkpeg_amd as K
    w, h, n = 256, 128, 11
    datas = [T.synth_jpeg(w, h, seed=40 + i, sigma=2.0 + 3 * (i % 4)) for i in range(n)]
    parsed = [K.host_parse(d) for d in datas]
    outs = ctx.decode_batch(parsed[0][1], [p[2] for p in parsed])
    for d, o in zip(datas, outs):
        st, want = T.oracle_decode(d)
        assert np.array_equal(o, want)
    # one of them again through the ordinary entry point: the lanes left the parent context intact
    assert np.array_equal(ctx.decode_scan(parsed[3][1], parsed[3][2]), outs[3])


@pytest.mark.parametrize("chunk", [0, 4])
def test_decode_batch_dev_fused(ctx, chunk):
    """Device-resident batch = the images as restart segments of one virtual stream, one set of launches
    (and, with the chunk size forced down by the test hook, several chunks)."""
    import torch
    import libkpeg_amd as K
    w, h, n = 320, 136, 11
    datas = [T.synth_jpeg(w, h, seed=300 + i, quality=80, sigma=1.0 + i) for i in range(n)]
    parsed = [K.host_parse(d) for d in datas]
    frame = parsed[0][1]
    d_scans = [torch.from_numpy(np.ascontiguousarray(p[2])).cuda() for p in parsed]
    d_rgbs = [torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda") for _ in range(n)]
    torch.cuda.synchronize()
    try:
        assert ctx.lib.kpeg_hip_debug_set(ctx._h, 3, chunk) == 0
        ctx.decode_batch_dev(frame, [t.data_ptr() for t in d_scans], [t.numel() for t in d_scans], [t.data_ptr() for t in d_rgbs])
        ctx.sync()
    finally:
        ctx.lib.kpeg_hip_debug_set(ctx._h, 3, 0)
    for d, o in zip(datas, d_rgbs):
        st, want = T.oracle_decode(d)
        assert np.array_equal(o.cpu().numpy(), want)


def test_decode_batch_dev_with_restart_markers(ctx):
    """Images that carry restart markers themselves (DRI extension) are outside the fused path's contract: the
    device batch then goes image by image over the context's lanes."""
    import torch
    import libkpeg_amd as K
    w, h, n, interval = 256, 64, 7, 8
    datas = [T.synth_jpeg(w, h, seed=500 + i, restart_interval=interval) for i in range(n)]
    wants, parsed = [], []
    for d in datas:
        want, p, _ = T.oracle_decode_rst(d, interval)
        wants.append(want)
        parsed.append(p)
    frame = T.make_frame(parsed[0], interval)
    d_scans = [torch.frombuffer(bytearray(p.scan), dtype=torch.uint8).cuda() for p in parsed]
    d_rgbs = [torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda") for _ in range(n)]
    torch.cuda.synchronize()
    ctx.decode_batch_dev(frame, [t.data_ptr() for t in d_scans], [t.numel() for t in d_scans], [t.data_ptr() for t in d_rgbs])
    ctx.sync()
    for want, o in zip(wants, d_rgbs):
        assert np.array_equal(o.cpu().numpy(), want)
    # a truncated image fails the batch at sync(); the lanes stay usable
    lens = [t.numel() for t in d_scans]
    lens[2] //= 2
    ctx.decode_batch_dev(frame, [t.data_ptr() for t in d_scans], lens, [t.data_ptr() for t in d_rgbs])
    with pytest.raises(RuntimeError):
        ctx.sync()
    ctx.decode_batch_dev(frame, [t.data_ptr() for t in d_scans], [t.numel() for t in d_scans], [t.data_ptr() for t in d_rgbs])
    ctx.sync()
    assert np.array_equal(d_rgbs[2].cpu().numpy(), wants[2])


def test_decode_batch_dev_and_error(ctx):
    """Device-resident batch on the caller's stream; a truncated scan in the middle fails the batch at sync()
    while the other images are still decoded."""
    import torch
    import libkpeg_amd as K
    w, h, n = 128, 128, 9
    datas = [T.synth_jpeg(w, h, seed=70 + i) for i in range(n)]
    parsed = [K.host_parse(d) for d in datas]
    frame = parsed[0][1]
    d_scans = [torch.from_numpy(np.ascontiguousarray(p[2])).cuda() for p in parsed]
    d_rgbs = [torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda") for _ in range(n)]
    torch.cuda.synchronize()
    ctx.decode_batch_dev(frame, [t.data_ptr() for t in d_scans], [t.numel() for t in d_scans], [t.data_ptr() for t in d_rgbs])
    ctx.sync()
    for d, o in zip(datas, d_rgbs):
        st, want = T.oracle_decode(d)
        assert np.array_equal(o.cpu().numpy(), want)
    lens = [t.numel() for t in d_scans]
    lens[4] //= 2   # truncated: the blocks of the second half never arrive
    for t in d_rgbs:
        t.zero_()
    torch.cuda.synchronize()
    ctx.decode_batch_dev(frame, [t.data_ptr() for t in d_scans], lens, [t.data_ptr() for t in d_rgbs])
    with pytest.raises(RuntimeError):
        ctx.sync()
    st, want = T.oracle_decode(datas[5])
    assert np.array_equal(d_rgbs[5].cpu().numpy(), want)
    # and the context is usable afterwards
    ctx.decode_batch_dev(frame, [t.data_ptr() for t in d_scans[:2]], [t.numel() for t in d_scans[:2]], [t.data_ptr() for t in d_rgbs[:2]])
    ctx.sync()


@pytest.mark.parametrize("weak", [False, True])
def test_bench_multi_gpu_path_rehearsal(weak):
    """bench.py's N>1 path (per-rank stripe synthesis with restart markers, DRI parse, stripe decode at a row offset, the
    banded decode + send to rank 0) as a 2-rank gloo rehearsal on this one GPU, verified against the oracle: the strong
    mode (one image, its rows split over the ranks: BASELINE config 5's shape at a small size) and the --weak mode."""
    import os, socket, subprocess, sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(T.ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse",
           "--width", "512", "--height", "256", "--no-cpu-baseline"] + (["--weak"] if weak else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, KPEG_BENCH_VERIFY="1"))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert "VERIFY_OK" in out.stderr
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    rows_per_rank = 256 if weak else 128
    assert d["n_gpus"] == 2 and d["scaling"] == ("weak" if weak else "strong")
    assert d["gather"]["bytes_into_root"] == 512 * rows_per_rank * 3 and d["gather"]["bands_per_rank"] == 2
    assert d["config"]["pixels_per_step"] == 512 * rows_per_rank * 2


@pytest.mark.parametrize("layout", [0, 2])
def test_parity_sweep_150_cases(ctx, layout):
    """The randomised sweep that used to be run by hand (tools/parity_sweep_dbg.py): 150 seeded cases over sizes up to
    712x472, qualities 5..100, noise levels, the dense-noise mode and restart intervals (none, one MCU row, 1, 5), with the
    coefficient layout chosen per call and with the compact stream forced wherever the width allows it."""
    assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, layout) == 0
    rng = np.random.default_rng(2026)
    try:
        done = 0
        while done < 150:
            w = int(rng.integers(1, 90)) * 8
            h = int(rng.integers(1, 60)) * 8
            q = int(rng.integers(5, 101))
            sigma = float(rng.choice([0.0, 2.0, 6.0, 20.0, 60.0]))
            mode = int(rng.integers(0, 2))
            ri = int(rng.choice([0, 0, w // 8, 1, 5]))
            seed = int(rng.integers(1, 1 << 30))
            try:
                data = T.synth_jpeg(w, h, seed=seed, quality=q, restart_interval=ri, sigma=sigma, mode=mode)
            except AssertionError:   # the test encoder's output buffer is too small for this case
                continue
            if ri:
                want, p, _ = T.oracle_decode_rst(data, ri)
            else:
                st, want = T.oracle_decode(data)
                assert st == T.DECODE_DONE
                p = T.oracle_parse(data)
            got = ctx.decode_scan(T.make_frame(p, ri), p.scan)
            assert np.array_equal(got, want), dict(w=w, h=h, q=q, sigma=sigma, mode=mode, ri=ri, seed=seed, layout=layout)
            done += 1
    finally:
        ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 0)


@pytest.mark.parametrize("q,sigma,layout", [(50, 2.0, 0), (90, 12.0, 2)])
def test_one_4k_image_three_times(ctx, q, sigma, layout):
    """tools/parity_large_dbg.py's soak as a test: a 3840x2160 image decoded three times and compared with the oracle every
    time (the fix-up passes patch bytes behind the tiles' own stores: an ordering bug there shows as a rare mismatch)."""
    data = T.synth_jpeg(3840, 2160, seed=9001, quality=q, sigma=sigma)
    st, want = T.oracle_decode(data, 16)
    assert st == T.DECODE_DONE
    p = T.oracle_parse(data)
    f = T.make_frame(p)
    assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, layout) == 0
    try:
        for rep in range(3):
            assert np.array_equal(ctx.decode_scan(f, p.scan), want), rep
    finally:
        ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 0)


@pytest.mark.parametrize("band_rows", [0, 8, 24, 1000])
def test_resident_decode_and_banded_download(ctx, band_rows):
    """kpeg_hip_decode_scan_resident + kpeg_hip_download_bands (what Image::dumpRawData streams to the PPM): the bands, put
    together, are the decode; band sizes that do not divide the height, one band for the whole image, the default size."""
    data = T.synth_jpeg(640, 200, seed=91, sigma=8.0)
    st, want = T.oracle_decode(data)
    assert st == T.DECODE_DONE
    p = T.oracle_parse(data)
    got, bands = ctx.decode_scan_banded(T.make_frame(p), p.scan, band_rows)
    assert np.array_equal(got, want)
    step = band_rows if band_rows else 200
    step = min(step, 200)
    assert bands == [(r, min(step, 200 - r)) for r in range(0, 200, step)]


def test_copies_of_an_image_own_their_pixels(tmp_path):
    """A copy of the decoder's image taken while the pixels are still on the GPU (lazy source), and the decoder's own image
    after dumpRawData() has left the source in place: both must still be picture A after the shared context has decoded
    picture B -- the copy fetches the pixels when it is made, the owner is made to fetch them before the buffer is reused, and
    a source that outlives its pixels answers "gone" (kpeg_hip_resident_generation) instead of handing out B's."""
    import ctypes, os
    import libkpeg_amd as K
    H = K.load_host()
    H.kpeg_host_test_image_holders.restype = ctypes.c_size_t
    H.kpeg_host_test_image_holders.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    w, h = 256, 128
    a, b = T.synth_jpeg(w, h, seed=5), T.synth_jpeg(w, h, seed=6, sigma=11.0)
    pa, pb = tmp_path / "a.jpg", tmp_path / "b.jpg"
    pa.write_bytes(a)
    pb.write_bytes(b)
    st, want = T.oracle_decode(a)
    assert st == T.DECODE_DONE
    out_copy, out_own = np.zeros(w * h * 3, np.uint8), np.zeros(w * h * 3, np.uint8)
    n = H.kpeg_host_test_image_holders(str(pa).encode(), str(pb).encode(), out_copy.ctypes.data, out_own.ctypes.data, out_copy.size)
    assert n == w * h * 3
    assert np.array_equal(out_copy.reshape(h, w, 3), want), "the copy holds another picture's pixels"
    assert np.array_equal(out_own.reshape(h, w, 3), want), "the decoder's own image holds another picture's pixels"
    assert T.sha256(open(tmp_path / "a.ppm", "rb").read()) == T.sha256(T.ppm_bytes(want))
