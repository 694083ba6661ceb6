"""BASELINE.json's configurations at their full sizes, against hashes produced by libKPEG's own decoder
(tests/golden/manifest_large.json, written by tests/golden/make_golden_large.py in the build container):

  config 2/3  1920x1080 and 7680x4320 synthetic files through the product parser + the C ABI: PPM SHA-256
  config 4    256 x 1080p (32 distinct scans) through kpeg_hip_decode_batch_dev: every output checked
  config 5    16384x16384 with one restart interval per MCU row, decoded whole and as 8 row stripes on one GPU:
              stripes == whole == the reference's per-interval decode

The synthetic generator is integer-only: the GPU box regenerates the inputs byte for byte (jpg_sha256 is
asserted first), so no large file travels.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import kpeg_testlib as T

pytestmark = pytest.mark.gpu
LARGE = json.load(open(os.path.join(T.GOLDEN, "manifest_large.json")))


def sha(b):
    return hashlib.sha256(b).hexdigest()


@pytest.fixture(scope="module")
def ctx():
    import libkpeg_amd
    c = libkpeg_amd.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("key", ["1920x1080_seed1234", "7680x4320_seed1234"])
def test_headline_inputs_match_the_reference_ppm(ctx, key):
    import libkpeg_amd as K
    g = LARGE["synth"][key]
    data = T.synth_jpeg(g["width"], g["height"], seed=g["seed"])
    assert sha(data) == g["jpg_sha256"], "the generator no longer reproduces the pinned input"
    rc, frame, scan = K.host_parse(data)
    assert rc == K.DECODE_DONE
    rgb = ctx.decode_scan(frame, scan)
    assert sha(T.ppm_bytes(rgb)) == g["ppm_sha256"]


@pytest.mark.parametrize("name", sorted(LARGE["natural"]))
def test_photographs_match_the_reference_ppm(ctx, name):
    """Natural coefficient statistics at 1-5 bits per pixel, Pillow's quantisers, optimised Huffman tables: K1 needs
    several rounds per workgroup on these."""
    import libkpeg_amd as K
    g = LARGE["natural"][name]
    data = open(os.path.join(T.GOLDEN, name + ".jpg"), "rb").read()
    assert sha(data) == g["jpg_sha256"]
    rc, frame, scan = K.host_parse(data)
    assert rc == K.DECODE_DONE
    for subseq in (0, 64, 96, 384):
        assert ctx.lib.kpeg_hip_debug_set(ctx._h, 4, subseq) == 0
        try:
            rgb = ctx.decode_scan(frame, scan)
        finally:
            ctx.lib.kpeg_hip_debug_set(ctx._h, 4, 0)
        assert sha(T.ppm_bytes(rgb)) == g["ppm_sha256"], subseq


@pytest.mark.parametrize("key", sorted(LARGE.get("natural_8k", {})))
def test_photographs_tiled_to_8k_match_the_reference(ctx, key):
    """The natural-content inputs of bench.py's default line (committed photographs tiled to 7680 x 4352, 1.0-3.5 bits per pixel):
    pixels against libKPEG's own decoder (tests/golden/make_golden_photos.py), with k_sync_write and through the separate
    launches.  On these some workgroup's entry assumption always fails and the grid is larger than what the device holds at
    once (1000-1400 workgroups): k_sync_write's workgroups repair themselves, and the launches behind it have nothing to do
    (one launch of K1 with work), with the 96-bit sub-sequences and -- the 3.5 bit/px case -- with the 384-bit ones."""
    import sys
    pytest.importorskip("PIL")
    sys.path.insert(0, T.ROOT)
    import bench
    import libkpeg_amd as K
    g = LARGE["natural_8k"][key]
    data = bench.tiled_photo_jpeg(g["source"], g["quality"], g["width"], g["height"])
    assert sha(data) == g["jpg_sha256"], "the tiled input is no longer reproduced byte for byte"
    rc, frame, scan = K.host_parse(data)
    assert rc == K.DECODE_DONE
    try:
        for fused in (1, 0):
            assert ctx.lib.kpeg_hip_debug_set(ctx._h, 9, fused) == 0
            ctx.set_profiling(True)
            rgb = ctx.decode_scan(frame, scan)
            rounds = int(ctx.timings()["sync_rounds"])
            ctx.set_profiling(False)
            assert sha(rgb.tobytes()) == g["rgb_sha256"], fused
            if fused:
                assert rounds == 1, rounds
    finally:
        ctx.lib.kpeg_hip_debug_set(ctx._h, 9, 1)


def test_batch_of_256_1080p_images(ctx):
    """BASELINE config 4 exactly as bench.py --batch 256 times it: 256 device-resident 1080p images, 32 distinct
    scans (seeds 1234..1265), one call of kpeg_hip_decode_batch_dev.  The first 32 outputs are hashed against the
    reference's pixels, the other 224 compared with them on the device."""
    import torch
    import libkpeg_amd as K
    n, uniq, w, h = 256, 32, 1920, 1080
    frame, scans, want = None, [], []
    for i in range(uniq):
        g = LARGE["synth"]["1920x1080_seed%d" % (1234 + i)]
        data = T.synth_jpeg(w, h, seed=1234 + i)
        assert sha(data) == g["jpg_sha256"]
        rc, frame, scan = K.host_parse(data)
        assert rc == K.DECODE_DONE
        scans.append(torch.from_numpy(np.ascontiguousarray(scan)).cuda())
        want.append(g["rgb_sha256"])
    d_scans = [scans[i % uniq] for i in range(n)]
    d_rgbs = [torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda") for _ in range(n)]
    torch.cuda.synchronize()
    ctx.decode_batch_dev(frame, [t.data_ptr() for t in d_scans], [t.numel() for t in d_scans], [t.data_ptr() for t in d_rgbs])
    ctx.sync()
    for i in range(uniq):
        assert sha(d_rgbs[i].cpu().numpy().tobytes()) == want[i], i
    for i in range(uniq, n):
        assert torch.equal(d_rgbs[i], d_rgbs[i % uniq]), i


def test_16k_restart_image_whole_and_as_8_stripes(ctx):
    """BASELINE config 5's image on one GPU: decoded whole (2048 restart intervals in one call) and as the 8 row stripes
    the 8 ranks would take (kpeg_hip_decode_stripe_dev on the bytes of each stripe's own intervals).  Both must equal
    the reference's per-interval decode, stripe by stripe."""
    import torch
    import libkpeg_amd as K
    g = LARGE["dri16k"]
    w, h, mw = g["width"], g["height"], g["width"] // 8
    data = T.synth_jpeg(w, h, seed=g["seed"], restart_interval=g["restart_interval"])
    assert sha(data) == g["jpg_sha256"]
    rc, frame, scan = K.host_parse(data, allow_dri=True)
    assert rc == K.DECODE_DONE and frame.restart_interval == mw
    del data
    d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda()
    d_rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.decode_scan_dev(frame, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
    ctx.sync()
    rows = h // 8
    whole = [sha(d_rgb[s * rows:(s + 1) * rows].cpu().numpy().tobytes()) for s in range(8)]
    assert whole == g["stripe8_rgb_sha256"]
    hh = hashlib.sha256()
    hh.update(T.ppm_header(w, h))
    hh.update(d_rgb.cpu().numpy().tobytes())
    assert hh.hexdigest() == g["ppm_sha256"]
    d_rgb.zero_()
    ranges = K.stripe_ranges(scan, h // 8, mw, frame.restart_interval, 8)
    for s, (r0, nr, b0, b1) in enumerate(ranges):
        assert (r0, nr) == (s * (h // 64), h // 64)
        sl = d_scan[b0:b1].clone()   # a stripe's bytes as its rank would hold them (own allocation, own alignment)
        ctx.decode_stripe_dev(frame, sl.data_ptr(), sl.numel(), r0, nr, d_rgb[r0 * 8:].data_ptr())
        ctx.sync()
        assert sha(d_rgb[r0 * 8:(r0 + nr) * 8].cpu().numpy().tobytes()) == g["stripe8_rgb_sha256"][s], s


def test_sharded_c_entry_with_two_contexts_on_one_gpu(ctx):
    """kpeg_hip_decode_sharded / _dev: ONE process, one context per GPU, host threads inside.  Rehearsed with two and
    three contexts on GPU 0 (the stripes then run concurrently on one device): the host-destination form (every GPU
    downloads its own rows) and the device-destination form (peer copies into the root GPU's buffer) must both equal the
    oracle, for a restart interval of one MCU row and of a quarter row."""
    import torch
    import libkpeg_amd as K
    others = [K.Context(0), K.Context(0)]
    try:
        for (w, h, interval) in ((1024, 520, 128), (512, 264, 16)):
            data = T.synth_jpeg(w, h, seed=61, restart_interval=interval, sigma=9.0)
            want, p, _ = T.oracle_decode_rst(data, interval)
            rc, frame, scan = K.host_parse(data, allow_dri=True)
            assert rc == K.DECODE_DONE
            for n in (1, 2, 3):
                cs = [ctx] + others[: n - 1]
                got = K.decode_sharded(cs, frame, scan)
                assert np.array_equal(got, want), (w, h, interval, n, "host")
                d = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()
                K.decode_sharded(cs, frame, scan, d_rgb_root=d.data_ptr())
                assert np.array_equal(d.cpu().numpy(), want), (w, h, interval, n, "device")
        # a stream without restart markers cannot be sharded
        data = T.synth_jpeg(256, 128, seed=62)
        rc, frame, scan = K.host_parse(data)
        with pytest.raises(K.KpegError) as ei:
            K.decode_sharded([ctx, others[0]], frame, scan)
        assert ei.value.code == K.E_UNSUPPORTED
        # ... one context is just a decode
        st, want1 = T.oracle_decode(data)
        assert np.array_equal(K.decode_sharded([ctx], frame, scan), want1)
    finally:
        for c in others:
            c.close()
