"""The compact coefficient stream between K2 and K4 (4-byte records per non-zero AC coefficient + dense DC array +
first record of every tile) against the oracle, and against the dense int16 layout on the same inputs.

The library picks the layout per call (sparse streams whose width is a multiple of 64); the test hook
kpeg_hip_debug_set(ctx, 7, layout) forces 1 = dense or 2 = compact wherever the geometry allows."""
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
    c.lib.kpeg_hip_debug_set(c._h, 7, 0)
    c.close()


def _both_layouts(ctx, frame, scan, want, what):
    for layout in (2, 1, 0):
        assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, layout) == 0
        got = ctx.decode_scan(frame, scan)
        bad = np.argwhere(got != want)
        assert bad.size == 0, "%s, layout %d: first mismatches (y,x,c) %s of %d" % (what, layout, bad[:8].tolist(), len(bad))
    ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 0)


@pytest.mark.parametrize("w,h,q,sigma,smode,interval", [
    (64, 8, 75, 6.0, 0, 0), (64, 64, 75, 6.0, 0, 0), (128, 72, 50, 12.0, 0, 0), (512, 512, 30, 3.0, 0, 0),
    (1920, 1080, 75, 6.0, 0, 0), (128, 64, 95, 0.0, 1, 0),        # dense noise: 40+ records per block, forced through the compact path
    (256, 128, 98, 40.0, 0, 0), (448, 200, 75, 6.0, 0, 56), (448, 200, 75, 6.0, 0, 5), (3840, 64, 85, 9.0, 0, 480)])
def test_compact_stream_matches_oracle(ctx, w, h, q, sigma, smode, interval):
    data = T.synth_jpeg(w, h, seed=17, quality=q, sigma=sigma, mode=smode, restart_interval=interval)
    if interval:
        want, p, _ = T.oracle_decode_rst(data, interval)
    else:
        st, want = T.oracle_decode(data)
        assert st == T.DECODE_DONE
        p = T.oracle_parse(data)
    _both_layouts(ctx, T.make_frame(p, interval), p.scan, want, "%dx%d q%d" % (w, h, q))


def _encode_and_check(ctx, coef_nat, qt_nat, w, h, what):
    """coef_nat [nmcu,3,8,8] natural order -> a JPEG through the test encoder -> both layouts vs the oracle."""
    zz = T.zz_table()
    coef_zz = np.ascontiguousarray(coef_nat.reshape(-1, 3, 64)[..., zz])
    data = T.encode_coefs(coef_zz, w, h, qt_nat[0], qt_nat[1])
    st, want = T.oracle_decode(data)
    assert st == T.DECODE_DONE
    p = T.oracle_parse(data)
    _both_layouts(ctx, T.make_frame(p), p.scan, want, what)
    return ctx.timings()


def test_structural_ties_through_the_compact_stream(ctx):
    """DC + equal and opposite (0,1)/(1,0) terms in every block: thousands of unsafe pixels of corner-only blocks, settled
    from the corner coefficients stashed in the queue entries (the compact stream has no block to re-read)."""
    w, h = 256, 64
    nmcu = (w // 8) * (h // 8)
    rng = np.random.default_rng(6)
    coef = np.zeros((nmcu, 3, 8, 8), np.int16)
    coef[:, :, 0, 0] = rng.integers(-60, 61, size=(nmcu, 3)) * 2 + 1
    s = rng.integers(1, 4, size=(nmcu, 3))
    coef[:, :, 0, 1] = s
    coef[:, :, 1, 0] = -s
    qt = np.full((2, 64), 9, np.uint16)
    qt[:, 0] = 4
    t = _encode_and_check(ctx, coef, qt, w, h, "structural ties")
    assert t["exact_pixels"] > 1000


def test_every_pixel_unsafe_through_the_compact_stream(ctx):
    """Huge dequantised values in full blocks: every bound exceeds 0.5, the queue overflows on every tile and every
    sample is rebuilt from its tile's records and evaluated in reference order by the whole wavefront."""
    rng = np.random.default_rng(5)
    w, h = 128, 16
    nmcu = (w // 8) * (h // 8)
    coef = rng.integers(-1000, 1001, size=(nmcu, 3, 8, 8)).astype(np.int16)
    coef[:, :, 0, 0] = rng.integers(-900, 901, size=(nmcu, 3))
    qt = np.full((2, 64), 255, np.uint16)
    t = _encode_and_check(ctx, coef, qt, w, h, "all unsafe")
    assert t["exact_pixels"] >= w * h


@pytest.mark.parametrize("seed", [1, 2])
def test_random_sparse_blocks_through_the_compact_stream(ctx, seed):
    """Blocks with 0..11 coefficients anywhere (non-corner unsafe samples take the record-scan path), zero blocks,
    DC-only blocks, DC differences of zero (quirk Q1 drops those blocks' AC terms: no records for them)."""
    rng = np.random.default_rng(seed)
    w, h = 192, 40
    nmcu = (w // 8) * (h // 8)
    coef = np.zeros((nmcu, 3, 64), np.int16)
    for b in range(nmcu * 3):
        n = rng.integers(0, 12)
        pos = rng.choice(np.arange(1, 64), size=n, replace=False)
        coef.reshape(-1, 64)[b, pos] = rng.integers(-40, 41, size=n)
    dc = rng.integers(-120, 121, size=(nmcu, 3))
    dc[rng.random((nmcu, 3)) < 0.3] = 7      # runs of equal DC values: coded difference 0
    coef[:, :, 0] = dc
    qt = rng.integers(1, 64, size=(2, 64)).astype(np.uint16)
    _encode_and_check(ctx, coef.reshape(nmcu, 3, 8, 8), qt, w, h, "random sparse")


def test_batch_through_the_compact_stream(ctx):
    """The fused batch (images = restart segments of one virtual stream) with the compact stream: tiles, DC array and
    record ordinals run across image boundaries."""
    import torch
    frames, scans, wants = None, [], []
    for i in range(5):
        data = T.synth_jpeg(192, 72, seed=100 + i, sigma=4.0 + i)
        st, want = T.oracle_decode(data)
        p = T.oracle_parse(data)
        frames = T.make_frame(p)
        scans.append(torch.frombuffer(bytearray(p.scan), dtype=torch.uint8).cuda())
        wants.append(want)
    for layout in (2, 1):
        assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, layout) == 0
        outs = [torch.zeros((72, 192, 3), dtype=torch.uint8, device="cuda") for _ in scans]
        torch.cuda.synchronize()
        ctx.decode_batch_dev(frames, [t.data_ptr() for t in scans], [t.numel() for t in scans], [t.data_ptr() for t in outs])
        ctx.sync()
        for i, o in enumerate(outs):
            assert np.array_equal(o.cpu().numpy(), wants[i]), (layout, i)
    ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 0)


def test_corrupt_streams_stay_inside_the_buffers(ctx):
    """Corrupted scans through the compact path: every decode reports KPEG_HIP_E_STREAM or succeeds, nothing faults, and
    the clean stream decodes correctly afterwards (the first-record table of a tile a corrupt stream never starts reads
    as empty; record counts bound what a lane may write)."""
    import libkpeg_amd as K
    data = T.synth_jpeg(256, 128, seed=77, sigma=12.0)
    st, want = T.oracle_decode(data)
    p = T.oracle_parse(data)
    frame = T.make_frame(p)
    clean = np.frombuffer(p.scan, dtype=np.uint8)
    rng = np.random.default_rng(7)
    assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 2) == 0
    failed = 0
    for case in range(80):
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
            ctx.decode_scan(frame, s)
        except K.KpegError as e:
            assert e.code == K.E_STREAM, e
            failed += 1
    assert failed > 10
    assert np.array_equal(ctx.decode_scan(frame, p.scan), want)
    ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 0)


def test_huge_blocks_split_over_lanes_and_workgroups_after_another_picture(ctx):
    """Blocks whose bound is +inf (A >= 4000) and which are split over sub-sequences and workgroups, on the one-kernel
    path (k_sync_write makes no presets): their bound must be written by whoever settles the block, not left as the
    previous picture of the same size wrote it.  The picture before has the same blocks WITHOUT the huge terms, i.e.
    finite bounds in exactly those slots."""
    w, h = 2048, 1024
    nmcu = (w // 8) * (h // 8)
    base = T.synth_jpeg(w, h, seed=31, quality=75, sigma=6.0)
    p0 = T.oracle_parse(base)
    rc, coef = T.oracle_entropy(p0)
    assert rc == 0
    zz = T.zz_table()
    qn = np.zeros((2, 64), np.uint16)
    qn[:, zz] = p0.qt[:2]
    qn[:, zz[5:10]] = 255                      # the five zig-zag positions the huge terms use
    coef = coef.copy()
    coef[:, :, 5:10] = 0
    plain = T.encode_coefs(coef, w, h, qn[0], qn[1])
    rng = np.random.default_rng(9)
    big = coef.copy()
    sel = np.arange(1, nmcu, 3)
    big[sel, 0, 5:9] = rng.choice(np.array([-1000, -700, 700, 1000], np.int16), size=(sel.size, 4))
    huge = T.encode_coefs(big, w, h, qn[0], qn[1])
    assert len(T.oracle_parse(huge).scan) * 8 < 3 * w * h, "must stay on the sparse (96-bit) path"
    st0, want0 = T.oracle_decode(plain)
    st1, want1 = T.oracle_decode(huge)
    assert st0 == T.DECODE_DONE and st1 == T.DECODE_DONE
    pp, ph = T.oracle_parse(plain), T.oracle_parse(huge)
    assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 2) == 0
    try:
        for fused in (1, 0):
            assert ctx.lib.kpeg_hip_debug_set(ctx._h, 9, fused) == 0
            for rep in range(2):
                assert np.array_equal(ctx.decode_scan(T.make_frame(pp), pp.scan), want0), (fused, rep, "plain")
                got = ctx.decode_scan(T.make_frame(ph), ph.scan)
                if fused:
                    assert int(ctx.timings()["sync_rounds"]) == 1, "k_sync_write did not finish this call itself"
                bad = np.argwhere(got != want1)
                assert bad.size == 0, "fused %d rep %d: first mismatches (y,x,c) %s of %d" % (fused, rep, bad[:8].tolist(), len(bad))
    finally:
        ctx.lib.kpeg_hip_debug_set(ctx._h, 9, 1)
        ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 0)


@pytest.mark.parametrize("warm", [0, 1, 3])
def test_workgroups_whose_entry_state_is_wrong_repair_themselves(ctx, warm):
    """k_sync_write with K1's lead-in cut to 0, 1 or 3 sub-sequences (debug key 2): nearly every workgroup then starts its first
    own sub-sequence from a state that is not the one the workgroup before ends in.  Each must find that out from its
    predecessor's published exit state, decode again from the right one as far as the change reaches, publish its totals a
    second time, and its successors must take those: the call is finished by that kernel alone (one launch of K1 with work)
    and every pixel is the oracle's.  Synthetic field (short re-synchronisation) and a photograph (long)."""
    import libkpeg_amd as K
    cases = [("synthetic 2048x1024", T.synth_jpeg(2048, 1024, seed=5, quality=75, sigma=6.0))]
    try:
        import PIL  # noqa: F401
        sys.path.insert(0, T.ROOT)
        import bench
        cases.append(("lena tiled to 2048x1024, q50", bench.tiled_photo_jpeg("lena.jpg", 50, 2048, 1024)))
        cases.append(("lena tiled to 2048x1024, q75", bench.tiled_photo_jpeg("lena.jpg", 75, 2048, 1024)))
    except ImportError:
        pass
    try:
        assert ctx.lib.kpeg_hip_debug_set(ctx._h, 2, warm) == 0
        assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 2) == 0   # the compact stream (these pictures are below the size that takes it by itself)
        for what, data in cases:
            st, want = T.oracle_decode(data)
            assert st == T.DECODE_DONE
            rc, frame, scan = K.host_parse(data)
            assert rc == K.DECODE_DONE
            for fused in (1, 0):
                assert ctx.lib.kpeg_hip_debug_set(ctx._h, 9, fused) == 0
                for rep in range(2):
                    ctx.set_profiling(True)
                    got = ctx.decode_scan(frame, scan)
                    rounds = int(ctx.timings()["sync_rounds"])
                    ctx.set_profiling(False)
                    bad = np.argwhere(got != want)
                    assert bad.size == 0, "%s warm %d fused %d: first mismatches (y,x,c) %s of %d" % (what, warm, fused, bad[:8].tolist(), len(bad))
                    if fused and len(scan) * 8 < 2.5 * frame.width * frame.height and len(scan) > 8 * 501 * 12:
                        assert rounds == 1, "%s warm %d: k_sync_write did not finish this call itself (%d launches of K1 had work)" % (what, warm, rounds)
    finally:
        ctx.lib.kpeg_hip_debug_set(ctx._h, 2, -1)
        ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 0)
        ctx.lib.kpeg_hip_debug_set(ctx._h, 9, 1)


def test_the_second_launch_of_k_sync_write_finishes_a_call_the_first_gives_up(ctx):
    """Since its workgroups repair themselves k_sync_write gives a call up only when a stream does not re-synchronise inside a
    workgroup or a wait expires -- nothing a test picture does.  Fault bit 3 (debug key 6, value 8) makes every third workgroup
    give up behind its hand-over, when some of its neighbours have written their coefficients and others have not: the kernel's
    second, strict launch must then decode the call (two launches of K1 with work), pixels the oracle's, and the call after it,
    without the fault, is the first launch's alone again.  With the short and the long sub-sequences, and with K1's lead-in cut
    to nothing, so that the results the first launch leaves are wrong at every workgroup's edge and the strict chain has to
    decode again wherever it goes.  A truncated stream must fail the same way through either."""
    import libkpeg_amd as K
    cases = [("synthetic 2048x1024", T.synth_jpeg(2048, 1024, seed=6, quality=75, sigma=6.0)),
             ("synthetic 4096x2048", T.synth_jpeg(4096, 2048, seed=7, quality=75, sigma=6.0))]
    try:
        import PIL  # noqa: F401
        sys.path.insert(0, T.ROOT)
        import bench
        cases.append(("lena tiled to 2048x1024, q75", bench.tiled_photo_jpeg("lena.jpg", 75, 2048, 1024)))
    except ImportError:
        pass
    try:
        assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 2) == 0
        for what, data in cases:
            st, want = T.oracle_decode(data)
            assert st == T.DECODE_DONE
            rc, frame, scan = K.host_parse(data)
            assert rc == K.DECODE_DONE
            for subseq in (0, 64, 384):
                for warm in (-1, 0):
                    assert ctx.lib.kpeg_hip_debug_set(ctx._h, 4, subseq) == 0
                    assert ctx.lib.kpeg_hip_debug_set(ctx._h, 2, warm) == 0
                    for fault, rounds_ok in ((8, lambda r: r == 2), (0, lambda r: r == 1), (8, lambda r: r == 2)):
                        assert ctx.lib.kpeg_hip_debug_set(ctx._h, 6, fault) == 0
                        ctx.set_profiling(True)
                        got = ctx.decode_scan(frame, scan)
                        rounds = int(ctx.timings()["sync_rounds"])
                        ctx.set_profiling(False)
                        bad = np.argwhere(got != want)
                        assert bad.size == 0, "%s subseq %d warm %d fault %d: first mismatches (y,x,c) %s of %d" % (what, subseq, warm, fault, bad[:8].tolist(), len(bad))
                        assert rounds_ok(rounds), (what, subseq, warm, fault, rounds)
            # a truncated stream: the same answer with and without the fault
            ctx.lib.kpeg_hip_debug_set(ctx._h, 4, 0)
            ctx.lib.kpeg_hip_debug_set(ctx._h, 2, -1)
            cut = scan[:len(scan) * 2 // 3]
            codes = []
            for fault in (0, 8):
                assert ctx.lib.kpeg_hip_debug_set(ctx._h, 6, fault) == 0
                try:
                    ctx.decode_scan(frame, cut)
                    codes.append(0)
                except K.KpegError as e:
                    codes.append(e.code)
            assert codes[0] == codes[1] == K.E_STREAM, (what, codes)
    finally:
        ctx.lib.kpeg_hip_debug_set(ctx._h, 6, 0)
        ctx.lib.kpeg_hip_debug_set(ctx._h, 4, 0)
        ctx.lib.kpeg_hip_debug_set(ctx._h, 2, -1)
        ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 0)
