"""Error reporting and hang protection of the device path: the status word is sticky until kpeg_hip_sync(), waits
between workgroups are bounded, a DC value outside the int16 coefficient layout is reported, switching streams
keeps the scratch buffers ordered."""
import os
import subprocess
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


def _case(w=512, h=256, seed=41):
    data = T.synth_jpeg(w, h, seed=seed, sigma=10.0)
    st, want = T.oracle_decode(data)
    assert st == T.DECODE_DONE
    p = T.oracle_parse(data)
    return T.make_frame(p), np.frombuffer(p.scan, np.uint8).copy(), want


def test_error_of_an_earlier_enqueued_call_survives_until_sync(ctx):
    """Several *_dev calls enqueued before one kpeg_hip_sync(): a corrupt stream in the FIRST of them must still be
    reported (the device-side error word is sticky until a sync has seen it), and the context works afterwards."""
    import torch
    import libkpeg_amd as K
    frame, scan, want = _case()
    bad = scan[: scan.size // 3].copy()          # truncated: the segment's last blocks are missing
    d_good = torch.from_numpy(scan).cuda()
    d_bad = torch.from_numpy(bad).cuda()
    out = [torch.zeros(want.shape, dtype=torch.uint8, device="cuda") for _ in range(4)]
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.decode_scan_dev(frame, d_bad.data_ptr(), d_bad.numel(), out[0].data_ptr())
    for k in (1, 2, 3):
        ctx.decode_scan_dev(frame, d_good.data_ptr(), d_good.numel(), out[k].data_ptr())
    with pytest.raises(K.KpegError) as ei:
        ctx.sync()
    assert ei.value.code == K.E_STREAM
    for k in (1, 2, 3):   # the clean calls behind the corrupt one decoded correctly all the same
        assert np.array_equal(out[k].cpu().numpy(), want), k
    # the flag has been seen: it is cleared before the next call
    ctx.decode_scan_dev(frame, d_good.data_ptr(), d_good.numel(), out[0].data_ptr())
    ctx.sync()
    assert np.array_equal(out[0].cpu().numpy(), want)


def test_k0_look_back_falls_back_when_a_predecessor_never_publishes(ctx):
    """Fault injection: K0's first workgroup never publishes its prefix (as if it had not been dispatched yet).  Its
    successors must not wait for it: after a short bound they compute its aggregate from its input bytes themselves
    (look-back with fallback) -- the decode is correct and nothing hangs, whatever the dispatch order."""
    frame, scan, want = _case()
    assert scan.size > 3 * 8192   # several K0 workgroups (8 KiB of scan each)
    assert ctx.lib.kpeg_hip_debug_set(ctx._h, 6, 1) == 0
    try:
        for bound_us in (0, 200):   # the default bound and a longer one
            assert ctx.lib.kpeg_hip_debug_set(ctx._h, 5, bound_us) == 0
            assert np.array_equal(ctx.decode_scan(frame, scan), want), bound_us
        # restart markers: the segment offsets come from the same prefixes
        data = T.synth_jpeg(512, 256, seed=44, sigma=10.0, restart_interval=7)
        want_r, p, _ = T.oracle_decode_rst(data, 7)
        assert np.array_equal(ctx.decode_scan(T.make_frame(p, 7), p.scan), want_r)
    finally:
        ctx.lib.kpeg_hip_debug_set(ctx._h, 5, 0)
        ctx.lib.kpeg_hip_debug_set(ctx._h, 6, 0)
    assert np.array_equal(ctx.decode_scan(frame, scan), want)


def test_k0_look_back_fallback_on_every_kind_of_chunk(ctx):
    """Fault injection: no K0 workgroup publishes its aggregate ahead of its prefix, and the bound is 1 us: look-backs
    compute many predecessors' aggregates from the input bytes themselves -- first, middle and last chunks, restart
    markers on chunk edges, the images of a fused batch.  The results must not change."""
    import torch
    assert ctx.lib.kpeg_hip_debug_set(ctx._h, 6, 4) == 0
    assert ctx.lib.kpeg_hip_debug_set(ctx._h, 5, 1) == 0
    try:
        for (w, h, interval, seed) in ((1024, 512, 0, 51), (1024, 512, 128, 52), (2048, 256, 3, 53), (512, 512, 1, 54)):
            data = T.synth_jpeg(w, h, seed=seed, sigma=10.0, restart_interval=interval)
            if interval:
                want, p, _ = T.oracle_decode_rst(data, interval)
            else:
                st, want = T.oracle_decode(data)
                p = T.oracle_parse(data)
            assert len(p.scan) > 5 * 8192
            got = ctx.decode_scan(T.make_frame(p, interval), p.scan)
            assert np.array_equal(got, want), (w, h, interval)
        frames, scans, wants = None, [], []
        for i in range(6):
            data = T.synth_jpeg(320, 200, seed=70 + i, sigma=12.0)
            st, want = T.oracle_decode(data)
            p = T.oracle_parse(data)
            frames = T.make_frame(p)
            scans.append(torch.frombuffer(bytearray(p.scan), dtype=torch.uint8).cuda())
            wants.append(want)
        outs = [torch.zeros((200, 320, 3), dtype=torch.uint8, device="cuda") for _ in scans]
        torch.cuda.synchronize()
        ctx.decode_batch_dev(frames, [t.data_ptr() for t in scans], [t.numel() for t in scans], [t.data_ptr() for t in outs])
        ctx.sync()
        for i, o in enumerate(outs):
            assert np.array_equal(o.cpu().numpy(), wants[i]), i
    finally:
        ctx.lib.kpeg_hip_debug_set(ctx._h, 5, 0)
        ctx.lib.kpeg_hip_debug_set(ctx._h, 6, 0)


_CHAIN = r"""
import sys
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import numpy as np, kpeg_testlib as T, libkpeg_amd as K
ctx = K.Context(0)
assert ctx.lib.kpeg_hip_debug_set(ctx._h, 4, 64) == 0
assert ctx.lib.kpeg_hip_debug_set(ctx._h, 2, 0) == 0      # no warm-up: every workgroup guesses wrong, the chained pass ripples
assert ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 1) == 0      # the dense layout: the separate launches (small pictures take the one kernel otherwise)
data = T.synth_jpeg(1024, 512, seed=21, quality=95, sigma=0.0, mode=1)
st, want = T.oracle_decode(data)
p = T.oracle_parse(data)
frame = T.make_frame(p)
assert np.array_equal(ctx.decode_scan(frame, p.scan), want)
assert int(ctx.timings()["sync_rounds"]) >= 3, "the chained pass did not ripple"
assert ctx.lib.kpeg_hip_debug_set(ctx._h, 5, 3000) == 0    # 3 ms
assert ctx.lib.kpeg_hip_debug_set(ctx._h, 6, 2) == 0       # the chained pass's first workgroup never publishes
try:
    ctx.decode_scan(frame, p.scan)
    print("NO_ERROR")
except K.KpegError as e:
    print("CODE", e.code)
ctx.lib.kpeg_hip_debug_set(ctx._h, 5, 0)
ctx.lib.kpeg_hip_debug_set(ctx._h, 6, 0)
assert np.array_equal(ctx.decode_scan(frame, p.scan), want)
print("RECOVERED")
"""


def test_chained_pass_wait_is_bounded():
    """The same for K1's chained pass, on the stress build (tiny workgroups: the only geometry that reaches a rippling
    chained pass): workgroup 0 never sets its done flag, the others time out, the call reports KPEG_HIP_E_DEVICE and
    the context decodes correctly afterwards."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "libkpeg_amd", "libkpeg_hip_stress.so")
    assert os.path.exists(lib), "run libkpeg_amd.build.build_all()"
    env = dict(os.environ, KPEG_HIP_LIB=lib)
    out = subprocess.run([sys.executable, "-c", _CHAIN % {"tests": os.path.join(root, "tests"), "root": root}], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "CODE -2" in out.stdout and "RECOVERED" in out.stdout, out.stdout


def _canonical_codes(counts, symbols):
    codes, code, k = {}, 0, 0
    for ln in range(1, 17):
        for _ in range(counts[ln - 1]):
            codes[symbols[k]] = (code, ln)
            code += 1
            k += 1
        code <<= 1
    return codes


def test_dc_value_outside_int16_is_reported(ctx):
    """Every block's DC difference is +2047 (category 11, the largest a baseline DC table codes): after 17 blocks of a
    component the predictor (an int in the reference, MCU.cpp:107-112) no longer fits the int16 coefficient layout.
    The stream is valid Huffman data; the call must report KPEG_HIP_E_STREAM instead of wrapping silently."""
    import libkpeg_amd as K
    w, h = 160, 8   # 20 MCUs
    p = T.oracle_parse(T.synth_jpeg(w, h, seed=3))
    frame = T.make_frame(p)
    bits = []
    for mcu in range(20):
        for c in range(3):
            dc = _canonical_codes(*p.dht[0][1 if c else 0][:2])
            ac = _canonical_codes(*p.dht[1][1 if c else 0][:2])
            code, ln = dc[11]
            bits += [(code >> (ln - 1 - i)) & 1 for i in range(ln)] + [1] * 11
            code, ln = ac[0]
            bits += [(code >> (ln - 1 - i)) & 1 for i in range(ln)]
    per_mcu = len(bits) // 20
    bits += [1] * (-len(bits) % 8)
    raw = np.packbits(np.array(bits, np.uint8)).tobytes()
    scan = raw.replace(b"\xff", b"\xff\x00")
    with pytest.raises(K.KpegError) as ei:
        ctx.decode_scan(frame, np.frombuffer(scan, np.uint8).copy())
    assert ei.value.code == K.E_STREAM
    # 15 MCUs of the same stream stay inside the range (15 * 2047 = 30705): accepted
    frame.width = 120
    ok_bits = bits[: 15 * per_mcu]
    ok_bits += [1] * (-len(ok_bits) % 8)
    ok = np.packbits(np.array(ok_bits, np.uint8)).tobytes().replace(b"\xff", b"\xff\x00")
    out = ctx.decode_scan(frame, np.frombuffer(ok, np.uint8).copy())
    assert out.shape == (8, 120, 3)


def test_switching_streams_keeps_the_scratch_ordered(ctx):
    """kpeg_hip_set_stream between calls: the new stream waits for what is queued on the old one (the scratch buffers
    are shared), so alternating between two streams without a sync in between still decodes correctly."""
    import torch
    frame, scan, want = _case(1024, 512, seed=43)
    d_scan = torch.from_numpy(scan).cuda()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.zeros(want.shape, dtype=torch.uint8, device="cuda") for _ in range(6)]
    torch.cuda.synchronize()
    for k in range(6):
        ctx.set_stream(streams[k & 1].cuda_stream)
        ctx.decode_scan_dev(frame, d_scan.data_ptr(), d_scan.numel(), outs[k].data_ptr())
    ctx.sync()
    torch.cuda.synchronize()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for k in range(6):
        assert np.array_equal(outs[k].cpu().numpy(), want), k


def test_one_kernel_path_beside_another_context():
    """k_sync_write waits for workgroups with smaller indices: that must hold up when another context's kernels share the chip
    and its grid is not resident all at once.  Two contexts on two streams, two pictures, four calls in flight each, every
    output compared (tools/fused_two_streams.py is the long version)."""
    import torch
    import libkpeg_amd as K
    pics = [T.synth_jpeg(2560, 1472, seed=31, quality=75, sigma=6.0), T.synth_jpeg(1920, 1088, seed=32, quality=80, sigma=4.0)]
    ctxs, frames, bufs, outs, want = [], [], [], [], []
    for d in pics:
        p = T.oracle_parse(d)
        st, w = T.oracle_decode(d)
        c = K.Context(0)
        assert c.lib.kpeg_hip_debug_set(c._h, 7, 2) == 0     # the compact stream, so that the one-kernel path applies
        stream = torch.cuda.Stream()
        c.set_stream(stream.cuda_stream)
        ctxs.append((c, stream))
        frames.append(T.make_frame(p))
        bufs.append(torch.frombuffer(bytearray(p.scan), dtype=torch.uint8).cuda())
        outs.append([torch.zeros(w.shape, dtype=torch.uint8, device="cuda") for _ in range(4)])
        want.append(w)
    torch.cuda.synchronize()
    for rnd in range(6):
        for k in range(4):
            for i, (c, _) in enumerate(ctxs):
                c.decode_scan_dev(frames[i], bufs[i].data_ptr(), bufs[i].numel(), outs[i][k].data_ptr())
        for c, _ in ctxs:
            c.sync()
        torch.cuda.synchronize()
        for i in range(2):
            for k in range(4):
                assert np.array_equal(outs[i][k].cpu().numpy(), want[i]), (rnd, i, k)
                outs[i][k].zero_()
        torch.cuda.synchronize()   # (the zeroing runs on torch's stream, the decodes on their own)
    assert [int(c.timings()["sync_rounds"]) for c, _ in ctxs] == [1, 1], "the one-kernel path was not taken"
