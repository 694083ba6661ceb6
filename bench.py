#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native libKPEG decode path.

Metric (BASELINE.json): Mpixels/s decoded, JFIF entropy-coded segment resident in HBM ->
RGB8 resident in HBM, on the synthetic 7680x4320 4:4:4 baseline JPEG of SURVEY.md 8(d), plus
the achieved HBM GB/s of the IDCT+colour kernel (K4) against the MI355X peak.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one full on-device decode (K0 unstuff, K1 sync, scan, K2 write, K4 IDCT+colour).
  N = 1   the headline: one 7680x4320 reference-compatible image (no restart markers), BASELINE config 3.
  N > 1   BASELINE config 5, strong scaling: ONE 16384x16384 image with a restart interval per MCU row; rank r
          decodes MCU rows [2048 r / N, 2048 (r+1) / N) from the bytes of its own restart intervals, no data-path
          collective; `value` = the whole image's pixels / the slowest rank's time, stripes resident in the HBM of
          the GPU that decoded them.  The gather of the stripes to rank 0 (the path's one exchange step: RCCL
          send/recv over xGMI, overlapped with the decode band by band) is timed as decode + gather in the
          "gather" object and `value_incl_gather`.  (--weak: the former weak-scaling mode, a 7680x4320 stripe per
          rank; --image16k: config 5's image on one GPU.)
Every rank hashes the pixels its timed loop produced against SHA-256s of libKPEG's own decoder's output
(tests/golden/manifest_large.json): "verified".

The JSON line also carries
  roofline     K4: 9 algorithmic bytes per pixel (6 B int16 coefficients in + 3 B RGB out)
               x pixels / K4's mean duration, measured with HIP events on the launch stream.
  cpu_baseline the real reference decoder (oracle/_ref/kpeg_ref, built from /root/reference in
               the build container; kind "reference") or, if that binary is absent, the CPU
               restatement (kind "port"), timed on a bounded crop of the same synthetic field.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W8K, H8K = 7680, 4320
SEED, QUALITY, SIGMA = 1234, 75, 6.0
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def _synth():
    so = os.path.join(ROOT, "tools", "libkpeg_synth.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools"), "all"])
    S = ctypes.CDLL(so)
    S.kpeg_synth_jpeg_rows.restype = ctypes.c_size_t
    S.kpeg_synth_jpeg_rows.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int,
                                       ctypes.c_uint32, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
    return S


def synth_jpeg(w, h, y0=0, restart_interval=0, seed=SEED, quality=QUALITY, sigma=SIGMA, mode=0):
    cap = w * h * 3 + (w * h) // 2 + 65536
    buf = np.empty(cap, np.uint8)
    n = _synth().kpeg_synth_jpeg_rows(w, h, y0, seed, quality, restart_interval, sigma, mode, buf.ctypes.data, cap)
    assert n > 0
    return buf[:n].tobytes()


# Natural content beside the synthetic field: the committed photographs (tests/golden: the reference's own lena.jpg and a Pillow-encoded
# sample photograph), tiled to 7680 x 4352 and re-encoded by the integer-only test encoder (tools/kpeg_synth.c) -- regenerable byte for
# byte wherever Pillow decodes the source files to the same pixels; tests/golden/make_golden_photos.py pinned their pixels to libKPEG's
# own decoder (manifest_large.json: "natural_8k").
PHOTO_CASES = [("lena.jpg", 50), ("lena.jpg", 75), ("nat_china_640x424_q90.jpg", 75), ("nat_china_640x424_q90.jpg", 90)]
PHOTO_W, PHOTO_H = 7680, 4352


def tiled_photo_jpeg(src, quality, w=PHOTO_W, h=PHOTO_H):
    from PIL import Image
    im = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", src)).convert("RGB"))
    big = np.ascontiguousarray(np.tile(im, (h // im.shape[0] + 1, w // im.shape[1] + 1, 1))[:h, :w])
    S = _synth()
    S.kpeg_synth_encode_rgb.restype = ctypes.c_size_t
    S.kpeg_synth_encode_rgb.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t]
    cap = w * h * 4 + 65536
    buf = np.empty(cap, np.uint8)
    n = S.kpeg_synth_encode_rgb(big.ctypes.data, w, h, quality, 0, buf.ctypes.data, cap)
    assert n > 0
    return buf[:n].tobytes()


def pinned_rgb_sha(w, h, world, rank, restart_stripe):
    """SHA-256 of the raw RGB bytes libKPEG's own decoder produces for this rank's part of the workload, if pinned."""
    mf = os.path.join(ROOT, "tests", "golden", "manifest_large.json")
    if not os.path.exists(mf):
        return None
    m = json.load(open(mf))
    if world == 1 and not restart_stripe:
        g = m.get("synth", {}).get("%dx%d_seed%d" % (w, h, SEED))
        return g["rgb_sha256"] if g and (g["quality"], g["sigma"]) == (QUALITY, SIGMA) else None
    return None


def pinned_stripe_shas(iw, ih, world, rank):
    """SHA-256s of the eighth-stripes of the 16384x16384 restart-interval image that make up this rank's rows."""
    mf = os.path.join(ROOT, "tests", "golden", "manifest_large.json")
    if not os.path.exists(mf) or 8 % world:
        return None
    g = json.load(open(mf)).get("dri16k")
    if not g or (g["width"], g["height"], g["seed"], g["quality"], g["sigma"]) != (iw, ih, SEED, QUALITY, SIGMA):
        return None
    per = 8 // world
    return g["stripe8_rgb_sha256"][rank * per:(rank + 1) * per]


def _port_decode(data, nthreads):
    """The CPU restatement (oracle/kpeg_oracle.c: the checker) as a timed baseline; returns seconds."""
    so = os.path.join(ROOT, "oracle", "libkpeg_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])
    L = ctypes.CDLL(so)
    L.kpeg_oracle_decode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)),
                                     ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]
    rgb = ctypes.POINTER(ctypes.c_uint8)()
    w, h = ctypes.c_uint32(), ctypes.c_uint32()
    t0 = time.perf_counter()
    st = L.kpeg_oracle_decode(data, len(data), ctypes.byref(rgb), ctypes.byref(w), ctypes.byref(h), nthreads)
    dt = time.perf_counter() - t0
    assert st == 4
    ctypes.CDLL(None).free(rgb)
    return dt


def cpu_baseline(sample_w=3840, sample_h=2160):
    """The CPU beside the GPU number, on a bounded crop (top-left sample_w x sample_h of the 8K field):
    `value` = libKPEG's own decoder, one thread as shipped (oracle/_ref, kind "reference"; the CPU restatement if that
    binary is absent, kind "port"); `all_cores` = the restatement with OpenMP over every host core (the reference is
    single-threaded; its per-MCU work is what the port spreads over threads), core count stated."""
    data = synth_jpeg(sample_w, sample_h)
    mp = sample_w * sample_h / 1e6
    ref = os.path.join(ROOT, "oracle", "_ref", "kpeg_ref")
    sample = "top-left %dx%d crop of the 7680x4320 synthetic field (seed %d, q%d), %.2f Mpixel" % (
        sample_w, sample_h, SEED, QUALITY, mp)
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    _port_decode(synth_jpeg(512, 512), ncores)   # thread pool up
    dt_all = min(_port_decode(data, ncores) for _ in range(2))
    all_cores = {"value": round(mp / dt_all, 3), "unit": "Mpixels/s", "cores": ncores, "kind": "port",
                 "what": "oracle/kpeg_oracle.c (bit-exact restatement: cos table precomputed, zero terms skipped), OpenMP over the MCU rows"}
    out = None
    if os.path.exists(ref):
        d = tempfile.mkdtemp(prefix="kpegbench")
        f = os.path.join(d, "sample.jpg")
        with open(f, "wb") as fh:
            fh.write(data)
        res = subprocess.run([ref, "decode", f], capture_output=True, text=True, timeout=600)
        try:
            info = json.loads(res.stdout.strip().splitlines()[-1])
            if info.get("status") == "DECODE_DONE":
                out = {"value": round(info["mpix_per_s"], 4), "unit": "Mpixels/s", "cores": 1, "kind": "reference",
                       "sample": sample + "; libKPEG's own decoder (oracle/_ref), 1 thread as shipped"}
        except Exception:
            pass
        finally:
            import shutil
            shutil.rmtree(d, ignore_errors=True)
    if out is None:
        # the reference binary did not travel: the CPU restatement on one thread (still only a baseline, never the product path)
        dt = _port_decode(data, 1)
        out = {"value": round(mp / dt, 4), "unit": "Mpixels/s", "cores": 1, "kind": "port",
               "sample": sample + "; oracle/kpeg_oracle.c (cos table precomputed, zero terms skipped), 1 thread"}
    out["all_cores"] = all_cores
    try:
        # the whole 7680x4320 file through libKPEG's own decoder, measured once where the reference is (tests/golden/make_golden_large.py)
        m = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest_large.json")))["synth"]["%dx%d_seed%d" % (W8K, H8K, SEED)]
        out["reference_full_image"] = {"seconds": m["ref_decode_s"], "value": round(W8K * H8K / 1e6 / m["ref_decode_s"], 4), "unit": "Mpixels/s", "cores": 1,
                                       "what": "the headline file itself, decoded by libKPEG in the build container when the golden hashes were made (a committed measurement, not taken in this run)"}
    except Exception:
        pass
    return out


def bench_batch(args, torch, K):
    """BASELINE config 4 (throughput mode): a batch of independent 1080p images resident in HBM."""
    n, w, h = args.batch, 1920, 1080
    uniq = min(n, 32)   # distinct images (host-side synthesis takes ~0.1 s each); the batch cycles through them
    frames, scans = None, []
    for i in range(uniq):
        rc, frame, scan = K.host_parse(synth_jpeg(w, h, seed=SEED + i))
        assert rc == K.DECODE_DONE, rc
        frames = frame
        scans.append(torch.from_numpy(np.ascontiguousarray(scan)).cuda())
    d_scans = [scans[i % uniq] for i in range(n)]
    d_rgbs = [torch.empty((h, w, 3), dtype=torch.uint8, device="cuda") for _ in range(n)]
    ctx = K.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    sp, sl, op = [t.data_ptr() for t in d_scans], [t.numel() for t in d_scans], [t.data_ptr() for t in d_rgbs]

    def lanes():
        ctx.decode_batch_dev(frames, sp, sl, op)

    def one_stream():
        for i in range(n):
            ctx.decode_scan_dev(frames, sp[i], sl[i], op[i])

    res = {}
    for name, fn in (("lanes", lanes), ("one_stream", one_stream)):   # "lanes": historical name of the batch entry point's leg
        for _ in range(args.warmup):
            fn()
        ctx.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        ctx.sync()
        res[name] = dt
        if name == "lanes":
            # the pixels of the timed batch against libKPEG's own decoder (tests/golden/manifest_large.json): the distinct
            # images by SHA-256, every repeat against its first copy on the device
            import hashlib
            mf = os.path.join(ROOT, "tests", "golden", "manifest_large.json")
            man = json.load(open(mf)).get("synth", {}) if os.path.exists(mf) else {}
            verified = True
            for i in range(uniq):
                g = man.get("%dx%d_seed%d" % (w, h, SEED + i))
                if not g or (g["quality"], g["sigma"]) != (QUALITY, SIGMA):
                    verified = None
                    break
                if hashlib.sha256(d_rgbs[i].cpu().numpy().tobytes()).hexdigest() != g["rgb_sha256"]:
                    raise SystemExit("bench.py --batch: image %d differs from the reference's pixels (SHA-256 mismatch)" % i)
            if verified:
                for i in range(uniq, n):
                    if not torch.equal(d_rgbs[i], d_rgbs[i % uniq]):
                        raise SystemExit("bench.py --batch: image %d differs from its first copy" % i)
    mp = n * w * h / 1e6
    print(json.dumps({
        "metric": "Mpixels/s decoded (JFIF->RGB), batch of 1080p 4:4:4 baseline", "value": round(mp / res["lanes"], 2), "unit": "Mpixels/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(res["lanes"] * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "batch of %d synthetic 1920x1080 4:4:4 baseline JPEGs q%d (%d distinct, seeds %d..), resident in HBM, "
                               "kpeg_hip_decode_batch_dev (the batch as restart segments of one virtual stream: one set of launches)" % (n, QUALITY, uniq, SEED), "images_per_step": n},
        "verified": verified, "images_per_s": round(n / res["lanes"], 1), "us_per_image": round(res["lanes"] / n * 1e6, 2),
        "one_stream": {"value": round(mp / res["one_stream"], 2), "us_per_image": round(res["one_stream"] / n * 1e6, 2)},
    }), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a timed region of ~40 ms.  Twenty steps (4 ms) end before the shader clock has settled under the load and
    # read 3-5 % low (0.207 against 0.198 ms per step, same build, same box); the whole default run still takes seconds.
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=None, help="image width (default: 7680, or 16384 for the sharded image)")
    ap.add_argument("--height", type=int, default=None, help="image height (default: 4320, or 16384 for the sharded image; --weak: rows per GPU)")
    ap.add_argument("--weak", action="store_true",
                    help="N>1: weak scaling instead of BASELINE config 5 -- every rank decodes a 7680x4320 stripe of a 7680 x (4320 N) image")
    ap.add_argument("--image16k", action="store_true",
                    help="N=1: decode config 5's 16384x16384 restart-interval image on one GPU (the N>1 runs' workload) instead of the 8K headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-photographs", action="store_true",
                    help="skip the natural-content figure (four committed photographs tiled to 7680x4352; ~15 s of host-side encoding)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--rehearse", action="store_true",
                    help="N>1 dry run on a one-GPU box: gloo backend, every rank on cuda:0 (the driver's runs use RCCL, one GPU per rank)")
    ap.add_argument("--idct-mode", type=int, default=0, help="kpeg_hip_set_idct_mode (2 = timing experiment, wrong pixels)")
    ap.add_argument("--idct-only", action="store_true", help="time K4 alone on resident coefficients (BASELINE config 2 style)")
    ap.add_argument("--restart-stripe", action="store_true",
                    help="N=1 only: decode the workload the N>1 ranks get (restart interval = one MCU row, DRI extension) instead of "
                         "the reference-compatible stream; tells what part of the N>1 per-rank time the restart machinery costs")
    ap.add_argument("--cold", action="store_true",
                    help="with --idct-only: overwrite a 1 GiB scratch buffer before every step, so that K4 finds neither its "
                         "coefficients nor its output lines in L2 / Infinity Cache (the kernel's own events exclude the fill)")
    ap.add_argument("--side-figures", action="store_true",
                    help="also report SURVEY 8(d)'s side figures: measured device-copy ceiling and the dense q95 noise stress input "
                         "(off by default so that a rocprofv3 summary of the default command holds the headline workload only)")
    ap.add_argument("--batch", type=int, default=0,
                    help="BASELINE config 4 instead of the headline: N synthetic 1920x1080 images (seeds 1234..), device-resident, "
                         "kpeg_hip_decode_batch_dev (fused launches) beside the one-stream loop; one step = the whole batch")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import libkpeg_amd as K

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path to measure")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # one explicit stream for torch's ops and the decoder's kernels alike (the default stream's handle is NULL, which the
    # C ABI reads as "the context's own stream": work on that would not be ordered with torch's)
    torch.cuda.set_stream(torch.cuda.Stream())
    if world > 1:
        if args.rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    coll_dev = "cpu" if args.rehearse else "cuda"

    if args.batch:
        return bench_batch(args, torch, K)

    # ---- workload --------------------------------------------------------------------------------------------
    #   N = 1            the headline: one 7680x4320 reference-compatible image (BASELINE config 3)
    #   N > 1            BASELINE config 5, strong scaling: ONE 16384x16384 image with a restart interval per MCU row,
    #                    rank r decodes MCU rows [r*2048/N, (r+1)*2048/N) from the bytes of its own restart intervals
    #   N > 1, --weak    every rank a 7680x4320 stripe of a 7680 x (4320 N) image
    # W x H is what THIS rank decodes, IW x IH the whole image.
    strong = (world > 1 and not args.weak) or args.image16k
    if strong:
        IW, IH = args.width or 16384, args.height or 16384
        if (IH // 8) % world:
            raise SystemExit("bench.py: %d MCU rows do not split evenly over %d ranks" % (IH // 8, world))
        W, H = IW, IH // world
    else:
        W, H = args.width or W8K, args.height or H8K
        IW, IH = W, H * world
    mw, mh = W // 8, H // 8
    if world == 1 and not args.restart_stripe and not strong:
        data = synth_jpeg(W, H)
        rc, frame, scan = K.host_parse(data)
        assert rc == K.DECODE_DONE, rc
        first_row, rows = 0, mh
    else:
        # restart interval = one MCU row; the stripe is generated directly (restart intervals are
        # independent), parsed with the DRI extension, decoded as rows [rank*mh, (rank+1)*mh) of the whole image
        data = synth_jpeg(W, H, y0=rank * H, restart_interval=mw)
        rc, frame, scan = K.host_parse(data, allow_dri=True)
        assert rc == K.DECODE_DONE, rc
        frame.height = IH
        first_row, rows = rank * mh, mh

    ctx = K.Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.set_idct_mode(args.idct_mode)
    d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda()
    d_rgb = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    d_coef = torch.empty(mw * mh * 192, dtype=torch.int16, device="cuda") if args.idct_only else None
    torch.cuda.synchronize()

    scratch = torch.empty(1 << 30, dtype=torch.uint8, device="cuda") if (args.idct_only and args.cold) else None

    def step():
        if args.idct_only:
            if scratch is not None:
                scratch.fill_(7)
            ctx.idct_colour_dev(frame1, d_coef.data_ptr(), d_rgb.data_ptr())
        else:
            ctx.decode_stripe_dev(frame, d_scan.data_ptr(), d_scan.numel(), first_row, rows, d_rgb.data_ptr())

    frame1 = None
    if args.idct_only:
        assert world == 1, "--idct-only is a single-GPU mode"
        frame1 = frame
        ctx.entropy_decode_dev(frame, d_scan.data_ptr(), d_scan.numel(), d_coef.data_ptr())
        ctx.sync()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.sync()
    # A short run (the driver's --steps 20 --warmup 5 is 4 ms of work) would be over before the shader clock has settled under the
    # load and read 3-5 % low.  The timed region stays `steps` steps and `warmup` stays what was asked for; what is missing to about
    # 25 ms of work before the timed region is spent on extra untimed steps (their count is reported as warmup_extra).
    warmup_extra = 0
    if not args.idct_only:
        torch.cuda.synchronize()
        e0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        est = max((time.perf_counter() - e0) / 5, 1e-6)
        warmup_extra = 5
        want = int(0.025 / est) - args.warmup - 5
        if world > 1:
            # every rank the same count (the slowest rank's estimate)
            wt = torch.tensor([want], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(wt, op=dist.ReduceOp.MIN)
            want = int(wt.item())
        for _ in range(max(0, min(want, 500))):
            step()
            warmup_extra += 1
        ctx.sync()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ctx.sync()  # deferred device status of every step enqueued above (the error word is sticky until this sync)
    # ---- the pixels the timed loop produced, against libKPEG's own decoder -------------------------------------
    # tests/golden/manifest_large.json holds SHA-256s of the reference's output on exactly these synthetic inputs
    # (tests/golden/make_golden_large.py, build container); None = this workload has no pinned hash
    verified = None
    if not args.idct_only and args.idct_mode == 0:
        import hashlib
        want_sha = pinned_rgb_sha(W, H, world, rank, args.restart_stripe or strong)
        parts = pinned_stripe_shas(IW, IH, world, rank) if strong else None
        if want_sha is not None:
            verified = hashlib.sha256(d_rgb.cpu().numpy().tobytes()).hexdigest() == want_sha
        elif parts is not None:
            # this rank's rows are whole eighths of the image: each against the reference's per-interval decode
            host = d_rgb.cpu().numpy()
            per = host.shape[0] // len(parts)
            verified = all(hashlib.sha256(host[i * per:(i + 1) * per].tobytes()).hexdigest() == parts[i] for i in range(len(parts)))
            del host
        if verified is False:
            raise SystemExit("bench.py: rank %d decoded pixels that differ from the reference's (SHA-256 mismatch)" % rank)
        if world > 1:
            # every rank's verdict: all True -> True, any unpinned -> None
            flags = [None] * world
            dist.all_gather_object(flags, verified)
            verified = None if any(f is None for f in flags) else all(flags)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel durations (HIP events on the launch stream), separate pass ---------------
    ctx.set_profiling(True)
    acc = {}
    nprof = max(5, min(args.steps, 20))
    for _ in range(nprof):
        step()
        ctx.sync()
        for k, v in ctx.timings().items():
            acc[k] = acc.get(k, 0.0) + v
    ctx.set_profiling(False)
    tm = {k: v / nprof for k, v in acc.items()}

    # ---- SURVEY 8(d) side figures (single GPU, default workload): device-copy ceiling and the dense stress input
    copy_gbs = stress = two_streams = photographs = None
    if args.side_figures and world == 1 and not args.idct_only and (W, H) == (W8K, H8K):
        # successive images alternating between two contexts/streams: one image's entropy kernels (latency-bound)
        # overlap the previous image's IDCT (instruction-bound).  The headline stays the one-stream number.
        ctx2 = K.Context(local_rank)
        st2 = torch.cuda.Stream()
        ctx2.set_stream(st2.cuda_stream)
        d_rgb2 = torch.empty_like(d_rgb)
        pair = [(ctx, d_rgb), (ctx2, d_rgb2)]
        def step2(i):
            c, o = pair[i & 1]
            c.decode_stripe_dev(frame, d_scan.data_ptr(), d_scan.numel(), first_row, rows, o.data_ptr())
        for i in range(4):
            step2(i)
        torch.cuda.synchronize()
        p0 = time.perf_counter()
        for i in range(40):
            step2(i)
        torch.cuda.synchronize()
        pms = (time.perf_counter() - p0) / 40 * 1e3
        ctx.sync()
        ctx2.sync()
        two_streams = {"ms_per_image": round(pms, 4), "value": round(W * H / (pms * 1e-3) / 1e6, 2), "unit": "Mpixels/s",
                       "what": "the same image decoded back to back, alternating between two contexts on two HIP streams"}
        del d_rgb2
        a = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
        b = torch.empty_like(a)
        for _ in range(3):
            b.copy_(a)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(20):
            b.copy_(a)
        torch.cuda.synchronize()
        copy_gbs = 2 * a.numel() * 20 / (time.perf_counter() - c0) / 1e9   # read + write
        del a, b
        sw, sh = 1920, 1080
        sdata = synth_jpeg(sw, sh, quality=95, sigma=0.0, mode=1)   # dense uniform noise, q95: long codes, slow re-synchronisation
        src, sframe, sscan = K.host_parse(sdata)
        assert src == K.DECODE_DONE, src
        sd_scan = torch.from_numpy(np.ascontiguousarray(sscan)).cuda()
        sd_rgb = torch.empty((sh, sw, 3), dtype=torch.uint8, device="cuda")
        for _ in range(3):
            ctx.decode_stripe_dev(sframe, sd_scan.data_ptr(), sd_scan.numel(), 0, sh // 8, sd_rgb.data_ptr())
        ctx.sync()
        s0 = time.perf_counter()
        for _ in range(20):
            ctx.decode_stripe_dev(sframe, sd_scan.data_ptr(), sd_scan.numel(), 0, sh // 8, sd_rgb.data_ptr())
        torch.cuda.synchronize()
        sms = (time.perf_counter() - s0) / 20 * 1e3
        ctx.sync()
        stress = {"workload": "%dx%d dense uniform noise, q95 (every coefficient non-zero; re-synchronises over thousands of bits: K1 is 95 %% of the time)" % (sw, sh),
                  "scan_bytes": int(sd_scan.numel()), "ms_per_step": round(sms, 4), "value": round(sw * sh / (sms * 1e-3) / 1e6, 2),
                  "unit": "Mpixels/s"}

        # the reference's own sample image and the committed photographs (tests/golden: small, so mostly the kernels'
        # fixed latencies; what they show is K1 on natural statistics -- bits per pixel, launches of K1 that had work)
        import hashlib
        gold = os.path.join(ROOT, "tests", "golden")
        man = json.load(open(os.path.join(gold, "manifest.json")))
        manl = json.load(open(os.path.join(gold, "manifest_large.json")))
        want_ppm = {"lena.jpg": man["lena"]["ppm_sha256"]}
        want_ppm.update({k + ".jpg": v["ppm_sha256"] for k, v in manl.get("natural", {}).items()})
        photographs = []
        for name in sorted(want_ppm):
            pdata = open(os.path.join(gold, name), "rb").read()
            prc, pframe, pscan = K.host_parse(pdata)
            if prc != K.DECODE_DONE:
                continue
            pw, ph = pframe.width, pframe.height
            pd_scan = torch.from_numpy(np.ascontiguousarray(pscan)).cuda()
            pd_rgb = torch.empty((ph, pw, 3), dtype=torch.uint8, device="cuda")
            for _ in range(3):
                ctx.decode_stripe_dev(pframe, pd_scan.data_ptr(), pd_scan.numel(), 0, ph // 8, pd_rgb.data_ptr())
            ctx.sync()
            q0 = time.perf_counter()
            for _ in range(50):
                ctx.decode_stripe_dev(pframe, pd_scan.data_ptr(), pd_scan.numel(), 0, ph // 8, pd_rgb.data_ptr())
            torch.cuda.synchronize()
            qms = (time.perf_counter() - q0) / 50 * 1e3
            ctx.sync()
            hdr = ("P6\n# PPM dump created using libKPEG: https://github.com/TheIllusionistMirage/libKPEG\n%d %d\n255\n" % (pw, ph)).encode()
            okp = hashlib.sha256(hdr + pd_rgb.cpu().numpy().tobytes()).hexdigest() == want_ppm[name]
            photographs.append({"file": name, "size": "%dx%d" % (pw, ph), "bits_per_pixel": round(len(pscan) * 8 / (pw * ph), 2),
                                "ms_per_image": round(qms, 4), "Mpixels_per_s": round(pw * ph / (qms * 1e-3) / 1e6, 1),
                                "k1_launches_with_work": ctx.timings().get("sync_rounds"), "verified": okp})

    # ---- natural content at the headline's size: the synthetic field is the friendly case (1.04 bits per pixel, re-synchronises after
    # ~64 bits); photographs carry 1-3 bits per pixel and re-synchronise more slowly.  Four committed photographs tiled to 7680 x 4352,
    # each verified against libKPEG's own decoder's pixels (tests/golden/make_golden_photos.py); the summary is their geometric mean.
    photos8k = None
    if world == 1 and not args.idct_only and not args.no_photographs and args.idct_mode == 0 and (W, H) == (W8K, H8K) and not strong and not args.restart_stripe:
        try:
            import hashlib
            man8 = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest_large.json"))).get("natural_8k", {})
            cases = []
            pd_rgb = torch.empty((PHOTO_H, PHOTO_W, 3), dtype=torch.uint8, device="cuda")
            for (src, q) in PHOTO_CASES:
                pdata = tiled_photo_jpeg(src, q)
                g = man8.get("%s_q%d_%dx%d" % (os.path.splitext(src)[0], q, PHOTO_W, PHOTO_H))
                prc, pframe, pscan = K.host_parse(pdata)
                assert prc == K.DECODE_DONE, prc
                pd_scan = torch.from_numpy(np.ascontiguousarray(pscan)).cuda()
                for _ in range(3):
                    ctx.decode_stripe_dev(pframe, pd_scan.data_ptr(), pd_scan.numel(), 0, PHOTO_H // 8, pd_rgb.data_ptr())
                ctx.sync()
                torch.cuda.synchronize()
                q0 = time.perf_counter()
                for _ in range(20):
                    ctx.decode_stripe_dev(pframe, pd_scan.data_ptr(), pd_scan.numel(), 0, PHOTO_H // 8, pd_rgb.data_ptr())
                torch.cuda.synchronize()
                qms = (time.perf_counter() - q0) / 20 * 1e3
                ctx.sync()
                okp = None
                if g and hashlib.sha256(pdata).hexdigest() == g["jpg_sha256"]:
                    okp = hashlib.sha256(pd_rgb.cpu().numpy().tobytes()).hexdigest() == g["rgb_sha256"]
                    if not okp:
                        raise SystemExit("bench.py: %s q%d tiled to 8K decodes to pixels that differ from the reference's" % (src, q))
                cases.append({"source": src, "quality": q, "bits_per_pixel": round(len(pscan) * 8 / (PHOTO_W * PHOTO_H), 2), "ms_per_image": round(qms, 4),
                              "Mpixels_per_s": round(PHOTO_W * PHOTO_H / (qms * 1e-3) / 1e6, 1), "k1_launches_with_work": ctx.timings().get("sync_rounds"),
                              "verified": okp})
            gm = float(np.exp(np.mean([np.log(c["Mpixels_per_s"]) for c in cases])))
            photos8k = {"geomean_Mpixels_per_s": round(gm, 1), "size": "%dx%d" % (PHOTO_W, PHOTO_H), "cases": cases,
                        "verified": None if any(c["verified"] is None for c in cases) else all(c["verified"] for c in cases),
                        "what": "committed photographs tiled to the headline's size and re-encoded (tools/kpeg_synth.c); their pixels pinned to libKPEG's own decoder"}
            del pd_rgb
        except ImportError:
            photos8k = {"skipped": "Pillow is not importable here"}

    # ---- decode + gather of the stripes to rank 0 (the path's one exchange step), timed apart --------------------
    # Every rank decodes its stripe as two bands of whole MCU rows; a band leaves for rank 0 (point-to-point send over
    # RCCL/xGMI, straight into its rows of the root's image) as soon as it is decoded, while the next band decodes.
    gather = None
    if world > 1 and not args.no_gather:
        nb = 2 if rows >= 2 else 1
        bands = K.stripe_ranges(scan, rows, mw, frame.restart_interval, nb)   # (first row in the stripe, rows, byte range)
        band_scans = [d_scan[b0:b1].clone() for (_, _, b0, b1) in bands]      # own allocations: 16-byte aligned for K0
        full = torch.empty((IH, IW, 3), dtype=torch.uint8, device="cuda") if rank == 0 else None
        mine = full[rank * H:(rank + 1) * H] if rank == 0 else d_rgb
        host_bufs = {}

        def one_round():
            # Band by band: the root posts the receives of ONE band from all its peers as one group (dist.batch_isend_irecv = one
            # ncclGroupStart/End: the seven xGMI links carry their stripes side by side; seven separate irecvs on an eagerly
            # initialised communicator are issued one after the other on one RCCL stream), the peers send that band behind its kernels.
            reqs, copies = [], []
            for bi, (r0, nr, _, _) in enumerate(bands):
                out_rows = mine[r0 * 8:(r0 + nr) * 8]
                ctx.decode_stripe_dev(frame, band_scans[bi].data_ptr(), band_scans[bi].numel(), first_row + r0, nr, out_rows.data_ptr())
                ops = []
                if rank == 0:
                    for src in range(1, world):
                        dst_rows = full[src * H + r0 * 8: src * H + (r0 + nr) * 8]
                        if args.rehearse:
                            hb = host_bufs.setdefault((src, r0), torch.empty(dst_rows.shape, dtype=torch.uint8))
                            ops.append(dist.P2POp(dist.irecv, hb, src))
                            copies.append((hb, dst_rows))
                        else:
                            ops.append(dist.P2POp(dist.irecv, dst_rows, src))
                else:
                    if args.rehearse:
                        ctx.sync()
                        ops.append(dist.P2POp(dist.isend, out_rows.cpu(), 0))
                    else:
                        ops.append(dist.P2POp(dist.isend, out_rows, 0))   # behind this band's kernels (same stream order); the next band decodes meanwhile
                if ops:
                    reqs.extend(dist.batch_isend_irecv(ops))
            for w in reqs:
                w.wait()
            for (hb, dst_rows) in copies:
                dst_rows.copy_(hb)
            torch.cuda.synchronize()

        for _ in range(2):
            one_round()
        ctx.sync()
        barrier()
        g0 = time.perf_counter()
        nrep = max(3, min(args.steps, 10))
        for _ in range(nrep):
            one_round()
        dist.barrier()
        gms = (time.perf_counter() - g0) / nrep * 1e3
        ctx.sync()
        t = torch.tensor([gms], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        gms = float(t.item())
        gathered_ok = None
        if rank == 0:
            import hashlib
            shas = pinned_stripe_shas(IW, IH, 1, 0) if strong else None
            if shas is not None:
                host = full.cpu().numpy()
                per = IH // 8
                gathered_ok = all(hashlib.sha256(host[i * per:(i + 1) * per].tobytes()).hexdigest() == shas[i] for i in range(8))
                del host
                if not gathered_ok:
                    raise SystemExit("bench.py: the gathered image differs from the reference's (SHA-256 mismatch)")
            if os.environ.get("KPEG_BENCH_VERIFY"):
                # rehearsal check on sizes without a pinned hash: the gathered image equals the oracle's decode
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import kpeg_testlib as T
                want = np.concatenate([T.oracle_decode_rst(synth_jpeg(W, H, y0=r * H, restart_interval=mw), mw, 16)[0] for r in range(world)], 0)
                assert np.array_equal(full.cpu().numpy(), want), "gathered stripes differ from the oracle"
                print("VERIFY_OK gathered %dx%d image equals the oracle" % (IW, IH), file=sys.stderr)
        gbytes = (world - 1) * d_rgb.numel()
        gather = {"ms_decode_and_gather": round(gms, 4), "bytes_into_root": gbytes, "bands_per_rank": nb, "verified": gathered_ok,
                  "how": "each rank decodes its stripe in %d bands; a band is sent to rank 0 (RCCL send/recv over xGMI, the root's receives of a band "
                         "grouped into one batch_isend_irecv, into its rows of the root's image) while the next band decodes" % nb}

    if rank == 0:
        pixels_per_step = W * H * world
        ms_per_step = elapsed / args.steps * 1e3
        value = pixels_per_step / (ms_per_step * 1e-3) / 1e6
        # K4's duration: the span between the HIP events in front of and behind its launch, less what such a span holds when no kernel is
        # launched in it -- the span in front of K4's (EV_WRITE -> EV_DC: there is no DC kernel any more) is exactly that, an event record
        # and the gap to the next one, measured in the same pass (4.6-5.5 us).  rocprofv3's average for the kernel on the same command
        # (profiles/r03_*_kernel_stats.csv) is what this has to agree with, and does within 2 %; the raw span read 8 % high.
        idct_span_ms = tm.get("idct_ms", 0.0)
        event_overhead_ms = tm.get("dc_ms", 0.0) if not args.idct_only else 0.0
        idct_ms = idct_span_ms - event_overhead_ms if idct_span_ms > 2 * event_overhead_ms else idct_span_ms
        alg_bytes = 9.0 * W * H  # per launch: this rank's stripe
        traffic = traffic_source = None
        tf = os.path.join(ROOT, "profiles", "k4_traffic.json")
        if world == 1 and (W, H) == (W8K, H8K) and not strong and os.path.exists(tf):
            # HBM bytes per launch from the PMC passes committed under profiles/ (same kernel, same workload;
            # counters cannot be read from inside this process)
            tj = json.load(open(tf))
            traffic = tj.get("traffic_bytes")
            traffic_source = "profiles/k4_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, gfx950 correction) on `%s`; a committed measurement, not taken in this run" % tj.get("workload", "?")
        roof = {"bound": "hbm", "achieved": round(alg_bytes / (idct_ms * 1e-3) / 1e9, 2) if idct_ms > 0 else None,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_bytes / (idct_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if idct_ms > 0 else None,
                "traffic": traffic, "traffic_source": traffic_source, "kernel": "k_idct_colour_fast", "kernel_ms": round(idct_ms, 5),
                "kernel_ms_how": "HIP-event span around the launch (%.5f ms) less the span an event record alone takes (%.5f ms, the empty span in front of it)" % (idct_span_ms, event_overhead_ms),
                "algorithmic_bytes": int(alg_bytes)}
        if traffic and traffic < alg_bytes:
            # the compact coefficient stream (the library's choice above 64 MiB of dense coefficients): K4 reads a tenth of the
            # algorithmic coefficient bytes, so achieved/frac -- algorithmic bytes / time, as the contract defines them -- overstate
            # what crosses the HBM pins; the kernel is bound by its instruction stream there
            roof["hbm_GBs_by_traffic"] = round(traffic / (idct_ms * 1e-3) / 1e9, 2) if idct_ms > 0 else None
        if copy_gbs:
            roof["device_copy_GBs"] = round(copy_gbs, 1)   # measured ceiling: 256 MiB device-to-device copy, read + write bytes
        # HIP events between the launches: each span holds its kernel(s) and the gap to the next event, so their sum exceeds ms_per_step
        # (the timed loop has no events).  Where k_sync_write did K1's pass 0 and K2 in one kernel, its second (strict) launch leaves at once and
        # no other entropy kernel is enqueued: the keys say so instead of naming kernels that did not run.
        one_k = (tm.get("sync_rounds") or 0) == 1 and (tm.get("huff_write_ms") or 1.0) < 0.015 and not args.idct_only
        names = {"huff_sync_ms": "k_sync_write_and_its_idle_second_launch_ms", "huff_scan_ms": "gap_behind_entropy_ms", "huff_write_ms": "gap_behind_entropy_2_ms",
                 "dc_ms": "gap_before_K4_ms", "unstuff_ms": "gap_before_entropy_ms"} if one_k else {"dc_ms": "gap_before_K4_ms"}
        kernels_ms = {names.get(k, k): round(v, 5) for k, v in tm.items() if k.endswith("_ms")}
        kernels_ms["note"] = "event-to-event spans (kernel + launch gap); their sum exceeds ms_per_step, which is timed without events"
        out = {
            "metric": "Mpixels/s decoded (JFIF->RGB) + achieved HBM GB/s, 8K 4:4:4 baseline",
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "warmup_extra": warmup_extra,   # untimed steps beyond `warmup`, so that ~25 ms of work precede the timed region (clock settled)
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("K4 only, " if args.idct_only else "") +
                       "%dx%d 4:4:4 baseline JPEG q%d, seed %d, %s" % (
                           IW, IH, QUALITY, SEED,
                           "no restart markers, full on-device Huffman + IDCT" if world == 1 and not args.restart_stripe and not strong else
                           "restart interval = 1 MCU row, %d row stripes of %d rows, one per GPU" % (world, H)),
                       "scan_bytes_per_gpu": int(d_scan.numel()), "pixels_per_step": pixels_per_step},
            "verified": verified,   # the timed loop's output == the reference decoder's pixels (SHA-256), None = not pinned
            "roofline": roof,
            "kernels_ms": kernels_ms,
            "sync_passes": tm.get("sync_rounds"), "exact_pixels_per_image": tm.get("exact_pixels"),
        }
        if not args.idct_only:
            # per-kernel and end-to-end achieved GB/s from algorithmic bytes (this rank's stripe):
            # K0 reads and writes the scan once, K1 reads it (its rounds run from LDS), K2 reads it and writes 6 B/pixel
            # of coefficients, K4 reads those and writes 3 B/pixel; fused minimum = scan in + 3 B/pixel out
            S, px = float(d_scan.numel()), float(W * H)
            def gbs(nbytes, ms):
                return round(nbytes / (ms * 1e-3) / 1e9, 1) if ms and ms > 0 else None
            # (one launch of K1 with work = k_sync_write did K1's pass 0 and K2 in one kernel: the events bracket that kernel and
            # the launches behind it, which left at once; K1 and K2 have no times of their own then)
            one_kernel = (tm.get("sync_rounds") or 0) == 1 and (tm.get("huff_write_ms") or 1.0) < 0.015
            out["algorithmic_GBs"] = {
                # (one image without restart markers runs no K0: K1 and K2 un-stuff what they stage; the event pair then brackets nothing)
                "K0_unstuff": gbs(2 * S, tm.get("unstuff_ms")) if (tm.get("unstuff_ms") or 0) > 0.008 else None,
                "K1_sync": None if one_kernel else gbs(S, tm.get("huff_sync_ms")),
                "K2_write": None if one_kernel else gbs(S + 6 * px, tm.get("huff_write_ms")),
                "K1_K2_one_kernel": gbs(2 * S + 6 * px, (tm.get("huff_sync_ms") or 0) + (tm.get("huff_write_ms") or 0)) if one_kernel else None,
                "K4_idct_colour": gbs(9 * px, idct_ms),
                "end_to_end_fused_minimum": gbs((S + 3 * px) * world, ms_per_step)}
        if two_streams:
            out["two_streams"] = two_streams
        if stress:
            out["stress"] = stress
        if photographs:
            out["photographs"] = photographs
        if photos8k:
            out["photographs_8k"] = photos8k
        if gather:
            out["gather"] = gather
            out["value_incl_gather"] = round(pixels_per_step / (gather["ms_decode_and_gather"] * 1e-3) / 1e6, 2)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
