#!/usr/bin/env python3
"""tools/k4_ab.py -- interleaved A/B timing of kernel variants in ONE process (GPU box).

    tools/variants.sh base="" x="-DKPEG_SOMETHING=1" ...      # build/ablate/libkpeg_hip_<name>.so
    python tools/k4_ab.py [--mode k4|decode] [--rounds 7] [--steps 20] [name ...]

Every variant library is loaded beside the others, gets its own context, and the variants are timed round-robin
(rounds x steps launches each; median and minimum of the per-round means are printed), which keeps clock and
placement drift common to all of them.  mode k4: K4 alone on resident coefficients (the library's own HIP events
around the kernel); mode decode: the whole K0..K4 decode (wall clock around `steps` back-to-back calls) plus the
per-kernel event times.  Each variant's output is hashed against the reference's pixels
(tests/golden/manifest_large.json): a fast variant with wrong pixels is reported as WRONG.
"""
import argparse
import glob
import hashlib
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the synthetic generator and the workload constants)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*")
    ap.add_argument("--mode", default="k4", choices=["k4", "decode"])
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--width", type=int, default=bench.W8K)
    ap.add_argument("--height", type=int, default=bench.H8K)
    ap.add_argument("--idct-mode", type=int, default=0, help="kpeg_hip_set_idct_mode: 2 = unsafe pixels are queued but not re-evaluated (wrong pixels: what the queueing alone costs)")
    ap.add_argument("--layouts", default="0", help="comma list of coefficient layouts per variant: 0 auto, 1 dense, 2 compact (debug key 7)")
    args = ap.parse_args()
    import torch
    import libkpeg_amd as K
    torch.cuda.set_stream(torch.cuda.Stream())   # a created stream: the default stream's handle 0 cannot be handed to the C ABI

    libs = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "build", "ablate", "libkpeg_hip_*.so"))):
        name = os.path.basename(f)[len("libkpeg_hip_"):-3]
        if not args.names or name in args.names:
            libs[name] = K.load_variant(f)
    assert libs, "no variants under build/ablate (tools/variants.sh)"
    W, H = args.width, args.height
    rc, frame, scan = K.host_parse(bench.synth_jpeg(W, H))
    assert rc == K.DECODE_DONE
    want = bench.pinned_rgb_sha(W, H, 1, 0, False)
    d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda()
    d_coef = torch.empty((W // 8) * (H // 8) * 192, dtype=torch.int16, device="cuda")
    d_rgb = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ctxs = {}
    layouts = [int(x) for x in args.layouts.split(",")]
    for name, lib in libs.items():
        for lay in layouts:
            c = K.Context(0, lib=lib)
            c.set_stream(stream)
            c.set_idct_mode(args.idct_mode)
            if lay:
                assert lib.kpeg_hip_debug_set(c._h, 7, lay) == 0
            ctxs[name + ("" if len(layouts) == 1 else "@%d" % lay)] = c
    first = next(iter(ctxs.values()))
    first.entropy_decode_dev(frame, d_scan.data_ptr(), d_scan.numel(), d_coef.data_ptr())
    first.sync()

    def sync(c):
        try:
            c.sync()
        except K.KpegError:      # an ablated variant may decode garbage: only its timing is of interest
            pass

    def run(c):
        if args.mode == "k4":
            c.idct_colour_dev(frame, d_coef.data_ptr(), d_rgb.data_ptr())
        else:
            c.decode_scan_dev(frame, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())

    ok = {}
    for name, c in ctxs.items():
        d_rgb.zero_()
        run(c)
        sync(c)
        ok[name] = None if want is None else hashlib.sha256(d_rgb.cpu().numpy().tobytes()).hexdigest() == want
    res = {n: {"wall": [], "k": {}} for n in ctxs}
    for r in range(args.rounds):
        for name, c in ctxs.items():
            for _ in range(3):
                run(c)
            sync(c)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                run(c)
            torch.cuda.synchronize()
            res[name]["wall"].append((time.perf_counter() - t0) / args.steps * 1e3)
            sync(c)
            c.set_profiling(True)
            acc = {}
            for _ in range(5):
                run(c)
                sync(c)
                for k, v in c.timings().items():
                    acc[k] = acc.get(k, 0.0) + v / 5
            c.set_profiling(False)
            for k, v in acc.items():
                res[name]["k"].setdefault(k, []).append(v)
    print("%-22s %-7s %10s %10s   per-kernel medians (ms)" % ("variant", "pixels", "wall med", "wall min"))
    for name in ctxs:
        w = res[name]["wall"]
        km = {k[:-3]: statistics.median(v) for k, v in res[name]["k"].items() if k.endswith("_ms") and statistics.median(v) > 0}
        print("%-22s %-7s %10.4f %10.4f   %s  unsafe=%d" % (
            name, {True: "ok", False: "WRONG", None: "n/a"}[ok[name]], statistics.median(w), min(w),
            " ".join("%s=%.4f" % kv for kv in km.items()), int(statistics.median(res[name]["k"].get("exact_pixels", [0])))), flush=True)


if __name__ == "__main__":
    main()
