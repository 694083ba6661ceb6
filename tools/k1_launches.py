#!/usr/bin/env python3
"""tools/k1_launches.py -- how many of K1's three launches have work on hard streams (GPU box)."""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench, libkpeg_amd as K
torch.cuda.set_stream(torch.cuda.Stream())
ctx = K.Context(0)
import kpeg_testlib as T
cases = [("q95 noise 1080p", bench.synth_jpeg(1920, 1080, quality=95, sigma=0.0, mode=1)),
         ("q98 sigma40 1080p", bench.synth_jpeg(1920, 1080, quality=98, sigma=40.0)),
         ("q90 sigma20 4K", bench.synth_jpeg(3840, 2160, quality=90, sigma=20.0)),
         ("lena", open("tests/golden/lena.jpg", "rb").read()),
         ("nat q96", open("tests/golden/nat_flower_320x208_q96.jpg", "rb").read())]
for name, data in cases:
    rc, f, scan = K.host_parse(data)
    ctx.set_profiling(True)
    ctx.decode_scan(f, scan)
    t = ctx.timings()
    print(name, "bits/px %.2f" % (len(scan) * 8 / (f.width * f.height)), "K1 launches with work:", t.get("sync_rounds"), "K1 ms %.3f" % t["huff_sync_ms"])
