#!/usr/bin/env python3
"""tools/host_batch_dbg.py -- wall time of the host-buffer batch entry point (PCIe and host copies included)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import libkpeg_amd as K
import bench
n, w, h = 64, 1920, 1080
parsed = [K.host_parse(bench.synth_jpeg(w, h, seed=1234 + i)) for i in range(8)]
frame = parsed[0][1]
scans = [parsed[i % 8][2] for i in range(n)]
ctx = K.Context(0)
ctx.decode_batch(frame, scans); ctx.decode_batch(frame, scans)
t0 = time.perf_counter(); outs = ctx.decode_batch(frame, scans); dt = time.perf_counter() - t0
print("decode_batch (lanes, pageable buffers): %.2f ms per image, %.1f Mpixel/s" % (dt / n * 1e3, n * w * h / dt / 1e6))
t0 = time.perf_counter()
for s in scans: ctx.decode_scan(frame, s)
dt = time.perf_counter() - t0
print("decode_scan loop: %.2f ms per image, %.1f Mpixel/s" % (dt / n * 1e3, n * w * h / dt / 1e6))
