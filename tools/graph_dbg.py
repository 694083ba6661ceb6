#!/usr/bin/env python3
"""tools/graph_dbg.py -- does replaying the 8K decode as a captured HIP graph shrink the launch gaps?  (experiment)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import libkpeg_amd as K
import bench
W, H = 7680, 4320
rc, frame, scan = K.host_parse(bench.synth_jpeg(W, H))
ctx = K.Context(0)
s = torch.cuda.Stream()
d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda()
d_rgb = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
def step():
    ctx.decode_stripe_dev(frame, d_scan.data_ptr(), d_scan.numel(), 0, H // 8, d_rgb.data_ptr())
with torch.cuda.stream(s):
    ctx.set_stream(s.cuda_stream)
    for _ in range(3): step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(50): step()
    s.synchronize()
    print("direct launches: %.4f ms per step" % ((time.perf_counter() - t0) / 50 * 1e3))
    ref = d_rgb.clone()
    g = torch.cuda.CUDAGraph()
    d_rgb.zero_()
    try:
        with torch.cuda.graph(g, stream=s):
            step()
        for _ in range(3): g.replay()
        s.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize()
        print("graph replay:    %.4f ms per step; same pixels: %s" % ((time.perf_counter() - t0) / 50 * 1e3, bool(torch.equal(ref, d_rgb))))
    except Exception as e:
        print("capture failed:", repr(e)[:300])
