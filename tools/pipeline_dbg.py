#!/usr/bin/env python3
"""tools/pipeline_dbg.py -- experiment: successive 8K decodes alternating between two contexts/streams (the entropy
kernels of one image overlap the IDCT of the previous one) vs the single-stream loop that bench.py reports."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import libkpeg_amd as K
import bench
W, H = 7680, 4320
rc, frame, scan = K.host_parse(bench.synth_jpeg(W, H))
d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda()
for nctx in (1, 2, 3):
    ctxs = [K.Context(0) for _ in range(nctx)]
    streams = [torch.cuda.Stream() for _ in range(nctx)]
    outs = [torch.empty((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(nctx)]
    for c, s in zip(ctxs, streams):
        c.set_stream(s.cuda_stream)
    def step(i):
        k = i % nctx
        ctxs[k].decode_stripe_dev(frame, d_scan.data_ptr(), d_scan.numel(), 0, H // 8, outs[k].data_ptr())
    for i in range(2 * nctx): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 60
    for i in range(n): step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    for c in ctxs: c.sync()
    print("%d context(s): %.4f ms per image, %.1f Gpixel/s" % (nctx, dt * 1e3, W * H / dt / 1e9))
