#!/usr/bin/env python3
"""tools/batch_breakdown.py -- per-kernel HIP-event times of one fused batch of 256 x 1080p (BASELINE config 4) on the GPU box."""
import sys, numpy as np, torch
sys.path.insert(0, "."); import bench, libkpeg_amd as K
torch.cuda.set_stream(torch.cuda.Stream())
n, w, h = 256, 1920, 1080
scans = []
for i in range(32):
    rc, frame, scan = K.host_parse(bench.synth_jpeg(w, h, seed=bench.SEED + i)); scans.append(torch.from_numpy(np.ascontiguousarray(scan)).cuda())
d_scans = [scans[i % 32] for i in range(n)]; d_rgbs = [torch.empty((h, w, 3), dtype=torch.uint8, device="cuda") for _ in range(n)]
ctx = K.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
sp, sl, op = [t.data_ptr() for t in d_scans], [t.numel() for t in d_scans], [t.data_ptr() for t in d_rgbs]
for _ in range(3): ctx.decode_batch_dev(frame, sp, sl, op)
ctx.sync(); ctx.set_profiling(True); ctx.decode_batch_dev(frame, sp, sl, op); ctx.sync()
print({k: round(v, 4) for k, v in ctx.timings().items() if k.endswith("_ms")})
