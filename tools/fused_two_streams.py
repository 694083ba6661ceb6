#!/usr/bin/env python3
"""tools/fused_two_streams.py -- k_sync_write while another context's kernels share the chip: two contexts, two streams, two
different pictures, many calls in flight; every output compared with the single-stream result (GPU box)."""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench, libkpeg_amd as K
pics = [bench.synth_jpeg(7680, 4320), bench.synth_jpeg(3840, 2176, seed=99, quality=80)]
ctxs, streams, bufs, outs, frames, want = [], [], [], [], [], []
for d in pics:
    rc, f, scan = K.host_parse(d)
    c = K.Context(0)
    want.append(c.decode_scan(f, scan).copy())
    st = torch.cuda.Stream()
    c.set_stream(st.cuda_stream)
    ctxs.append(c); streams.append(st); frames.append(f)
    bufs.append(torch.frombuffer(bytearray(scan), dtype=torch.uint8).cuda())
    outs.append([torch.zeros((f.height, f.width, 3), dtype=torch.uint8, device="cuda") for _ in range(4)])
torch.cuda.synchronize()
bad = 0
for rnd in range(40):
    for k in range(4):
        for i, c in enumerate(ctxs):
            c.decode_scan_dev(frames[i], bufs[i].data_ptr(), bufs[i].numel(), outs[i][k].data_ptr())
    for c in ctxs: c.sync()
    torch.cuda.synchronize()
    for i in range(2):
        for k in range(4):
            got = outs[i][k].cpu().numpy()
            if not np.array_equal(got, want[i]):
                bad += 1
                d = np.argwhere(got != want[i])
                print("round", rnd, "context", i, "call", k, "wrong pixels", len(d), "first", d[0].tolist(), "last", d[-1].tolist(), "all zero:", bool((got == 0).all()), flush=True)
    for i in range(2):
        for k in range(4): outs[i][k].zero_()
    torch.cuda.synchronize()   # (the zeroing runs on torch's stream, the decodes on their own)
print("rounds 40 x 4 calls x 2 contexts, wrong outputs:", bad, " launches of K1 with work (last call):", [int(c.timings()["sync_rounds"]) for c in ctxs])
