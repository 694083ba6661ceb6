import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench, libkpeg_amd as K
torch.cuda.set_stream(torch.cuda.Stream())
ctx = K.Context(0); ctx.set_profiling(True)
for name, data in (("8K synthetic", bench.synth_jpeg(7680, 4320)), ("lena q75 8K", bench.tiled_photo_jpeg("lena.jpg", 75)), ("china q90 8K", bench.tiled_photo_jpeg("nat_china_640x424_q90.jpg", 90))):
    rc, f, scan = K.host_parse(data)
    for fault in (0, 8):
        ctx.lib.kpeg_hip_debug_set(ctx._h, 6, fault)
        best = None
        for _ in range(4):
            ctx.decode_scan(f, scan); t = ctx.timings()
            if best is None or t["total_ms"] < best["total_ms"]: best = dict(t)
        print(name, "fault", fault, "sync %.4f total %.4f rounds %d" % (best["huff_sync_ms"], best["total_ms"], best["sync_rounds"]))
ctx.lib.kpeg_hip_debug_set(ctx._h, 6, 0)
