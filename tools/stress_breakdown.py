import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench, libkpeg_amd as K
torch.cuda.set_stream(torch.cuda.Stream())
ctx = K.Context(0); ctx.set_profiling(True)
for name, data in (("q95 noise 1080p", bench.synth_jpeg(1920, 1080, quality=95, sigma=0.0, mode=1)), ("q90 sigma20 4K", bench.synth_jpeg(3840, 2160, quality=90, sigma=20.0)), ("lena 512", open("tests/golden/lena.jpg","rb").read())):
    rc, f, scan = K.host_parse(data)
    best=None
    for _ in range(5):
        ctx.decode_scan(f, scan); t = ctx.timings()
        if best is None or t["total_ms"] < best["total_ms"]: best = dict(t)
    print(name, "  ".join("%s=%.4f" % (k, v) if isinstance(v, float) else "%s=%s" % (k, v) for k, v in best.items()))
