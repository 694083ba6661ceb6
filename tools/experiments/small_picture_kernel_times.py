import sys, os, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, libkpeg_amd as K, kpeg_testlib as T, bench
torch.cuda.set_stream(torch.cuda.Stream())
ctx = K.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for name, data in [("lena.jpg", open(os.path.join(T.GOLDEN, "lena.jpg"), "rb").read()), ("synthetic 512x512", bench.synth_jpeg(512, 512)), ("synthetic 1920x1088", bench.synth_jpeg(1920, 1088))]:
    rc, f, scan = K.host_parse(data)
    d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda(); d_rgb = torch.empty((f.height, f.width, 3), dtype=torch.uint8, device="cuda")
    for _ in range(20): ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
    ctx.sync(); ctx.set_profiling(True)
    acc = {}
    for _ in range(10):
        ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr()); ctx.sync()
        for k, v in ctx.timings().items():
            if k.endswith("_ms"): acc[k] = acc.get(k, 0) + v / 10
    ctx.set_profiling(False)
    print(name, {k: round(v, 4) for k, v in acc.items()})
