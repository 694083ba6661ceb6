import io, sys, time, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, libkpeg_amd as K
from PIL import Image
torch.cuda.set_stream(torch.cuda.Stream())
rng = np.random.default_rng(1)
w, h = 4000, 3000
y, x = np.mgrid[0:h, 0:w]
px = np.stack([x * 255.0 / (w - 1), y * 255.0 / (h - 1), ((x // 64 + y // 64) % 2) * 255.0], -1) + rng.normal(0, 8, (h, w, 3))
b = io.BytesIO(); Image.fromarray(np.clip(px, 0, 255).astype(np.uint8)).save(b, "JPEG", quality=85, subsampling=2)
rc, f, scan = K.host_parse(b.getvalue(), allow_420=True)
ctx = K.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda(); d_rgb = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
for _ in range(3): ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
ctx.sync(); t0 = time.perf_counter()
for _ in range(20): ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3; ctx.sync()
ctx.set_profiling(True); ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr()); ctx.sync()
print("4:2:0 %dx%d, %.2f bits/px: %.3f ms per image = %.1f Gpixel/s" % (w, h, len(scan) * 8 / (w * h), ms, w * h / ms / 1e6), {k: round(v, 4) for k, v in ctx.timings().items() if k.endswith("_ms")})
print("K1 launches with work:", ctx.timings().get("sync_rounds"))
ctx.set_profiling(False)
for ss in (96, 384):
    ctx.lib.kpeg_hip_debug_set(ctx._h, 4, ss)
    for _ in range(3): ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(10): ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3; ctx.sync()
    ctx.set_profiling(True); ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr()); ctx.sync(); ctx.set_profiling(False)
    print("sub-sequences of %d bits: %.3f ms, K1 %.3f ms, launches with work %s" % (ss, ms, ctx.timings()["huff_sync_ms"], ctx.timings().get("sync_rounds")))
if hasattr(ctx.lib, "kpeg_hip_debug_entropy_stamps"):
    import ctypes
    ctx.lib.kpeg_hip_debug_set(ctx._h, 4, 0)
    ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr()); ctx.sync()
    a = np.zeros(8192 * 16, np.uint64)
    ctx.lib.kpeg_hip_debug_entropy_stamps(0, a.ctypes.data_as(ctypes.c_void_p), a.size)
    a = a.reshape(8192, 16); a = a[a[:, 0] != 0]
    rounds = (a[:, 5] >> np.uint64(32)).astype(int); runs = (a[:, 5] & np.uint64(0xFFFFFFFF)).astype(int)
    print("K1 pass 0 (stats build): %d wavefronts, rounds per wave: median %d p90 %d max %d; decodes per wave median %d" % (
        len(a), np.median(rounds), np.percentile(rounds, 90), rounds.max(), np.median(runs)))
