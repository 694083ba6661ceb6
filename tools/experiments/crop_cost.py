"""What the any-size extension's crop kernel costs: a 3999x2999 picture (padded 4000x3000) against the 4000x3000 one, device to device."""
import io, sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, libkpeg_amd as K
from PIL import Image
torch.cuda.set_stream(torch.cuda.Stream())
rng = np.random.default_rng(2)
ctx = K.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
y, x = np.mgrid[0:3000, 0:4000]
px = np.clip(np.stack([x * 255.0 / 3999, y * 255.0 / 2999, (x + y) % 256], -1) + rng.normal(0, 6, (3000, 4000, 3)), 0, 255).astype(np.uint8)
for (w, h) in ((4000, 3000), (3999, 2999), (3993, 2993)):
    b = io.BytesIO(); Image.fromarray(px[:h, :w]).save(b, "JPEG", quality=80, subsampling=0)
    rc, f, scan = K.host_parse(b.getvalue(), allow_any_size=True)
    d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda(); d_rgb = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    for _ in range(5): ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
    ctx.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 50 * 1e3; ctx.sync()
    print("%dx%d: %.4f ms per picture" % (w, h, ms), flush=True)
