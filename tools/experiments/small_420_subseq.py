"""4:2:0 pictures of a few sizes with the sub-sequence size forced (debug key 4): 64 / 96 / 384 bits and the library's choice."""
import io, sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, libkpeg_amd as K, kpeg_testlib as T
from PIL import Image
from test_420 import _photo
torch.cuda.set_stream(torch.cuda.Stream())
ctx = K.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ph = _photo()
rng = np.random.default_rng(3)
cases = [("photograph 640x424 q85", ph[:424, :640], 85), ("photograph 320x208 q75", ph[:208, :320], 75)]
y, x = np.mgrid[0:1080, 0:1920]
cases.append(("synthetic 1920x1080 q80", np.clip(np.stack([x * 255.0 / 1919, y * 255.0 / 1079, (x + y) % 256], -1) + rng.normal(0, 8, (1080, 1920, 3)), 0, 255).astype(np.uint8), 80))
cases.append(("photograph tiled 2560x1696 q85", np.tile(ph[:424, :640], (4, 4, 1)), 85))
cases.append(("photograph tiled 1024x768 q85", np.tile(ph[:424, :640], (2, 2, 1))[:768, :1024], 85))
cases.append(("photograph tiled 1280x848 q85", np.tile(ph[:424, :640], (2, 2, 1)), 85))
y2, x2 = np.mgrid[0:720, 0:1280]
cases.append(("synthetic 1280x720 q80", np.clip(np.stack([x2 * 255.0 / 1279, y2 * 255.0 / 719, (x2 + y2) % 256], -1) + rng.normal(0, 8, (720, 1280, 3)), 0, 255).astype(np.uint8), 80))
cases.append(("synthetic 800x600 q80", np.clip(np.stack([x2 * 255.0 / 1279, y2 * 255.0 / 719, (x2 + y2) % 256], -1) + rng.normal(0, 8, (720, 1280, 3)), 0, 255).astype(np.uint8)[:600, :800], 80))
for name, px, q in cases:
    b = io.BytesIO(); Image.fromarray(px).save(b, "JPEG", quality=q, subsampling=2)
    rc, f, scan = K.host_parse(b.getvalue(), allow_420=True)
    st, want = T.oracle_decode_420(b.getvalue())
    h, w = px.shape[:2]
    d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda(); d_rgb = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    out = []
    for ss in (64, 96, 384, 0):
        ctx.lib.kpeg_hip_debug_set(ctx._h, 4, ss)
        ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr()); ctx.sync()
        assert np.array_equal(d_rgb.cpu().numpy(), want), (name, ss)
        best = 1e9
        for rep in range(3):
            for _ in range(5): ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
            ctx.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): ctx.decode_scan_dev(f, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 30 * 1e3); ctx.sync()
        out.append(best)
    print("%-34s %.2f bits/px  ms per picture with 64 / 96 / 384 / the library's choice: %.4f / %.4f / %.4f / %.4f" % (name, len(scan) * 8 / (w * h), *out), flush=True)
ctx.lib.kpeg_hip_debug_set(ctx._h, 4, 0)
