"""Which kernels a one-component picture takes (run under rocprofv3 --kernel-trace --stats): layout 2 must show k_write<..., true, ...>."""
import io, sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import libkpeg_amd as K, kpeg_testlib as T
from PIL import Image
rng = np.random.default_rng(1)
px = np.clip(rng.normal(128, 30, (1080, 1920)), 0, 255).astype(np.uint8)
b = io.BytesIO(); Image.fromarray(px, "L").save(b, "JPEG", quality=75)
st, want = T.oracle_decode_gray(b.getvalue())
rc, f, scan = K.host_parse(b.getvalue(), allow_gray=True)
ctx = K.Context(0)
for layout in (1, 2):
    ctx.lib.kpeg_hip_debug_set(ctx._h, 7, layout)
    for _ in range(3): got = ctx.decode_scan(f, scan)
    assert np.array_equal(got, want), layout
    t0 = time.perf_counter()
    for _ in range(20): ctx.decode_scan(f, scan)
    print("layout", layout, "%.3f ms per decode_scan (host buffers)" % ((time.perf_counter() - t0) / 20 * 1e3))
