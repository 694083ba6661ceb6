import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, kpeg_testlib as T, libkpeg_amd, time
ctx = libkpeg_amd.Context(0)
for (w,h) in [(2048,1024),(7680,4320)]:
    rgb = np.empty((h, w, 3), np.uint8); rgb[:] = (200,30,77)
    data = T.encode_rgb(rgb, quality=75)
    p = T.oracle_parse(data)
    for warm in (-1, 0):
        ctx.lib.kpeg_hip_debug_set(ctx._h, 2, warm)
        ctx.decode_scan(T.make_frame(p), p.scan)
        t0=time.time(); got = ctx.decode_scan(T.make_frame(p), p.scan); dt=time.time()-t0
        print(w,h,"warm",warm,"scan bytes",len(p.scan),"sync passes",ctx.timings()["sync_rounds"], "wall ms %.2f"%(dt*1e3), "ok", bool((got==np.array([200,30,77],np.uint8)).mean()>0.5))
