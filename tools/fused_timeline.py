#!/usr/bin/env python3
"""tools/fused_timeline.py -- per workgroup (wavefront 0 of each), k_sync_write on the 8K image (stats build, KPEG_FUSED=1):
when K1's part ends, when K2's loop starts and ends, on the workgroup's own clock, relative to its own start."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch, libkpeg_amd as K, bench
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
data = bench.synth_jpeg(W, H)
rc, frame, scan = K.host_parse(data)
ctx = K.Context(0)
ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 2)
for _ in range(3): ctx.decode_scan(frame, scan)
def stamps(which):
    a = np.zeros(8192 * 16, np.uint64)
    assert ctx.lib.kpeg_hip_debug_entropy_stamps(which, a.ctypes.data_as(ctypes.c_void_p), a.size) == 0
    return a.reshape(8192, 16)
k1, k2 = stamps(0), stamps(1)
nwg = int((k1[::8, 0] != 0).sum())
w0 = np.arange(0, nwg * 8, 8)   # wavefront 0 of every workgroup
t_start = k1[w0, 0].astype(np.int64)
k1_end = k1[w0, 3].astype(np.int64) - t_start
k2_beg = k2[w0, 2].astype(np.int64) - t_start
k2_end = k2[w0, 4].astype(np.int64) - t_start
print("wg   K1 part ends   K2 loop starts   K2 ends   (wait)   (K2 loop + epilogue)")
for g in list(range(0, nwg, max(1, nwg // 20))) + [nwg - 1]:
    print("%4d %10d %14d %12d %9d %12d" % (g, k1_end[g], k2_beg[g], k2_end[g], k2_beg[g] - k1_end[g], k2_end[g] - k2_beg[g]))
print("slowest K1 part: wg", int(np.argmax(k1_end)), int(k1_end.max()), " last K2 end:", int(k2_end.max()), "wg", int(np.argmax(k2_end)))
