#!/usr/bin/env python3
"""tools/sync_dbg.py -- loop counts of k_sync_pass on the bench image (needs a -DKPEG_SYNC_STATS=1 build,
KPEG_HIP_LIB=build/ablate/libkpeg_hip_STATS.so).  Experiment tool, not part of the product."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import libkpeg_amd as K
import bench

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
data = bench.synth_jpeg(W, H)
rc, frame, scan = K.host_parse(data)
ctx = K.Context(0)
rgb = ctx.decode_scan(frame, scan)
out = (ctypes.c_uint32 * 16)()
ctx.lib.kpeg_hip_debug_words(ctx._h, out, 16)
w = list(out)
nwg = (len(scan) * 8 + 255) // 256 // 256 + 1
print("words", w)
print("WG rounds: sum %d max %d (~%d WGs -> mean %.2f)" % (w[8], w[9], nwg, w[8] / nwg))
print("round 0: waves %d mean max-lane iterations %.1f" % (w[11], w[10] / max(1, w[11])))
print("jacobi: active wave-rounds %d, mean iterations %.1f, total wave-iterations %d (round 0 total %d)" % (w[13], w[12] / max(1, w[13]), w[12], w[10]))
print("pass 0 per workgroup (s_memtime ticks): setup %.0f  round 0 %.0f  later rounds %.0f" % (w[13] * 16 / max(1, w[11] // 2), w[14] * 16 / max(1, w[11] // 2), w[15] * 16 / max(1, w[11] // 2)))
