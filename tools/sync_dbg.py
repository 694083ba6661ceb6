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
nwg = w[11]
print("words", w)
print("pass 0: %d workgroups, rounds per workgroup mean %.2f max %d" % (nwg, w[8] / max(1, nwg), w[9]))
print("sub-sequence decodes %d, symbol steps %d (%.1f per decode)" % (w[10], w[12], w[12] / max(1, w[10])))
print("per workgroup (s_memtime ticks): setup %.0f  round 0 %.0f  later rounds %.0f" % (w[13] * 16 / max(1, nwg), w[14] * 16 / max(1, nwg), w[15] * 16 / max(1, nwg)))
