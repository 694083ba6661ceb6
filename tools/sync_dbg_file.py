#!/usr/bin/env python3
"""tools/sync_dbg_file.py <file.jpg> -- K1 loop counters on a JPEG file (needs a -DKPEG_SYNC_STATS=1 build via KPEG_HIP_LIB)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import libkpeg_amd as K
data = open(sys.argv[1], "rb").read()
rc, frame, scan = K.host_parse(data)
ctx = K.Context(0)
ctx.decode_scan(frame, scan)
out = (ctypes.c_uint32 * 16)()
ctx.lib.kpeg_hip_debug_words(ctx._h, out, 16)
w = list(out)
nwg = max(1, w[11])
print("%s %dx%d scan %d bytes (%.2f bits/pixel)" % (sys.argv[1], frame.width, frame.height, len(scan), len(scan) * 8.0 / (frame.width * frame.height)))
print("pass 0: %d workgroups, rounds per workgroup mean %.2f max %d" % (nwg, w[8] / nwg, w[9]))
print("sub-sequence decodes %d, symbol steps %d (%.1f per decode)" % (w[10], w[12], w[12] / max(1, w[10])))
