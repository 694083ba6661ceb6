#!/usr/bin/env python3
"""tools/sync_dbg.py -- per-wavefront timelines of k_sync_pass (pass 0) and k_write on the bench image (needs a
-DKPEG_SYNC_STATS=1 build: KPEG_HIP_LIB=build/ablate/libkpeg_hip_stats.so).  Experiment tool, not part of the product."""
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
for _ in range(3):
    rgb = ctx.decode_scan(frame, scan)
ctx.set_profiling(True)
ctx.decode_scan(frame, scan)
print({k: round(v, 4) for k, v in ctx.timings().items() if k.endswith("_ms")})


def stamps(which):
    a = np.zeros(8192 * 16, np.uint64)
    assert ctx.lib.kpeg_hip_debug_entropy_stamps(which, a.ctypes.data_as(ctypes.c_void_p), a.size) == 0
    a = a.reshape(8192, 16)
    return a[a[:, 0] != 0]


def pct(x):
    return "min %.0f  p10 %.0f  med %.0f  p90 %.0f  max %.0f" % tuple(np.percentile(x, [0, 10, 50, 90, 100]))


k1 = stamps(0)
t0 = k1[:, 0].min()
print("K1 pass 0: %d wavefronts; clock ticks (s_memtime) relative to the first wavefront's start" % len(k1))
print("  start            ", pct((k1[:, 0] - t0).astype(float)))
print("  tables+stage     ", pct((k1[:, 1] - k1[:, 0]).astype(float)))
print("  first decode     ", pct((k1[:, 2] - k1[:, 1]).astype(float)))
print("  clear            ", pct((k1[:, 8] - k1[:, 2]).astype(float)))
print("  rounds + waits   ", pct((k1[:, 15] - k1[:, 8]).astype(float)))
print("  counts' decode   ", pct((k1[:, 3] - k1[:, 15]).astype(float)))
M = np.uint64((1 << 56) - 1)
prev = k1[:, 8]
for q in range(6):
    tq = k1[:, 9 + q] & M
    ok = tq != 0
    if ok.sum() == 0:
        break
    print("  round %d: %4d waves, lanes decoding %s;  ticks since the round before %s" % (q + 1, ok.sum(), pct((k1[ok, 9 + q] >> np.uint64(56)).astype(float)), pct((tq[ok] - prev[ok]).astype(float))))
    prev = np.where(ok, tq, prev)
print("  end              ", pct((k1[:, 3] - t0).astype(float)))
rounds = (k1[:, 5] >> np.uint64(32)).astype(float)
runs = (k1[:, 5] & np.uint64(0xFFFFFFFF)).astype(float)
mx = (k1[:, 6] >> np.uint64(32)).astype(float)
it = (k1[:, 6] & np.uint64(0xFFFFFFFF)).astype(float)
print("  rounds per wave  ", pct(rounds), " decodes per wave", pct(runs))
print("  steps: busiest lane", pct(mx), " all lanes", pct(it), " idle polls", pct(k1[:, 7].astype(float)))
k2 = stamps(1)
t0 = k2[:, 0].min()
print("K2: %d wavefronts" % len(k2))
print("  start            ", pct((k2[:, 0] - t0).astype(float)))
print("  tables+stage     ", pct((k2[:, 1] - k2[:, 0]).astype(float)))
print("  scan             ", pct((k2[:, 2] - k2[:, 1]).astype(float)))
print("  decode loop      ", pct((k2[:, 3] - k2[:, 2]).astype(float)))
print("  epilogue         ", pct((k2[:, 4] - k2[:, 3]).astype(float)))
print("  end              ", pct((k2[:, 4] - t0).astype(float)))
print("  steps: busiest lane", pct((k2[:, 6] >> np.uint64(32)).astype(float)), " all lanes", pct((k2[:, 6] & np.uint64(0xFFFFFFFF)).astype(float)))
