#!/usr/bin/env python3
"""tools/fused_timeline.py -- per workgroup (wavefront 0 of each), k_sync_write on the 8K image (stats build, KPEG_FUSED=1):
when K1's part ends, when K2's loop starts and ends, on the workgroup's own clock, relative to its own start."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch, libkpeg_amd as K, bench
# python tools/fused_timeline.py [W H]  |  python tools/fused_timeline.py photo lena.jpg 75   (a committed photograph tiled to 7680x4352)
if len(sys.argv) > 3 and sys.argv[1] == "photo":
    data = bench.tiled_photo_jpeg(sys.argv[2], int(sys.argv[3]))
else:
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
k1_end = k1[w0, 3].astype(np.int64) - t_start      # wavefront 0's own K1 part
pub = k2[w0, 0].astype(np.int64) - t_start         # the workgroup's K1 part is over (all wavefronts), totals published
got = k2[w0, 1].astype(np.int64) - t_start         # wavefront 0 has every predecessor's record it takes
k2_beg = k2[w0, 2].astype(np.int64) - t_start
k2_end = k2[w0, 4].astype(np.int64) - t_start
# s_memtime counts per XCD (the XCDs' counters are far apart); s_memrealtime (100 MHz) is one clock for the chip.  Every wavefront stamped
# both at the end of its K1 part (k1[:, 3] / k1[:, 4]) and at the end of K2 (k2[:, 4] / k2[:, 5]): ticks per 10 ns from those, then every
# s_memtime stamp of a wavefront goes to the common clock.
tm3, rt3 = k1[w0, 3].astype(np.float64), k1[w0, 4].astype(np.float64)
tw4, rt4 = k2[w0, 4].astype(np.float64), k2[w0, 5].astype(np.float64)
ratio = np.median((tw4 - tm3) / np.maximum(rt4 - rt3, 1))
r0 = (rt3 - (tm3 - k1[w0, 0].astype(np.float64)) / ratio).min()   # the first workgroup's start
def absolute(col):   # 10 ns units since the first workgroup started
    return rt3 + (col.astype(np.float64) - tm3) / ratio - r0
a_k1, a_pub, a_got, a_beg, a_end = absolute(k1[w0, 3]), absolute(k2[w0, 0]), absolute(k2[w0, 1]), absolute(k2[w0, 2]), absolute(k2[w0, 4])
latest_pub_before = np.concatenate([[0.0], np.maximum.accumulate(a_pub)[:-1]])   # when the last of workgroups 0..g-1 published
print("shader clock %.0f MHz; times in us since the first workgroup started" % (ratio * 100))
print("  wg  wave0 K1 ends  published  last pred. published  wave0 records in  K2 loop starts  K2 ends  hand-over (K2 start - max(published, last pred.))")
ho = a_beg - np.maximum(a_pub, latest_pub_before)
for g in list(range(0, nwg, max(1, nwg // 24))) + [nwg - 1]:
    print("%4d %12.2f %10.2f %20.2f %17.2f %15.2f %8.2f %10.2f" % (g, a_k1[g] / 100, a_pub[g] / 100, latest_pub_before[g] / 100, a_got[g] / 100, a_beg[g] / 100, a_end[g] / 100, ho[g] / 100))
print("last to publish: wg %d at %.2f us; last K2 end %.2f us" % (int(np.argmax(a_pub)), a_pub.max() / 100, a_end.max() / 100))
print("hand-over: median %.2f us, 90 %% %.2f, max %.2f; of the workgroups behind the last to publish: median %.2f" % (
    np.median(ho) / 100, np.percentile(ho, 90) / 100, ho.max() / 100, np.median(ho[int(np.argmax(a_pub)) + 1:]) / 100 if int(np.argmax(a_pub)) + 1 < nwg else 0))
# K2's part, every wavefront: loop and epilogue in shader cycles, symbol steps of its busiest lane
allw = np.arange(nwg * 8)
loop = (k2[allw, 3].astype(np.int64) - k2[allw, 2].astype(np.int64))
epi = (k2[allw, 4].astype(np.int64) - k2[allw, 3].astype(np.int64))
mx = (k2[allw, 6] >> np.uint64(32)).astype(np.int64)
sm = (k2[allw, 6] & np.uint64(0xFFFFFFFF)).astype(np.int64)
ok = k2[allw, 2] != 0
pc = lambda a: " ".join("%7d" % np.percentile(a, q) for q in (0, 10, 50, 90, 99, 100))
print("K2 per wavefront (cycles; percentiles 0 10 50 90 99 100):")
print("   loop            %s" % pc(loop[ok]))
print("   epilogue        %s" % pc(epi[ok]))
print("   steps, busiest lane %s   mean steps per lane %.1f" % (pc(mx[ok]), sm[ok].sum() / (64.0 * ok.sum())))
print("   cycles per step of the busiest lane: median %.0f" % np.median(loop[ok] / np.maximum(mx[ok], 1)))
for name, sel in (("workgroups 0..255", allw < 256 * 8), ("256..511", (allw >= 256 * 8) & (allw < 512 * 8)), ("512..", allw >= 512 * 8)):
    s2 = ok & sel
    if s2.any():
        print("   %-18s loop median %6d  epilogue median %6d  cycles per step %4.0f" % (name, np.median(loop[s2]), np.median(epi[s2]), np.median(loop[s2] / np.maximum(mx[s2], 1))))
# per workgroup: the slowest wavefront's loop end against wavefront 0's
wl = loop.reshape(nwg, 8)
print("   slowest wavefront's loop / wavefront 0's, per workgroup: median %.2f" % np.median(wl.max(1) / np.maximum(wl[:, 0], 1)))
