#!/usr/bin/env python3
"""tools/umask_stats.py -- how K4's unsafe pixels are spread over the picture (GPU box).

Decodes the bench's 8K picture (or --width/--height), reads K4's unsafe-pixel mask back (test hook kpeg_hip_debug_umask) and prints
what k_fixup's work distribution sees: marked pixels per tile and per chunk of FX_CHUNK_TILES tiles, for consecutive and
for interleaved chunks."""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=bench.W8K)
    ap.add_argument("--height", type=int, default=bench.H8K)
    ap.add_argument("--chunk", type=int, default=16)
    args = ap.parse_args()
    import torch
    import libkpeg_amd as K
    torch.cuda.set_stream(torch.cuda.Stream())
    W, H = args.width, args.height
    rc, frame, scan = K.host_parse(bench.synth_jpeg(W, H))
    assert rc == K.DECODE_DONE
    ctx = K.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_scan = torch.from_numpy(np.ascontiguousarray(scan)).cuda()
    d_rgb = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    ctx.decode_scan_dev(frame, d_scan.data_ptr(), d_scan.numel(), d_rgb.data_ptr())
    ctx.sync()
    ntiles = ((W // 8 + 7) // 8) * (H // 8)
    buf = np.zeros(ntiles * 64, np.uint8)
    ctx.lib.kpeg_hip_debug_umask.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    assert ctx.lib.kpeg_hip_debug_umask(ctx._h, buf.ctypes.data, buf.size) == 0
    per_tile = np.unpackbits(buf).reshape(ntiles, 512).sum(1)
    print("tiles %d, marked pixels %d, tiles with any %d (%.1f %%), max per tile %d" % (
        ntiles, per_tile.sum(), (per_tile > 0).sum(), 100.0 * (per_tile > 0).mean(), per_tile.max()))
    print("per tile histogram (0,1,2,3-4,5-8,9-16,17-32,33-64,65+):",
          [int(((per_tile >= a) & (per_tile <= b)).sum()) for a, b in ((0, 0), (1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 32), (33, 64), (65, 512))])
    C = args.chunk
    pad = (-ntiles) % C
    pt = np.concatenate([per_tile, np.zeros(pad, per_tile.dtype)])
    cons = pt.reshape(-1, C).sum(1)
    nch = cons.size
    inter = pt.reshape(C, nch).sum(0)   # chunk c = tiles c, c + nch, c + 2 nch, ...
    for name, v in (("consecutive", cons), ("interleaved", inter)):
        b = (v + 63) // 64
        print("%s chunks of %d tiles: %d chunks, mean %.1f, max %d marked; batches of 64: mean %.2f, max %d, chunks with > 1 batch %d" % (
            name, C, v.size, v.mean(), v.max(), b.mean(), b.max(), (b > 1).sum()))
    mcu = np.unpackbits(buf).reshape(ntiles, 8, 64).sum(2)   # marked pixels per MCU (lane >> 3 = MCU: 8 lanes x 8 bits)
    print("MCUs with any marked pixel: %d of %d; marked pixels per such MCU: mean %.2f" % ((mcu > 0).sum(), mcu.size, mcu[mcu > 0].mean()))


if __name__ == "__main__":
    main()
