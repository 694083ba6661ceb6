#!/usr/bin/env python3
"""tools/experiments/sharded_scan_time.py -- the host part of kpeg_hip_decode_sharded_dev on BASELINE config 5's image (16384 x 16384, a
restart interval per MCU row): with KPEG_DEBUG set the library prints what its restart-marker scan took; here the whole call with one,
two and four contexts on one GPU, pixels hashed against the reference's stripe hashes (tests/golden/manifest_large.json)."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["KPEG_DEBUG"] = "1"
import bench  # noqa: E402


def main():
    import torch
    import libkpeg_amd as K
    W = H = 16384
    data = bench.synth_jpeg(W, H, restart_interval=W // 8)
    rc, frame, scan = K.host_parse(data, allow_dri=True)
    assert rc == K.DECODE_DONE
    want = bench.pinned_stripe_shas(W, H, 1, 0)
    ctxs = [K.Context(0) for _ in range(4)]
    d = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    for n in (1, 2, 4):
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            K.decode_sharded(ctxs[:n], frame, scan, d_rgb_root=d.data_ptr())
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
        ok = None
        if want:
            rows = H // 8
            got = [hashlib.sha256(d[k * rows:(k + 1) * rows].cpu().numpy().tobytes()).hexdigest() for k in range(8)]
            ok = got == want
        print("%d context(s): %.2f ms for the call (scan of %d bytes uploaded from pageable host memory inside it); pixels %s" % (n, ms, len(scan), {True: "verified", False: "WRONG", None: "not pinned"}[ok]), flush=True)


if __name__ == "__main__":
    main()
