#!/usr/bin/env python3
"""tools/parity_large_dbg.py [n] -- a few large images (3840x2160, several seeds, qualities and noise levels), each decoded
three times on the GPU and compared with the oracle every time (soak for the fix-up passes' store ordering).  Experiment tool."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import kpeg_testlib as T, libkpeg_amd as K
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = K.Context(0)
bad = 0
for i in range(n):
    q = [75, 50, 90, 30, 85, 95][i % 6]; sigma = [6.0, 2.0, 12.0, 0.0, 20.0, 6.0][i % 6]
    data = T.synth_jpeg(3840, 2160, seed=9000 + i, quality=q, sigma=sigma)
    st, want = T.oracle_decode(data, 16)
    assert st == T.DECODE_DONE
    p = T.oracle_parse(data); f = T.make_frame(p)
    for rep in range(3):
        got = ctx.decode_scan(f, p.scan)
        nb = int((got != want).sum())
        if nb:
            bad += 1
            print("MISMATCH image", i, "rep", rep, nb)
    print("image", i, "q", q, "sigma", sigma, "ok", flush=True)
print("mismatching decodes:", bad)
sys.exit(1 if bad else 0)
