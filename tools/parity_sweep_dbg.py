#!/usr/bin/env python3
"""tools/parity_sweep_dbg.py [cases] [seed] -- extra randomised GPU-vs-oracle parity cases beyond tests/ (sizes, qualities,
noise levels, dense mode, restart intervals).  Experiment tool: prints the failures and exits non-zero if there are any."""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import kpeg_testlib as T, libkpeg_amd as K
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
ctx = K.Context(0)
bad = 0
for i in range(n):
    w = int(rng.integers(1, 90)) * 8; h = int(rng.integers(1, 60)) * 8
    q = int(rng.integers(5, 101)); sigma = float(rng.choice([0.0, 2.0, 6.0, 20.0, 60.0])); mode = int(rng.integers(0, 2))
    ri = int(rng.choice([0, 0, w // 8, 1, 5]))
    try:
        data = T.synth_jpeg(w, h, seed=int(rng.integers(1, 1 << 30)), quality=q, restart_interval=ri, sigma=sigma, mode=mode)
    except AssertionError:   # the test encoder's output buffer is too small for this case
        continue
    if ri:
        want, p, _ = T.oracle_decode_rst(data, ri)
    else:
        st, want = T.oracle_decode(data); p = T.oracle_parse(data)
        assert st == T.DECODE_DONE
    got = ctx.decode_scan(T.make_frame(p, ri), p.scan)
    nb = int((got != want).sum())
    if nb:
        bad += 1
        print("MISMATCH", dict(w=w, h=h, q=q, sigma=sigma, mode=mode, ri=ri), nb)
print("cases %d, mismatching %d" % (n, bad))
sys.exit(1 if bad else 0)
