#!/usr/bin/env python3
"""tools/soak.py [seconds] [seed] -- randomised soak on the GPU box: random sizes, qualities, noise levels, restart
intervals, coefficient layouts, and the extensions (grayscale, any size, 4:2:0) against the oracle, until the time is
up.  Prints a line per mismatch (none expected) and a summary; exit code 1 on any mismatch."""
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kpeg_testlib as T  # noqa: E402
import libkpeg_amd as K  # noqa: E402
from PIL import Image  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
    ctx = K.Context(0)
    t0, n, bad, kinds = time.time(), 0, 0, {}
    while time.time() - t0 < budget:
        kind = str(rng.choice(["444", "444", "dri", "gray", "anysize", "420"]))
        q = int(rng.integers(5, 101))
        try:
            if kind in ("444", "dri"):
                w, h = int(rng.integers(1, 200)) * 8, int(rng.integers(1, 120)) * 8
                ri = 0 if kind == "444" else int(rng.choice([w // 8, 1, 3, 17]))
                try:
                    data = T.synth_jpeg(w, h, seed=int(rng.integers(1, 1 << 30)), quality=q, restart_interval=ri,
                                        sigma=float(rng.choice([0.0, 2.0, 6.0, 20.0, 60.0])), mode=int(rng.integers(0, 2)))
                except AssertionError:
                    continue
                if ri:
                    want, p, _ = T.oracle_decode_rst(data, ri)
                else:
                    st, want = T.oracle_decode(data)
                    p = T.oracle_parse(data)
                frame, scan = T.make_frame(p, ri), p.scan
                ctx.lib.kpeg_hip_debug_set(ctx._h, 7, int(rng.choice([0, 1, 2])))
                ctx.lib.kpeg_hip_debug_set(ctx._h, 2, int(rng.choice([-1, -1, 0, 1, 3])))   # K1's lead-in: cut short, k_sync_write's workgroups repair themselves
                ctx.lib.kpeg_hip_debug_set(ctx._h, 6, int(rng.choice([0, 0, 0, 8])))         # every third workgroup of k_sync_write gives up: its second launch decodes the call
            else:
                w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
                if kind == "gray":      # the grayscale oracle wants whole blocks
                    w, h = ((w + 7) & ~7), ((h + 7) & ~7)
                    if rng.random() < 0.5:
                        w = max(64, w & ~63)    # whole K4 tiles: the compact stream is possible (layout 2 asks for it)
                    ctx.lib.kpeg_hip_debug_set(ctx._h, 7, int(rng.choice([0, 1, 2])))
                y, x = np.mgrid[0:h, 0:w]
                px = np.stack([(x * 3 + y) % 256, (y * 2) % 256, (x + y * 5) % 256], -1) * float(rng.random()) + rng.normal(128, float(rng.choice([1, 10, 40])), (h, w, 3)) * float(rng.random())
                px = np.clip(px, 0, 255).astype(np.uint8)
                kw = {}
                if rng.random() < 0.3:
                    kw["optimize"] = True
                if rng.random() < 0.3:
                    kw["restart_marker_blocks"] = int(rng.integers(1, 40))
                buf = io.BytesIO()
                if kind == "gray":
                    Image.fromarray(px[..., 0], "L").save(buf, "JPEG", quality=q, **kw)
                    # the grayscale oracle wants whole blocks
                    if (w & 7) or (h & 7):
                        continue
                    st, want = T.oracle_decode_gray(buf.getvalue())
                elif kind == "anysize":
                    kw.pop("restart_marker_blocks", None)
                    Image.fromarray(px).save(buf, "JPEG", quality=q, subsampling=0, **kw)
                    st, want = T.oracle_decode_any_size(buf.getvalue())
                else:
                    Image.fromarray(px).save(buf, "JPEG", quality=q, subsampling=2, **kw)
                    st, want = T.oracle_decode_420(buf.getvalue())
                if st != T.DECODE_DONE:
                    continue
                rc, frame, scan = K.host_parse(buf.getvalue(), allow_dri=True, allow_gray=True, allow_any_size=True, allow_420=True)
                if rc != K.DECODE_DONE:
                    print("parser rejects what the oracle decodes:", kind, w, h, q, kw, rc, flush=True)
                    bad += 1
                    continue
            got = ctx.decode_scan(frame, scan)
        except OSError:      # Pillow's encoder gives up on some parameter combinations
            continue
        except K.KpegError as e:
            print("ERROR", kind, e, flush=True)
            bad += 1
            continue
        finally:
            ctx.lib.kpeg_hip_debug_set(ctx._h, 7, 0)
            ctx.lib.kpeg_hip_debug_set(ctx._h, 2, -1)
            ctx.lib.kpeg_hip_debug_set(ctx._h, 6, 0)
        n += 1
        kinds[kind] = kinds.get(kind, 0) + 1
        if not np.array_equal(got, want):
            bad += 1
            print("MISMATCH", kind, got.shape, q, int((got != want).sum()), flush=True)
        if n % 200 == 0:
            print("%d cases, %d bad, %.0f s" % (n, bad, time.time() - t0), flush=True)
    print("soak: %d cases %s, %d bad" % (n, kinds, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
