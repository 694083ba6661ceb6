"""Worker for tests/test_sharding.py: world_size-N gloo job on CPU.  The per-stripe decoder is the
oracle (there is no GPU here); the splitting, the stripe geometry and the gather are the product's
libkpeg_amd.sharded code."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kpeg_testlib as T  # noqa: E402
from libkpeg_amd import sharded  # noqa: E402


def oracle_stripe_decoder(interval):
    def decode(frame, scan_slice, first_row, rows):
        # a stripe is a self-contained DRI stream of `rows` MCU rows
        p = T.Parsed(T.DECODE_DONE, frame.width, rows * 8,
                     np.array([[frame.qt[t][k] for k in range(64)] for t in range(2)] * 2, dtype=np.uint16),
                     [[(bytes(frame.dht[c][i].counts), bytes(frame.dht[c][i].symbols), 1) for i in range(2)] for c in range(2)],
                     scan_slice.tobytes(), 2)
        rc, coef = T.oracle_entropy(p, interval)
        assert rc == 0, rc
        return torch.from_numpy(T.oracle_idct_colour(coef, p.qt, frame.width, rows * 8, 1))
    return decode


def main():
    w, h, interval, seed = (int(x) for x in sys.argv[1:5])
    dist.init_process_group(backend="gloo")
    data = T.synth_jpeg(w, h, seed=seed, restart_interval=interval)
    full = sharded.decode_sharded(data, oracle_stripe_decoder(interval))
    if dist.get_rank() == 0:
        want, _, _ = T.oracle_decode_rst(data, interval)
        assert tuple(full.shape) == want.shape, (full.shape, want.shape)
        assert np.array_equal(full.numpy(), want)
        print("SHARD_OK", dist.get_world_size(), w, h, interval)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
