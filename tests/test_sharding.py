"""N>1 path on CPU: world_size-2 (and 3) gloo jobs run the product's stripe splitting + gather with
the oracle as the per-stripe decoder."""
import os
import socket
import subprocess
import sys

import pytest

import kpeg_testlib as T


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,w,h,interval", [(2, 64, 64, 8), (2, 128, 48, 16), (3, 64, 72, 8), (2, 64, 40, 4)])
def test_stripe_sharding_and_gather(world, w, h, interval):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(T.ROOT, "tests", "_shard_worker.py"), str(w), str(h), str(interval), "31"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "SHARD_OK %d %d %d %d" % (world, w, h, interval) in out.stdout
