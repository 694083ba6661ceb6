"""Row-stripe sharding of one large DRI image over the ranks of a torch.distributed job
(one process per GPU, RCCL over xGMI on the GPU box; gloo on CPU in the tests).

Restart intervals are byte-aligned and reset the DC predictors, so a stripe of whole MCU rows can
be decoded from the bytes of its own restart intervals alone: no halo, no data-path collective.
The path's one exchange step is the gather of the decoded RGB rows to rank 0.
(The reference has no counterpart: it rejects DRI.  SURVEY.md 8e.)
"""
import numpy as np

import libkpeg_amd as K


def plan(frame, scan, world):
    """[(first_mcu_row, mcu_rows, byte_begin, byte_end)] for ranks 0..world-1."""
    mw, mh = frame.width // 8, frame.height // 8
    return K.stripe_ranges(scan, mh, mw, frame.restart_interval, world)


def gpu_stripe_decoder(ctx):
    """decode_fn for decode_sharded(): K0..K4 on this rank's GPU, result stays in HBM."""
    import torch

    def decode(frame, scan_slice, first_row, rows):
        d_scan = torch.from_numpy(np.ascontiguousarray(scan_slice)).cuda()
        d_rgb = torch.empty((rows * 8, frame.width, 3), dtype=torch.uint8, device="cuda")
        h = torch.cuda.current_stream().cuda_stream
        if h:
            ctx.set_stream(h)            # ordered with the torch ops around it
        else:
            torch.cuda.synchronize()     # default stream (handle 0): the context keeps its own stream, so wait for the upload
            ctx.use_own_stream()
        ctx.decode_stripe_dev(frame, d_scan.data_ptr(), d_scan.numel(), first_row, rows, d_rgb.data_ptr())
        ctx.sync()
        return d_rgb

    return decode


def decode_sharded(data, decode_fn, dst=0):
    """Every rank passes the same JFIF bytes; returns the full image tensor on rank `dst`, None elsewhere.
    decode_fn(frame, scan_slice, first_mcu_row, mcu_rows) -> uint8 tensor [mcu_rows*8, width, 3]."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    rc, frame, scan = K.host_parse(data, allow_dri=True)
    if rc != K.DECODE_DONE:
        raise K.KpegError(K.E_ARG, "marker parser returned %d" % rc)
    if frame.restart_interval == 0:
        raise K.KpegError(K.E_UNSUPPORTED, "a stream without restart markers is one serial bit string: it cannot be sharded")
    ranges = plan(frame, scan, world)
    first_row, rows, b0, b1 = ranges[rank]
    max_rows = max(r[1] for r in ranges)
    stripe = decode_fn(frame, scan[b0:b1], first_row, rows) if rows else None
    # equal-sized gather buffers (the last stripe may be shorter)
    dev = stripe.device if stripe is not None else torch.device("cpu")
    buf = torch.zeros((max_rows * 8, frame.width, 3), dtype=torch.uint8, device=dev)
    if stripe is not None:
        buf[: rows * 8] = stripe
    glist = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, glist, dst=dst)
    if rank != dst:
        return None
    return torch.cat([glist[r][: ranges[r][1] * 8] for r in range(world)], dim=0)
