"""N > 1 path on CPU: 2, 3 and 8 gloo ranks exchange packed band buffers and must reassemble the exact image
(8 ranks at the headline height 2160: ranks 0-5 own 34 bands, ranks 6-7 own 33, so the padded all-gather is exercised)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from computeraytracer_amd import dist as cdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, h, w, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if world == 8 and h == 2160:
        assert cdist.local_rows(h, rank, world) == (34 if rank < 6 else 33) * 8
    full = np.random.default_rng(7).random((h, w, 4)).astype(np.float32)        # the same "rendered frame" on every rank
    band = torch.from_numpy(cdist.pack_bands(full, rank, world))                  # what this rank's context would hold
    img = cdist.gather_image(band, h, world)
    ok = bool(np.array_equal(img.numpy(), full))
    t = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        q.put(float(t[0]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h,w", [(2, 64, 24), (2, 225, 40), (3, 100, 16), (8, 2160, 8)])
def test_band_gather_reassembles_the_frame(world, h, w):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, h, w, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert q.get(timeout=10) == 1.0


def test_pack_and_row_index_are_inverse():
    for h, world in [(7, 2), (64, 8), (225, 3), (2160, 8)]:
        full = np.arange(h * 2 * 1, dtype=np.float32).reshape(h, 2, 1)
        max_rows = cdist.local_rows(h, 0, world)
        cat = np.zeros((world * max_rows, 2, 1), np.float32)
        for r in range(world):
            b = cdist.pack_bands(full, r, world)
            cat[r * max_rows:r * max_rows + b.shape[0]] = b
        assert np.array_equal(cat[cdist.row_index(h, world, max_rows)], full)
