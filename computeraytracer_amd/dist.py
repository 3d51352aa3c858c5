"""Tile sharding across the GPUs of one node.

One process per GPU.  Every rank holds the whole scene + BVH and renders the 8-row bands b of the frame with
b % world == rank (crt_config.band_rank / band_world); pixels are independent given the scene and the RNG is keyed by
the global pixel position, so the partition does not change any pixel.  The only exchange step is one all-gather of the
packed band buffers (RCCL over xGMI on GPUs, gloo on CPU in the tests) after the last iteration, followed by an
un-interleave.  torch / torch.distributed are used for that plumbing only.

A process that uses both must `import torch` BEFORE computeraytracer_amd.abi loads libcrt.so: PyTorch-ROCm ships its own
HIP runtime under the same soname, the first one loaded is the one both share, and libcrt.so loaded first leaves the
process with two runtimes of which the second cannot open the GPU (crt_create then reports "no usable HIP device").
"""
import numpy as np

from . import abi


def local_rows(height, rank, world):
    """Number of rows in rank's packed band buffer (whole 8-row bands)."""
    nb = (height + 7) // 8
    return 8 * len(range(rank, nb, world))


def pack_bands(full, rank, world):
    """numpy reference of the packed band layout: full[H, W, C] -> packed[local_rows, W, C] (rows past H stay zero)."""
    h = full.shape[0]
    out = np.zeros((local_rows(h, rank, world),) + full.shape[1:], dtype=full.dtype)
    for prow in range(out.shape[0]):
        y = ((prow // 8) * world + rank) * 8 + prow % 8
        if y < h:
            out[prow] = full[y]
    return out


def row_index(height, world, max_rows):
    """For every image row y: its position in the concatenation of the ranks' (padded to max_rows) band buffers."""
    idx = np.zeros(height, dtype=np.int64)
    for y in range(height):
        band = y // 8
        idx[y] = (band % world) * max_rows + (band // world) * 8 + y % 8
    return idx


def assemble(gathered, height, world):
    """gathered: torch tensor [world * max_rows, W, C] (rank-major) -> image [height, W, C] on the same device."""
    import torch
    max_rows = gathered.shape[0] // world
    idx = torch.from_numpy(row_index(height, world, max_rows)).to(gathered.device)
    return gathered.index_select(0, idx)


def gather_image(band, height, world, group=None):
    """All-gather the ranks' packed band tensors ([rows_r, W, C], rows_r may differ by 8) and un-interleave them into
    the full image on every rank."""
    import torch
    import torch.distributed as dist
    max_rows = local_rows(height, 0, world)
    if band.shape[0] < max_rows:
        pad = torch.zeros((max_rows - band.shape[0],) + tuple(band.shape[1:]), dtype=band.dtype, device=band.device)
        band = torch.cat([band, pad], 0)
    out = torch.empty((world * max_rows,) + tuple(band.shape[1:]), dtype=band.dtype, device=band.device)
    dist.all_gather_into_tensor(out, band.contiguous(), group=group)
    return assemble(out, height, world)


def context_band_tensor(ctx, width, device):
    """Copy a context's packed band buffer into a fresh torch tensor on `device` (device-to-device)."""
    import torch
    _, nbytes = ctx.image_device_ptr()
    rows = nbytes // (width * 16)
    t = torch.empty((rows, width, 4), dtype=torch.float32, device=device)
    ctx.image_copy_to_device(t.data_ptr(), nbytes)
    return t
