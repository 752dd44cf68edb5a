"""Device selection + the batch-axis sharding the reference lacks.

``dev()`` / ``setup_dist()`` keep the reference's semantics (utils/dist_util.py:18-51: one device
picked by index).  Sampling is embarrassingly parallel over the batch, so the
multi-GPU path is: one process per GPU, every rank samples its contiguous slice of the batch with
noise keyed by GLOBAL sample index, and ONE RCCL all-gather reassembles the result.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

used_device = 0


def setup_dist(device=0):
    """Select the device; if launched under torchrun (RANK/WORLD_SIZE set) also join the process
    group (backend nccl = RCCL on ROCm, gloo on CPU)."""
    global used_device
    used_device = device
    if dist.is_available() and not dist.is_initialized() and "RANK" in os.environ \
            and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            used_device = int(os.environ.get("LOCAL_RANK", device))
            torch.cuda.set_device(used_device)
        dist.init_process_group(backend=backend, init_method="env://")


def dev():
    if torch.cuda.is_available() and used_device >= 0:
        return torch.device(f"cuda:{used_device}")
    return torch.device("cpu")


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n: int, rank: int, world_size: int):
    """Contiguous slice [lo, hi) of a length-n batch owned by `rank` (sizes differ by at most 1)."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(obj, rank: int, world_size: int, n: int):
    """Slice every tensor / list whose leading dimension is the batch; recurse into dicts."""
    lo, hi = shard_bounds(n, rank, world_size)
    if isinstance(obj, dict):
        return {k: shard_batch(v, rank, world_size, n) for k, v in obj.items()}
    if torch.is_tensor(obj) and obj.dim() > 0 and obj.shape[0] == n:
        return obj[lo:hi]
    if isinstance(obj, (list, tuple)) and len(obj) == n:
        return type(obj)(obj[lo:hi])
    return obj


def shard_call(shape, model_kwargs, noise=None, init_image=None):
    """This rank's share of ``diffusion.p_sample_loop(model, shape, model_kwargs=..., noise=..., init_image=...)``:
    returns (local_shape, local_model_kwargs, local_noise, local_init_image).  Every batch-leading tensor / list in
    model_kwargs (``y`` entries, ``obs_x0`` / ``obs_mask``) is sliced to the rank's contiguous range and
    ``y['first_sample']`` is set to the GLOBAL index of its first sample, which the sampler forwards to the engine's
    counter-based generator: with the same torch seed on every rank (utils.fixseed) the gathered result equals the
    single-device batch bit for bit.  Reassemble with all_gather_batch(local_sample, shape[0])."""
    rank, world_size = world()
    n = int(shape[0])
    lo, hi = shard_bounds(n, rank, world_size)
    local = shard_batch(model_kwargs, rank, world_size, n)
    local = dict(local)
    local['y'] = dict(local.get('y', {}), first_sample=lo)
    cut = lambda t: None if t is None else t[lo:hi]
    return (hi - lo,) + tuple(shape[1:]), local, cut(noise), cut(init_image)


def all_gather_batch(local: torch.Tensor, n: int) -> torch.Tensor:
    """Reassemble the full batch on every rank: ONE all-gather (RCCL over xGMI on GPUs).  Ranks may
    own slices that differ by one sample, so slices are padded to the largest and trimmed after."""
    rank, world_size = world()
    if world_size == 1 and not (dist.is_available() and dist.is_initialized()):
        return local          # (an initialised 1-rank group still takes the collective: same code path as N > 1)
    sizes = [shard_bounds(n, r, world_size) for r in range(world_size)]
    biggest = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < biggest:
        pad = torch.cat([local, local.new_zeros((biggest - local.shape[0],) + tuple(local.shape[1:]))])
    out = local.new_empty((world_size * biggest,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, pad.contiguous())
    parts = [out[r * biggest:r * biggest + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, dim=0)
