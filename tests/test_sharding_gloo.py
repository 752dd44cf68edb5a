"""N>1 path on CPU: batch sharding + the one all-gather, world_size 2 over gloo."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import sub


def test_shard_bounds_cover_batch():
    du = sub("utils.dist_util")
    for n in (1, 2, 7, 32, 1024, 1025):
        for w in (1, 2, 3, 8):
            spans = [du.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_shard_batch_slices_nested_kwargs():
    du = sub("utils.dist_util")
    y = {"y": {"mask": torch.arange(6).view(6, 1, 1, 1), "text": list("abcdef"),
               "text_scale": torch.ones(6), "imputate": True, "inpainted_motion": torch.zeros(6, 3, 1, 2)}}
    part = du.shard_batch(y, rank=1, world_size=2, n=6)
    assert part["y"]["mask"].flatten().tolist() == [3, 4, 5]
    assert part["y"]["text"] == ["d", "e", "f"] and part["y"]["imputate"] is True
    assert part["y"]["inpainted_motion"].shape == (3, 3, 1, 2)


def _worker(rank, world, port, n):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    import importlib
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    du = importlib.import_module("diffusion-motion-inbetweening_amd.utils.dist_util")
    dist.init_process_group("gloo", init_method="env://")
    try:
        full = torch.arange(n * 3 * 2, dtype=torch.float32).view(n, 3, 1, 2)
        lo, hi = du.shard_bounds(n, rank, world)
        out = du.all_gather_batch(full[lo:hi].clone(), n)
        assert torch.equal(out, full), (rank, out.shape)
        assert du.world() == (rank, world)
        # the documented torchrun flow: shard_call sets the global sample offset the sampler forwards to the engine
        kw = {"y": {"mask": torch.ones(n, 1, 1, 2, dtype=torch.bool), "text": ["t%d" % i for i in range(n)],
                    "imputate": True}, "obs_x0": full.clone()}
        shape, local, nz, init = du.shard_call((n, 3, 1, 2), kw, noise=full)
        assert shape == (hi - lo, 3, 1, 2) and local["y"]["first_sample"] == lo and init is None
        assert torch.equal(nz, full[lo:hi]) and torch.equal(local["obs_x0"], full[lo:hi])
        assert local["y"]["text"] == ["t%d" % i for i in range(lo, hi)] and local["y"]["imputate"] is True
        assert "first_sample" not in kw["y"]          # the caller's dict is not mutated
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [8, 7])
def test_all_gather_batch_world2(n):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, n), nprocs=2, join=True)
