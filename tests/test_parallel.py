"""Multi-process (gloo, world size 2, CPU) tests of the multi-GPU layer: ray sharding, the fused tile
all-gather (equal and ragged), and the sharded-render driver.  The renderer here is a deterministic fake
(a pure function of the ray data): the point is the partition / exchange logic, the HIP march itself is
covered by the -m gpu tests and is bit-identical under ray_range slicing."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neuralbody_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FakeRenderer:
    """rgb = f(ray_o, ray_d) so that any correct sharding reproduces the serial result exactly."""

    def render(self, batch, ray_range=None):
        b, e = ray_range if ray_range is not None else (0, batch["ray_o"].shape[1])
        o, d = batch["ray_o"][0, b:e], batch["ray_d"][0, b:e]
        rgb = torch.sin(o * 3.0 + d * 7.0)
        return {"rgb_map": rgb[None], "acc_map": (o * d).sum(-1)[None]}


def _worker(rank, world, port, n, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        batch = {"ray_o": torch.randn(1, n, 3, generator=g), "ray_d": torch.randn(1, n, 3, generator=g)}
        ren = FakeRenderer()
        full = ren.render(batch)
        got = parallel.render_sharded(ren, batch, keys=("rgb_map", "acc_map"))
        assert torch.equal(got["rgb_map"], full["rgb_map"]), "sharded rgb differs"
        assert torch.equal(got["acc_map"], full["acc_map"]), "sharded acc differs"
        # equal-size fast path
        tile = torch.full((5, 3), float(rank))
        allt = parallel.all_gather_tiles(tile)
        assert allt.shape == (5 * world, 3) and torch.equal(allt[5 * rank:5 * rank + 5], tile)
        # ragged path with explicit sizes
        sizes = [3 + r for r in range(world)]
        t = torch.arange(sizes[rank] * 2, dtype=torch.float32).view(sizes[rank], 2) + 100 * rank
        cat = parallel.all_gather_tiles(t, sizes=sizes)
        assert cat.shape == (sum(sizes), 2)
        off = sum(sizes[:rank])
        assert torch.equal(cat[off:off + sizes[rank]], t)
        # bench.py's cross-rank bookkeeping: MAX-reduced elapsed time, per-rank rows in rank order, one arithmetic per job
        elapsed, rows = parallel.reduce_timings(1.0 + rank, [10.0 * rank, 0.5, 7.0 + rank], precision_code=1)
        assert elapsed == float(world) and len(rows) == world
        assert all(rows[r] == [10.0 * r, 0.5, 7.0 + r] for r in range(world))
        try:
            parallel.reduce_timings(1.0, [0.0], precision_code=rank)  # the ranks disagree
            raise AssertionError("ranks with different arithmetics must be refused")
        except RuntimeError as e:
            assert "different decoder arithmetics" in str(e)
        np.save(os.path.join(out_dir, "ok_%d.npy" % rank), np.ones(1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [1000, 7, 1])
def test_sharded_render_gloo_world2(tmp_path, n):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert os.path.exists(tmp_path / ("ok_%d.npy" % r))


def test_reduce_timings_without_a_process_group():
    assert parallel.reduce_timings(2.5, [1.0, 2.0], 1) == (2.5, [[1.0, 2.0]])


def test_shard_range_tiles_exactly():
    for n in (0, 1, 7, 64, 262144, 262145):
        for world in (1, 2, 3, 4, 8):
            ranges = [parallel.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 0
    with pytest.raises(ValueError):
        parallel.shard_range(10, 2, 2)
