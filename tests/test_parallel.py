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


class FakeImageRenderer(FakeRenderer):
    """... with an image geometry, as the real Renderer has: render_sharded then hands out whole 8-row tile bands."""

    def __init__(self, H, W):
        from types import SimpleNamespace

        self.cfg = SimpleNamespace(H=H, W=W)
        self.seen = None

    def render(self, batch, ray_range=None):
        self.seen = ray_range
        return FakeRenderer.render(self, batch, ray_range)


def _worker_tiled(rank, world, port, H, W, covered, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(1)
        mask = torch.ones(H * W, dtype=torch.bool) if covered else (torch.rand(H * W, generator=g) < 0.4)
        n = int(mask.sum())
        batch = {"ray_o": torch.randn(1, n, 3, generator=g), "ray_d": torch.randn(1, n, 3, generator=g), "mask_at_box": mask[None]}
        ren = FakeImageRenderer(H, W)
        got = parallel.render_sharded(ren, batch, keys=("rgb_map", "acc_map"))
        b, e = ren.seen
        full = FakeRenderer().render(batch)
        assert torch.equal(got["rgb_map"], full["rgb_map"]) and torch.equal(got["acc_map"], full["acc_map"])
        # the range is the rays of whole 8-row bands: it starts with the first ray of a band and ends with the last ray of one
        pix = torch.nonzero(mask).reshape(-1)
        band = torch.div(pix, 8 * W, rounding_mode="floor")
        if e > b:
            assert b == 0 or int(band[b - 1]) < int(band[b])
            assert e == n or int(band[e - 1]) < int(band[e])
        if covered:
            r0, r1 = parallel.shard_tile_rows(H, rank, world)
            assert (b, e) == (r0 * W, r1 * W)
        np.save(os.path.join(out_dir, "ok_%d.npy" % rank), np.array([b, e]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W,covered", [(2, 32, 24, True), (4, 40, 16, True), (4, 20, 16, False), (2, 8, 8, True)])
def test_tile_aligned_sharding_gloo(tmp_path, world, H, W, covered):
    """world-2 and world-4: every rank gets whole 8-row tile bands (ragged: 5 bands over 4 ranks, 3 bands over 4, 1 band over 2),
    the gathered image equals the serial one, and the ranges tile the ray list."""
    port = _free_port()
    mp.spawn(_worker_tiled, args=(world, port, H, W, covered, str(tmp_path)), nprocs=world, join=True)
    ranges = [np.load(tmp_path / ("ok_%d.npy" % r)) for r in range(world)]
    assert ranges[0][0] == 0 and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))


def test_partially_covered_view_is_split_by_rays_not_by_rows():
    """a mask_at_box view (the subject in the middle third of the rows) — equal band counts would give the outer ranks
    nothing and the middle ranks everything; the cuts follow the cumulative ray count instead, still on band borders."""
    H, W, world = 256, 64, 8
    mask = torch.zeros(H, W, dtype=torch.bool)
    mask[88:168, 10:50] = True  # 80 rows x 40 pixels = 3200 rays in rows 88..167 (bands 11..20)
    n = int(mask.sum())
    ranges = parallel.shard_ranges_tiled(n, world, H, W, mask.reshape(-1))
    assert ranges[0][0] == 0 and ranges[-1][1] == n and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    sizes = [e - b for b, e in ranges]
    assert max(sizes) <= 2 * 320 and min(sizes) >= 320  # 10 bands of 320 rays over 8 ranks: one or two bands each
    assert all(b % 320 == 0 and e % 320 == 0 for b, e in ranges)  # whole bands
    rows_only = [parallel.shard_tile_rows(H, r, world) for r in range(world)]  # what round 5 did: ranks 0, 1, 6, 7 got no ray at all
    assert sum(int(mask[r0:r1].sum()) == 0 for r0, r1 in rows_only) >= 4
    assert parallel.shard_range_tiled(n, 3, world, H, W, mask.reshape(-1)) == ranges[3]
    # the cuts are the nearest band borders to r / world of the rays; one read-back serves every rank
    cum = parallel.band_ray_counts(mask.reshape(-1), H, W)
    assert cum.tolist()[11] == 0 and cum.tolist()[21] == n and len(cum) == H // 8 + 1
    assert parallel.balanced_band_cuts([0, 10, 10, 10, 20], 2) == [0, 1, 4]  # ties go to the earlier border
    assert parallel.balanced_band_cuts([0, 0, 0], 4) == [0, 0, 0, 0, 2]  # an empty view: the bands go to the last rank, which holds no ray either
    # an image height that is not a multiple of the band height
    m2 = torch.ones(20, 16, dtype=torch.bool)
    r2 = parallel.shard_ranges_tiled(320, 2, 20, 16, m2.reshape(-1))
    assert r2 == [(0, 128), (128, 320)]  # borders at 0 / 128 / 256 / 320 rays: 128 is nearest to 160 (tie with none)


def test_shard_tile_rows_balance():
    for H in (8, 20, 512, 520, 1024):
        for world in (1, 2, 3, 4, 8):
            rows = [parallel.shard_tile_rows(H, r, world) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == H and all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
            bands = [(b - a + 7) // 8 for a, b in rows]
            assert max(bands) - min(bands) <= 1
    assert parallel.shard_range_tiled(512 * 512, 3, 8, 512, 512) == (3 * 64 * 512, 4 * 64 * 512)
    with pytest.raises(ValueError):
        parallel.shard_range_tiled(100, 0, 2, 16, 16)


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


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(argv, env_extra=None, timeout=300):
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_launches_itself_for_more_than_one_gpu():
    """`python bench.py --gpus 2` with no launcher around it (what a driver that ran `python bench.py --gpus 1` will type next):
    the process re-executes itself under torch.distributed.run, two ranks rendezvous on 127.0.0.1 (gloo here: --dry-run runs the
    launch / barrier / MAX-over-ranks / per-rank bookkeeping with stub steps and no HIP path), rank 0 prints ONE JSON line."""
    import json

    out = _bench(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["dry_run"] is True and j["value"] is None and j["n_gpus"] == 2 and j["backend"] == "gloo"
    assert [r["rank"] for r in j["per_rank"]] == [0, 1] and [r["march_ms"] for r in j["per_rank"]] == [2.0, 4.0]
    assert j["ms_per_step"] >= 4.0  # the MAX over the ranks: rank 1's stub step sleeps 4 ms
    assert "re-executing under torch.distributed.run" in out.stderr


def test_bench_refuses_more_gpus_than_the_box_has():
    if torch.cuda.is_available() and torch.cuda.device_count() >= 64:
        pytest.skip("a box with 64 devices")
    out = _bench(["--gpus", "64", "--steps", "1"])
    assert out.returncode != 0 and "HIP device(s) visible" in out.stderr and "Traceback" not in out.stderr


def test_bench_names_a_world_size_mismatch():
    out = _bench(["--gpus", "2", "--dry-run"], {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                                "MASTER_PORT": str(_free_port())})
    assert out.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in out.stderr


def test_balanced_band_cuts_properties():
    """For any band histogram and world size: the cuts are monotone band borders from 0 to the last band, and no rank's ray count
    is further from the ideal n / world than the largest band (the best whole bands allow, up to the nearest-border rule)."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=200, deadline=None)
    @given(st.lists(st.integers(min_value=0, max_value=5000), min_size=1, max_size=80), st.integers(min_value=1, max_value=16))
    def check(per_band, world):
        cum = [0]
        for v in per_band:
            cum.append(cum[-1] + v)
        cuts = parallel.balanced_band_cuts(cum, world)
        assert len(cuts) == world + 1 and cuts[0] == 0 and cuts[-1] == len(per_band)
        assert all(a <= b for a, b in zip(cuts, cuts[1:]))
        total, biggest = cum[-1], max(per_band)
        for r in range(1, world):
            assert abs(cum[cuts[r]] - total * r / world) <= biggest / 2 + 1e-9 or cuts[r] in (cuts[r - 1],)  # nearest border (or pinned by monotonicity)
        sizes = [cum[cuts[r + 1]] - cum[cuts[r]] for r in range(world)]
        assert sum(sizes) == total and all(sz <= total / world + biggest + 1e-9 for sz in sizes)

    check()
