"""Multi-GPU layer of the hot path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

Rays are independent given the (replicated, read-only) feature volume and weights, so the path shards
with NO data-path collective: every rank marches a contiguous range of the compacted ray list (or its
own views) and one all-gather of the rendered tiles assembles the image(s).  The reference has no
multi-GPU inference at all; training there is stock DDP (lib/train/trainers/trainer.py:13-18).

Message sizes: rgb tiles are n_rays x 3 fp32 — 3 MB per 512x512 view, 12.6 MB at 1024x1024 — far below
what saturates a 153 GB/s xGMI link, so a single fused all-gather per image is the right granularity
(latency-bound; splitting it into per-tile messages would only multiply launch latency).
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced [begin, end) of `n` rays for `rank` (sizes differ by at most one; ranges tile
    [0, n) exactly; empty when n < world for the trailing ranks)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world %d/%d" % (rank, world))
    q, r = divmod(int(n), world)
    b = rank * q + min(rank, r)
    return b, b + q + (1 if rank < r else 0)


TILE = 8  # ops.TILE: the march takes the rays of an image in 8 x 8 pixel tiles (one workgroup each)


def shard_tile_rows(H, rank, world, tile=TILE):
    """Pixel rows [r0, r1) of `rank`: whole bands of `tile` rows, as balanced as whole bands allow (band counts differ by at most
    one; the last band may be cut by the image border; trailing ranks are empty when there are fewer bands than ranks)."""
    bands = (int(H) + tile - 1) // tile
    b0, b1 = shard_range(bands, rank, world)
    return min(b0 * tile, int(H)), min(b1 * tile, int(H))


def band_ray_counts(mask, H, W, tile=TILE):
    """Cumulative ray count in front of every `tile`-row band of the H x W image: host int64 [bands + 1] (ONE device -> host
    read-back when the mask lives on the device).  mask [H*W] bool / uint8: which pixels carry a ray."""
    H, W = int(H), int(W)
    m = (torch.as_tensor(mask).reshape(H, W) != 0).sum(1).to(torch.int64)  # rays per pixel row
    bands = (H + tile - 1) // tile
    if H % tile:
        m = torch.cat([m, torch.zeros(bands * tile - H, dtype=torch.int64, device=m.device)])
    per_band = m.view(bands, tile).sum(1).cpu()
    return torch.cat([torch.zeros(1, dtype=torch.int64), per_band.cumsum(0)])


def balanced_band_cuts(cum, world):
    """Band indices c_0 = 0 <= c_1 <= ... <= c_world = bands that cut the image into `world` runs of whole bands holding as equal
    a number of RAYS as whole bands allow: c_r = the band border whose cumulative count is nearest to r / world of the total
    (ties: the earlier border).  A fully covered image gets equal band counts; a partially covered view (mask_at_box: every
    real test view) gets narrow shares where the subject is and wide ones over the empty top / bottom rows."""
    import bisect

    cum = [int(v) for v in cum]
    bands, total = len(cum) - 1, cum[-1]
    scaled = [c * world for c in cum]  # exact integer comparison of cum[k] / total against r / world
    cuts = [0]
    for r in range(1, world):
        t = total * r
        i = bisect.bisect_left(scaled, t)  # first border with at least r / world of the rays in front of it
        k = i if i <= bands and (i == 0 or scaled[i] - t < t - scaled[i - 1]) else i - 1
        cuts.append(max(min(k, bands), cuts[-1]))
    return cuts + [bands]


def shard_ranges_tiled(n, world, H, W, mask=None, tile=TILE):
    """The ray range [begin, end) of EVERY rank such that each range is exactly the rays of whole `tile`-row bands of the
    H x W image: every 8 x 8 pixel tile of the march then belongs to ONE rank with all of its rays, so the rank marches the same
    workgroups over the same voxel lists as a single GPU rendering the whole image would — its share of the image is
    bit-identical to that render (a range that cuts tiles agrees to rounding only: DESIGN.md §3).  `mask` [H*W] (bool / uint8,
    host or device; None = every pixel has a ray, n == H * W): which pixels carry a ray.  With a mask the band borders are
    chosen so that the ranks hold equal numbers of rays (`balanced_band_cuts`), from ONE read-back of the per-band counts."""
    if mask is None:
        if int(n) != int(H) * int(W):
            raise ValueError("%d rays for a %d x %d image: pass the pixel mask of a partially covered view" % (n, H, W))
        rows = [shard_tile_rows(H, r, world, tile) for r in range(world)]
        return [(r0 * int(W), r1 * int(W)) for r0, r1 in rows]
    cum = band_ray_counts(mask, H, W, tile)
    if int(cum[-1]) != int(n):
        raise ValueError("the mask holds %d rays, the batch %d" % (int(cum[-1]), n))
    cuts = balanced_band_cuts(cum, world)
    return [(int(cum[cuts[r]]), int(cum[cuts[r + 1]])) for r in range(world)]


def shard_range_tiled(n, rank, world, H, W, mask=None, tile=TILE):
    """One rank's entry of `shard_ranges_tiled` (callers that need every rank's range take the list: one read-back)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world %d/%d" % (rank, world))
    return shard_ranges_tiled(n, world, H, W, mask, tile)[rank]


def all_gather_tiles(tile, group=None, sizes=None):
    """All-gather per-rank tiles [n_r, ...] into one [sum n_r, ...] tensor on every rank.  Equal-size
    tiles go through a single all_gather_into_tensor; ragged ones are padded to the largest tile
    (`sizes` = per-rank row counts, computed with shard_range on the host — no size exchange)."""
    if not (dist.is_available() and dist.is_initialized()):
        return tile
    world = dist.get_world_size(group)
    tile = tile.contiguous()  # a group of one still goes through the collective: the single-GPU tests exercise RCCL
    if sizes is None or len(set(sizes)) == 1:
        out = torch.empty((world * tile.shape[0],) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
        dist.all_gather_into_tensor(out, tile, group=group)
        return out
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
    pad[:tile.shape[0]] = tile
    out = torch.empty((world * m,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * m:r * m + sizes[r]] for r in range(world)], 0)


def render_sharded(renderer, batch, group=None, keys=("rgb_map",), prefetched=None):
    """Render one batch with its rays split across the ranks of `group`; every rank encodes the (cheap,
    deterministic) feature volume locally (or takes it from `prefetched`, the ticket of renderer.prefetch(batch)), marches
    its ray range and receives the full maps."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = batch["ray_o"].shape[1]
    ranges = shard_ranges(renderer, batch, world)
    b, e = ranges[rank]
    extra = {} if prefetched is None else {"prefetched": prefetched}  # renderers with the reference's signature stay usable
    part = renderer.render(batch, ray_range=(b, e), **extra)
    if not dist.is_initialized():
        return {k: part[k] for k in keys}
    sizes = [r[1] - r[0] for r in ranges]
    return {k: all_gather_tiles(part[k][0], group, sizes)[None] for k in keys}


def shard_ranges(renderer, batch, world):
    """The ray range of every rank: whole 8-row tile bands when the renderer knows the image geometry (cfg.H, cfg.W: the march
    then works in 8 x 8 pixel tiles and a rank's share is bit-identical to the single-GPU render; a partially covered view's
    bands are dealt so that the ranks hold equal numbers of rays), plain balanced ranges otherwise (ray lists without an image,
    e.g. training batches)."""
    n = batch["ray_o"].shape[1]
    cfg = getattr(renderer, "cfg", None)
    H, W = (getattr(cfg, "H", None), getattr(cfg, "W", None)) if cfg is not None else (None, None)
    mask = batch.get("mask_at_box")
    if H and W and mask is not None and mask.numel() == int(H) * int(W) and n >= 1:
        if n == int(H) * int(W):
            return shard_ranges_tiled(n, world, H, W, None)
        # one read-back of the per-band ray counts per mask: kept on the renderer with the mask tensor itself (identity +
        # version + the caller's frame token, like Renderer._tile_order: the entry holds the tensor, its address cannot be recycled)
        key = (mask._version, int(world), int(H), int(W), batch.get("frame_token"))
        c = getattr(renderer, "_shard_cache", None)
        if c is not None and c[0] is mask and c[1] == key:
            return list(c[2])
        ranges = shard_ranges_tiled(n, world, H, W, mask.reshape(-1))
        try:
            renderer._shard_cache = (mask, key, ranges)
        except AttributeError:  # a renderer that takes no attributes
            pass
        return list(ranges)
    return [shard_range(n, r, world) for r in range(world)]


def reduce_timings(elapsed_s, local, precision_code, group=None, device=None):
    """The cross-rank bookkeeping of bench.py: the MAX over the ranks of the elapsed wall time (the whole-job time of the bench
    contract) by one all_reduce, and every rank's `local` floats (march ms, all-gather ms, median step ms ...) in rank order
    by one all_gather.  Raises if the ranks did not all run the same arithmetic (`precision_code`: its number in
    _lib.PRECISIONS) — an image stitched from tiles of different arithmetic would still be inside the tolerance, but it is not
    what a single GPU renders.  Without a process group: (elapsed_s, [local])."""
    local = [float(v) for v in local]
    if not (dist.is_available() and dist.is_initialized()):
        return float(elapsed_s), [local]
    world = dist.get_world_size(group)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    mine = torch.tensor(local + [float(precision_code)], dtype=torch.float64, device=device)
    allr = torch.empty(world * mine.numel(), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(allr, mine, group=group)
    allr = allr.view(world, mine.numel()).cpu()
    codes = sorted(set(int(c) for c in allr[:, -1].tolist()))
    if len(codes) != 1:
        raise RuntimeError("the ranks of this job ran different decoder arithmetics: precision codes %s" % codes)
    return float(t.item()), [[float(v) for v in row[:-1]] for row in allr.tolist()]
