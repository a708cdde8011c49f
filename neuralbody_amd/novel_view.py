"""Novel-view driving: the camera path, the rotating-SMPL frames and the per-view loop that
the reference runs in DataLoader workers + a visualizer, with everything per-pixel on the device:

    host (numpy, fp64 like the reference)          device (HIP)
    gen_path / load_cam / rotate_smpl_frame  -->   nb_raygen -> Renderer.render -> nb_image_assemble  --> image [H,W,3]

Host functions restate lib/utils/render_utils.py:29-106 and lib/datasets/light_stage/monocular_demo_dataset.py:33-86
operation by operation (they are camera / pose algebra on a handful of matrices and 6890 vertices, microseconds of
numpy); the per-pixel and per-sample work never touches the host.  Views are independent: `render_views_sharded` deals
them round-robin to the ranks and gathers finished images with one all-gather per round of views."""
import math

import numpy as np
import torch

from . import ops
from .parallel import shard_range


# ------------------------------------------------------------------------------------------- host camera algebra
def _normalize(x):
    return x / np.linalg.norm(x)


def _viewmatrix(z, up, pos):
    """lib/utils/render_utils.py:15-21"""
    vec2 = _normalize(z)
    vec1 = _normalize(np.cross(vec2, up))
    vec0 = _normalize(np.cross(vec1, vec2))
    return np.stack([vec0, vec1, vec2, pos], 1)


def load_cam(cams, ratio):
    """lib/utils/render_utils.py:29-50 on an already-loaded `cams` dict (keys K, R, T; T in millimetres):
    -> (K list with the first two rows scaled by `ratio`, RT list of 4x4 world-to-camera matrices in metres)."""
    Ks, RTs = [], []
    lower = np.array([[0.0, 0.0, 0.0, 1.0]])
    for i in range(len(cams["K"])):
        K = np.array(cams["K"][i], dtype=np.float64)
        K[:2] = K[:2] * ratio
        r = np.array(cams["R"][i], dtype=np.float64)
        t = np.array(cams["T"][i], dtype=np.float64).reshape(3, 1) / 1000.0
        Ks.append(K)
        RTs.append(np.concatenate([np.concatenate([r, t], 1), lower], 0))
    return Ks, RTs


def gen_path(RT, num_render_views, center=None):
    """lib/utils/render_utils.py:61-106: spiral of `num_render_views` world-to-camera 4x4 matrices around the training
    cameras `RT` (list/array of 4x4 world-to-camera).  Unlike the reference this does not overwrite its argument."""
    lower = np.array([[0.0, 0.0, 0.0, 1.0]])
    c2w_all = np.linalg.inv(np.array(RT, dtype=np.float64))
    c2w_all = np.concatenate([c2w_all[:, :, 1:2], c2w_all[:, :, 0:1], -c2w_all[:, :, 2:3], c2w_all[:, :, 3:4]], 2)
    up = _normalize(c2w_all[:, :3, 0].sum(0))
    z = _normalize(c2w_all[0, :3, 2])
    vec1 = _normalize(np.cross(z, up))
    vec2 = _normalize(np.cross(up, vec1))
    z_off = 0
    if center is None:
        center = c2w_all[:, :3, 3].mean(0)
        z_off = 1.3
    c2w = np.stack([up, vec1, vec2, center], 1)
    # radii of the spiral: 80th percentile of the training cameras' offsets in the path frame (ptstocam, :24-26)
    pts = c2w_all[:, :3, 3]
    tt = np.matmul(c2w[:3, :3].T, (pts - c2w[:3, 3])[..., np.newaxis])[..., 0].T
    rads = np.percentile(np.abs(tt), 80, -1) * 1.3
    rads = np.array(list(rads) + [1.0])
    out = []
    for theta in np.linspace(0.0, 2 * np.pi, num_render_views + 1)[:-1]:
        cam_pos = np.array([0, np.sin(theta), np.cos(theta), 1] * rads)
        cam_pos_world = np.dot(c2w[:3, :4], cam_pos)
        zz = _normalize(cam_pos_world - np.dot(c2w[:3, :4], np.array([z_off, 0, 0, 1.0])))
        mat = _viewmatrix(zz, up, cam_pos_world)
        mat = np.concatenate([mat[:, 1:2], mat[:, 0:1], -mat[:, 2:3], mat[:, 3:4]], 1)
        out.append(np.linalg.inv(np.concatenate([mat, lower], 0)))
    return out


def rodrigues(rvec):
    """axis-angle [3] -> rotation matrix [3,3] (cv2.Rodrigues forward), fp64."""
    rvec = np.asarray(rvec, np.float64).reshape(3)
    th = float(np.linalg.norm(rvec))
    if th < 1e-12:
        return np.eye(3)
    k = rvec / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * Kx + (1 - math.cos(th)) * (Kx @ Kx)


def rotate_smpl_frame(xyz, Rh, Th, t, voxel_size=(0.005, 0.005, 0.005)):
    """lib/datasets/light_stage/monocular_demo_dataset.py:33-86 (People-Snapshot turntable: the BODY rotates by `t`
    about the vertical axis through its centroid, the camera stays): world vertices `xyz` [V,3] f32 of the frame, its
    SMPL global rotation `Rh` [3] and translation `Th` [3] -> dict(coord [V,3] i32 (dhw), out_sh [3] i32, can_bounds
    [2,3] f32 (world, rotated), bounds [2,3] f32 (SMPL space), R [3,3] f32, Th [3] f32) — the `sp_input` fields of the
    rotated frame.  (The reference converts R -> Rh -> R through cv2.Rodrigues, :62 and :125; that round trip is the
    identity up to fp32 rounding and is skipped.)"""
    xyz = np.asarray(xyz).astype(np.float32)
    rot_ = np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]])
    rot = np.eye(3)
    rot[[0, 0, 2, 2], [0, 2, 0, 2]] = rot_.ravel()
    center = np.mean(xyz, axis=0)
    xyz = (np.dot(xyz - center, rot.T) + center).astype(np.float32)

    def padded_bounds(p):
        lo, hi = np.min(p, axis=0), np.max(p, axis=0)
        lo[1] -= 0.1
        hi[1] += 0.1
        return lo, hi

    can_bounds = np.stack(padded_bounds(xyz), axis=0)
    R = np.dot(rot, rodrigues(Rh).astype(np.float32))
    Th = (np.sum(rot * (np.asarray(Th).astype(np.float32) - center), axis=1) + center).astype(np.float32)
    xyz = np.dot(xyz - Th, R).astype(np.float32)
    min_xyz, max_xyz = padded_bounds(xyz)
    bounds = np.stack([min_xyz, max_xyz], axis=0)
    dhw, min_dhw, max_dhw = xyz[:, [2, 1, 0]], min_xyz[[2, 1, 0]], max_xyz[[2, 1, 0]]
    vs = np.array(voxel_size)
    coord = np.round((dhw - min_dhw) / vs).astype(np.int32)
    out_sh = (np.ceil((max_dhw - min_dhw) / vs).astype(np.int32) | 31) + 1
    return {"coord": coord, "out_sh": out_sh, "can_bounds": can_bounds, "bounds": bounds, "R": R.astype(np.float32), "Th": Th}


# ------------------------------------------------------------------------------------------- device per-view loop
class NovelViewRenderer:
    """One finished image per call, nothing per-pixel on the host:
    nb_raygen (image_rays) -> renderer.render (any Renderer / RendererMmsk / RendererMsk) -> nb_image_assemble
    (if_nerf_demo.py:15-30).  `H, W` are the reduced image size int(cfg.H * cfg.ratio), int(cfg.W * cfg.ratio)."""

    def __init__(self, renderer, H, W, device="cuda:0", reuse_volumes=False):
        """reuse_volumes: encode a frame once and reuse its feature volumes for every further view of the same frame
        (same `coord` tensor, out_sh, latent-independent encoder weights).  The reference re-encodes per view
        (if_clight_renderer.py:99-100); the volumes are identical (the encoder is deterministic), only the BatchNorm
        running-statistics side effect of the skipped passes is lost — off by default."""
        self.renderer, self.H, self.W, self.device = renderer, int(H), int(W), torch.device(device)
        self.reuse_volumes = bool(reuse_volumes)
        # render_views: the next view's encoder is enqueued on a second HIP stream before this view's march (Renderer.prefetch)
        self.prefetch_encoder = self.device.type == "cuda" and hasattr(renderer, "prefetch")
        self._vol_key, self._vols = None, None

    def _frame_volumes(self, batch):
        if not self.reuse_volumes:
            return None
        net = self.renderer.net
        coord, out_sh = batch["coord"], batch["out_sh"]
        # The cache entry HOLDS the tensors it was computed from and compares identity + version counters: a new
        # frame's `coord` is a different tensor object even when the allocator hands it the old one's address
        # (round 1 keyed on data_ptr, which a freed-and-reallocated batch reproduces).
        params = list(net.xyzc_net.parameters()) + [net.c.weight]
        src = (coord, out_sh, net.training, params)
        ver = (coord._version, out_sh._version, tuple(p._version for p in params))
        old = self._vol_key
        same = old is not None and old[0][0] is coord and old[0][1] is out_sh and old[0][2] == net.training and \
            len(old[0][3]) == len(params) and all(a is b for a, b in zip(old[0][3], params)) and old[1] == ver
        if not same:
            self._vols = net.encode_sparse_voxels(self.renderer.prepare_sp_input(batch))
            self._vol_key = (src, ver)
        return self._vols

    def _launch_rays(self, K, RT, can_bounds):
        """nb_raygen of one view + an asynchronous 4-byte copy of its ray count into pinned host memory; the event marks the
        copy, so waiting for it does not wait for anything enqueued later (render_views enqueues it a whole view ahead)."""
        RT = np.asarray(RT, np.float64)
        rays = ops.raygen(self.H, self.W, K, RT[:3, :3], RT[:3, 3], can_bounds, self.device)
        slot = self._count_slot = (getattr(self, "_count_slot", -1) + 1) % 4
        if getattr(self, "_count_host", None) is None:
            pin = self.device.type == "cuda"
            self._count_host = [torch.empty(1, dtype=torch.int32, pin_memory=pin) for _ in range(4)]
        host = self._count_host[slot]
        host.copy_(rays[5], non_blocking=True)
        ev = None
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record()
        return rays, host, ev

    @staticmethod
    def _finish_batch(launched, frame):
        (ray_o, ray_d, near, far, mask, _), host, ev = launched
        if ev is not None:
            ev.synchronize()
        n = int(host[0])  # the march is launched over exactly n rays
        batch = dict(frame)
        batch.update(ray_o=ray_o[None, :n], ray_d=ray_d[None, :n], near=near[None, :n], far=far[None, :n],
                     mask_at_box=mask[None])
        return batch

    def view_batch(self, K, RT, can_bounds, frame):
        """image_rays on device + the frame's sp_input fields -> the batch dict Renderer.render consumes.
        `frame`: dict with device tensors coord [1,V,3] i32, out_sh [1,3] i32, bounds [1,2,3], R [1,3,3], Th [1,*,3],
        latent_index [1] (+ the mask-culling keys for the _mmsk/_msk renderers).  One 4-byte host read per view (the ray
        count); a single call waits for everything enqueued before it — loops over views should use render_views."""
        return self._finish_batch(self._launch_rays(K, RT, can_bounds), frame)

    def render_views(self, views, bgr=False, scale=1.0):
        """Generator over finished views with the ray generation running ONE VIEW AHEAD: view k + 1's nb_raygen and the copy
        of its ray count are enqueued before view k's encoder and march, so the host learns the count while the previous
        march is still running and never drains the queue (the per-call render_view waits for the previous march before it
        can enqueue the next view's ~110 encoder launches).  `views`: iterable of (K, RT, can_bounds, frame)."""
        it = iter(views)
        cur = next(it, None)
        launched = None if cur is None else self._launch_rays(cur[0], cur[1], cur[2])
        ticket = None
        while cur is not None:
            batch = self._finish_batch(launched, cur[3])
            nxt = next(it, None)
            cur_ticket, ticket = ticket, None
            ahead = nxt is not None and self.prefetch_encoder and not self.reuse_volumes
            if nxt is not None:
                launched = self._launch_rays(nxt[0], nxt[1], nxt[2])
            fence = self.renderer.fence(self.device) if ahead else None
            with torch.no_grad():  # the render runs without autograd; the consumer's loop body keeps ITS grad mode (yielding from
                out = self._render_batch(batch, bgr, scale, None, cur_ticket)  # inside the block would leak no_grad into it)
                if ahead:  # view k + 1's encoder on a second stream, beside view k's march (enqueued behind it, ordered before it)
                    ticket = self.renderer.prefetch(nxt[3], after=fence)
            yield out
            cur = nxt

    def render_view(self, K, RT, can_bounds, frame, bgr=False, scale=1.0, t_rand=None):
        """-> dict(img [H,W,3], depth [H,W], mask_at_box [H,W] uint8, n_rays) — device tensors."""
        with torch.no_grad():  # inference loop (run.py:66,98 wraps renderer.render the same way)
            return self._render_view(K, RT, can_bounds, frame, bgr, scale, t_rand)

    def _render_view(self, K, RT, can_bounds, frame, bgr, scale, t_rand):
        return self._render_batch(self.view_batch(K, RT, can_bounds, frame), bgr, scale, t_rand)

    def _render_batch(self, batch, bgr, scale, t_rand, prefetched=None):
        n = batch["ray_o"].shape[1]
        if n == 0:
            out = {"rgb_map": torch.zeros((1, 0, 3), device=self.device), "depth_map": torch.zeros((1, 0), device=self.device)}
        else:
            extra = {} if prefetched is None else {"prefetched": prefetched}
            out = self.renderer.render(batch, t_rand=t_rand, feature_volume=self._frame_volumes(batch), **extra)
        img, depth = ops.image_assemble(batch["mask_at_box"][0], out["rgb_map"][0].contiguous(), out["depth_map"][0].contiguous(),
                                        white_bkgd=self.renderer.cfg.white_bkgd, bgr=bgr, scale=scale)
        return {"img": img.view(self.H, self.W, 3), "depth": depth.view(self.H, self.W),
                "mask_at_box": batch["mask_at_box"][0].view(self.H, self.W), "n_rays": n}


def view_assignment(n_views, rank, world_size):
    """Views a rank renders: round-robin (view v -> rank v % world_size), so every round of `world_size` consecutive
    views finishes together and can be gathered with one collective."""
    return list(range(rank, n_views, world_size))


def render_views_sharded(render_view, n_views, H, W, device, gather=True):
    """Render `n_views` images with the ranks of the default process group: rank r renders views r, r+world, ...
    (`render_view(v)` -> device tensor [H,W,3]); after each round the finished images are exchanged with ONE
    all_gather_into_tensor (RCCL on GPUs, gloo in the CPU tests).  Returns the list of all images on every rank
    (gather=True) or only this rank's {view: image} (gather=False).  Without a process group: a plain loop."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        imgs = [render_view(v) for v in range(n_views)]
        return imgs if gather else dict(enumerate(imgs))
    rank, world = dist.get_rank(), dist.get_world_size()
    mine, out = {}, [None] * n_views
    for base in range(0, n_views, world):
        v = base + rank
        img = render_view(v) if v < n_views else torch.zeros((H, W, 3), dtype=torch.float32, device=device)
        if v < n_views:
            mine[v] = img
        if gather:
            buf = torch.empty((world * H, W, 3), dtype=torch.float32, device=device)  # concatenated along dim 0
            dist.all_gather_into_tensor(buf, img.contiguous())
            buf = buf.view(world, H, W, 3)
            for r in range(min(world, n_views - base)):
                out[base + r] = buf[r]
    return out if gather else mine
