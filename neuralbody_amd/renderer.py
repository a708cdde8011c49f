"""`Renderer` — drop-in for zju3dv/neuralbody lib/networks/renderer/if_clight_renderer.py::Renderer.

`render(batch)` returns the same dict (rgb_map, disp_map, acc_map, weights, depth_map; if_clight_
renderer.py:84-90) but runs ONE fused HIP launch over all rays (nb_march) instead of the reference's
Python loop over 2048-ray chunks (:107-118) that materialises every intermediate in HBM.

The overridable pieces the reference's subclasses rely on are kept with the same signatures
(if_clight_renderer_mmsk.py:8, if_mesh_renderer.py:11): get_sampling_points, prepare_sp_input,
get_density_color, get_pixel_value.  get_pixel_value (points decoded through the Network API, then
nb_composite) is what a subclass that culls samples would call; render() itself uses the fused path.
"""
import torch

from . import ops


class RenderConfig:
    """The cfg keys the hot path reads."""

    def __init__(self, N_samples=64, perturb=0.0, raw_noise_std=0.0, white_bkgd=False, H=None, W=None, mesh_th=50.0):
        # H, W (optional): image size of the view the rays come from (cfg.H * cfg.ratio in the reference); when
        # batch['mask_at_box'] covers H*W pixels, rays are marched in 8x8 pixel tiles
        self.H, self.W = H, W
        self.N_samples = int(N_samples)
        self.perturb = float(perturb)
        self.raw_noise_std = float(raw_noise_std)
        self.white_bkgd = bool(white_bkgd)
        self.mesh_th = float(mesh_th)  # lib/config/config.py:45; the shipped configs override it to 5


class SpInput(dict):
    """The sp_input dict of prepare_sp_input (if_clight_renderer.py:29-52) whose 'coord' entry — [B * n, 4] = (batch index, d, h, w),
    a concatenation launch per frame — is built when somebody INDEXES it (`sp_input['coord']`, as the reference's
    encode_sparse_voxels does, latent_xyzc.py:31): this package's encoder takes the [n, 3] coordinates themselves (`_coord_dhw`).
    `in`, `get`, `keys` see the entry only once it exists, like any dict with a `__missing__`."""

    def __missing__(self, key):
        if key == "coord" and "_coord_parts" in self:
            zc, coord = dict.__getitem__(self, "_coord_parts")
            val = torch.cat([zc, coord], dim=1)
            self[key] = val
            return val
        raise KeyError(key)


class Renderer:
    def __init__(self, net, cfg=None):
        self.net = net
        self.cfg = cfg if cfg is not None else RenderConfig()

    # -- if_clight_renderer.py:11-27 (host-visible sampling; the fused path does this in-kernel)
    def get_sampling_points(self, ray_o, ray_d, near, far, t_rand=None):
        t_vals = torch.linspace(0.0, 1.0, steps=self.cfg.N_samples).to(near)
        z_vals = near[..., None] * (1.0 - t_vals) + far[..., None] * t_vals
        if self.cfg.perturb > 0.0 and self.net.training:
            mids = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
            upper = torch.cat([mids, z_vals[..., -1:]], -1)
            lower = torch.cat([z_vals[..., :1], mids], -1)
            if t_rand is None:
                t_rand = torch.rand(z_vals.shape, device=z_vals.device)
            z_vals = lower + (upper - lower) * t_rand.to(upper)
        pts = ray_o[:, :, None] + ray_d[:, :, None] * z_vals[..., None]
        return pts, z_vals

    # -- if_clight_renderer.py:29-52
    def prepare_sp_input(self, batch):
        sp_input = SpInput()
        sh = batch["coord"].shape
        # built on the coordinates' device: the reference makes these on the host and copies them over, and a copy from
        # pageable host memory makes the launch thread wait for everything already enqueued (the previous view's march) —
        # the encoder's ~90 launches then cannot be enqueued under it (tools/experiments/cpu_ahead.py: 18.2 ms of host time
        # per render() instead of 1.5)
        coord = batch["coord"].view(-1, sh[-1])
        if sh[0] == 1:
            # batch size 1 (every shipped config): the batch-index column is a constant, kept per (length, dtype, device); the
            # encoder takes the [n, 3] coordinates themselves (`_coord_dhw`) instead of cutting the column off again
            zc = getattr(self, "_zero_col", None)
            if zc is None or zc.shape[0] != sh[1] or zc.dtype != coord.dtype or zc.device != coord.device:
                zc = self._zero_col = torch.zeros((sh[1], 1), dtype=coord.dtype, device=coord.device)
            sp_input["_coord_parts"] = (zc, coord)  # 'coord' itself: SpInput.__missing__, on first use
            sp_input["_coord_dhw"] = coord
        else:
            idx = [torch.full([sh[1]], i, dtype=coord.dtype, device=coord.device) for i in range(sh[0])]
            sp_input["coord"] = torch.cat([torch.cat(idx)[:, None], coord], dim=1)
        sp_input["out_sh"] = self._host_out_sh(batch["out_sh"])
        sp_input["batch_size"] = sh[0]
        sp_input["bounds"] = batch["bounds"]
        sp_input["R"] = batch["R"]
        sp_input["Th"] = batch["Th"]
        sp_input["latent_index"] = batch["latent_index"]
        return sp_input

    def _host_out_sh(self, t):
        """max over the batch of out_sh as a host list (if_clight_renderer.py:40-41: one device -> host sync per render() when
        the tensor lives on the device).  The last answer is kept together with the tensor OBJECT it was read from (identity
        and version, the entry holds the tensor so its address cannot be recycled): a caller that renders many views of one
        frame from the same batch tensors does not stall the launch queue on every view; a DataLoader loop, which makes fresh
        tensors per frame, syncs once per frame like the reference."""
        c = getattr(self, "_out_sh_cache", None)
        token = getattr(self, "_frame_token", None)  # set by render(): a caller that rewrites out_sh in place through a raw pointer
        if c is not None and c[0] is t and c[1] == (t._version, token):
            return list(c[2])
        out_sh, _ = torch.max(t, dim=0)
        val = out_sh.tolist()
        # the host has just waited for the device: the moment to look at the previous frame's fold-plane saturation counter
        # (precision 'auto'; Network._auto_checks_planes parks it instead of draining the queue once per frame)
        chk = getattr(self.net, "check_pending_saturation", None)
        if chk is not None and t.is_cuda:
            chk()
        self._out_sh_cache = (t, (t._version, token), val)
        return list(val)

    # -- if_clight_renderer.py:54-60
    def get_density_color(self, wpts, viewdir, raw_decoder):
        n_batch, n_pixel, n_sample = wpts.shape[:3]
        wpts = wpts.view(n_batch, n_pixel * n_sample, -1)
        viewdir = viewdir[:, :, None].repeat(1, 1, n_sample, 1).contiguous()
        viewdir = viewdir.view(n_batch, n_pixel * n_sample, -1)
        return raw_decoder(wpts, viewdir)

    # -- if_clight_renderer.py:62-92 through the Network API + nb_composite (unfused, for subclasses)
    def get_pixel_value(self, ray_o, ray_d, near, far, feature_volume, sp_input, batch, raw_noise=None):
        """raw_noise: standard-normal tensor [B, n_pixel, N_samples] used instead of torch.randn when cfg.raw_noise_std > 0
        (nerf_net_utils.py:31-35), the same way `t_rand` replaces the stratified jitter's torch.rand."""
        wpts, z_vals = self.get_sampling_points(ray_o, ray_d, near, far)
        viewdir = ray_d / torch.norm(ray_d, dim=2, keepdim=True)
        raw_decoder = lambda x_point, viewdir_val: self.net.calculate_density_color(  # noqa: E731
            x_point, viewdir_val, feature_volume, sp_input)
        wpts_raw = self.get_density_color(wpts, viewdir, raw_decoder)
        n_batch, n_pixel, n_sample = wpts.shape[:3]
        fix = getattr(self.net, "fix_last_densities", None)
        if fix is not None and float(self.cfg.raw_noise_std) == 0.0:
            # the densities the 1e10 last interval makes sign-critical, re-decoded at fp32 (the fused march does this in nb_march)
            wpts_raw = fix(wpts_raw.reshape(n_batch, n_pixel, n_sample, 4), wpts, feature_volume, sp_input)
        raw = self._add_raw_noise(wpts_raw.reshape(-1, n_sample, 4), raw_noise).contiguous()
        rgb, disp, acc, weights, depth = ops.composite(raw, z_vals.reshape(-1, n_sample).contiguous(),
                                                       ray_d.reshape(-1, 3).contiguous(), self.cfg.white_bkgd)
        return {"rgb_map": rgb.view(n_batch, n_pixel, -1), "disp_map": disp.view(n_batch, n_pixel),
                "acc_map": acc.view(n_batch, n_pixel), "weights": weights.view(n_batch, n_pixel, -1),
                "depth_map": depth.view(n_batch, n_pixel)}

    def _add_raw_noise(self, raw, raw_noise):
        """raw [n, S, 4] -> the same with sigma + randn * raw_noise_std (nerf_net_utils.py:31-35); identity when the std is 0."""
        std = float(self.cfg.raw_noise_std)
        if std == 0.0:
            return raw
        noise = torch.randn(raw.shape[:2], device=raw.device) if raw_noise is None else raw_noise.reshape(raw.shape[:2]).to(raw)
        raw = raw.clone()
        raw[..., 3] += noise * std
        return raw

    # -- if_clight_renderer.py:94-122
    def render(self, batch, t_rand=None, want_raw=False, ray_range=None, feature_volume=None, raw_noise=None, prefetched=None):
        """ray_range = (begin, end) renders a contiguous slice of the rays (multi-GPU sharding).
        prefetched: ticket of `prefetch(batch)` — this frame's encoder pass, enqueued earlier on a second stream.
        feature_volume: volumes of a previous `net.encode_sparse_voxels` of the SAME frame (same coord, out_sh, weights):
        the encoder is then skipped — novel-view loops render many views of one frame (NovelViewRenderer.reuse_volumes)."""
        ray_o, ray_d, near, far = batch["ray_o"], batch["ray_d"], batch["near"], batch["far"]
        n_batch, n_pixel = ray_o.shape[:2]
        if n_batch != 1:
            return self._render_frames(batch, n_batch, t_rand, want_raw, ray_range, feature_volume, raw_noise)
        if torch.is_grad_enabled() and ray_range is None and any(p.requires_grad for p in self.net.parameters()):
            # training step (lib/train/trainers/if_nerf_clight.py:18-36): differentiable HIP path
            from . import training

            if feature_volume is not None or want_raw or self.make_cull(batch) is not None:
                raise NotImplementedError("the differentiable path renders all samples of the batch from its own encoder pass: "
                                          "feature_volume / want_raw / sample culling are inference-only (wrap the call in "
                                          "torch.no_grad(), as run.py:66,98 does)")
            if self.cfg.perturb > 0.0 and self.net.training and t_rand is None:
                t_rand = torch.rand((n_batch, n_pixel, self.cfg.N_samples), device=ray_o.device)
            self._queue_behind_prefetch(ray_o.device)
            ret = training.render_train(self, batch, t_rand, raw_noise)
            self._mark_inline_encode(ray_o.device)
            return ret
        self._frame_token = batch.get("frame_token")
        ahead = self._take_prefetched(batch, prefetched) if feature_volume is None else None
        if ahead is not None:
            sp_input, feature_volume = ahead
        else:
            sp_input = self.prepare_sp_input(batch)
            if feature_volume is None:
                self._queue_behind_prefetch(ray_o.device)
                feature_volume = self.net.encode_sparse_voxels(sp_input)
                self._mark_inline_encode(ray_o.device)
        b, e = (0, n_pixel) if ray_range is None else ray_range
        if self.cfg.perturb > 0.0 and self.net.training:
            if t_rand is None:
                t_rand = torch.rand((n_batch, n_pixel, self.cfg.N_samples), device=ray_o.device)
            tr = t_rand[0, b:e].float().contiguous()
        else:
            tr = None
        ray_order = self._tile_order(batch, n_pixel, b, e)
        # a fully covered image's slot list names every ray of the range by construction; a mask-derived list names as many rays
        # as the mask holds pixels, which the batch may contradict (then the surplus rays read 0: ops.march zero-fills)
        covers = ray_order is not None and n_pixel == int(getattr(self.cfg, "H", 0) or 0) * int(getattr(self.cfg, "W", 0) or 0)
        cull = self.make_cull(batch)
        noisy = self.cfg.raw_noise_std != 0.0
        ret = self.net.render_rays(ray_o[0, b:e].contiguous(), ray_d[0, b:e].contiguous(), near[0, b:e].contiguous(),
                                   far[0, b:e].contiguous(), feature_volume, sp_input, self.cfg.N_samples, t_rand=tr,
                                   white_bkgd=self.cfg.white_bkgd, want_raw=want_raw or noisy, ray_order=ray_order, cull=cull,
                                   order_covers_all=covers)
        self.last_ill = getattr(self.net, "last_ill", None)  # header of the march's last-sample fix-up (diagnostics)
        if noisy:
            # raw_noise_std > 0 (nerf_net_utils.py:31-35; no shipped config): the march delivers `raw`, the noise is added to the
            # densities and the rays are composited again by nb_composite with the z values of the same sampling
            _, z_vals = self.get_sampling_points(ray_o[:, b:e], ray_d[:, b:e], near[:, b:e], far[:, b:e],
                                                 t_rand=None if tr is None else tr[None])
            raw = self._add_raw_noise(ret["raw"], None if raw_noise is None else raw_noise[0, b:e]).contiguous()
            rgb, disp, acc, weights, depth = ops.composite(raw, z_vals[0].contiguous(), ray_d[0, b:e].contiguous(), self.cfg.white_bkgd)
            ret = {"rgb_map": rgb, "disp_map": disp, "acc_map": acc, "weights": weights, "depth_map": depth,
                   **({"raw": raw} if want_raw else {})}
        return {k: v[None] for k, v in ret.items()}

    def _render_frames(self, batch, n_batch, t_rand, want_raw, ray_range, feature_volume, raw_noise):
        """A batch of B > 1 frames (lib/config/config.py:81 defaults train.batch_size to 4; every shipped YAML sets 1): B renders of
        one frame each — differentiable or not — stacked along the batch dimension.  The reference itself cannot run this case (its
        Network pairs ONE set of 6890 vertex codes with the B * 6890 coordinates of the batch, latent_xyzc.py:35-36, which spconv
        indexes out of bounds), so the semantics are its own B = 1 results frame by frame, with the batch's common out_sh
        (if_clight_renderer.py:40-41: the maximum over the batch) — BatchNorm statistics are per frame."""
        from .network import frame_volumes

        out_sh = torch.max(batch["out_sh"], dim=0, keepdim=True)[0]
        outs = []
        for b in range(n_batch):
            sub = {k: (v[b:b + 1] if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == n_batch else v) for k, v in batch.items()}
            sub["out_sh"] = out_sh
            fv = None if feature_volume is None else frame_volumes(feature_volume, b)
            outs.append(self.render(sub, t_rand=None if t_rand is None else t_rand[b:b + 1], want_raw=want_raw, ray_range=ray_range,
                                    feature_volume=fv, raw_noise=None if raw_noise is None else raw_noise[b:b + 1]))
        return {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}

    # -- the encoder of the NEXT frame under the march of this one (no counterpart in the reference, whose render() is one
    # synchronous pass: if_clight_renderer.py:94-122)
    _FRAME_KEYS = ("coord", "out_sh", "bounds", "R", "Th", "latent_index")

    def fence(self, device=None):
        """An event on the current stream, for `prefetch(batch, after=fence)`: everything enqueued so far (the previous march,
        the copies that produced the next frame's tensors), and nothing enqueued later."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        return ev

    def prefetch(self, batch, after=None):
        """Enqueue the encoder of `batch`'s frame (prepare_sp_input, encode_sparse_voxels and, for the folded arithmetic, the
        fold planes) on a SECOND HIP stream and return a ticket for `render(batch, prefetched=ticket)`.  The second stream first
        waits for `after` (an event of `fence()` taken BEFORE the render() of the frame in front was enqueued) or, without one,
        for everything enqueued on the current stream so far — so either call prefetch before that render(), or take the fence
        before it and call prefetch behind it (the launch thread then hands the march to the device first, its ~60 encoder
        launches afterwards: what a loop that starts on an idle device wants).  What it waits for must include the previous
        march: the volumes that march read are recycled by this pass.  Its ~60 small launches (1.2 ms on an empty chip, a quarter of the
        CUs busy) run beside the march that render() enqueues next — mostly in the slots the march's last workgroups leave free
        (tools/experiments/overlap_check.py: 13.26 -> 12.72 ms per 512 x 512 view).  render() marches the ticket's volumes when its
        batch carries the SAME frame tensors (identity and version counters of coord, out_sh, bounds, R, Th, latent_index; the
        rays may differ) and encodes in place otherwise.  Inference only (the differentiable path runs its own encoder pass);
        only the frame keys of `batch` are read."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.net.parameters()):
            raise RuntimeError("Renderer.prefetch is inference-only: call it under torch.no_grad() (the training step encodes inside autograd)")
        dev = batch["coord"].device
        if dev.type != "cuda":
            raise RuntimeError("Renderer.prefetch needs device tensors")
        if batch["coord"].dim() == 3 and batch["coord"].shape[0] != 1:
            raise NotImplementedError("Renderer.prefetch takes ONE frame (a batch of B > 1 frames is rendered frame by frame: prefetch "
                                      "each frame's batch, or call render() without a ticket)")
        side = getattr(self, "_side_stream", None)
        if side is None or side.device != dev:
            side = self._side_stream = torch.cuda.Stream(device=dev)
        if after is None:
            side.wait_stream(torch.cuda.current_stream(dev))
        else:
            side.wait_event(after)
        pre = getattr(self, "_pre_march", None)
        if pre is not None:
            side.wait_event(pre)  # whatever `after` is: never in front of the march before the one being enqueued / just enqueued
        inl = getattr(self, "_inline_enc", None)
        if inl is not None:
            # ... and never beside an encoder pass that a render() ran inline on the main stream (first view of a loop, a ticket
            # that did not match, a training step): both would update the BatchNorm running statistics in place
            side.wait_event(inl)
        self._frame_token = batch.get("frame_token")
        with torch.cuda.stream(side):
            sp_input = self.prepare_sp_input(batch)
            replayed = self._replay_encoder_graph(batch, sp_input, side) if getattr(self, "use_encoder_graph", False) else None
            if replayed is not None:
                sp_input, feature_volume = replayed
            else:
                feature_volume = self._encode_for_ticket(sp_input)
            ready = torch.cuda.Event()
            ready.record(side)
        return (self._frame_key(batch), sp_input, feature_volume, ready)

    def _encode_for_ticket(self, sp_input):
        feature_volume = self.net.encode_sparse_voxels(sp_input)
        if self.net.march_precision() == "f16f6":
            self.net.make_scene(feature_volume, sp_input, "f16f6")  # leaves the fold planes on the FeatureVolumes object
        return feature_volume

    # -- the prefetched pass as a HIP GRAPH (`use_encoder_graph = True`).  The pass is ~48 small launches; enqueued one by one they
    # cost the launch thread ~20 us each (~1 ms per frame), which a 12 ms march hides — but a rank's share of a view split over 8
    # GPUs marches in 1.6 ms, and then the launch thread, not the device, sets the step.  Captured once per (frame tensors, weights,
    # BatchNorm mode) the pass is ONE launch.  A graph owns its memory: every replay rewrites the same index grids, rows and fold
    # planes, so TWO graphs alternate — the planes the march in front is still reading belong to the other one, and prefetch()
    # already orders a pass behind the march before the one being enqueued (`_pre_march`), i.e. behind the last reader of the pool
    # it rewrites.  Keyed on the frame tensors' identity and versions (a graph replays addresses: another frame needs another
    # capture; loops over the views of ONE frame — turntables, the strong split — reuse it) and on the parameters' versions.
    ENCODER_GRAPH_SLOTS = 2

    def _encoder_graph_key(self, batch):
        net = self.net
        return (self._frame_key(batch), bool(net.training), net.march_precision(),
                tuple((p.data_ptr(), p._version) for p in net.parameters()))

    def _replay_encoder_graph(self, batch, sp_input, side):
        """(sp_input, feature_volume) of the frame from a captured pass, or None (the caller then enqueues the pass launch by launch:
        the first pass of a frame warms the caches — packed weights, the out_sh read-back, 'auto''s one-time checks — that must not
        sit inside a capture; the next two are the captures themselves, each behind a device synchronisation)."""
        st = getattr(self, "_enc_graphs", None)
        key = self._encoder_graph_key(batch)
        if st is None or not self._same_graph_key(st["key"], key):
            st = self._enc_graphs = {"key": key, "seen": 0, "slots": [], "turn": 0}
        st["seen"] += 1
        if st["seen"] <= 1:
            return None
        if len(st["slots"]) < self.ENCODER_GRAPH_SLOTS:
            g = torch.cuda.CUDAGraph()
            sp = type(sp_input)(sp_input)
            pend = getattr(self.net, "_sat_pending", None)
            with torch.cuda.graph(g, stream=side):
                fv = self._encode_for_ticket(sp)
            self.net._sat_pending = pend  # (a counter parked during capture belongs to no pass)
            st["slots"].append((g, sp, fv))
            g.replay()  # capture records, it does not run
            return sp, fv
        g, sp, fv = st["slots"][st["turn"]]
        st["turn"] = (st["turn"] + 1) % len(st["slots"])
        g.replay()
        return sp, fv

    @staticmethod
    def _same_graph_key(a, b):
        fa, fb = a[0], b[0]
        return (len(fa) == len(fb) and all((x[0] is y[0] and x[1] == y[1]) if isinstance(x, tuple) else x == y for x, y in zip(fa, fb))
                and a[1:] == b[1:])

    def _frame_key(self, batch):
        return tuple((batch[k], batch[k]._version) for k in self._FRAME_KEYS) + (batch.get("frame_token"),)

    def _queue_behind_prefetch(self, device):
        """An encoder pass on the current stream: a prefetched pass may be in flight on the second stream, and two passes at
        once would race on the BatchNorm running statistics and counters — this one queues behind it."""
        side = getattr(self, "_side_stream", None)
        if side is not None and device.type == "cuda":
            torch.cuda.current_stream(device).wait_stream(side)

    def _mark_inline_encode(self, device):
        """An encoder pass was just enqueued on the current stream: later prefetches queue behind it (the other direction of
        `_queue_behind_prefetch`)."""
        if device.type == "cuda":
            self._inline_enc = torch.cuda.Event()
            self._inline_enc.record(torch.cuda.current_stream(device))

    def _release_held(self, device):
        """A render() without a ticket: the volumes of the last ticket are let go (their march is in front of this point of the
        current stream, which is what later prefetches wait for)."""
        if getattr(self, "_held", None) is not None:
            self._held, self._pre_march = None, torch.cuda.Event()
            self._pre_march.record(torch.cuda.current_stream(device))

    def _ticket_is_for(self, ticket, batch):
        """The ticket's frame tensors are the batch's: the same objects at the same version counters, the same frame token."""
        key = self._frame_key(batch)
        return all(a[0] is b[0] and a[1] == b[1] for a, b in zip(ticket[0][:-1], key[:-1])) and ticket[0][-1] == key[-1]

    def _take_prefetched(self, batch, ticket):
        if ticket is None or any(k not in batch for k in self._FRAME_KEYS):
            self._release_held(batch["ray_o"].device)
            return None
        if not self._ticket_is_for(ticket, batch):
            self._release_held(batch["ray_o"].device)
            return None  # another frame, or this frame's tensors rewritten in place since: the volumes are not this batch's
        main = torch.cuda.current_stream(batch["coord"].device)
        # The volumes live in the second stream's memory pool.  They stay referenced here until the NEXT ticket is taken, and an
        # event in front of this march is what every later prefetch waits for at least: the pass that recycles them is then
        # ordered behind the march that reads them, whenever the caller drops its ticket
        self._held, self._pre_march = ticket, torch.cuda.Event()
        self._pre_march.record(main)
        main.wait_event(ticket[3])
        return ticket[1], ticket[2]

    def make_cull(self, batch):
        """Sample-culling description for nb_march (None: the base renderer decodes every sample)."""
        return None

    def _tile_order(self, batch, n_pixel, b, e):
        """The slot list that marches the rays [b, e) in 8 x 8 pixel tiles (ops.tile_slots; None when the image geometry is
        unknown: the march then takes the rays in list order, 64 consecutive rays per workgroup).  Results do not depend on it
        beyond rounding: it decides which rays share a workgroup, i.e. a voxel list."""
        H, W = getattr(self.cfg, "H", None), getattr(self.cfg, "W", None)
        mask = batch.get("mask_at_box")
        if not H or not W or mask is None or mask.numel() != int(H) * int(W) or e <= b:
            return None
        H, W, dev = int(H), int(W), mask.device
        geo = getattr(self, "_slot_pixels", None)
        if geo is None or geo[0] != (H, W, str(dev)):
            geo = self._slot_pixels = ((H, W, str(dev)), ops.tile_pixels(H, W, dev))
        if n_pixel == H * W:
            # every pixel of the image is a ray: the list depends on the image geometry and the ray range only
            full = getattr(self, "_order_full", None)
            if full is None:
                full = self._order_full = {}
            key = (H, W, b, e, str(dev))
            if key not in full:
                while len(full) >= 8:  # a handful of geometries / ray ranges per process; callers that vary them must not grow it
                    full.pop(next(iter(full)))
                full[key] = ops.tile_slots(torch.ones(H * W, dtype=torch.bool, device=dev), geo[1], b, e)
            else:
                full[key] = full.pop(key)  # most recently used last
            return full[key]
        # cached on the mask tensor's IDENTITY: the entry holds the tensor itself, so its address cannot be recycled for
        # another frame's mask while the entry lives.  A caller that rewrites the same tensor in place through a raw pointer
        # (nb_raygen) bumps nothing, so the frame token `batch.get("frame_token")` is part of the key when given.
        cached = getattr(self, "_order_cache", None)
        token = batch.get("frame_token")
        if cached is not None and cached[0] is mask and cached[1] == (mask._version, W, b, e, token):
            return cached[2]
        order = ops.tile_slots(mask, geo[1], b, e)  # the ray count is whatever the mask holds: no read-back
        self._order_cache = (mask, (mask._version, W, b, e, token), order)
        return order


class RendererMmsk(Renderer):
    """lib/networks/renderer/if_clight_renderer_mmsk.py::Renderer — novel views of multi-view (ZJU) subjects: samples
    that project outside any of the dilated training-view masks are culled (batch keys msks [B,nv,H,W], Ks [B,nv,3,3],
    RT [B,nv,3,4], multi_view_demo_dataset.py:176).  The culling runs inside nb_march (nb_cull)."""

    def make_cull(self, batch):
        if "Ks" not in batch:
            raise KeyError("the _mmsk renderer needs batch['msks'], batch['Ks'], batch['RT']")
        return ops.make_cull(batch["msks"][0], batch["RT"][0], batch["Ks"][0])


class RendererMsk(Renderer):
    """lib/networks/renderer/if_clight_renderer_msk.py::Renderer — novel views of monocular (People-Snapshot) subjects:
    the sample is carried into the snapshot frame's pose (R0_snap, Th0_snap) and tested against that frame's mask."""

    def make_cull(self, batch):
        if "R0_snap" not in batch:
            raise KeyError("the _msk renderer needs batch['msk'], batch['K'], batch['RT'], batch['R0_snap'], batch['Th0_snap']")
        return ops.make_cull(batch["msk"][:1], batch["RT"][:1], batch["K"][:1], R0=batch["R0_snap"][0],
                             Th0=batch["Th0_snap"][0])


class RendererMesh(Renderer):
    """lib/networks/renderer/if_mesh_renderer.py::Renderer — density on the lattice of
    multi_view_mesh_dataset.py:142-158 for mesh extraction.  The hot part (encoder + `calculate_density` of every
    lattice point flagged `inside`) runs on HIP in ONE nb_decode_points launch (the reference chunks 131 072 points per
    call, :37); marching cubes is CPU post-processing of the reference's own dependencies (PyMCubes, trimesh)."""

    def batchify_rays(self, wpts, alpha_decoder, chunk=1024 * 32):
        """if_mesh_renderer.py:15-24; kept for subclasses — chunking is not needed for memory here."""
        return torch.cat([alpha_decoder(wpts[:, i:i + chunk]) for i in range(0, wpts.shape[1], chunk)], 1)

    def density_cube(self, batch, pad=10):
        """-> DEVICE float32 tensor [X+2*pad, Y+2*pad, Z+2*pad]: alpha at the inside lattice points, 0 elsewhere."""
        pts = batch["pts"]
        inside = batch["inside"][0].bool()
        wpts = pts[0][inside][None].contiguous()
        sp_input = self.prepare_sp_input(batch)
        feature_volume = self.net.encode_sparse_voxels(sp_input)
        alpha = self.net.calculate_density(wpts, feature_volume, sp_input)
        cube = torch.zeros(pts.shape[1:-1], dtype=torch.float32, device=pts.device)
        cube[inside] = alpha[0, :, 0]
        return torch.nn.functional.pad(cube, (pad,) * 6)

    def render(self, batch):
        import mcubes  # CPU marching cubes, as in the reference (if_mesh_renderer.py:6,46)
        import trimesh

        cube = self.density_cube(batch).double().cpu().numpy()
        vertices, triangles = mcubes.marching_cubes(cube, self.cfg.mesh_th)
        return {"cube": cube, "mesh": trimesh.Trimesh(vertices, triangles)}
