"""The multi-GPU code paths executed for real on the one GPU of the test box: a RCCL ("nccl") process
group of world_size 1 runs parallel.render_sharded with the REAL Renderer, the tile all-gather, bench.py under
torch.distributed.run, and DistributedDataParallel(NetworkWrapper(net)) for one training step whose gradients must equal
the non-DDP step's (lib/train/trainers/trainer.py:13-18 wraps the same module the same way)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist

from tests import helpers as H
from tests.golden import scenes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nccl_group():
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    yield dist.group.WORLD
    dist.destroy_process_group()


def test_render_sharded_and_tile_gather_with_the_real_renderer(nccl_group):
    from neuralbody_amd import parallel

    r, sd, body, batch, cam, _ = scenes.build("small")
    net = H.make_network(sd, DEV, True)
    rend = H.make_renderer(net, r)
    bd = H.device_batch(batch, DEV)
    with torch.no_grad():
        full = rend.render(bd)
        got = parallel.render_sharded(rend, bd, nccl_group, keys=("rgb_map", "acc_map"))
        # the ragged path of the gather (sizes differ) through RCCL as well
        n = bd["ray_o"].shape[1]
        tiles = parallel.all_gather_tiles(full["rgb_map"][0], nccl_group, sizes=[n])
    torch.cuda.synchronize()
    assert H.same_bits(got["rgb_map"], full["rgb_map"]) and H.same_bits(got["acc_map"], full["acc_map"])
    assert H.same_bits(tiles, full["rgb_map"][0])
    # a ray range renders the same rays as the full render; bit for bit when the range keeps the 64-ray workgroups of the
    # full render together (the fc_0-folded march contracts over a workgroup's voxel list), to rounding otherwise
    with torch.no_grad():
        part = rend.render(bd, ray_range=(128, 384))
        odd = rend.render(bd, ray_range=(100, 357))
    assert H.same_bits(part["rgb_map"], full["rgb_map"][:, 128:384])
    assert H.same_result(odd["rgb_map"], full["rgb_map"][:, 100:357], H.DEFAULT_PRECISION)


def test_ddp_training_step_equals_the_plain_step(nccl_group):
    from torch.nn.parallel import DistributedDataParallel as DDP

    mod = H.load_plugin("if_nerf_clight.py")
    sd, bd = H.training_batch(seed=1, n_rand=256, size=48)
    t_rand = torch.rand((1, 256, 64), generator=torch.Generator().manual_seed(3)).to(DEV)

    def one_step(wrap):
        net = H.make_network(sd, DEV, True, "f32")
        wrapper = mod.NetworkWrapper(net)
        wrapper.renderer.render = (lambda orig: (lambda batch, **kw: orig(batch, t_rand=t_rand, **kw)))(wrapper.renderer.render)
        model = DDP(wrapper, device_ids=[0]) if wrap else wrapper
        opt = torch.optim.Adam(net.parameters(), lr=5e-4)
        ret, loss, stats, _ = model(bd)
        opt.zero_grad()
        loss.mean().backward()
        grads = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        torch.nn.utils.clip_grad_value_(net.parameters(), 40)
        opt.step()
        return float(loss.detach()), grads, {n: p.detach().clone() for n, p in net.named_parameters()}

    l0, g0, p0 = one_step(False)
    l1, g1, p1 = one_step(True)
    torch.cuda.synchronize()
    assert abs(l0 - l1) <= 1e-6 * max(1.0, abs(l0)), (l0, l1)
    assert set(g0) == set(g1) and len(g0) == 69
    for name in g0:
        # fp32 atomics in the weight-gradient kernels make a run reproducible to a few ulp of the largest entry, not bitwise
        scale = float(g0[name].abs().max()) + 1e-30
        assert float((g0[name] - g1[name]).abs().max()) <= 2e-5 * scale, name
        assert float((p0[name] - p1[name]).abs().max()) <= 1e-6 + 2e-3 * 5e-4, name  # one Adam step of lr 5e-4


def test_bench_under_torch_distributed_run():
    """bench.py launched the way the driver launches it for N > 1 (one process here): env:// RCCL init, barrier-bracketed
    timed region, tile all-gather, max-over-ranks reduction; both scaling modes."""
    for scaling in ("weak", "strong"):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
               "--no-cpu-baseline", "--no-extras", "--size", "128", "--scaling", scaling]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        j = json.loads(line)
        assert j["n_gpus"] == 1 and j["scaling"] == scaling and j["value"] > 0 and j["roofline"]["avg_launch_ms"] > 0
        if scaling == "weak":  # a weak run under a launcher carries the strong leg (one view per step, rays split over the ranks)
            sl = j["strong_leg"]
            assert sl["ms_per_step"] > 0 and sl["per_rank"][0]["march_ms"] > 0 and sl["per_rank"][0]["allgather_ms"] > 0
            assert 0.2 < sl["measured_speedup_vs_one_gpu_view"] < 5.0  # one rank, a 128 x 128 view, 2 steps: the same work, give or take the launch thread
        else:
            assert "strong_leg" not in j


def test_bare_bench_command():
    """`python bench.py --gpus 1 ...` exactly as the driver types it (no launcher, no RANK in the environment): one process, no
    process group, the JSON line with roofline and parity objects."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-extras",
                          "--no-cpu-baseline", "--size", "128"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["value"] > 0 and "per_rank" not in j and "strong_leg" not in j
    # --gpus 2 on this one-GPU box: refused with a message, not a traceback (the launcher checks the device count first)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "HIP device(s) visible" in out.stderr
