"""Why the training leg inside `python bench.py` read 8.9-9.6 ms when `--mode train` reads 6.6 (round 4): the same leg after the
main loop of bench.py, with and without what the main loop leaves behind."""
import argparse
import gc
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench  # noqa: E402

a = argparse.Namespace(size=512, samples=64, steps=6, warmup=2, reuse_volumes=False, precision=None)
dev = torch.device("cuda", 0)
print("train first              %.3f ms" % bench.train_bench(a, dev)["value"], flush=True)
sd, body, net, rend, bd, n_rays = bench.build_scene(dev, 512, 512, 64, None)
poses = bench.build_poses(dev, body, bd, 512, 512)
with torch.no_grad():
    t = None
    for i in range(16):
        cur, t = t, rend.prefetch(poses[(i + 1) % 8])
        rend.render(poses[i % 8], prefetched=cur)
    torch.cuda.synchronize()
print("train after render loop  %.3f ms" % bench.train_bench(a, dev)["value"], flush=True)
print("train again              %.3f ms" % bench.train_bench(a, dev)["value"], flush=True)
del t, cur, rend, net, poses
gc.collect()
torch.cuda.empty_cache()
print("train after empty_cache  %.3f ms" % bench.train_bench(a, dev)["value"], flush=True)
