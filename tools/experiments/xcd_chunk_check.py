"""Chunk size of the XCD remap (NB_XCD_CHUNK builds, NB_LIB_PATH): march of the full bench view, of the two mask-culled renderers and
of one rank's 1/8 share."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from neuralbody_amd import ops, parallel  # noqa: E402

a = argparse.Namespace(size=512, samples=64, precision=None, steps=10, warmup=3)
dev = torch.device("cuda:0")
sd, body, net, rend, bd, n = bench.build_scene(dev, 512, 512, 64, None)


def march_ms(rng):
    with torch.no_grad():
        for _ in range(3):
            rend.render(bd, ray_range=rng)
        ops.MARCH_EVENTS = []
        for _ in range(10):
            rend.render(bd, ray_range=rng)
        torch.cuda.synchronize()
        ev, ops.MARCH_EVENTS = ops.MARCH_EVENTS, None
    return float(np.mean([x.elapsed_time(y) for x, y in ev]))


full = march_ms(None)
share = max(march_ms(parallel.shard_range_tiled(n, r, 8, 512, 512)) for r in (0, 3, 7))
c = bench.culled_bench(a, dev)
print("full %.3f ms | 1/8 share %.3f ms | mmsk %.3f ms | msk %.3f ms" % (full, share, c["mmsk_march_ms"], c["msk_march_ms"]))
