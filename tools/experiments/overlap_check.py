"""Does the encoder of view i + 1 hide behind the march of view i?  (round 4)

The march fills every CU with two 80-KiB workgroups; the encoder is ~90 small launches that use a fraction of the chip for 1.2 ms.
Runs the bench views (a) serially on one stream, (b) with the next view's encoder (+ fold planes) on a second stream of the same
priority, (c) on a second stream of high priority, and prints ms per view for each.

    python tools/experiments/overlap_check.py [--steps 24]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=24)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    sd, body, net, rend, bd, n_rays = bench.build_scene(dev, 512, 512, 64, None)
    poses = bench.build_poses(dev, body, bd, 512, 512)

    def encode(b):
        sp = rend.prepare_sp_input(b)
        fv = net.encode_sparse_voxels(sp)
        net.make_scene(fv, sp, net.march_precision())  # builds (and caches on fv) the fold planes
        return fv

    def serial(n):
        for i in range(n):
            rend.render(poses[i % len(poses)])

    def piped(n, side):
        main_s = torch.cuda.current_stream()
        marched = None  # event after the march of the previous view
        with torch.cuda.stream(side):
            side.wait_stream(main_s)
            fv = encode(poses[0])
            ready = torch.cuda.Event()
            ready.record(side)
        for i in range(n):
            cur_fv, cur_ready = fv, ready
            with torch.cuda.stream(side):
                if marched is not None:
                    side.wait_event(marched)  # the volumes freed by now were read by the march BEFORE the one this encoder overlaps
                fv = encode(poses[(i + 1) % len(poses)])
                ready = torch.cuda.Event()
                ready.record(side)
            main_s.wait_event(cur_ready)
            rend.render(poses[i % len(poses)], feature_volume=cur_fv)
            ev = torch.cuda.Event()
            ev.record(main_s)
            marched = ev
        main_s.wait_stream(side)

    def timed(fn, *args):
        fn(4, *args[1:]) if args else fn(4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(*args) if args else fn(a.steps)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3

    with torch.no_grad():
        serial(3)
        torch.cuda.synchronize()
        for rep in range(2):
            print("serial, one stream          %.3f ms per view" % timed(serial), flush=True)
            s0 = torch.cuda.Stream(device=dev)
            print("encoder on a second stream  %.3f ms per view" % timed(piped, a.steps, s0), flush=True)
            s1 = torch.cuda.Stream(device=dev, priority=-1)
            print("... of high priority        %.3f ms per view" % timed(piped, a.steps, s1), flush=True)


if __name__ == "__main__":
    main()
