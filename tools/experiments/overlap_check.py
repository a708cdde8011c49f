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


if __name__ == "__main__" and "--trace" not in sys.argv and "--alternate" not in sys.argv and "--two-sides" not in sys.argv:
    main()


def alternate(steps=32):
    """Marches alternating between two streams (the head of march i + 1 in the tail of march i), encoders two frames ahead on a
    third: ms per view against the one-stream pipeline of main()."""
    dev = torch.device("cuda", 0)
    sd, body, net, rend, bd, n_rays = bench.build_scene(dev, 512, 512, 64, None)
    poses = bench.build_poses(dev, body, bd, 512, 512)
    enc_s = torch.cuda.Stream(device=dev)
    ms = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]

    def encode(b):
        sp = rend.prepare_sp_input(b)
        fv = net.encode_sparse_voxels(sp)
        net.make_scene(fv, sp, net.march_precision())
        return fv

    def run(n, depth):
        ready, fvs, done = {}, {}, {}

        def prefetch(j, after):
            with torch.cuda.stream(enc_s):
                if after is not None:
                    enc_s.wait_event(after)
                fvs[j] = encode(poses[j % len(poses)])
                ready[j] = torch.cuda.Event()
                ready[j].record(enc_s)

        for j in range(depth):
            prefetch(j, None)
        outs = []
        for i in range(n):
            prefetch(i + depth, done.get(i - 1))
            s = ms[i % 2] if depth > 1 else ms[0]
            with torch.cuda.stream(s):
                s.wait_event(ready.pop(i))
                outs.append(rend.render(poses[i % len(poses)], feature_volume=fvs.pop(i))["rgb_map"])
                done[i] = torch.cuda.Event()
                done[i].record(s)
            del outs[:-4]
        torch.cuda.synchronize()
        return outs[-1]

    with torch.no_grad():
        ref = rend.render(poses[(steps - 1) % len(poses)])["rgb_map"].clone()
        torch.cuda.synchronize()
        for rep in range(2):
            for depth in (1, 2):
                run(6, depth)
                t0 = time.perf_counter()
                out = run(steps, depth)
                dt = (time.perf_counter() - t0) / steps * 1e3
                print("marches on %s, encoders %d ahead: %.3f ms per view (max |rgb - serial| %.1e)" % (
                    "two alternating streams" if depth > 1 else "one stream", depth, dt, float((out - ref).abs().max())), flush=True)


def two_side_streams(steps=32):
    """Encoders two views ahead on TWO alternating side streams (the chains of consecutive views overlap each other across two
    march tails) against one side stream one view ahead.  (The BatchNorm running statistics of concurrent passes race here:
    timing experiment only.)"""
    dev = torch.device("cuda", 0)
    sd, body, net, rend, bd, n_rays = bench.build_scene(dev, 512, 512, 64, None)
    poses = bench.build_poses(dev, body, bd, 512, 512)
    sides = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]

    marchers = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]

    def run(n, depth, n_streams, n_march=1):
        tickets = {}
        for i in range(n):
            cur = tickets.pop(i, None)
            ms = marchers[i % 2] if n_march == 2 else torch.cuda.current_stream()
            with torch.cuda.stream(ms):
                fence = rend.fence()
                out = rend.render(poses[i % len(poses)], prefetched=cur)["rgb_map"]
                rend._side_stream = sides[(i + depth) % n_streams]
                tickets[i + depth] = rend.prefetch(poses[(i + depth) % len(poses)], after=fence)
        torch.cuda.synchronize()
        return out

    with torch.no_grad():
        ref = rend.render(poses[(steps - 1) % len(poses)])["rgb_map"].clone()
        for rep in range(2):
            for depth, ns, nm in ((1, 1, 1), (2, 2, 1), (2, 2, 2), (3, 2, 2), (4, 2, 2)):
                run(8, depth, ns, nm)
                t0 = time.perf_counter()
                out = run(steps, depth, ns, nm)
                dt = (time.perf_counter() - t0) / steps * 1e3
                print("encoders %d ahead on %d side stream(s), marches on %d stream(s): %.3f ms per view (max |rgb - serial| %.1e)" % (
                    depth, ns, nm, dt, float((out - ref).abs().max())), flush=True)


def trace(steps=14, depth=1):
    """Where the encoder of a later step actually runs: HIP events on both streams (fence / render with ticket / prefetch `depth`
    views ahead, the calling order of bench.py), times in ms after the start of the first traced march."""
    from neuralbody_amd import ops

    dev = torch.device("cuda", 0)
    sd, body, net, rend, bd, n_rays = bench.build_scene(dev, 512, 512, 64, None)
    poses = bench.build_poses(dev, body, bd, 512, 512)
    rows, enc = [], {}
    marks = []  # (view, event) recorded on the encoder's stream behind every BatchNorm launch
    cur_view = [0]
    for fname in ("enc_bn_relu", "enc_bn_relu_split"):
        def wrap(f):
            def g(*a, **k):
                r = f(*a, **k)
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append((cur_view[0], e))
                return r
            return g
        setattr(ops, fname, wrap(getattr(ops, fname)))
    tickets = {}
    with torch.no_grad():
        for i in range(3):
            rend.render(poses[i])
        torch.cuda.synchronize()
        for i in range(steps):
            cur = tickets.pop(i, None)
            fence = rend.fence()
            ops.MARCH_EVENTS = []
            rend.render(poses[i % len(poses)], prefetched=cur)
            (m0, m1), = ops.MARCH_EVENTS
            ops.MARCH_EVENTS = None
            cur_view[0] = i + depth
            tickets[i + depth] = rend.prefetch(poses[(i + depth) % len(poses)], after=fence)
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(rend._side_stream)
            enc[i + depth] = e1
            rows.append((m0, m1))
        torch.cuda.synchronize()
    base = rows[4][0]
    print("prefetch depth %d" % depth)
    for k in (6, 7):
        m0, m1 = rows[k]
        print("view %d: march %.3f .. %.3f; the 17 layers of the encoder of view %d (enqueued behind this march) finish at (ms after the march's start):" % (
            k, base.elapsed_time(m0), base.elapsed_time(m1), k + depth))
        print("   " + " ".join("%.2f" % m0.elapsed_time(e) for v, e in marks if v == k + depth))
    print("view | march start | march end | its own encoder done (side stream) | gap to the next march start")
    for k in range(4, len(rows) - 1):
        m0, m1 = rows[k]
        print("%4d | %10.3f | %9.3f | %9.3f | %6.3f" % (k, base.elapsed_time(m0), base.elapsed_time(m1),
                                                          base.elapsed_time(enc[k]) if k in enc else float("nan"), m1.elapsed_time(rows[k + 1][0])))


if __name__ == "__main__" and "--trace" in sys.argv:
    trace(depth=2 if "--depth2" in sys.argv else 1)
if __name__ == "__main__" and "--alternate" in sys.argv:
    alternate()
if __name__ == "__main__" and "--two-sides" in sys.argv:
    two_side_streams()
