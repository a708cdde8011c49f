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


if __name__ == "__main__" and "--trace" not in sys.argv and "--alternate" not in sys.argv:
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


def trace(steps=12):
    """Where the encoder of step i + 1 actually runs: HIP events on both streams (Renderer.prefetch / render with tickets),
    times in ms after the start of the first traced march."""
    from neuralbody_amd import ops

    dev = torch.device("cuda", 0)
    sd, body, net, rend, bd, n_rays = bench.build_scene(dev, 512, 512, 64, None)
    poses = bench.build_poses(dev, body, bd, 512, 512)
    rows = []
    marks = []  # (step, layer, event) recorded on the encoder's stream behind every BatchNorm launch
    cur_step = [0]
    for fname in ("enc_bn_relu", "enc_bn_relu_split"):
        def wrap(f):
            def g(*a, **k):
                r = f(*a, **k)
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append((cur_step[0], e))
                return r
            return g
        setattr(ops, fname, wrap(getattr(ops, fname)))
    with torch.no_grad():
        for i in range(3):
            rend.render(poses[i])
        torch.cuda.synchronize()
        ticket = None
        orig = rend.prefetch

        def traced_prefetch(batch):
            t = orig(batch)
            return t

        for i in range(steps):
            cur, ticket_next = ticket, None
            side = getattr(rend, "_side_stream", None)
            e_enc0 = torch.cuda.Event(enable_timing=True)
            e_enc1 = torch.cuda.Event(enable_timing=True)
            cur_step[0] = i
            ticket_next = rend.prefetch(poses[(i + 1) % len(poses)])
            side = rend._side_stream
            # the ticket's ready event marks the end; a start marker cannot be inserted behind the wait inside prefetch, so the
            # encoder's own duration is taken from a second event pair around an identical pass below
            ops.MARCH_EVENTS = []
            rend.render(poses[i % len(poses)], prefetched=cur)
            (m0, m1), = ops.MARCH_EVENTS
            ops.MARCH_EVENTS = None
            e_enc1.record(side)
            rows.append((m0, m1, e_enc1))
            ticket = ticket_next
        torch.cuda.synchronize()
    base = rows[2][0]
    for k in (4, 5):
        m0, m1, _ = rows[k]
        print("step %d: march %.3f .. %.3f; the 17 layers of the encoder enqueued with it finish at (ms after the march's start):" % (
            k, base.elapsed_time(m0), base.elapsed_time(m1)))
        print("   " + " ".join("%.2f" % m0.elapsed_time(e) for st, e in marks if st == k))
    print("step | march start | march end | encoder of the next step done (side stream) | gap to the next march start")
    for k in range(2, len(rows) - 1):
        m0, m1, e1 = rows[k]
        print("%4d | %10.3f | %9.3f | %9.3f | %6.3f" % (k, base.elapsed_time(m0), base.elapsed_time(m1), base.elapsed_time(e1),
                                                          m1.elapsed_time(rows[k + 1][0])))


if __name__ == "__main__" and "--trace" in sys.argv:
    trace()
if __name__ == "__main__" and "--alternate" in sys.argv:
    alternate()
