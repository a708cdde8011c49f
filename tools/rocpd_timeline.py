"""Launch-by-launch timeline of ONE Renderer.render step from a rocprofv3 --kernel-trace database: every kernel between the
second-last and the last march launch, with its duration and the gap to the previous kernel's end (run on the GPU box).

    python tools/rocpd_timeline.py gpurun_out/prof/x_results.db [march kernel substring] > profiles/rNN_step_timeline.md
"""
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if "(" in name and not name.startswith("void at::"):
        name = name[:name.index("(")]
    return name.replace("void ", "")[:90]


def main(path, march="nb_march_fold_kernel"):  # (not "nb_march": nb_march_fixup_kernel follows every march since round 6)
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if march in r[0]]
    if len(idx) < 2:
        print("need two march launches, found", len(idx))
        return
    a, b = idx[-2], idx[-1]
    seg = rows[a:b + 1]
    print("# one render step: the launches from the end of a march to the end of the next (%d launches)\n" % (len(seg) - 1))
    print("| # | kernel | start after previous march end (us) | duration (us) | gap before (us) |\n|---|---|---|---|---|")
    t0, prev_end = seg[0][2], seg[0][2]
    busy = 0.0
    for i, (n, s, e) in enumerate(seg[1:], 1):
        print("| %d | %s | %.1f | %.1f | %.1f |" % (i, short(n), (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
        busy += (e - s) / 1e3
        prev_end = max(prev_end, e)
    span = (seg[-1][1] - t0) / 1e3
    print("\nfrom the end of a march to the start of the next: %.1f us, of which kernels %.1f us (%d launches), idle %.1f us; march %.1f us"
          % (span, busy - (seg[-1][2] - seg[-1][1]) / 1e3, len(seg) - 2, span - (busy - (seg[-1][2] - seg[-1][1]) / 1e3),
             (seg[-1][2] - seg[-1][1]) / 1e3))


if __name__ == "__main__":
    main(*sys.argv[1:3])
