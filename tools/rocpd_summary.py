"""Summarise a rocprofv3 (rocpd SQLite) result into the per-kernel table `rocprofv3 --stats` prints:
name, calls, total/avg/min/max duration (us), share of GPU kernel time.

    python tools/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/rNN_<what>.md
"""
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if "(" in name and not name.startswith("void at::"):
        name = name[:name.index("(")]
    return name.replace("void ", "")[:110]


def main(path, extra=""):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(scratch_size), max(lds_size), "
                     "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("# rocprofv3 --kernel-trace --stats summary (%s)%s\n" % (path.split("/")[-1], extra))
    print("Register columns are what rocprofv3 records per dispatch.  On gfx950 `accum_vgpr_count` comes back 0 and `vgpr_count` does not "
          "include the accumulation registers a kernel addresses as a[..] (nb_march_f6_kernel: 240 here, 256 VGPR + ~215 AGPR in its "
          "code object): the authoritative numbers are the code object's (`hipcc -Rpass-analysis=kernel-resource-usage`).\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | VGPR (as reported) | AGPR (as reported) | SGPR | scratch B | LDS B | grid x wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for n, cnt, tot, avg, mn, mx, vg, ag, sg, sc, lds, gx, wx in rows:
        print("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.2f | %s | %s | %s | %s | %s | %sx%s |" % (
            short(n), cnt, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, vg, ag, sg, sc, lds, gx, wx))
    print("\nTotal GPU kernel time: %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))
    try:
        pmc = c.execute("select pmc_info.name, kernels.name, count(*), avg(pmc_events.value), sum(pmc_events.value) "
                        "from pmc_events join pmc_info on pmc_events.pmc_id = pmc_info.id "
                        "join kernels on kernels.dispatch_id = pmc_events.event_id group by 1, 2 order by 5 desc limit 40").fetchall()
        if pmc:
            print("\n| counter | kernel | dispatches | avg per dispatch | sum |\n|---|---|---|---|---|")
            for cn, kn, cnt, avg, sm in pmc:
                print("| %s | %s | %d | %.6g | %.6g |" % (cn, short(kn), cnt, avg, sm))
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main(sys.argv[1], " — " + " ".join(sys.argv[2:]) if len(sys.argv) > 2 else "")
