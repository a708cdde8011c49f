# round 6, job j: the prefetched encoder pass as one HIP graph: default bench with / without (step time, strong8 proxy), frame tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6j; mkdir -p $O
python bench.py --no-cpu-baseline > $O/bench_graph.json 2> $O/bench_graph.err
python bench.py --no-cpu-baseline --no-encoder-graph > $O/bench_nograph.json 2> $O/bench_nograph.err
python bench.py --no-cpu-baseline > $O/bench_graph2.json 2> $O/bench_graph2.err
python - <<'PY'
import json
for f in ("bench_graph","bench_nograph","bench_graph2"):
    try:
        r=json.load(open("gpurun_out/r6j/%s.json"%f)); e=r["extras"]
        print(f, "ms/step %.3f median %.3f march %.3f serial %.3f | strong8 rank %.3f full %.3f pred %.2f | turntable %s" % (r["ms_per_step"], r["median_ms_per_step"], r["roofline"]["avg_launch_ms"], r.get("serial_ms_per_step",0), e.get("strong8_rank_ms",0), e.get("strong8_full_view_ms",0), e.get("strong8_predicted_speedup",0), e.get("turntable_ms_per_view")))
    except Exception as ex:
        print(f, "ERR", ex, open("gpurun_out/r6j/%s.err"%f).read()[-800:])
PY
