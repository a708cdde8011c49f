cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3ah; mkdir -p $O
( rocm-smi --showclocks --showpower --showmaxpower --showmemvendor 2>/dev/null | grep -v "^$" | head -30; rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | grep -i partition ) > $O/smi.txt 2>&1
bash tools/pmc_traffic.sh f16f6 r3ah_traffic > /dev/null 2>&1; cp gpurun_out/r3ah_traffic_summary.txt $O/traffic_ms6_raw.txt; rm -rf gpurun_out/r3ah_traffic_*
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
( python bench.py --steps 500 --no-extras --no-cpu-baseline > /dev/null 2>&1 ) & pid=$!; sleep 10; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power \(W\)" | sed 's/ \+/ /g' >> $O/smi.txt; wait $pid
cat $O/smi.txt $O/traffic_ms6_raw.txt
python - <<'PY'
import json
for line in open('gpurun_out/r3ah/bench.json'):
    if line.startswith('{'):
        j = json.loads(line)
        print({k: j[k] for k in ('value', 'ms_per_step', 'parity_linf', 'parity_linf_all')}); r = j['roofline']; print(r['kernel'], r['avg_launch_ms'], r['frac'])
        e = j['extras']; print({k: round(v, 3) for k, v in e.items() if isinstance(v, float)})
PY
