cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3ag; mkdir -p $O
bash tools/pmc_march.sh f16f6r r3ag_pmc > /dev/null 2>&1; cp gpurun_out/r3ag_pmc_summary.txt $O/pmc_march_raw.txt
bash tools/pmc_march.sh f16f6 r3ag_pmc6 > /dev/null 2>&1; cp gpurun_out/r3ag_pmc6_summary.txt $O/pmc_march_ms6_raw.txt
cat $O/pmc_march_raw.txt
rm -rf gpurun_out/r3ag_pmc*
timeout 600 python -m pytest tests -q -x -m gpu -k "f16f6 or points or decode or cube" 2>&1 | tail -2
for p in f16f6 f16f6r; do timeout 200 python bench.py --no-cpu-baseline --no-extras --precision $p 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$p', j['roofline']['kernel'], 'march %.2f ms' % j['roofline']['avg_launch_ms'], 'step %.2f' % j['ms_per_step'], 'parity', j.get('parity_linf'), j.get('parity_linf_all'))
"; done
