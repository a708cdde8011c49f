cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4t; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
timeout 900 python bench.py --steps 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
r=json.load(open("$O/bench.json"))
print("ms_per_step %.3f median %.3f serial %.3f march %.3f encoder_ms %.3f launches %d turntable %.3f train %.3f" % (r["ms_per_step"], r["median_ms_per_step"], r["serial_ms_per_step"], r["roofline"]["avg_launch_ms"], r["extras"]["encoder_ms"], r["extras"]["launches_per_view"], r["extras"]["turntable_ms_per_view"], r["extras"]["train_step_ms"]))
PY
