export TMPDIR=/tmp
out=gpurun_out/r2af; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $out/bench_20.json 2> $out/bench_20.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r2af/bench_20.json"))
print({k: d[k] for k in ("value", "ms_per_step", "steps", "warmup")}, d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches_timed"], d["cpu_baseline"]["value"], d["unfused_classifier"]["value"])
PY
