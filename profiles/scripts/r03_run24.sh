set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3x; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -k "bench_sweep_workload" > $out/alone.txt 2>&1; echo "alone (fresh box, find-db disabled) rc=$?"; tail -3 $out/alone.txt; grep "^E  " $out/alone.txt | head -3
timeout -k 10 300 python bench.py --lean --steps 5 > $out/bench_lean.json 2> $out/bench_lean.err; python3 -c "import json;d=json.load(open('$out/bench_lean.json'));print('find-db disabled:', d['value'], d['roofline']['frac'])"
XAI_MIOPEN_DISABLE_FIND_DB=0 timeout -k 10 300 python bench.py --lean --steps 5 > $out/bench_lean2.json 2> $out/bench_lean2.err; python3 -c "import json;d=json.load(open('$out/bench_lean2.json'));print('find-db enabled, private db:', d['value'])"
