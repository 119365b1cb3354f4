set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3ad; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd $R
timeout -k 10 400 python bench.py --steps 5 --no-cpu-baseline > $out/bench_line.json 2> $out/bench.err; tail -8 $out/bench.err; python3 -c "import json;d=json.load(open('$out/bench_line.json'));print(d['sweep_strong']['forward_batches_warmup_and_timed'], d['sweep_strong']['value'])"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; rc=$?; tail -6 $out/pytest.txt; exit $rc
