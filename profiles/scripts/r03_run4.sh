# round 3, call 4: multi-stream flows (bit-identity tests), the restructured bench line
set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3d; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_e2e.py -m gpu -x -q -k "config or bench or streams or sweep" > $out/pytest.txt 2>&1; rc=$?; tail -15 $out/pytest.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 700 python bench.py --steps 5 --warmup 1 > $out/bench_line.json 2> $out/bench.err; rc=$?; tail -14 $out/bench.err; cut -c1-600 $out/bench_line.json; exit $rc
