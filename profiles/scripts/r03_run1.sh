# round 3, call 1: the changed paths (streaming IG, K16, benched-composition oracle test) and the new bench legs
set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3a; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; rc=$?; tail -15 $out/pytest.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --steps 5 --warmup 1 > $out/bench_line.json 2> $out/bench.err; rc=$?; tail -12 $out/bench.err; cut -c1-1500 $out/bench_line.json; exit $rc
