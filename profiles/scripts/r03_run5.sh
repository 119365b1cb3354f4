# round 3, call 5: whole GPU suite with the reference-Counter mode, special_version, 4-rank rehearsal, bench at 1 and 4 ranks
set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3e; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=12 > $out/pytest.txt 2>&1; rc=$?; tail -30 $out/pytest.txt; exit $rc
