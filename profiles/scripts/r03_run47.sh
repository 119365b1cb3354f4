set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3au; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd $R
for i in 1 2; do
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/pytest_$i.txt 2>&1; rc=$?; tail -2 $out/pytest_$i.txt; [ $rc -eq 0 ] || exit $rc
done
