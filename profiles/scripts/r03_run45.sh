set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3as; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd $R
XAI_FUZZ_SCALE=100 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > $out/fuzz.txt 2>&1; rc=$?; tail -4 $out/fuzz.txt; exit $rc
