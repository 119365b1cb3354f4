export TMPDIR=/tmp
out=gpurun_out/r2ad; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
XAI_FUZZ_SCALE=300 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider --durations=5 > $out/pytest_soak.log 2>&1; echo "soak rc=$?"; tail -12 $out/pytest_soak.log
