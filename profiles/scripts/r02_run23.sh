export TMPDIR=/tmp
out=gpurun_out/r2ae; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
XAI_FUZZ_SCALE=1 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider -k "ig_family" > $out/pytest1.log 2>&1; echo "x1 rc=$?"; tail -15 $out/pytest1.log
XAI_FUZZ_SCALE=1500 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider -k "ig_family" > $out/pytest1500.log 2>&1; echo "x40 rc=$?"; tail -15 $out/pytest1500.log
