set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3w; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -k "bench_sweep_workload" > $out/alone.txt 2>&1; echo "alone (fresh box) rc=$?"; tail -3 $out/alone.txt
timeout -k 10 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -k "bench_contract_line" > $out/contract.txt 2>&1; echo "contract rc=$?"; tail -3 $out/contract.txt
ls ~/.config/miopen 2>&1 | head -3
