set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3v; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd $R
ls -la ~/.config/miopen ~/.cache/miopen 2>&1 | head -5
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -k "bench_sweep_workload" > $out/alone.txt 2>&1; echo "alone rc=$?"; tail -3 $out/alone.txt
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -k "bench_sweep_workload" > $out/alone2.txt 2>&1; echo "alone again rc=$?"; tail -3 $out/alone2.txt
ls ~/.config/miopen 2>&1 | head; 
timeout -k 10 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -k "bench_contract_line or bench_sweep_workload" > $out/after_contract.txt 2>&1; echo "after contract rc=$?"; tail -3 $out/after_contract.txt
ls -la ~/.config/miopen 2>&1 | head; find /tmp -maxdepth 1 -name "xai_miopen*" | head
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -k "bench_sweep_workload" > $out/alone3.txt 2>&1; echo "alone after rc=$?"; tail -3 $out/alone3.txt
