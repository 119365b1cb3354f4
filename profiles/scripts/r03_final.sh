# Round-3 measurement pass (gpurun -- bash profiles/scripts/r03_final.sh [part]): ledger of the whole GPU suite, the contract line, the
# other configurations, the per-kernel table, the 4-rank rehearsal on one GPU.  Outputs under gpurun_out/r3z/; copied to profiles/r03_*.
set -o pipefail
export TMPDIR=/tmp
R=/root/repo; out=$R/gpurun_out/r3z; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd $R
part=${1:-all}
if [ $part = all ] || [ $part = tests ]; then
XAI_PARITY_REPORT=$out/parity_deterministic.json timeout -k 10 1000 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $out/pytest_deterministic.log 2>&1; rc=$?
echo "pytest deterministic rc=$rc"; tail -3 $out/pytest_deterministic.log; [ $rc -eq 0 ] || exit $rc
XAI_TEST_DETERMINISTIC=0 XAI_PARITY_REPORT=$out/parity_default.json timeout -k 10 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider > $out/pytest_default.log 2>&1
echo "pytest default-mode rc=$?"; tail -3 $out/pytest_default.log
fi
if [ $part = all ] || [ $part = bench ]; then
timeout -k 10 900 python bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?"; tail -12 $out/bench.err; cut -c1-400 $out/bench_line.json
timeout -k 10 900 python bench_configs.py > $out/bench_configs.jsonl 2> $out/bench_configs.err; echo "configs rc=$?"; cut -c1-260 $out/bench_configs.jsonl
timeout -k 10 900 python bench_configs.py --deterministic 1 --configs 2,3,4,5 --check 0 > $out/bench_configs_deterministic.jsonl 2> $out/bench_configs_det.err; echo "configs det rc=$?"; cut -c1-260 $out/bench_configs_deterministic.jsonl
timeout -k 10 600 python profiles/bench_kernels.py > $out/kernels.txt 2> $out/kernels.err; echo "kernels rc=$?"; tail -12 $out/kernels.txt
fi
if [ $part = all ] || [ $part = ranks ]; then
export XAI_DIST_BACKEND=gloo XAI_FORCE_DEVICE=0 OMP_NUM_THREADS=2
timeout -k 10 500 python bench.py --gpus 4 --steps 2 --warmup 1 --images 8 --strong-images 32 > $out/rehearsal_4_ranks_bench.out 2> $out/rehearsal_4_ranks_bench.err; echo "4 ranks ig rc=$?"; tail -1 $out/rehearsal_4_ranks_bench.out | cut -c1-300
timeout -k 10 500 python bench.py --gpus 4 --workload sweep --sweep-images 24 --sweep-methods grad,ig,gc --steps 1 --warmup 1 > $out/rehearsal_4_ranks_sweep.out 2> $out/rehearsal_4_ranks_sweep.err; echo "4 ranks sweep rc=$?"; tail -1 $out/rehearsal_4_ranks_sweep.out | cut -c1-300
fi
