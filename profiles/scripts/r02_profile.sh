# Round-2 measurement pass (gpurun -- bash profiles/scripts/r02_profile.sh): kernel-trace stats of the contract command, PMC
# traffic (separate passes, --kernel-trace only, as MI355X_MICROARCH.md prescribes), the other configurations, the sweep workload.
set -o pipefail
export TMPDIR=/tmp
R=/root/repo; out=$R/gpurun_out/r2p; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_trace -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_line_under_rocprof.json 2> $out/bench_under_rocprof.err; echo "trace rc=$?"
f=$(find $out/bench_trace -name "*kernel_trace.csv" | head -1); python3 $R/profiles/summarize_trace.py $f ig_accum 0 3 > $out/bench_timed_region.txt 2>&1; head -12 $out/bench_timed_region.txt
cp $(find $out/bench_trace -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv; rm -rf $out/bench_trace
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_kernels_$c -o t -- python3 $R/profiles/bench_kernels.py > $out/pmc_kernels_$c.txt 2> $out/pmc_kernels_$c.err; echo "pmc kernels $c rc=$?"
done
cd $R
python3 profiles/pmc_kernel_table.py $out/pmc_kernels_FETCH_SIZE $out/pmc_kernels_WRITE_SIZE > $out/pmc_kernels.csv 2> $out/pmc_kernels.err; grep -i "rise_apply\|ig_accum\|blur" $out/pmc_kernels.csv | cut -c1-200
rm -rf $out/pmc_kernels_FETCH_SIZE $out/pmc_kernels_WRITE_SIZE
timeout -k 10 900 python bench_configs.py > $out/bench_configs.jsonl 2> $out/bench_configs.err; echo "configs rc=$?"; cut -c1-300 $out/bench_configs.jsonl
timeout -k 10 900 python bench.py --workload sweep --sweep-images 24 --steps 1 --warmup 1 --no-cpu-baseline > $out/bench_sweep_line.json 2> $out/bench_sweep.err; echo "sweep rc=$?"; cut -c1-600 $out/bench_sweep_line.json
timeout -k 10 900 python bench.py --workload sweep --sweep-images 24 --steps 1 --warmup 1 --no-cpu-baseline --deterministic 1 > $out/bench_sweep_line_deterministic.json 2>> $out/bench_sweep.err; echo "sweep det rc=$?"; cut -c1-300 $out/bench_sweep_line_deterministic.json
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_bench_$c -o t -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --miopen-db 0 --fuse-bn-relu 0 > $out/pmc_bench_$c.json 2> $out/pmc_bench_$c.err; echo "pmc bench $c rc=$?"
  f=$(find $out/pmc_bench_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $out/pmc_header.csv; grep -i "ig_accum" $f | head -4 > $out/pmc_ig_accum_$c.csv; }
  rm -rf $out/pmc_bench_$c
done
cat $out/pmc_header.csv $out/pmc_ig_accum_*.csv | cut -c1-400
