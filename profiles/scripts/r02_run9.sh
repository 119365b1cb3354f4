set -o pipefail
export TMPDIR=/tmp
R=/root/repo; out=$R/gpurun_out/r2q; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $out/pytest_kernels.log 2>&1; echo "pytest kernels rc=$?"; tail -4 $out/pytest_kernels.log
timeout -k 10 300 python profiles/bench_kernels.py --json $out/kernels.json > $out/kernels.txt 2>&1; grep -i "blur\|rise_apply" $out/kernels.txt
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_kernels_$c -o t -- python3 $R/profiles/bench_kernels.py > $out/pmc_kernels_$c.txt 2> $out/pmc_kernels_$c.err; echo "pmc kernels $c rc=$?"
done
cd $R
python3 profiles/pmc_kernel_table.py $out/pmc_kernels_FETCH_SIZE $out/pmc_kernels_WRITE_SIZE > $out/pmc_kernels.csv 2> $out/pmc_kernels.err; cat $out/pmc_kernels.err; cut -c1-220 $out/pmc_kernels.csv
rm -rf $out/pmc_kernels_FETCH_SIZE $out/pmc_kernels_WRITE_SIZE
