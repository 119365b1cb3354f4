export TMPDIR=/tmp
R=/root/repo; out=$R/gpurun_out/r2ap; mkdir -p $out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_kernels_$c -o t -- python3 $R/profiles/bench_kernels.py > $out/pmc_kernels_$c.txt 2> $out/pmc_kernels_$c.err; echo "pmc kernels $c rc=$?"
done
cd $R
python3 profiles/pmc_kernel_table.py $out/pmc_kernels_FETCH_SIZE $out/pmc_kernels_WRITE_SIZE > $out/pmc_kernels.csv 2> $out/pmc_kernels.err; grep -i "causal\|blur\|interp\|rise_apply" $out/pmc_kernels.csv | cut -c1-160
rm -rf $out/pmc_kernels_FETCH_SIZE $out/pmc_kernels_WRITE_SIZE
timeout -k 10 300 python profiles/bench_kernels.py --json $out/kernels.json > $out/kernels.txt 2>&1; echo "kernels rc=$?"
