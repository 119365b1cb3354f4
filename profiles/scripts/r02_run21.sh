export TMPDIR=/tmp
R=/root/repo; out=$R/gpurun_out/r2ac; mkdir -p $out
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/sweep_trace -o t -- python3 $R/bench.py --workload sweep --sweep-images 8 --sweep-methods ig --steps 1 --warmup 1 --no-cpu-baseline > $out/sweep_line.json 2> $out/sweep.err; echo "trace rc=$?"
f=$(find $out/sweep_trace -name "*kernel_trace.csv" | head -1)
python3 $R/profiles/summarize_trace.py $f segment_sums 2 9 > $out/sweep_timed_region.txt 2>&1; head -24 $out/sweep_timed_region.txt
rm -rf $out/sweep_trace
