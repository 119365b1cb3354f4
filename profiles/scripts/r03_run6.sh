# round 3, call 6: (a) stream count for the sweep, (b) PMC traffic of the IG flows, (c) rocprof stats of the contract command, (d) PMC of K2 in bench
set -o pipefail
export TMPDIR=/tmp
R=/root/repo; out=$R/gpurun_out/r3f; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd $R
for n in 1 2 3 4 6; do
  timeout -k 10 300 python bench.py --workload sweep --sweep-methods ig --sweep-images 64 --steps 1 --warmup 1 --streams $n > $out/sweep_streams_$n.json 2> $out/sweep_streams_$n.err; echo "sweep streams $n rc=$?"; python3 -c "import json;d=json.load(open('$out/sweep_streams_$n.json'));print($n, d['value'], 'images/s')"
done
cd /tmp
for flow in streaming buffered; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${flow}_$c -o t -- python3 $R/profiles/experiments/exp_ig_flows.py $flow > $out/pmc_${flow}_$c.txt 2> $out/pmc_${flow}_$c.err; echo "pmc $flow $c rc=$?"
  done
  python3 $R/profiles/pmc_ig_flows.py $out/pmc_${flow}_FETCH_SIZE $out/pmc_${flow}_WRITE_SIZE $flow | tee -a $out/ig_flows_traffic.jsonl | cut -c1-400
  rm -rf $out/pmc_${flow}_FETCH_SIZE $out/pmc_${flow}_WRITE_SIZE
done
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_trace -o t -- python3 $R/bench.py --steps 3 --warmup 1 --lean > $out/bench_line_under_rocprof.json 2> $out/bench_under_rocprof.err; echo "trace rc=$?"
f=$(find $out/bench_trace -name "*kernel_trace.csv" | head -1); python3 $R/profiles/summarize_trace.py $f ig_accum 0 3 > $out/bench_timed_region.txt 2>&1; head -14 $out/bench_timed_region.txt
cp $(find $out/bench_trace -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv; rm -rf $out/bench_trace
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_bench_$c -o t -- python3 $R/bench.py --steps 1 --warmup 1 --lean --fuse-bn-relu 0 > $out/pmc_bench_$c.json 2> $out/pmc_bench_$c.err; echo "pmc bench $c rc=$?"
  f=$(find $out/pmc_bench_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && { head -1 $f > $out/pmc_header.csv; grep -i "ig_accum" $f | head -4 > $out/pmc_ig_accum_$c.csv; }
  rm -rf $out/pmc_bench_$c
done
cat $out/pmc_header.csv $out/pmc_ig_accum_*.csv | cut -c1-300
