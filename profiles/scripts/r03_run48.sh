set -o pipefail
export TMPDIR=/tmp
R=/root/repo; out=$R/gpurun_out/r3av; mkdir -p $out
cd /tmp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_thr -o t -- python3 $R/bench.py --mode throughput --lean --steps 1 --warmup 1 > $out/pmc_thr.json 2> $out/pmc_thr.err; echo "pmc + find-db rc=$?"
tail -5 $out/pmc_thr.err | cut -c1-200; head -c 300 $out/pmc_thr.json; echo
f=$(find $out/pmc_thr -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -c "" $f; rm -rf $out/pmc_thr
