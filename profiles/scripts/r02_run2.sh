set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r2b; mkdir -p $out
XAI_PARITY_REPORT=$out/parity_deterministic.json timeout -k 10 1000 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=15 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $out/pytest.log
cd /tmp
for cfg in "immediate 24" "finddb 50"; do set -- $cfg
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof_$1_$2 -o trace -- python3 /root/repo/tests/parity_report.py kernels --mode $1 --batch $2 --out /root/repo/$out/kernels_$1_$2.json > /root/repo/$out/kernels_$1_$2.log 2>&1; echo "kernels $1 $2 rc=$?"
done
cd /root/repo
for f in $out/prof_*/*/*kernel_trace.csv $out/prof_*/*kernel_trace.csv; do [ -f "$f" ] && python3 profiles/between_markers.py $f > ${f%.csv}_between_markers.txt 2>&1; done
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -size +20M -delete
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --deterministic 1 > $out/bench_deterministic.json 2> $out/bench_deterministic.err; echo "bench det rc=$?"
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --miopen-db 0 > $out/bench_immediate.json 2> $out/bench_immediate.err; echo "bench imm rc=$?"
cat $out/bench_*.json | cut -c1-400
