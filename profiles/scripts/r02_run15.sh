export TMPDIR=/tmp
out=gpurun_out/r2w; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -k "dispatch or directory or cli or sweep" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
timeout -k 10 900 python bench.py --workload sweep --sweep-images 24 --steps 1 --warmup 1 --no-cpu-baseline > $out/bench_sweep_line.json 2> $out/bench_sweep.err; echo "sweep rc=$?"; cut -c1-330 $out/bench_sweep_line.json
timeout -k 10 900 python bench.py --workload sweep --sweep-images 24 --steps 1 --warmup 1 --no-cpu-baseline --deterministic 1 > $out/bench_sweep_line_deterministic.json 2>> $out/bench_sweep.err; echo "sweep det rc=$?"; cut -c1-330 $out/bench_sweep_line_deterministic.json
