export TMPDIR=/tmp
out=gpurun_out/r2u; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $out/pytest_kernels.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_kernels.log
timeout -k 10 300 python profiles/bench_kernels.py --json $out/kernels.json > $out/kernels.txt 2>&1; cat $out/kernels.txt
