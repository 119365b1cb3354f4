export TMPDIR=/tmp
out=gpurun_out/r2aq; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider -k "rise" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest.log
