export TMPDIR=/tmp
out=gpurun_out/r2z; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "plain_c" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $out/pytest.log
