export TMPDIR=/tmp
out=gpurun_out/r2al; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -k "cli or directory" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
