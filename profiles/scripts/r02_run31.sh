export TMPDIR=/tmp
out=gpurun_out/r2am; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -k "two_ranks or directory or cli" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $out/pytest.log
