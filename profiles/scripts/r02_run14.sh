export TMPDIR=/tmp
out=gpurun_out/r2v; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -k "rccl or two_ranks" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $out/pytest.log
