export TMPDIR=/tmp
out=gpurun_out/r2ao; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -k "tied or single_run or embeddings" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $out/pytest.log
