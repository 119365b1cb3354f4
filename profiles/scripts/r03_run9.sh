set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3i; mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -k "captured or smoothGrad" > $out/pytest.txt 2>&1; rc=$?; tail -25 $out/pytest.txt; exit $rc
