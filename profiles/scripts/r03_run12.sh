set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3l; mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "captured" > $out/pytest.txt 2>&1; rc=$?; tail -25 $out/pytest.txt; exit $rc
