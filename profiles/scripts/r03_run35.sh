set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3aj; mkdir -p $out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -x -q -k "stream_workers or get_CNN_attr_dispatch or smoothGrad" > $out/pytest.txt 2>&1; rc=$?; tail -6 $out/pytest.txt; exit $rc
