set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3ar; mkdir -p $out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -k "csv or evaluate_perturbation or cli" > $out/pytest.txt 2>&1; rc=$?; tail -5 $out/pytest.txt; exit $rc
