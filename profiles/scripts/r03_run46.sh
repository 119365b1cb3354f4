set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3at; mkdir -p $out
cd $R
XAI_FUZZ_SCALE=150 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "stream_workers" > $out/fuzz.txt 2>&1; rc=$?; tail -12 $out/fuzz.txt; exit $rc
