export TMPDIR=/tmp
R=/root/repo; out=$R/gpurun_out/r2s; mkdir -p $out
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "blur" > $out/pytest_blur.log 2>&1; echo "pytest blur rc=$?"; tail -3 $out/pytest_blur.log
timeout -k 10 300 python profiles/bench_kernels.py > $out/kernels.txt 2>&1; grep -i "blur" $out/kernels.txt
bash profiles/scripts/r02_run10.sh
