export TMPDIR=/tmp
out=gpurun_out/r2aa; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
timeout -k 10 400 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -k "causal or ViT_CX or vitcx" > $out/pytest2.log 2>&1; echo "pytest2 rc=$?"; tail -3 $out/pytest2.log
timeout -k 10 300 python profiles/bench_kernels.py --json $out/kernels.json > $out/kernels.txt 2>&1; grep -i "causal\|rownorm\|cluster" $out/kernels.txt
