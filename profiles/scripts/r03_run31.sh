set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3af; mkdir -p $out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "stream_workers or reentrant" > $out/pytest.txt 2>&1; rc=$?; tail -8 $out/pytest.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench_configs.py --configs 3,5 --check 0 2> $out/c.err | cut -c1-230
timeout -k 10 300 python bench_configs.py --configs 3,5 --check 0 --deterministic 1 2> $out/cd.err | cut -c1-230
python -c "import __graft_entry__ as g; g.smoke()"
