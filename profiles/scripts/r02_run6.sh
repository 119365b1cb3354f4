set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r2f; mkdir -p $out
./image-classification-xai_amd/csrc/tune/repro_graph_memset2_memset > $out/repro_graph_memset2.txt 2>&1; echo "repro memset rc=$?"
./image-classification-xai_amd/csrc/tune/repro_graph_memset2_kernel >> $out/repro_graph_memset2.txt 2>&1; echo "repro kernel rc=$?"
cat $out/repro_graph_memset2.txt
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -k "bench or cli" > $out/pytest_bench.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_bench.log
