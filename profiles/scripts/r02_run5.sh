set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r2e; mkdir -p $out
XAI_PARITY_REPORT=$out/parity_deterministic.json timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=6 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -14 $out/pytest.log
./image-classification-xai_amd/csrc/tune/repro_graph_memset2_memset > $out/repro_graph_memset2.txt 2>&1; echo "repro memset rc=$?"
./image-classification-xai_amd/csrc/tune/repro_graph_memset2_kernel >> $out/repro_graph_memset2.txt 2>&1; echo "repro kernel rc=$?"
cat $out/repro_graph_memset2.txt
timeout -k 10 500 python profiles/bench_kernels.py --json $out/kernels.json > $out/kernels.txt 2>&1; echo "bench_kernels rc=$?"; cat $out/kernels.txt
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --deterministic 1 > $out/bench_finddb_deterministic.json 2> $out/bench_finddb_deterministic.err; echo "bench rc=$?"
timeout -k 10 500 python tests/parity_report.py resnet --mode finddb_deterministic --out $out/resnet_finddb_deterministic.json > $out/resnet_finddb_deterministic.log 2>&1; echo "resnet rc=$?"; head -30 $out/resnet_finddb_deterministic.log
cut -c1-700 $out/bench_finddb_deterministic.json
