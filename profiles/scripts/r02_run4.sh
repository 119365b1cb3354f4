set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r2d; mkdir -p $out
XAI_PARITY_REPORT=$out/parity_deterministic.json timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $out/pytest.log
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof_immediate_24 -o trace -- python3 /root/repo/tests/parity_report.py kernels --mode immediate --batch 24 --module layer2.0.conv2 --out /root/repo/$out/kernels_immediate_24.json > /root/repo/$out/kernels_immediate_24.log 2>&1; echo "kernels rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof_deterministic_24 -o trace -- python3 /root/repo/tests/parity_report.py kernels --mode deterministic --batch 24 --module layer2.0.conv2 --out /root/repo/$out/kernels_deterministic_24.json > /root/repo/$out/kernels_deterministic_24.log 2>&1; echo "kernels det rc=$?"
cd /root/repo
for f in $(find $out -name "*kernel_trace.csv"); do python3 profiles/between_markers.py $f > ${f%.csv}_between_markers.txt 2>&1; echo $f; cat ${f%.csv}_between_markers.txt; done
find $out -name "*kernel_trace.csv" -size +8M -delete
./image-classification-xai_amd/csrc/tune/tune_rise > $out/tune_rise.txt 2>&1; echo "tune rc=$?"; grep -v "V1 sep" $out/tune_rise.txt
timeout -k 10 600 python profiles/experiments/exp_graph_memset.py $out/memset > $out/memset.log 2>&1; echo "memset rc=$?"; tail -45 $out/memset.log
