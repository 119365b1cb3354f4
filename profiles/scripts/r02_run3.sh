set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r2c; mkdir -p $out
XAI_PARITY_REPORT=$out/parity_deterministic.json timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -40 $out/pytest.log
cd /tmp
for cfg in "immediate 24" "finddb 50"; do set -- $cfg
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof_$1_$2 -o trace -- python3 /root/repo/tests/parity_report.py kernels --mode $1 --batch $2 --out /root/repo/$out/kernels_$1_$2.json > /root/repo/$out/kernels_$1_$2.log 2>&1; echo "kernels $1 $2 rc=$?"
done
cd /root/repo
for f in $(find $out -name "*kernel_trace.csv"); do python3 profiles/between_markers.py $f > ${f%.csv}_between_markers.txt 2>&1; cat ${f%.csv}_between_markers.txt; done
find $out -name "*kernel_trace.csv" -size +8M -delete
./image-classification-xai_amd/csrc/tune/tune_rise > $out/tune_rise.txt 2>&1; echo "tune rc=$?"; cat $out/tune_rise.txt
