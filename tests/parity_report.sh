#!/bin/bash
# Runs on the GPU box (gpurun -- bash tests/parity_report.sh): produces the measured-error ledgers and the evidence files
# that the tolerances of tests/test_gpu_e2e.py are tied to.  Outputs under gpurun_out/parity/; copy them to profiles/r02_*.
set -o pipefail
out=gpurun_out/parity
mkdir -p $out
export TMPDIR=/tmp
XAI_PARITY_REPORT=$out/parity_deterministic.json timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $out/pytest_deterministic.log 2>&1
echo "pytest deterministic rc=$?" | tee -a $out/summary.txt
tail -3 $out/pytest_deterministic.log | tee -a $out/summary.txt
XAI_TEST_DETERMINISTIC=0 XAI_PARITY_REPORT=$out/parity_default.json timeout -k 10 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider > $out/pytest_default.log 2>&1
echo "pytest default rc=$?" | tee -a $out/summary.txt
tail -3 $out/pytest_default.log | tee -a $out/summary.txt
timeout -k 10 600 python tests/parity_report.py gates --out $out/gate_flips.json > $out/gates.log 2>&1
echo "gates rc=$?" | tee -a $out/summary.txt
for mode in immediate deterministic finddb; do
  timeout -k 10 900 python tests/parity_report.py resnet --mode $mode --out $out/resnet_$mode.json > $out/resnet_$mode.log 2>&1
  echo "resnet $mode rc=$?" | tee -a $out/summary.txt
done
tail -40 $out/gates.log
