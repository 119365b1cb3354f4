set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3ao; mkdir -p $out
cd $R
timeout -k 10 300 python profiles/experiments/exp_k2_over_steps.py 3 30 2>/dev/null | tee $out/k2_3streams.json | cut -c1-700
timeout -k 10 300 python profiles/experiments/exp_k2_over_steps.py 1 30 2>/dev/null | tee $out/k2_1stream.json | cut -c1-700
