set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3aq; mkdir -p $out
cd $R
XAI_EXP_FUSED=1 XAI_EXP_BATCH=1 timeout -k 10 300 python profiles/experiments/exp_cold_start_which_layer.py 2> $out/e.err | tee $out/cold_start_fused_b1.jsonl | cut -c1-330; tail -2 $out/e.err
