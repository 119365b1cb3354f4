set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3ap; mkdir -p $out
cd $R
timeout -k 10 300 python profiles/experiments/exp_cold_start_which_layer.py 2> $out/e.err | tee $out/cold_start_which_layer.jsonl; tail -2 $out/e.err
