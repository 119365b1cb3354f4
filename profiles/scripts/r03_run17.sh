set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3q; mkdir -p $out
cd $R
timeout -k 10 300 python profiles/experiments/exp_bwd_concurrency_variants.py 2> $out/v.err | tee $out/bwd_concurrency_variants.jsonl; tail -2 $out/v.err
