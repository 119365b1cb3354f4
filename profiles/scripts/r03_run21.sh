set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3u; mkdir -p $out
cd $R
timeout -k 10 800 python profiles/experiments/exp_sweep_determinism_soak.py 2> $out/soak.err | tee $out/sweep_determinism_soak.jsonl; tail -2 $out/soak.err
