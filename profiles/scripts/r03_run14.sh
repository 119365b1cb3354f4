set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3n; mkdir -p $out
cd $R
timeout -k 10 600 python profiles/experiments/exp_graph_concurrency_bisect.py deterministic 2> $out/bisect.err | tee $out/graph_concurrency_bisect.jsonl | cut -c1-260; tail -3 $out/bisect.err
