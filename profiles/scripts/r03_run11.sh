set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3k; mkdir -p $out
cd $R
timeout -k 10 500 python profiles/experiments/exp_ig_graph_streams.py deterministic own 2> $out/gs.err | tee $out/graph_streams_own.jsonl | grep '"flow": "graph"' | cut -c1-330; tail -2 $out/gs.err
