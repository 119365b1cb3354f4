set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3m; mkdir -p $out
cd $R
XAI_EXP_PLAIN=1 XAI_EXP_IPP=1 timeout -k 10 500 python profiles/experiments/exp_ig_graph_streams.py deterministic own 2> $out/gs.err | tee $out/graph_streams_plain_model.jsonl | grep '"flow": "graph"' | cut -c1-330; tail -2 $out/gs.err
