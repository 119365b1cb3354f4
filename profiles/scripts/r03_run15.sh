set -o pipefail
export TMPDIR=/tmp
R=/root/repo; out=$R/gpurun_out/r3o; mkdir -p $out
cd $R
XAI_EXP_EAGER=1 timeout -k 10 600 python profiles/experiments/exp_graph_concurrency_bisect.py deterministic 2> $out/eager.err | tee $out/eager_concurrency_all_layers.jsonl | grep -v '"fwd": 0, "bwd": 0' | cut -c1-300; echo "(all-layer eager pass done: lines above are the mismatching modules)"
XAI_EXP_EAGER=1 XAI_EXP_ONLY=layer4.0.conv3 XAI_EXP_TRIALS=200 XAI_EXP_REPS=4 timeout -k 10 600 python profiles/experiments/exp_graph_concurrency_bisect.py deterministic 2>> $out/eager.err | tee $out/eager_concurrency_layer4_0_conv3.jsonl | cut -c1-300
cd /tmp
XAI_EXP_ONLY=layer4.0.conv3 XAI_EXP_TRIALS=2 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $R/profiles/experiments/exp_graph_concurrency_bisect.py deterministic > $out/trace_run.jsonl 2> $out/trace.err
f=$(find $out/trace -name "*kernel_stats.csv" | head -1); cut -c1-160 $f | head -20; cp $f $out/layer4_0_conv3_kernel_stats.csv; rm -rf $out/trace
