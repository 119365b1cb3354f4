set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3c; mkdir -p $out
cd $R
timeout -k 10 500 python profiles/experiments/exp_ig_graph_streams.py deterministic 2> $out/gs_det.err | tee $out/graph_streams_deterministic.jsonl; rc=$?; tail -5 $out/gs_det.err; [ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python profiles/experiments/exp_ig_graph_streams.py finddb 2> $out/gs_fdb.err | tee $out/graph_streams_finddb.jsonl; rc=$?; tail -5 $out/gs_fdb.err; exit $rc
