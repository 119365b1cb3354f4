set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3b; mkdir -p $out
cd $R
timeout -k 10 500 python profiles/experiments/exp_ig_streams.py deterministic 2> $out/streams_det.err | tee $out/streams_deterministic.jsonl; rc=$?; tail -3 $out/streams_det.err; [ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python profiles/experiments/exp_ig_streams.py finddb 2> $out/streams_fdb.err | tee $out/streams_finddb.jsonl; rc=$?; tail -3 $out/streams_fdb.err; exit $rc
