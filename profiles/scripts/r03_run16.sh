set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3p; mkdir -p $out
cd $R
timeout -k 10 300 python profiles/experiments/exp_handle_per_thread.py 2> $out/hpt.err | tee $out/handle_per_thread.jsonl; tail -2 $out/hpt.err
timeout -k 10 500 python profiles/experiments/exp_ig_streams.py deterministic 2> $out/streams_det.err | tee $out/streams_threads_deterministic.jsonl | cut -c1-250; tail -3 $out/streams_det.err
