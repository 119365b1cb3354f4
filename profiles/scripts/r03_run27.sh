set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3ab; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd $R
timeout -k 10 500 python profiles/experiments/exp_ig_streams.py deterministic 2> $out/streams_det.err | tee $out/streams_workers_graphs_deterministic.jsonl | cut -c1-220; tail -3 $out/streams_det.err
timeout -k 10 400 python bench.py --steps 5 --no-cpu-baseline > $out/bench_line.json 2> $out/bench.err; tail -9 $out/bench.err
