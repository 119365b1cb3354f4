export TMPDIR=/tmp
out=gpurun_out/r2aj; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
timeout -k 10 600 python bench.py --workload sweep --sweep-images 96 --sweep-methods ig --steps 1 --warmup 1 --no-cpu-baseline > $out/sweep_ig_96.json 2> $out/sweep_ig_96.err; echo "rc=$?"; cut -c1-260 $out/sweep_ig_96.json
timeout -k 10 600 python bench.py --workload sweep --sweep-images 96 --sweep-methods ig --steps 1 --warmup 1 --no-cpu-baseline --deterministic 1 > $out/sweep_ig_96_det.json 2>> $out/sweep_ig_96.err; echo "rc=$?"; cut -c1-260 $out/sweep_ig_96_det.json
timeout -k 10 600 python bench.py --workload sweep --sweep-images 96 --sweep-methods gc --steps 1 --warmup 1 --no-cpu-baseline > $out/sweep_gc_96.json 2>> $out/sweep_ig_96.err; echo "rc=$?"; cut -c1-260 $out/sweep_gc_96.json
