export TMPDIR=/tmp
out=gpurun_out/r2ab; mkdir -p $out
export XAI_DIST_BACKEND=gloo XAI_FORCE_DEVICE=0 OMP_NUM_THREADS=2
for n in 4; do
  timeout -k 10 400 python bench.py --gpus $n --steps 1 --warmup 1 --images 2 --no-cpu-baseline --miopen-db 0 > $out/bench_ig_$n.json 2> $out/bench_ig_$n.err; echo "ig $n rc=$?"; cut -c1-260 $out/bench_ig_$n.json
  timeout -k 10 400 python bench.py --gpus $n --workload sweep --sweep-images 6 --sweep-methods grad,gc --steps 1 --warmup 0 --deterministic 1 --no-cpu-baseline > $out/bench_sweep_$n.json 2> $out/bench_sweep_$n.err; echo "sweep $n rc=$?"; cut -c1-260 $out/bench_sweep_$n.json
done
tail -3 $out/bench_ig_4.err
