set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3g; mkdir -p $out
cd $R
timeout -k 10 500 python profiles/experiments/exp_fwd_graph_streams.py 2> $out/fwd.err | tee $out/fwd_graph_streams.jsonl; rc=$?; tail -3 $out/fwd.err; [ $rc -eq 0 ] || exit $rc
export XAI_DIST_BACKEND=gloo XAI_FORCE_DEVICE=0 OMP_NUM_THREADS=2
for cfg in "2 2" "2 3" "3 2"; do set -- $cfg
  timeout -k 10 300 python bench.py --gpus $1 --workload sweep --sweep-methods ig --sweep-images 96 --steps 1 --warmup 1 --streams $2 > $out/sweep_procs_$1_streams_$2.json 2> $out/sweep_procs_$1_$2.err; echo "procs $1 streams $2 rc=$?"; python3 -c "import json;d=json.load(open('$out/sweep_procs_$1_streams_$2.json'));print('procs',$1,'streams',$2, d['value'], 'images/s')"
done
