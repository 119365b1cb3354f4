set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3aa; mkdir -p $out
cd $R
export OMP_NUM_THREADS=2
FUSE=${1:-1}; STREAMS=${2:-3}
run() { # tag
  timeout -k 10 300 python bench.py --gpus 1 --workload sweep --sweep-images 4 --sweep-methods grad,gc --steps 1 --warmup 0 --deterministic 1 --no-cpu-baseline --streams $STREAMS --fuse-bn-relu $FUSE 2> $out/err_$1.txt | grep '^{' > $out/sweep_$1.json || echo "rc=$?"
}
run cold; run warm1; run warm2
python3 - <<'PY'
import json
o='/root/repo/gpurun_out/r3aa/'
runs={k:json.load(open(o+f'sweep_{k}.json'))['metric_means'] for k in ('cold','warm1','warm2')}
ref=runs['warm2']
for k,v in runs.items():
    print(k,'max |diff| vs warm2', {m:max(abs(v[m][key]-ref[m][key]) for key in v[m]) for m in v})
PY
