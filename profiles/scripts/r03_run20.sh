set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3t; mkdir -p $out
cd $R
export OMP_NUM_THREADS=2
run() { # ranks streams tag
  if [ $1 -gt 1 ]; then export XAI_DIST_BACKEND=gloo XAI_FORCE_DEVICE=0; else unset XAI_DIST_BACKEND XAI_FORCE_DEVICE; fi
  timeout -k 10 300 python bench.py --gpus $1 --workload sweep --sweep-images 4 --sweep-methods grad,gc --steps 1 --warmup 0 --deterministic 1 --no-cpu-baseline --streams $2 2> $out/err_$3.txt | grep '^{' > $out/sweep_$3.json || echo "rc=$?"
}
run 1 1 r1s1; run 1 3 r1s3; run 2 1 r2s1a; run 2 1 r2s1b; run 2 3 r2s3a; run 2 3 r2s3b; run 2 2 r2s2
python3 - <<'PY'
import json
o='/root/repo/gpurun_out/r3t/'
runs={k:json.load(open(o+f'sweep_{k}.json'))['metric_means'] for k in ('r1s1','r1s3','r2s1a','r2s1b','r2s3a','r2s3b','r2s2')}
ref=runs['r1s1']
for k,v in runs.items():
    print(k,'max |diff| vs r1s1', {m:max(abs(v[m][key]-ref[m][key]) for key in v[m]) for m in v})
PY
