set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3s; mkdir -p $out
cd $R
for s in 1 3; do for rep in a b; do
timeout -k 10 200 python bench.py --workload sweep --sweep-images 4 --sweep-methods grad,gc --steps 1 --warmup 0 --deterministic 1 --no-cpu-baseline --streams $s > $out/sweep_s${s}_$rep.json 2> $out/err_s${s}_$rep.txt || echo "rc=$?"
done; done
python3 - <<'PY'
import json
o='/root/repo/gpurun_out/r3s/'
runs={k:json.load(open(o+f'sweep_{k}.json'))['metric_means'] for k in ('s1_a','s1_b','s3_a','s3_b')}
ref=runs['s1_a']
for k,v in runs.items():
    worst=max(abs(v[m][key]-ref[m][key]) for m in v for key in v[m])
    print(k,'max |diff| vs s1_a', worst, {m:max(abs(v[m][key]-ref[m][key]) for key in v[m]) for m in v})
PY
