set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3j; mkdir -p $out
cd $R
for n in 3 4; do
timeout -k 10 300 python bench.py --workload sweep --sweep-methods ig --sweep-images 96 --steps 1 --warmup 1 --streams $n > $out/sweep_$n.json 2> $out/sweep_$n.err; echo "rc=$?"; python3 -c "import json;d=json.load(open('$out/sweep_$n.json'));print($n, d['value'], 'images/s')"
done
timeout -k 10 300 python bench.py --lean --steps 5 > $out/bench_lean.json 2> $out/bench_lean.err; python3 -c "import json;d=json.load(open('$out/bench_lean.json'));print(d['value'], d['roofline']['frac'])"
