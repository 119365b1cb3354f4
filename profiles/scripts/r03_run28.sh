set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3ac; mkdir -p $out
cd $R
timeout -k 10 400 python bench.py --steps 5 --lean > $out/bench_lean.json 2> $out/bench.err; tail -4 $out/bench.err
