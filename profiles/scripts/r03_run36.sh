set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3ak; mkdir -p $out
cd $R
timeout -k 10 400 python bench.py --mode throughput --lean --steps 5 > $out/thr.json 2> $out/thr.err; tail -3 $out/thr.err | cut -c1-250
