set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3ah; mkdir -p $out
cd $R
timeout -k 10 400 python bench_configs.py --configs 4 --check 0 --deterministic 1 2> $out/c4d.err | cut -c1-600; tail -3 $out/c4d.err
timeout -k 10 400 python bench_configs.py --configs 4 --check 0 2> $out/c4.err | cut -c1-600
