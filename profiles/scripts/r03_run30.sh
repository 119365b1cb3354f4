set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3ae; mkdir -p $out
cd $R
for s in 1 3; do
timeout -k 10 300 python bench_configs.py --configs 3 --check 0 --streams $s 2> $out/c3_s$s.err | cut -c1-200
timeout -k 10 300 python bench_configs.py --configs 3 --check 0 --streams $s --deterministic 1 2> $out/c3d_s$s.err | cut -c1-200
done
tail -3 $out/c3_s3.err
