export TMPDIR=/tmp
out=gpurun_out/r2x; mkdir -p $out
export MIOPEN_DEBUG_CONVOLUTION_DETERMINISTIC=1
timeout -k 10 600 python tests/parity_report.py resnet --mode finddb --out $out/resnet_finddb_envdet.json > $out/resnet_finddb_envdet.log 2>&1; echo "resnet rc=$?"; head -48 $out/resnet_finddb_envdet.log | grep -v "^ *$" | tr -s ' ' | head -60
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_finddb_envdet.json 2> $out/bench_finddb_envdet.err; echo "bench rc=$?"; cut -c1-120 $out/bench_finddb_envdet.json
tail -3 $out/bench_finddb_envdet.err
