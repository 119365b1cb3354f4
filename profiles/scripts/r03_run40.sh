set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3an; mkdir -p $out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -k "uncapturable or IG_streams or IG_signature" > $out/pytest.txt 2>&1; rc=$?; tail -8 $out/pytest.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --steps 5 --no-cpu-baseline > $out/bench_line.json 2> $out/bench.err; tail -8 $out/bench.err | cut -c1-200
