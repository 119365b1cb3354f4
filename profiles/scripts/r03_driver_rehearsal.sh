# what the driver does at round end, on a fresh box: the GPU suite, smoke(), the contract command
set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3drv${1:-}; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd $R
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > $out/pytest.txt 2>&1; rc=$?; tail -4 $out/pytest.txt; [ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 2 > $out/bench_line.json 2> $out/bench.err; rc=$?; tail -9 $out/bench.err; python3 -c "import json;d=json.load(open('$out/bench_line.json'));print('value',d['value'],'frac',d['roofline']['frac'],'strong',d['sweep_strong']['value'],'thr',d['throughput_mode'].get('value'),'cpu',d['cpu_baseline']['value'])"; exit $rc
