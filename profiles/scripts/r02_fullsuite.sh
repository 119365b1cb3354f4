set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/full; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
XAI_PARITY_REPORT=$out/parity_deterministic.json timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -16 $out/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout -k 10 600 python bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-300 $out/bench_line.json
