export TMPDIR=/tmp
out=gpurun_out/r2ai; mkdir -p $out
( while true; do echo "[heartbeat $(date +%T)]"; sleep 60; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
for ipp in 1 2 4 8; do
  timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --images-per-pass $ipp > $out/bench_ipp$ipp.json 2> $out/bench_ipp$ipp.err; echo "ipp $ipp rc=$?"
  python3 -c "import json; d=json.load(open('$out/bench_ipp$ipp.json')); print($ipp, round(d['value'],2), round(d['unfused_classifier']['value'],2), round(d['roofline']['frac'],3))"
done
