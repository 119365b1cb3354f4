export TMPDIR=/tmp
out=gpurun_out/r2ag; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "stamped or accum" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $out/pytest.log
timeout -k 10 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r2ag/bench.json"))
print(d["value"], json.dumps(d["roofline"], indent=1))
PY
