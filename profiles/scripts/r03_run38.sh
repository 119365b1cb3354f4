set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3am; mkdir -p $out
cd $R/image-classification-xai_amd && timeout -k 10 300 python -m xai_engine.selfcheck 2>&1 | tail -8
cd $R && timeout -k 10 300 python bench.py --lean --steps 3 2>/dev/null | python3 -c "import sys,json;d=json.loads(sys.stdin.read());print(d['value'], d['config']['peak_device_memory_gib_rank0'])"
