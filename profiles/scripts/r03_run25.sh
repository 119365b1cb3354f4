set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3y; mkdir -p $out
cd $R
timeout -k 10 300 python profiles/experiments/exp_first_call_vs_later.py 2> $out/a.err | tee $out/first_call_cold_box.jsonl | cut -c1-600
echo "--- second process on the same (now warm) box"
timeout -k 10 300 python profiles/experiments/exp_first_call_vs_later.py 2> $out/b.err | tee $out/first_call_warm_box.jsonl | cut -c1-600
