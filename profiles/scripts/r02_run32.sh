export TMPDIR=/tmp
out=gpurun_out/r2an; mkdir -p $out
for m in deterministic finddb; do timeout -k 10 300 python profiles/experiments/exp_cold_start.py $m 2>&1 | grep -v amdgpu.ids | tee -a $out/cold_start.txt; done
