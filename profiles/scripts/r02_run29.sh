export TMPDIR=/tmp
out=gpurun_out/r2ak; mkdir -p $out
timeout -k 10 500 python profiles/experiments/exp_sweep_batch.py > $out/exp_sweep_batch.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $out/exp_sweep_batch.txt
