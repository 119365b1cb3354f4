export TMPDIR=/tmp
out=gpurun_out/r2y; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "resnext or full_size_rise or cli" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $out/pytest.log
mkdir -p /tmp/val && python - <<'PY'
import numpy as np
from PIL import Image
rng = np.random.default_rng(3)
for i in range(1, 5):
    Image.fromarray(rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)).save(f"/tmp/val/ILSVRC2012_val_{i:08d}.JPEG", format="PNG")
PY
cd image-classification-xai_amd
for m in R101 RNXT VIT32; do
  a=ig; [ $m = VIT32 ] && a=rollout
  timeout -k 10 300 python -m xai_engine.evaluate_perturbation --model $m --attr_func $a --image_count 2 --cuda_num 0 --dataset_path /tmp/val --out_dir /tmp/res > ../$out/cli_$m.log 2>&1; echo "cli $m rc=$?"; tail -1 ../$out/cli_$m.log | cut -c1-200
done
ls /tmp/res/*/ 2>/dev/null
