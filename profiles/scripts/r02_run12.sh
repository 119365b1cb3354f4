export TMPDIR=/tmp
out=gpurun_out/r2t; mkdir -p $out
XAI_PARITY_REPORT=$out/parity_vit.json timeout -k 10 600 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 -k "vit or VIT or ViT or TIS or embeddings or config4" > $out/pytest_vit.log 2>&1; echo "pytest rc=$?"; tail -12 $out/pytest_vit.log
timeout -k 10 600 python bench_configs.py --configs 4,6 > $out/configs46.jsonl 2> $out/configs46.err; echo "configs rc=$?"; cut -c1-500 $out/configs46.jsonl
