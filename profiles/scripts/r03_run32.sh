set -o pipefail
R=/root/repo; out=$R/gpurun_out/r3ag; mkdir -p $out
cd $R
export XAI_EXP_TRIALS=20
echo "--- default"; timeout -k 10 200 python profiles/experiments/exp_bwd_concurrency_variants.py 2>/dev/null | grep "one_caller" | tee $out/default.jsonl
echo "--- ROCBLAS_STREAM_ORDER_ALLOC=1"; ROCBLAS_STREAM_ORDER_ALLOC=1 timeout -k 10 200 python profiles/experiments/exp_bwd_concurrency_variants.py 2>/dev/null | grep "one_caller" | tee $out/rocblas_stream_order_alloc.jsonl
echo "--- MIOPEN_DEBUG_CONV_GEMM=0"; MIOPEN_DEBUG_CONV_GEMM=0 timeout -k 10 200 python profiles/experiments/exp_bwd_concurrency_variants.py 2>/dev/null | grep "one_caller" | tee $out/miopen_no_gemm.jsonl
