mkdir -p gpurun_out/r06
for i in 1 2 3 4; do
  timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -s -k "frozen or hard_regimes or trained or fused_reference or graphed_iteration or c_caller" 2>&1 | grep -E "^sweep|passed|failed|FAILED" 
done
