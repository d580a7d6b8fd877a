mkdir -p gpurun_out/r06
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r06/gpu_test_run_new.txt
cat gpurun_out/r06/gpu_test_run_new.txt
