mkdir -p gpurun_out/r06
python bench.py > gpurun_out/r06/r06_bench.json 2> gpurun_out/r06/bench.err
tail -c 400 gpurun_out/r06/r06_bench.json
python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -16 > gpurun_out/r06/r06_gpu_test_run.txt
tail -4 gpurun_out/r06/r06_gpu_test_run.txt
