mkdir -p gpurun_out/r06
for i in $(seq 1 12); do
  RDG_TRAINED_TEST_DETERMINISTIC=$((i % 2)) timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -k "trained" > gpurun_out/r06/trained_$i.log 2>&1
  if grep -q "1 passed" gpurun_out/r06/trained_$i.log; then echo "$i passed"; rm gpurun_out/r06/trained_$i.log; else echo "$i FAILED"; grep -E "violations|AssertionError|assert " gpurun_out/r06/trained_$i.log | cut -c1-2500; fi
done
