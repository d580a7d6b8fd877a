set -u
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -x -q 2>&1 | tail -25 > $O/t_r6.log
cat $O/t_r6.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python3 scripts/trace_one_step.py $O/kt > $O/one_step.txt 2>&1
rm -rf $O/kt
cat $O/one_step.txt | tail -60
