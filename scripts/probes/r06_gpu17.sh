cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06/kt_early -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-sub-records --no-live-pmc > /dev/null 2> gpurun_out/r06/kt_early.err
python3 scripts/trace_one_step.py gpurun_out/r06/kt_early > gpurun_out/r06/one_step_early.txt 2>&1
tail -45 gpurun_out/r06/one_step_early.txt
