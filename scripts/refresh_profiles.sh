#!/bin/bash
# Regenerates the measurement evidence under gpurun_out/prof_<tag>/ on the GPU box (copy what is to be judged into
# profiles/): bench line, rocprofv3 kernel stats of the same command, one-step kernel timeline, PMC passes, the two
# extra scenes, the launch-bound regime with and without the captured graph, the long-list probe.
#   usage: scripts/refresh_profiles.sh <tag>
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD
T=${1:-r06}
O=$R/gpurun_out/prof_$T
mkdir -p $O
# the PMC passes first: bench.py reads profiles/${T}_pmc_*.json (stamped with the hash of the kernel sources) for `roofline.traffic`
# and the VALU-issue figures, so they have to exist (in the box's copy of profiles/) before the headline line is printed
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err
python3 scripts/pmc_hbm_traffic.py $O/pmc_fetch $O/pmc_write $O/${T}_pmc_hbm_traffic.json 1000000 1920 1080 > $O/pmc_traffic.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pmc_sq.err
python3 scripts/pmc_summary.py $O/pmc_sq $O/${T}_pmc_sq_counters.json 1000000 1920 1080 > /dev/null 2>&1
cp $O/${T}_pmc_hbm_traffic.json $O/${T}_pmc_sq_counters.json $R/profiles/ 2>/dev/null
python3 bench.py > $O/${T}_bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_bench_under_rocprof.json 2> $O/kt.err
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/${T}_bench_kernel_stats.csv
python3 scripts/trace_one_step.py $O/kt > $O/${T}_one_step.txt 2>&1
# the train LOOP (densify_and_prune every 100 steps), eager / per-phase profile / graph re-captured per densification, and the
# reference's iteration (static + dynamic sub-steps over the two-segment cloud)
python3 bench.py --loop 600 > $O/${T}_loop_bench.json 2>/dev/null
python3 bench.py --loop 2500 > $O/${T}_loop_long_bench.json 2>/dev/null
python3 bench.py --loop 600 --loop-profile > $O/${T}_loop_profile.json 2>/dev/null
python3 bench.py --loop 600 --graph > $O/${T}_loop_graph_bench.json 2>/dev/null
python3 bench.py --loop 600 --points 100000 > $O/${T}_loop_100k_bench.json 2>/dev/null
python3 bench.py --loop 600 --points 100000 --graph > $O/${T}_loop_100k_graph_bench.json 2>/dev/null
# fixed capacity (dead rows, densification in place): eager, and ONE captured graph for the whole loop (re-captured only when the
# capacity itself is outgrown: the 2 500-step runs double the cloud)
python3 bench.py --loop 600 --fixed-capacity 1.2 > $O/${T}_loop_fixed_bench.json 2>/dev/null
python3 bench.py --loop 600 --fixed-capacity 1.2 --graph > $O/${T}_loop_fixed_graph_bench.json 2>/dev/null
python3 bench.py --loop 2500 --fixed-capacity 1.3 --graph > $O/${T}_loop_fixed_graph_long_bench.json 2>/dev/null
python3 bench.py --loop 600 --points 100000 --fixed-capacity 1.2 > $O/${T}_loop_100k_fixed_bench.json 2>/dev/null
python3 bench.py --loop 600 --points 100000 --fixed-capacity 1.2 --graph > $O/${T}_loop_100k_fixed_graph_bench.json 2>/dev/null
python3 bench.py --loop 2500 --points 100000 --fixed-capacity 1.3 --graph > $O/${T}_loop_100k_fixed_graph_long_bench.json 2>/dev/null
python3 bench.py --loop 2500 --points 100000 > $O/${T}_loop_100k_long_bench.json 2>/dev/null
python3 bench.py --iteration reference > $O/${T}_reference_iteration_bench.json 2>/dev/null
python3 bench.py --iteration reference --points 200000 > $O/${T}_reference_iteration_200k_bench.json 2>/dev/null
python3 bench.py --iteration reference --points 200000 --graph > $O/${T}_reference_iteration_200k_graph_bench.json 2>/dev/null
python3 bench.py --iteration reference-v1 > $O/${T}_reference_iteration_v1_bench.json 2>/dev/null
python3 bench.py --iteration reference-v1 --points 200000 > $O/${T}_reference_iteration_v1_200k_bench.json 2>/dev/null
# the tight tile rectangles against the reference's (RdgRasterSettings.cull), alternating on this box
for i in 1 2; do
  RDG_CULL=1 python3 bench.py --steps 60 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=b['stage_ms']; print('cull 1', 'step %.4f ms' % b['ms_per_step'], 'D', b['config']['D'], 'D_composited', b['config']['D_composited'], 'binning %.1f us' % (1e3*(s['scan_dup']+s['sort']+s['ranges'])), 'render_fwd %.1f' % (1e3*s['render_fwd']), 'render_bwd %.1f' % (1e3*s['render_bwd']))"
  RDG_CULL=0 python3 bench.py --steps 60 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=b['stage_ms']; print('cull 0', 'step %.4f ms' % b['ms_per_step'], 'D', b['config']['D'], 'D_composited', b['config']['D_composited'], 'binning %.1f us' % (1e3*(s['scan_dup']+s['sort']+s['ranges'])), 'render_fwd %.1f' % (1e3*s['render_fwd']), 'render_bwd %.1f' % (1e3*s['render_bwd']))"
done > $O/${T}_cull_ab.txt
# the randomised sweeps on seeds of their own (the -m gpu test runs 300 + 300 on 610000 / 620000 under frozen rules)
python3 scripts/parity_sweep.py 1500 630000 > $O/${T}_parity_sweep_regular.txt 2>&1
RDG_SWEEP_PROFILE=aniso python3 scripts/parity_sweep.py 1500 640000 > $O/${T}_parity_sweep_aniso.txt 2>&1
# the two scenes the survey's generator never enters (not the headline): bench line + kernel table each
for SC in sheets dense; do
  python3 bench.py --scene $SC --no-cpu-baseline > $O/${T}_scene_${SC}_bench.json 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$SC -- python3 bench.py --scene $SC --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/kt_$SC.err
  cp $(find $O/kt_$SC -name "*kernel_stats.csv" | head -1) $O/${T}_scene_${SC}_kernel_stats.csv
  rm -rf $O/kt_$SC
done
# launch-bound regime (the reference's real clouds are ~100 k points): eager vs one captured graph per step
python3 bench.py --points 100000 --no-cpu-baseline > $O/${T}_photometric_100k_bench.json 2>/dev/null
python3 bench.py --points 100000 --no-cpu-baseline --graph > $O/${T}_photometric_100k_graph_bench.json 2>/dev/null
python3 bench.py --no-cpu-baseline --graph > $O/${T}_graph_bench.json 2>/dev/null
# the config-5 loss set under the graph (four replayed steps between two eager rigidity steps), at the reference's real cloud size
python3 bench.py --points 100000 --full-losses --no-cpu-baseline > $O/${T}_full_losses_100k_bench.json 2>/dev/null
python3 bench.py --points 100000 --full-losses --no-cpu-baseline --graph > $O/${T}_full_losses_100k_graph_bench.json 2>/dev/null
# the step on a cloud that went through one densify-and-prune (not a BASELINE config)
python3 bench.py --no-cpu-baseline --densify-first > $O/${T}_densified_bench.json 2>/dev/null
# the radix binning forced on the headline frame (the default picks bucket binning there)
RDG_BIN_MODE=radix python3 bench.py --no-cpu-baseline > $O/${T}_radix_binning_bench.json 2>/dev/null
python3 bench.py --points 4000000 --width 3840 --height 2160 --steps 10 --warmup 5 --gt-frames 4 --no-cpu-baseline > $O/${T}_photometric_4m_4k_bench.json 2>/dev/null
python3 bench.py --full-losses --no-cpu-baseline > $O/${T}_full_losses_bench.json 2>/dev/null
python3 bench.py --full-losses --points 4000000 --width 3840 --height 2160 --steps 10 --warmup 5 --gt-frames 4 --no-cpu-baseline > $O/${T}_full_losses_4m_4k_bench.json 2>/dev/null
RDG_DETERMINISTIC=1 python3 bench.py --no-cpu-baseline > $O/${T}_deterministic_bench.json 2>/dev/null
# kernel tables of the config-5 loss set (every 5th step is a rigidity step)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_fl -- python3 bench.py --full-losses --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/kt_fl.err
cp $(find $O/kt_fl -name "*kernel_stats.csv" | head -1) $O/${T}_full_losses_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_fl4 -- python3 bench.py --full-losses --points 4000000 --width 3840 --height 2160 --steps 10 --warmup 5 --gt-frames 4 --no-cpu-baseline > /dev/null 2> $O/kt_fl4.err
cp $(find $O/kt_fl4 -name "*kernel_stats.csv" | head -1) $O/${T}_full_losses_4m_4k_kernel_stats.csv
rm -rf $O/kt_fl $O/kt_fl4
# the two binning algorithms on the three scenes (stage timers of the bench line), one box
{
  for SC in uniform sheets dense; do
    for M in bucket radix; do
      RDG_BIN_MODE=$M python3 bench.py --scene $SC --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=b['stage_ms']; print('$SC', '$M', 'binning %.1f us' % (1e3*(s.get('scan_dup',0)+s.get('sort',0)+s.get('ranges',0))), 'step %.4f ms' % b['ms_per_step'])"
    done
  done
} > $O/${T}_binning_modes_raw.txt
# probes of the rigidity step's kernels, the neighbour search and the sort (n = 2 M and 500 k: config-5 sample sizes)
{
  python3 scripts/rig_rows_probe.py 2000000; python3 scripts/rig_rows_probe.py 500000
  python3 scripts/knn_probe.py 2000000; python3 scripts/knn_probe.py 500000
  python3 scripts/sort_probe.py 17000000 16; python3 scripts/sort_probe.py 2000000 48
} > $O/${T}_rigidity_probes.txt 2>/dev/null
python3 scripts/heavy_tile_probe.py > $O/${T}_heavy_tile.json 2> $O/heavy.err
rm -rf $O/kt $O/pmc_fetch $O/pmc_write $O/pmc_sq
ls -la $O
tail -c 300 $O/${T}_bench.json
