#!/bin/bash
# Runs on the GPU box (gpurun): the round's evidence for bench.py's headline, into gpurun_out/prof_<tag>/.
#   scripts/profile_round.sh r01d
# 1. un-profiled default bench line; 2. rocprofv3 --kernel-trace --stats of the same command;
# 3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ counters) with kernel-trace only.
# python3 is named directly after `--` (no env / bash -c hop: the profiler's preload initialises the GPU).
set -o pipefail
tag=${1:-r01x}
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
python3 bench.py --extras-out "$out/bench_extras.json" > "$out/bench.json" 2> "$out/bench.err" || exit 1
# which kind of box this is, FIRST: a box that throttles under the combined load (kernel_over_memory_only >= 1.15; ordinary boxes read 1.03-1.06) reads 0.65-0.75
# of the HBM peak where an ordinary one reads 0.77-0.81; collect_profiles.py puts it in the tag's README row
python3 -c "import json,sys; d=json.load(open('$out/bench.json')); b=d.get('box') or {}; print('[box] kernel_over_memory_only', b.get('kernel_over_memory_only'), 'clock_held_ghz', b.get('clock_held_ghz'), 'throttles', b.get('throttles_under_combined_load'), '| frac', d['roofline']['frac'], 'line bytes', len(open('$out/bench.json').read()))"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --cpu-seconds 0 --no-extras --no-configs --no-roofline-4m --no-live-traffic > "$out/bench_stats.json" 2> "$out/stats.err" || exit 1
short="--steps 40 --warmup 8 --spinup-seconds 0.2 --cpu-seconds 0 --no-extras --no-configs --no-roofline-4m --no-live-traffic"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 bench.py $short > "$out/bench_fetch.json" 2> "$out/fetch.err" || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 bench.py $short > "$out/bench_write.json" 2> "$out/write.err" || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/sq" -- python3 bench.py $short > "$out/bench_sq.json" 2> "$out/sq.err" || exit 1
# the second roofline object (bench.py roofline_4m): 4 194 304 bodies, fp16 coefficients, two rotating replicas
big="--bodies 4194304 --scenes 2 --steps 40 --warmup 8 --spinup-seconds 0.2 --cpu-seconds 0 --no-extras --no-configs --no-roofline-4m --no-live-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats4m" -- python3 bench.py --bodies 4194304 --scenes 2 --steps 400 --warmup 40 --cpu-seconds 0 --no-extras --no-configs --no-roofline-4m --no-live-traffic > "$out/bench_stats4m.json" 2> "$out/stats4m.err" || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch4m" -- python3 bench.py $big > "$out/bench_fetch4m.json" 2> "$out/fetch4m.err" || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write4m" -- python3 bench.py $big > "$out/bench_write4m.json" 2> "$out/write4m.err" || exit 1
# every engine kernel in one run: the full default bench (headline + roofline_4m + all extras) under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/allkernels" -- python3 bench.py --cpu-seconds 0 --no-live-traffic --extras-out "$out/bench_allkernels_extras.json" > "$out/bench_allkernels.json" 2> "$out/allkernels.err" || exit 1
find "$out/allkernels" -name "*kernel_trace.csv" -delete
bash scripts/profile_aos.sh "$tag" || exit 1
# auxiliary kernels (round 4): the stand-alone kinetic energy (one launch), the resident closed loop (VALU-bound: SQ counters),
# several scenes in one launch
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/ke1m" -- python3 scripts/run_aux.py ke 1048576 2000 > "$out/aux_ke1m.json" 2> "$out/ke1m.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/ke4m" -- python3 scripts/run_aux.py ke 4194304 1000 > "$out/aux_ke4m.json" 2> "$out/ke4m.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/resident" -- python3 scripts/run_aux.py resident 1048576 40 > "$out/aux_resident.json" 2> "$out/resident.err" || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/resident_sq" -- python3 scripts/run_aux.py resident 1048576 8 > "$out/aux_resident_sq.json" 2> "$out/resident_sq.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/batch" -- python3 scripts/run_aux.py batch 1048576 400 > "$out/aux_batch.json" 2> "$out/batch.err" || exit 1
find "$out/ke1m" "$out/ke4m" "$out/resident" "$out/batch" -name "*kernel_trace.csv" -delete
find "$out/stats4m" -name "*kernel_trace.csv" -size +20M -delete
# keep the merge-back small: the per-dispatch traces of the stats run are large
find "$out/stats" -name "*kernel_trace.csv" -size +20M -delete
ls -R "$out" | head -40
