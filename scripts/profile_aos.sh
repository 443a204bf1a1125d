#!/bin/bash
# Runs on the GPU box (called by scripts/profile_round.sh; can be run alone): kernel time and HBM-side bytes of the
# simulator-facing entry, into gpurun_out/prof_<tag>/aos_*.      scripts/profile_aos.sh r03a
set -o pipefail
tag=${1:-r01x}
out=gpurun_out/prof_$tag
mkdir -p "$out"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
# the simulator-facing entry (hydro_step_wrench_aos, fp32 parameters) at 1 048 576 and 4 194 304 bodies: kernel time + HBM-side bytes
aos1="--layout aos --workload c5-f32 --steps 400 --warmup 40 --cpu-seconds 0 --no-extras --no-configs --no-roofline-4m --no-live-traffic"
aos1s="--layout aos --workload c5-f32 --steps 40 --warmup 8 --spinup-seconds 0.2 --cpu-seconds 0 --no-extras --no-configs --no-roofline-4m --no-live-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/aos_stats" -- python3 bench.py $aos1 > "$out/bench_aos_stats.json" 2> "$out/aos_stats.err" || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/aos_fetch" -- python3 bench.py $aos1s > "$out/bench_aos_fetch.json" 2> "$out/aos_fetch.err" || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/aos_write" -- python3 bench.py $aos1s > "$out/bench_aos_write.json" 2> "$out/aos_write.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/aos_stats4m" -- python3 bench.py --bodies 4194304 --scenes 2 $aos1 > "$out/bench_aos_stats4m.json" 2> "$out/aos_stats4m.err" || exit 1
find "$out/aos_stats" "$out/aos_stats4m" -name "*kernel_trace.csv" -size +20M -delete
