#!/bin/bash
# DEV-ONLY: what the closing barrier of the timed region costs at the driver's --steps 20: a ONE-rank RCCL group on the
# one GPU of the box (HYDRO_BENCH_FORCE_GROUP=1; a lower bound of what 8 ranks pay), torch.distributed's barrier against
# the node-local shared-memory one, alternating.
export HYDRO_DIST_ALWAYS=1 HYDRO_BENCH_FORCE_GROUP=1 MASTER_ADDR=127.0.0.1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
port=29611
for r in 1 2 3 4; do
for v in dist node none; do
  port=$((port+1)); export MASTER_PORT=$port
  if [ $v = dist ]; then export HYDRO_BARRIER=dist; else unset HYDRO_BARRIER; fi
  if [ $v = none ]; then unset HYDRO_DIST_ALWAYS HYDRO_BENCH_FORCE_GROUP; else export HYDRO_DIST_ALWAYS=1 HYDRO_BENCH_FORCE_GROUP=1; fi
  python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-extras --no-configs --no-live-traffic --no-roofline-4m --no-strong-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$v', d.get('barrier'), round(d['ms_per_step']*1000,3), round(r['kernel_us'],3), round(r['frac'],4))"
done; done
