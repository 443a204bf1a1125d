#!/bin/bash
# DEV-ONLY: the driver's own command (--steps 20 --warmup 5) with the runtime's interrupt-driven host waits (default) and
# with polling waits (HSA_ENABLE_INTERRUPT=0), alternating; prints ms_per_step, the HIP-event figure and both fractions.
for r in 1 2 3; do
for v in default poll; do
  if [ $v = poll ]; then export HSA_ENABLE_INTERRUPT=0; else unset HSA_ENABLE_INTERRUPT; fi
  python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-extras --no-configs --no-live-traffic --no-roofline-4m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$v', round(d['ms_per_step']*1000,3), round(r['kernel_us'],3), round(r['frac'],4), round(r['frac_contract_steps'],4))"
done; done
