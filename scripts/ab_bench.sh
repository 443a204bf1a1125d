set -e
for r in 1 2 3; do
for v in nt wt sc1; do
  HYDRO_LIBRARY=$PWD/scripts/_variants/libvar_$v.so python bench.py --cpu-seconds 0 --no-extras --no-configs --no-live-traffic --no-roofline-4m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step']*1000, d['roofline']['frac'])"
done; done
