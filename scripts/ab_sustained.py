#!/usr/bin/env python3
"""DEV-ONLY: SUSTAINED rate of whole-library variants (scripts/_variants/libvar_<name>.so): every variant steps the C5
headline (1 048 576 bodies, fp16 coefficients, 4 rotating replicas) back to back for `seconds`, and the mean step time of
each tenth of that span is printed - boxes whose power management throttles under sustained combined load show it as a
drift within the span.  Variants run one after the other with an idle pause in between, twice, in alternating order.
    python scripts/ab_sustained.py name1 name2 ...      -> gpurun_out/ab_sustained.log"""
import ctypes, os, statistics, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
from silver2_isaacsim_amd import _native as nat
from silver2_isaacsim_amd.engine import HydroEngine
import bench

names = sys.argv[1:]
SECONDS = float(os.environ.get("HYDRO_AB_SECONDS", "2.0"))
dev = torch.device("cuda:0"); stream = torch.cuda.Stream(dev)
LOG = open(os.path.join(REPO, "gpurun_out", "ab_sustained.log"), "a")
def say(s):
    print(s, flush=True); LOG.write(s + "\n"); LOG.flush()

sc = bench.build_scene("c5", 1048576, 11)
reps = {}
full = dict(nat.SIGNATURES)
for nm in names:
    raw = ctypes.CDLL(os.path.join(REPO, "scripts", "_variants", f"libvar_{nm}.so"))
    for k in [k for k in full if not hasattr(raw, k)]:
        del nat.SIGNATURES[k]
    nat._lib = nat.load(os.path.join(REPO, "scripts", "_variants", f"libvar_{nm}.so"))
    nat.SIGNATURES.update(full)
    reps[nm] = [bench.Replica(sc, "f16", dev, roll=7919 * k) for k in range(4)]
    with torch.cuda.stream(stream):
        for r in reps[nm]: r.step()
    stream.synchronize()

def sustained(nm):
    R = reps[nm]; marks = []
    with torch.cuda.stream(stream):
        t_end = time.perf_counter() + SECONDS
        k = 0
        while time.perf_counter() < t_end:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(400):
                R[k % 4].step(); k += 1
            e1.record(stream)
            marks.append((e0, e1))
            if len(marks) % 8 == 0:
                marks[-8][1].synchronize()                       # keep the host at most ~8 chunks ahead
        stream.synchronize()
    us = [a.elapsed_time(b) * 1e3 / 400 for a, b in marks]
    tenth = max(1, len(us) // 10)
    return [statistics.mean(us[i:i + tenth]) for i in range(0, tenth * 10, tenth)], statistics.mean(us[len(us) // 2:])

for order in (names, names[::-1]):
    for nm in order:
        time.sleep(1.5)                                          # idle: every variant starts from the same cooled-down state
        prof, late = sustained(nm)
        say(f"{nm:10s} sustained {SECONDS:.0f} s: second half {late:6.2f} us/step   tenths: " + " ".join(f"{x:.1f}" for x in prof))
