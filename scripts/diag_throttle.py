#!/usr/bin/env python3
"""DEV-ONLY: what helps on a box that throttles under sustained combined load?  First the product kernel is stepped
back to back for 2 s; if its second half runs under 23.5 us per C5 step the box is an ordinary one and the script stops.
Otherwise every variant (library@tuning, e.g. diet3w4@w3 = waves cap 3, @b128 = 128-thread blocks, @nt0 = temporal accesses)
gets the same 2 s, in alternating order, plus the in-kernel clocks.      -> gpurun_out/diag_throttle.log"""
import ctypes, os, re, statistics, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch
from silver2_isaacsim_amd import _native as nat
import bench

names = sys.argv[1:] or ["diet3w4", "base", "diet3", "diet3w4@w3", "diet3w4@b128", "diet3w4@nt0"]
dev = torch.device("cuda:0"); stream = torch.cuda.Stream(dev)
LOG = open(os.path.join(REPO, "gpurun_out", "diag_throttle.log"), "a")
def say(s):
    print(s, flush=True); LOG.write(s + "\n"); LOG.flush()

sc = bench.build_scene("c5", 1048576, 11)
full = dict(nat.SIGNATURES)
def replicas(nm):
    libname, _, opt = nm.partition("@")
    raw = ctypes.CDLL(os.path.join(REPO, "scripts", "_variants", f"libvar_{libname}.so"))
    for k in [k for k in full if not hasattr(raw, k)]:
        del nat.SIGNATURES[k]
    nat._lib = nat.load(os.path.join(REPO, "scripts", "_variants", f"libvar_{libname}.so"))
    nat.SIGNATURES.update(full)
    R = [bench.Replica(sc, "f16", dev, roll=7919 * k) for k in range(4)]
    if opt:
        w = re.search(r"w(\d)", opt); b = re.search(r"b(\d+)", opt); nt = re.search(r"nt(\d)", opt)
        for r in R:
            r.engine.set_tuning(0, int(b.group(1)) if b else 0, int(nt.group(1)) if nt else -1, int(w.group(1)) if w else -1)
    with torch.cuda.stream(stream):
        for r in R: r.step()
    stream.synchronize()
    return R

def sustained(R, seconds=2.0):
    marks = []
    with torch.cuda.stream(stream):
        t_end = time.perf_counter() + seconds; k = 0
        while time.perf_counter() < t_end:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(400):
                R[k % 4].step(); k += 1
            e1.record(stream); marks.append((e0, e1))
            if len(marks) % 8 == 0:
                marks[-8][1].synchronize()
        stream.synchronize()
    us = [a.elapsed_time(b) * 1e3 / 400 for a, b in marks]
    tenth = max(1, len(us) // 10)
    return [statistics.mean(us[i:i + tenth]) for i in range(0, tenth * 10, tenth)], statistics.mean(us[len(us) // 2:])

reps = {names[0]: replicas(names[0])}
prof, late = sustained(reps[names[0]])
say(f"[box check] {names[0]}: second half {late:.2f} us/step  tenths " + " ".join(f"{x:.1f}" for x in prof))
if late < 23.5 and os.environ.get("HYDRO_THROTTLE_ANYWAY") != "1":
    say("ordinary box: nothing to learn here"); sys.exit(0)
say("THROTTLING BOX - running the variants")
for nm in names[1:]:
    reps[nm] = replicas(nm)
for order in (names, names[::-1]):
    for nm in order:
        time.sleep(1.0)
        prof, late = sustained(reps[nm])
        say(f"{nm:16s}: second half {late:6.2f} us/step   tenths: " + " ".join(f"{x:.1f}" for x in prof))
from scripts import probes
c = probes.clock_probes(1048576, dev, stream, seconds=1.5)
say("clocks: " + "  ".join(f"{k} {v['ghz']:.3f} GHz" for k, v in c.items()))
