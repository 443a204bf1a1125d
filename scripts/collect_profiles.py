#!/usr/bin/env python3
"""Container side: turn gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into the committed summaries under
profiles/ and refresh profiles/traffic.json.      python scripts/collect_profiles.py r01d"""
import csv, glob, json, os, shutil, statistics, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(REPO, "gpurun_out", f"prof_{tag}")
dst = os.path.join(REPO, "profiles")
KERNEL = "wrench_tiled_kernel"


def one(pattern):
    """newest match: gpurun merges a new run's files NEXT TO an older run's (different pid prefixes)"""
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    if not hits:
        raise SystemExit(f"missing {pattern}")
    return hits[-1]


shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
if os.path.exists(os.path.join(src, "bench_extras.json")):          # the side file (round 6: extras no longer ride on the line)
    shutil.copy(os.path.join(src, "bench_extras.json"), os.path.join(dst, f"{tag}_bench_extras.json"))
shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_c5_kernel_stats.csv"))
shutil.copy(one("stats/**/*domain_stats.csv"), os.path.join(dst, f"{tag}_c5_domain_stats.csv"))


def pmc(sub, label):
    """per-dispatch counter values of the wrench kernel -> profiles/<tag>_c5_pmc_<label>.csv; returns {counter: [values]}"""
    rows = []
    for path in [one(os.path.join(sub, "**", "*counter_collection.csv"))]:
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if KERNEL in r["Kernel_Name"]:
                    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0].replace(", ", ",")
                    rows.append((int(r["Dispatch_Id"]), name, r["Counter_Name"], float(r["Counter_Value"])))
    rows.sort()
    with open(os.path.join(dst, f"{tag}_c5_pmc_{label}.csv"), "w", newline="") as f:
        w = csv.writer(f); w.writerow(["dispatch_id", "kernel", "counter", "value"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.6f}"])
    out = {}
    for _, _, c, v in rows:
        out.setdefault(c, []).append(v)
    return out


fetch = pmc("fetch", "FETCH_SIZE")["FETCH_SIZE"]
write = pmc("write", "WRITE_SIZE")["WRITE_SIZE"]
sq = pmc("sq", "SQ")
bench = json.load(open(os.path.join(src, "bench.json")))
n = bench["config"]["bodies_per_gpu"]
fetch_raw = statistics.median(fetch) * 1024.0          # FETCH_SIZE / WRITE_SIZE are in KB
write_b = statistics.median(write) * 1024.0
tiles = (n + 63) // 64
expected_read = tiles * 64 * (11 * 4 + 6 * 4) + tiles * 1920          # state minus px,py + prev + fp16 parameter record
# the timed region of the stats run = its last `steps` launches of the wrench kernel
trace = one("stats/**/*kernel_trace.csv") if glob.glob(os.path.join(src, "stats", "**", "*kernel_trace.csv"), recursive=True) else None
mean_us = None
if trace:
    d = []
    with open(trace, newline="") as f:
        for r in csv.DictReader(f):
            if KERNEL in r["Kernel_Name"]:
                d.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    d.sort()
    steps = json.load(open(os.path.join(src, "bench_stats.json")))["steps"]
    mean_us = statistics.mean(x[1] for x in d[-steps:]) / 1e3
path = os.path.join(dst, "traffic.json")
tr = json.load(open(path))
prev = tr.get("c5:tiled") or {}
if prev.get("kernel_trace_timed_region_mean_us") and "c5:tiled_timings" not in tr:      # carry the last pre-round-6 record over
    tr["c5:tiled_timings"] = {"by_tag": {"r05b": {"kernel_trace_timed_region_mean_us": prev["kernel_trace_timed_region_mean_us"],
                                                   "in_bench_event_us_unprofiled": prev.get("in_bench_event_us", {}).get("unprofiled")}}}
tr["c5:tiled"] = {
    "hbm_bytes_per_launch": 2.0 * fetch_raw + write_b,
    "fetch_size_bytes_raw": fetch_raw, "fetch_size_bytes_corrected": 2.0 * fetch_raw, "write_size_bytes": write_b,
    "expected_read_bytes": expected_read, "expected_write_bytes": tiles * 64 * 24, "algorithmic_bytes": n * 130,
    "source": f"profiles/{tag}_c5_pmc_FETCH_SIZE.csv + {tag}_c5_pmc_WRITE_SIZE.csv (rocprofv3 --pmc, separate passes, "
              f"scripts/profile_round.sh; FETCH_SIZE x2 per MI355X_MICROARCH.md HBM section; 2 x FETCH vs the known read "
              f"bytes of this access pattern: {100.0 * (2.0 * fetch_raw / expected_read - 1.0):+.2f}%)",
    "note": "read side is 98 B/body, not 106: px and py are provably unused by the wrench; their two 256-B runs of every state tile are skipped",
    "sq_per_launch_median": {k: statistics.median(v) for k, v in sorted(sq.items())},
    "kernel_trace_timed_region_mean_us": mean_us,
    "in_bench_event_us": {"unprofiled": bench["roofline"]["kernel_us"],
                          "under_kernel_trace": json.load(open(os.path.join(src, "bench_stats.json")))["roofline"]["kernel_us"]},
    "tag": tag,
}
# timings are a property of the BOX a tag ran on (DVFS): kept per tag, never overwritten by the next round's run
by_tag = (tr_old.get("by_tag") or {}) if (tr_old := tr.get("c5:tiled_timings")) else {}
box = bench.get("box") or {}
by_tag[tag] = {"kernel_trace_timed_region_mean_us": mean_us, "in_bench_event_us_unprofiled": bench["roofline"]["kernel_us"],
               "frac": bench["roofline"]["frac"], "kernel_over_memory_only": box.get("kernel_over_memory_only"),
               "clock_held_ghz": box.get("clock_held_ghz"), "throttles_under_combined_load": box.get("throttles_under_combined_load")}
tr["c5:tiled_timings"] = {"by_tag": by_tag}
# second roofline object: 4 194 304 bodies, fp16 coefficients
if os.path.isdir(os.path.join(src, "fetch4m")):
    shutil.copy(one("stats4m/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_f16_4m_kernel_stats.csv"))

    def med(sub, counter):
        vals = []
        with open(one(os.path.join(sub, "**", "*counter_collection.csv")), newline="") as f:
            for r in csv.DictReader(f):
                if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    vals.append(float(r["Counter_Value"]))
        return statistics.median(vals) * 1024.0
    f4, w4 = med("fetch4m", "FETCH_SIZE"), med("write4m", "WRITE_SIZE")
    b4 = json.load(open(os.path.join(src, "bench_stats4m.json")))
    n4 = b4["config"]["bodies_per_gpu"]; t4 = (n4 + 63) // 64
    exp4 = t4 * 64 * (11 * 4 + 6 * 4) + t4 * 1920
    tr["f16_4m:tiled"] = {
        "hbm_bytes_per_launch": 2.0 * f4 + w4, "fetch_size_bytes_raw": f4, "fetch_size_bytes_corrected": 2.0 * f4, "write_size_bytes": w4,
        "expected_read_bytes": exp4, "expected_write_bytes": t4 * 64 * 24, "algorithmic_bytes": n4 * 130,
        "in_bench_event_us": {"under_kernel_trace": b4["roofline"]["kernel_us"]},
        "frac_of_8TBs_algorithmic": b4["roofline"]["frac"],
        "source": f"python bench.py --bodies 4194304 --scenes 2 (fp16 coefficients, two rotating replicas = 1.1 GB); rocprofv3 --pmc FETCH_SIZE / "
                  f"WRITE_SIZE in separate passes, FETCH x2 (gfx950; 2 x FETCH vs the known read bytes: {100.0 * (2.0 * f4 / exp4 - 1.0):+.2f}%); "
                  f"profiles/{tag}_f16_4m_kernel_stats.csv",
    }
# every engine kernel in one run (full bench incl. extras)
if os.path.isdir(os.path.join(src, "allkernels")):
    shutil.copy(one("allkernels/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_allkernels_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "bench_allkernels.json"), os.path.join(dst, f"{tag}_bench_under_kernel_trace.json"))
# the simulator-facing entry: hydro_step_wrench_aos, fp32 parameters
if os.path.isdir(os.path.join(src, "aos_fetch")):
    shutil.copy(one("aos_stats/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_aos_1m_kernel_stats.csv"))
    shutil.copy(one("aos_stats4m/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_aos_4m_kernel_stats.csv"))

    def med_aos(sub, counter):
        vals = []
        with open(one(os.path.join(sub, "**", "*counter_collection.csv")), newline="") as f:
            for r in csv.DictReader(f):
                if "wrench_aos" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    vals.append(float(r["Counter_Value"]))
        return statistics.median(vals) * 1024.0
    def avg_us(path):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if "wrench_aos" in r["Name"]:
                    return float(r["AverageNs"]) / 1e3, int(r["Calls"])
        return None, 0
    k1, c1 = avg_us(os.path.join(dst, f"{tag}_aos_1m_kernel_stats.csv")); k4, c4 = avg_us(os.path.join(dst, f"{tag}_aos_4m_kernel_stats.csv"))
    fa, wa = med_aos("aos_fetch", "FETCH_SIZE"), med_aos("aos_write", "WRITE_SIZE")
    ba = json.load(open(os.path.join(src, "bench_aos_stats.json"))); ba4 = json.load(open(os.path.join(src, "bench_aos_stats4m.json")))
    na = ba["config"]["bodies_per_gpu"]
    tr["c5-f32:aos"] = {
        "hbm_bytes_per_launch": 2.0 * fa + wa, "fetch_size_bytes_raw": fa, "fetch_size_bytes_corrected": 2.0 * fa, "write_size_bytes": wa,
        "expected_read_bytes": na * (12 + 16 + 24 + 24 + 44), "expected_write_bytes": na * (24 + 24), "algorithmic_bytes": na * 168,
        # kernel durations from the trace itself (all launches of the run, spin-up included): under the profiler the host
        # side of this entry's Python call is slower than the 30 us kernel at 1 M, so the in-bench per-step figure of the
        # profiled run is host-bound and says nothing about the kernel
        "kernel_trace_mean_us": {"1m": k1, "4m": k4}, "kernel_trace_calls": {"1m": c1, "4m": c4},
        "frac_of_8TBs": {"1m": na * 168 / (k1 * 1e-6) / 8e12, "4m": ba4["config"]["bodies_per_gpu"] * 168 / (k4 * 1e-6) / 8e12},
        "in_bench_step_us_under_the_profiler": {"1m": ba["roofline"]["kernel_us"], "4m": ba4["roofline"]["kernel_us"]},
        "source": f"python bench.py --layout aos --workload c5-f32 (hydro_step_wrench_aos, fp32 parameters, engine-owned previous velocity); rocprofv3 "
                  f"--pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x2 (gfx950; the x2 calibration is that of the 4-byte streaming pattern; "
                  f"2 x FETCH vs the known read bytes here: {100.0 * (2.0 * fa / (na * 120) - 1.0):+.2f}%); profiles/{tag}_aos_1m_kernel_stats.csv, {tag}_aos_4m_kernel_stats.csv",
    }
# auxiliary kernels (round 4): kinetic energy in one launch, the resident closed loop (SQ counters), several scenes per launch
if os.path.isdir(os.path.join(src, "ke1m")):
    for sub, name in (("ke1m", "ke_1m"), ("ke4m", "ke_4m"), ("resident", "resident_1m"), ("batch", "batch_4x1m")):
        shutil.copy(one(f"{sub}/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"))
    aux = {k: json.load(open(os.path.join(src, f"aux_{k}.json"))) for k in ("ke1m", "ke4m", "resident", "batch")}
    rows = []
    with open(one("resident_sq/**/*counter_collection.csv"), newline="") as f:
        for r in csv.DictReader(f):
            if "step_fused_multi_tiled_kernel" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"])))
    rows.sort()
    with open(os.path.join(dst, f"{tag}_resident_pmc_SQ.csv"), "w", newline="") as f:
        w = csv.writer(f); w.writerow(["dispatch_id", "kernel", "counter", "value"])
        for d_, c_, v_ in rows:
            w.writerow([d_, "step_fused_multi_tiled_kernel (64 steps per launch, 1 048 576 bodies)", c_, f"{v_:.6f}"])
    sqr = {}
    for _, c_, v_ in rows:
        sqr.setdefault(c_, []).append(v_)
    aux["resident"]["sq_per_launch_median"] = {k: statistics.median(v) for k, v in sorted(sqr.items())}
    if "SQ_INSTS_VALU" in sqr and "SQ_WAVES" in sqr:
        aux["resident"]["valu_instructions_per_wave_per_step_from_counters"] = statistics.median(sqr["SQ_INSTS_VALU"]) / statistics.median(sqr["SQ_WAVES"]) / 64.0
    json.dump(aux, open(os.path.join(dst, f"{tag}_aux_kernels.json"), "w"), indent=1)
    print(json.dumps(aux, indent=1)[:3000])
json.dump(tr, open(path, "w"), indent=1)
print(json.dumps(tr.get("c5-f32:aos"), indent=1))
print(json.dumps(tr["c5:tiled"], indent=1))
print(json.dumps(tr.get("f16_4m:tiled"), indent=1))
