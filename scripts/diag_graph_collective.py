#!/usr/bin/env python3
"""DEV-ONLY (GPU box): can an RCCL all-reduce live inside a HIP graph next to the step kernels?  One-rank RCCL group; two graphs of
10 steps whose last step samples the kinetic energy, followed - inside the capture - by dist.all_reduce of the pair and its copy
to pinned host memory.  Yes: 20 steps + 2 complete samples in 161 us of wall time with 22-29 us of host time (host-driven
sampling: 290 us).  What KineticEnergyMonitor.capture_sample came from.      python scripts/diag_graph_collective.py"""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.update(HYDRO_DIST_ALWAYS="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29677")
import torch, torch.distributed as dist
import bench
from silver2_isaacsim_amd import distributed as hd, scenes
hd.init_process_group(force=True, node_barrier=False)
dev = torch.device("cuda:0"); stream = torch.cuda.Stream(dev)
full = bench.build_scene("c4", 262144, 4)
reps = [bench.Replica(full, "f32", dev, roll=0) for _ in range(2)]
ke = [torch.zeros(2, dtype=torch.float64, device=dev) for _ in range(2)]
host = [torch.zeros(2, dtype=torch.float64, pin_memory=True) for _ in range(2)]
bench.spin_up(reps, stream, 0.3)
G = 10
graphs = []
with torch.cuda.stream(stream):
    for j in range(2):
        reps[0].step_sampling(ke[j]); reps[1].step_sampling(ke[j])
    dist.all_reduce(ke[0]); dist.all_reduce(ke[1])          # warm the communicator on these buffers
    stream.synchronize()
    for j in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
            for k in range(G):
                (reps[k % 2].step_sampling(ke[j]) if k == G - 1 else reps[k % 2].step())
            dist.all_reduce(ke[j])                           # captured: RCCL inside the graph
            host[j].copy_(ke[j], non_blocking=True)
        graphs.append(g)
    for g in graphs: g.replay()
    stream.synchronize()
    want = scenes.kinetic_energy_fp64(full.state, full.params)
    print("captured collective ok:", host[0].tolist(), host[1].tolist(), want)
    for trial in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); e1.record(stream); torch.cuda.synchronize()
        t0 = time.perf_counter(); e0.record(stream)
        graphs[0].replay(); graphs[1].replay()
        e1.record(stream); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(json.dumps({"20 steps + 2 samples, all in graphs": {"host_us": round((t1 - t0) * 1e6, 1), "wall_us": round((t2 - t0) * 1e6, 1), "events_us": round(e0.elapsed_time(e1) * 1e3, 1)}}))
dist.destroy_process_group()
