#!/usr/bin/env python3
"""DEV-ONLY (GPU box): bench_strong.c4_strong_leg by itself on one rank - without a process group (the monitor's all-reduce is a no-op)
and with a one-rank RCCL group (HYDRO_DIST_ALWAYS=1) - at the driver's 20 steps and at 600: where the time of the leg goes.
    python scripts/diag_strong_leg.py [nccl]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench  # noqa: E402,F401
from scripts import bench_strong  # noqa: E402
from silver2_isaacsim_amd import distributed as hd  # noqa: E402

use_nccl = len(sys.argv) > 1 and sys.argv[1] == "nccl"
if use_nccl:
    os.environ.update(HYDRO_DIST_ALWAYS="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    hd.init_process_group(force=True, node_barrier=True)
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
stream = torch.cuda.Stream(dev)
for steps, warmup in ((20, 5), (20, 5), (600, 8)):
    r = bench_strong.c4_strong_leg(0, 1, dev, stream, steps, warmup, collectives=use_nccl)
    ke = r["ke"]
    print(json.dumps({"group": "nccl x1" if use_nccl else "none", "steps": steps, "ms_per_step_us": r["ms_per_step"] * 1e3, "kernel_us": r["kernel_us_rank0"],
                      "samples": ke["samples"], "host_waits": ke["host_waits"], "graph_steps": r["graph_steps"], "rel": ke["rel_err"],
                      "captured": r["captured"]}), flush=True)
