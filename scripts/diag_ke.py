#!/usr/bin/env python3
"""DEV: the stand-alone kinetic-energy entry against its memory-only probe (56 B per body), interleaved, at 1 M and 4 M
bodies - the figure bench.py reports as extras.bound_probes_*.ke_kernel_us / ke_memory_only_us (VERDICT r3 item 3:
kernel <= 1.1 x probe).  python scripts/diag_ke.py [n ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from scripts import probes  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0"); stream = torch.cuda.Stream(dev)
    for n in [int(x) for x in sys.argv[1:]] or [1048576, 4194304]:
        r = probes.bound_probes(n, dev, stream, rounds=7, which=(), with_aos=False, with_ke=True)
        us = r["us"]
        probe = [v for k, v in us.items() if "KE reads" in k][0]
        kern = [v for k, v in us.items() if "hydro_kinetic_energy_tiled" in k][0]
        for name, v in us.items():
            print(f"n={n:9d} {name:58s}: {v:8.2f} us", flush=True)
        print(f"n={n:9d} kernel / probe = {kern / probe:.3f}   ke_frac = {n * 56 / (kern * 1e-6) / 8e12:.3f} of 8 TB/s", flush=True)
        torch.cuda.empty_cache()
