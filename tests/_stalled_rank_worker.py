"""Worker of tests/test_closed_loop_gpu.py::test_a_rank_that_never_joins_a_sample_is_a_timeout_not_a_hang: the sharded closed-loop
example with rank 1 made to stall before its first replay (it never joins the kinetic-energy all-reduce)."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "examples"))
import sharded_closed_loop as example                               # noqa: E402
from silver2_isaacsim_amd.simulate import ClosedLoopSim            # noqa: E402

if int(os.environ.get("RANK", "0")) == 1:
    ClosedLoopSim.run = lambda self, steps, graph_steps=64: time.sleep(3600)
example.main(sys.argv[1:])
