"""Runs bench.py in PLUMBING mode: gloo + CPU tensors + the torch emulation of the op layer (tests/emulate.py) on tiny meshes.
It exercises launch / sharding / all-gather / JSON assembly of bench.py without a GPU; the line it prints says it is not a
measurement. Only tests/ use this entry; bench.py itself never imports anything from tests/."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE]
os.environ["MORIG_BENCH_PLUMBING"] = "1"
os.environ["MORIG_BENCH_ENTRY"] = os.path.abspath(__file__)      # bench.py's self-launch re-enters through this file

from emulate import EmuOps                # noqa: E402
from morig_amd import runtime             # noqa: E402

runtime._test_ops = EmuOps()
sys.argv[0] = os.path.join(ROOT, "bench.py")
runpy.run_path(sys.argv[0], run_name="__main__")
