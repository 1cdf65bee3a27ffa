"""Which op layer the network plans run on: always the HIP one (morig_amd.native.NativeOps).

``_test_ops`` is a seam for the CPU unit tests of the HOST logic (packing, column placement,
plan wiring): tests/ install a torch emulation of the op interface there (tests/emulate.py; also
tests/bench_plumbing.py, which runs bench.py's launch / sharding / collective code on CPU).
Nothing in the product or in bench.py sets it, and there is no automatic fallback: without the HIP library or a GPU, get_ops() raises.
"""
from __future__ import annotations

_test_ops = None


def get_ops():
    if _test_ops is not None:
        return _test_ops
    from . import native
    return native.get_ops()
