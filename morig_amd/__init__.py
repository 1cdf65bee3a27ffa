"""morig_amd -- MI355X-native (gfx950) forward path of MoRig's geometric networks.

Layout: csrc/ (HIP kernels + C ABI, built to lib/libmorig_hip.so), native.py (ctypes binding + op
layer), packing.py (parameter layout for the kernels), models/ (drop-in mirror of the reference's
``models`` package), dist.py (mesh sharding + RCCL all-gather), harness.py (post-ops and writers of
the reference's eval loop), synth.py (seeded synthetic inputs)."""
__version__ = "0.1.0"
