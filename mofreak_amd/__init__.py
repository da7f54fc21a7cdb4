"""mofreak_amd -- MI355X-native MoFREAK descriptor extraction (the hot path of ChrisWhiten/MoFREAK).

Layout:
  csrc/      gfx950 HIP kernels + the C ABI of libmofreak_hip.so (include/mofreak_hip.h)
  api.py     ctypes binding of that C ABI (numpy / torch buffers in, no compute of its own)
  host/      C++ MoFREAKUtilities facade with the reference's class interface, over the C ABI
  harness.py the Python mirror of MoFREAKUtilities / computeMoFREAKFiles + multi-GPU sharding
  launch.py  one process per GPU from a parent that makes no GPU call (what `bench.py --gpus N` uses)
  synth.py   deterministic synthetic frame stacks and keypoint grids
  build.py   in-tree hipcc build

Nothing here falls back to the CPU: without libmofreak_hip.so and a GPU the compute calls raise.
"""
from . import synth  # noqa: F401
from .api import (BITS_NATURAL, BITS_SSE, BITS_SSE_SIGNED, KEYPOINT_DTYPE, ROW_DTYPE, TABLES_ONLY, Context,  # noqa: F401
                  MoFREAKError, format_rows, load, parse_rows)
