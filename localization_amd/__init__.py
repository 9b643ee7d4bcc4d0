"""localization_amd — MI355X-native batched range-localization solver (hot path of sair-lab/localization).

The product is the C-ABI shared library ``liblocalization_amd.so`` (HIP kernels for gfx950 + C++ host code,
see ``include/localization_amd.h``).  This package is the thin Python (ctypes) harness around it that tests,
``bench.py`` and ``__graft_entry__.py`` use; PyTorch is used only for device memory, streams and
``torch.distributed``.  There is no CPU fallback: without the built library or without a HIP device the
solver constructors raise.
"""
from ._lib import LocalizationAmdError, abi_version, device_count, lib, library_path
from .config import LocalizationConfig, load_config
from .snapshot import SnapshotSolver, pack_ranges, unpack_ranges
from .window import WindowBatch, WindowSolver
from .node import LocalizationNode, release_batch_cache, solve_batch
from .fusion import FusionSolver
from . import ate, bag

__all__ = [
    "LocalizationAmdError", "abi_version", "device_count", "lib", "library_path",
    "LocalizationConfig", "load_config", "SnapshotSolver", "pack_ranges", "unpack_ranges", "WindowBatch", "WindowSolver", "LocalizationNode", "solve_batch", "release_batch_cache", "FusionSolver",
]
