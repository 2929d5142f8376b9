"""bluerov2_amd -- MI355X-native batched NMPC (SQP-RTI) solver for the BlueROV2 OCP.

The compute path is the HIP library bluerov2_amd/lib/libbluerov2_nmpc.so (C ABI: include/bluerov2_nmpc.h).  This package
is the thin host-side mirror of that ABI; there is no CPU or PyTorch fallback -- without the library or without a GPU the
solver raises.
"""
from .solver import (BatchSolver, SolverOptions, RESULT_DTYPE, P_NOMINAL, build_library, library_path,  # noqa: F401
                     NoDeviceError, thrust_allocation, PATH_AUTO, PATH_STREAMING, PATH_FUSED, PATH_WINDOWED)

from .ekf import BatchEkf, EkfParams  # noqa: F401,E402
from .group import SolverGroup, GATHER_RECORDS, GATHER_PACKED, rccl_version, unique_id  # noqa: F401,E402

__all__ = ["SolverGroup", "GATHER_RECORDS", "GATHER_PACKED", "rccl_version", "unique_id", "BatchEkf", "EkfParams", "BatchSolver", "SolverOptions", "RESULT_DTYPE", "P_NOMINAL", "build_library", "library_path",
           "NoDeviceError", "thrust_allocation", "PATH_AUTO", "PATH_STREAMING", "PATH_FUSED", "PATH_WINDOWED"]
