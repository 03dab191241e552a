"""ptmcmc_amd -- MI355X-native parallel-tempering step engine behind ptmcmc's chain::step() plug-in surface.

Only what the hot path needs lives here: csrc/ (HIP kernels + the C ABI of include/ptm_engine.h),
engine.py (ctypes binding), problems.py (synthetic workloads), parallel.py (ladder sharding over
torch.distributed / RCCL), host/ (C++ facade with the reference's class names).
"""
from .engine import Engine, PtmError, device_count, load  # noqa: F401

__all__ = ["Engine", "PtmError", "device_count", "load"]
