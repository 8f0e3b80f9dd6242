"""gpirt_amd -- host-side mirror of the ``gpirtMCMC`` interface over the HIP library."""
from ._abi import (GpirtConfig, GpirtOutputs, default_config, KERNEL_RBF, KERNEL_MATERN52,
                   THETA_ESS, THETA_GRID, RETURN_F, RETURN_FSTAR, ALLREDUCE_ON_STREAM, EXACT_FSTAR)

__all__ = ["GpirtConfig", "GpirtOutputs", "default_config", "KERNEL_RBF", "KERNEL_MATERN52",
           "THETA_ESS", "THETA_GRID", "RETURN_F", "RETURN_FSTAR", "ALLREDUCE_ON_STREAM", "EXACT_FSTAR"]
