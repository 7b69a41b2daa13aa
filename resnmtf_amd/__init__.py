"""resnmtf_amd -- MI355X-native (gfx950) multiplicative-update inner loop of ResNMTF.

Only what the hot path needs: ``csrc/`` (HIP kernels + the C-ABI of
``include/resnmtf_hip.h``), a ctypes engine over it, the host-side mirror of the reference's
``res_nmtf_inner`` / ``apply_resnmtf`` entry points, and the view-sharded driver over
``torch.distributed``.  There is no CPU compute path in this package.
"""
from .api import apply_resnmtf, res_nmtf_inner  # noqa: F401
from .engine import Engine, device_count  # noqa: F401
from ._lib import ResnmtfError  # noqa: F401

__all__ = ["apply_resnmtf", "res_nmtf_inner", "Engine", "device_count", "ResnmtfError"]
