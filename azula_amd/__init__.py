r"""azula_amd -- MI355X-native sampling engine, drop-in for the sampling path of azula.

    from azula_amd.denoise import KarrasDenoiser
    from azula_amd.noise import VPSchedule
    from azula_amd.sample import DDIMSampler, DDPMSampler

The per-step hot path (schedule arithmetic, preconditioning, backbone forward, DDPM/DDIM
transition) runs in hand-written gfx950 HIP kernels behind the C ABI of ``include/azula_amd.h``
(``azula_amd/csrc/libazula_amd.so``, loaded with ctypes).  There is no eager/CPU fallback for
device tensors: a missing shared object raises.
"""

__version__ = "0.1.0"

from . import denoise, noise, sample  # noqa: F401
from . import nn  # noqa: F401
