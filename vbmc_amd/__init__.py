"""vbmc_amd: MI355X-native ELBO inner loop of VBMC behind the reference's call surface.

Host-side mirror (Python, because the reference's MATLAB toolchain is absent here) of
negelcbo_vbmc / gplogjoint / entmc_vbmc / entlb_vbmc / gplite_post / gplite_pred / gplite_nlZ /
vpsieve_vbmc / vpoptimize_vbmc / fminadam over the C ABI of libvbmc_hip.so
(include/vbmc_hip.h).  All numerics run in hand-written HIP kernels on gfx950; there is
no CPU fallback in this package.
"""
from . import _lib  # noqa: F401
from ._lib import Context, DeviceGP, VbmcHipError, VbmcUnsupported  # noqa: F401
from .vp import get_vptheta, make_vp, rescale_params, vpbounds  # noqa: F401
from .elbo import (Engine, PreparedObjective, default_engine, entlb_vbmc, entmc_vbmc, fminadam_device, gplogjoint,  # noqa: F401
                   negelcbo_batch, negelcbo_shard, negelcbo_vbmc)
from .gplite import gplite_hypprior, gplite_nlZ, gplite_post, gplite_post_rank1, gplite_pred, sq_dist  # noqa: F401,E402
from .acq import acq_info, acqwrapper_vbmc, activeimportancesampling_vbmc, ensemble_slice_sample, vbmc_rnd  # noqa: F401,E402
from .optimize import (eval_fullelcbo, fminadam, gethpd_vbmc, sieve_evaluate, vbinit_vbmc, vpoptimize_vbmc,  # noqa: F401,E402
                       vpsieve_vbmc)
