"""primitive/psf.rs:34-71: the PSF trait; only its verify half (f_a, check_domain) is on the hot path."""
from .gpv_ring import GadgetParametersRing, PSFGPVRing  # noqa: F401
from .gpv import GadgetParameters, PSFGPV  # noqa: F401
