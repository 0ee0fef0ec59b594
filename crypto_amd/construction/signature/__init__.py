"""construction/signature.rs: SignatureScheme trait; FDH with the ring GPV PSF (verify path)."""
from .fdh import FDH  # noqa: F401
