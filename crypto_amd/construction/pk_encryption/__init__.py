"""construction/pk_encryption.rs: the PKEncryptionScheme trait and the schemes on the hot path."""
from .base import GenericMultiBitEncryption, PKEncryptionScheme  # noqa: F401
from .regev import Regev  # noqa: F401
from .dual_regev import DualRegev  # noqa: F401
from .lpr import LPR  # noqa: F401
from .ring_lpr import RingLPR  # noqa: F401
from .regev_discrete_gauss import DualRegevWithDiscreteGaussianRegularity, RegevWithDiscreteGaussianRegularity  # noqa: F401
