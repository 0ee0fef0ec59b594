from .common_moduli import new_anticyclic, new_cyclic  # noqa: F401
from .rotation_matrix import rot_minus, rot_minus_matrix  # noqa: F401
