"""construction/hash.rs: HashInto trait, SIS hash, SHA-256 based hashes."""
from .sis import SISHash  # noqa: F401
from . import sha256  # noqa: F401
