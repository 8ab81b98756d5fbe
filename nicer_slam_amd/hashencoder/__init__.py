"""Drop-in for the reference's ``code/hashencoder`` package (same module and attribute names)."""
from .hashgrid import HashEncoder, hash_encode  # noqa: F401
