"""Drop-in for the reference's `config` module."""
from recombiner_amd.config import configs  # noqa: F401
