"""Drop-in for the reference's `utils` module."""
from recombiner_amd.utils import *  # noqa: F401,F403
