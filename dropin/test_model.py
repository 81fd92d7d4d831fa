"""Drop-in for the reference's `test_model` module."""
from recombiner_amd.test_model import *  # noqa: F401,F403
from recombiner_amd.test_model import Sine, TestBNNmodel  # noqa: F401
import numpy as np  # noqa: F401
import torch  # noqa: F401
