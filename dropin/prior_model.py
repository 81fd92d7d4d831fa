"""Drop-in for the reference's `prior_model` module: `from prior_model import *` keeps working and
pickles that name `prior_model.LinearTransform` / `prior_model.Upsample` resolve to the MI355X classes."""
from recombiner_amd.prior_model import *  # noqa: F401,F403
from recombiner_amd.prior_model import (LinearTransform, PriorBNNmodel, Upsample, get_grouping,  # noqa: F401
                                        get_grouping_by_kl, group_parameters)
import numpy as np  # noqa: F401  (the reference's drivers rely on these names leaking through the star import)
import torch  # noqa: F401
import torch.nn.functional as F  # noqa: F401

# pickles written while this drop-in is active carry the reference's class paths, so prior checkpoints
# interchange with the reference in both directions (main_prior_training.py:334-335, main_compression.py:44-45)
LinearTransform.__module__ = "prior_model"
Upsample.__module__ = "prior_model"
