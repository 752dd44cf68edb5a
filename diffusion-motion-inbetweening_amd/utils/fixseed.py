"""Seeding helper, same behaviour as the reference's ``utils/fixseed.py:6-11``."""
import random

import numpy as np
import torch


def fixseed(seed):
    torch.backends.cudnn.benchmark = False
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
