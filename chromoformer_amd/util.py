import os
import random

import numpy as np
import torch


def seed_everything(seed=42):
    """Same seeding contract as the reference (chromoformer/util.py:6-13); the cudnn flags it
    sets have no meaning on this path (determinism is by construction: no atomics)."""
    random.seed(seed)
    np.random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
