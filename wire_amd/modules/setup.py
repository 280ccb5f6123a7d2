"""Seeding helper (reference modules/setup.py:7-13)."""
import random

import numpy as np
import torch


def seed_everything(seed: int = 0) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)            # also seeds every visible GPU generator
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    from .utils import log
    log(f"Seed set to {seed}")
