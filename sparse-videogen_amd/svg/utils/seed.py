"""ref: svg/utils/seed.py:8-15"""
import os
import random

import numpy as np
import torch


def seed_everything(seed: int) -> None:
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    # the device-switched SVG1 path draws its profiler rows from a private generator derived from this seed
    from ..models._core import reseed_switch_generator

    reseed_switch_generator(seed)
