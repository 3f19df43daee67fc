"""`seed_everything(seed)` as the reference's entry scripts call it (ref: svg/utils/seed.py:8-15): Python's, numpy's and torch's
generators on the host and on every visible GPU, the hash seed of child interpreters, and the deterministic-convolution flags the
reference sets (MIOpen reads the same `torch.backends.cudnn` switches on ROCm).  One thing more than the reference: the private generator
the device-switched SVG1 path draws its profiler rows from follows the same seed."""
import os
import random

import numpy as np
import torch


def seed_everything(seed: int) -> None:
    seed = int(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    cudnn = torch.backends.cudnn
    cudnn.deterministic, cudnn.benchmark = True, False
    from ..models._core import reseed_switch_generator

    reseed_switch_generator(seed)
