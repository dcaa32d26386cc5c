"""AverageMeter / Logger / setup_seed with the reference's behaviour
(/root/reference/utils/utils.py:2-34, utils/benchmark_utils.py:9-18)."""
import os
import random

import numpy as np
import torch


class AverageMeter:
    """Running mean of per-pair values (mean of means; NaNs propagate, as upstream)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val, self.avg, self.sum, self.sq_sum, self.count = 0, 0, 0.0, 0.0, 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.sq_sum += val ** 2 * n
        self.count += n
        self.avg = self.sum / self.count


class Logger:
    def __init__(self, log_path):
        if os.path.exists(log_path):
            os.remove(log_path)
        self.fw = open(log_path, "a")

    def write(self, text):
        self.fw.write(text)
        self.fw.flush()

    def close(self):
        self.fw.close()


def setup_seed(seed):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)
