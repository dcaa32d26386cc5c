import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deformationpyramid_amd.parallel import aggregate, shard_range  # noqa: E402

dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
lo, hi = shard_range(10, r, w)
vals = torch.tensor([float(hi - lo), float(sum(range(lo, hi)))], dtype=torch.float64)
tot, tmax = aggregate(vals, elapsed=1.0 + r, device=torch.device("cpu"))
if r == 0:
    assert tot.tolist() == [10.0, 45.0], tot
    assert tmax == 2.0
    print("AGG_OK")
dist.destroy_process_group()
