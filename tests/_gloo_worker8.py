"""Worker of test_eight_rank_gloo_bench_aggregation: every rank calls bench.py's aggregation path (parallel.job_summary)
with its own pair count and a skewed elapsed time; rank 0 checks value = sum(pairs) / max(elapsed)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deformationpyramid_amd.parallel import job_summary  # noqa: E402

dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
pairs = 4096 - 7 * r                                   # ranks may finish different numbers of pairs
elapsed = 4.0 + 0.37 * ((5 * r) % w)                   # skewed, the slowest is not the last rank
s = job_summary(pairs, elapsed, [float(r), 2.0], torch.device("cpu"))
want_pairs = sum(4096 - 7 * k for k in range(w))
want_t = max(4.0 + 0.37 * ((5 * k) % w) for k in range(w))
assert s["world_size"] == w and s["pairs"] == want_pairs and s["elapsed"] == want_t, s
assert abs(s["value"] - want_pairs / want_t) < 1e-9
assert s["sums"].tolist() == [sum(range(w)), 2.0 * w]
assert s["pairs_per_rank"] == [4096.0 - 7 * k for k in range(w)]
assert s["elapsed_per_rank"] == [4.0 + 0.37 * ((5 * k) % w) for k in range(w)]
rates = [p / t for p, t in zip(s["pairs_per_rank"], s["elapsed_per_rank"])]
assert s["rank_pairs_per_s_min"] == min(rates) and s["rank_pairs_per_s_max"] == max(rates)
assert w * min(rates) * 0.9 < s["value"] <= sum(rates)       # the job rate is set by the slowest rank, never above the sum
dist.barrier()
if r == 0:
    print("AGG8_OK", s["value"])
dist.destroy_process_group()
