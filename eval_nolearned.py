#!/usr/bin/env python3
"""Drop-in counterpart of the reference's eval_nolearned.py (/root/reference/eval_nolearned.py:26-159)
for the NDP model on the MI355X path.

    python eval_nolearned.py --config config/NDP.yaml [--batched] [--synthetic N]

Same flow as upstream: seed once, load the YAML (with `!join`), build `Registration(config)`, loop over
the benchmark pairs, compute the ground-truth scene flow and the overlap mask, `load_pcds` + `register`,
then `compute_flow_metrics` averaged by `AverageMeter`, then the timer report.  Differences:

* data: the 4DMatch `*.npz` split (keys s_pc, t_pc, s2t_flow, rot, trans, correspondences --
  correspondence/datasets/_4dmatch.py:60-73) is read when `config.data_root` exists; otherwise
  (the 14 GB download is not available offline) `--synthetic N` seeded pairs stand in;
* `--batched` registers all pairs through `Registration.register_batch` (many pairs resident on the
  GPU); without it the loop calls `register()` pair by pair exactly like upstream;
* under `torchrun` (WORLD_SIZE > 1) the pairs of each benchmark are sharded over the ranks, one GPU per rank
  (`parallel.shard_range`), and the per-metric sums are combined by ONE all-reduce (RCCL; `NDP_BENCH_BACKEND=gloo` for a
  rehearsal on a box with one GPU) -- BASELINE.json config 3, "4DMatch-F full split batched across 8 GPUs";
* `deformation_model: NDP` and the `NSFP` / `Nerfies` baselines (config/baselines/*.yaml) are served; the other models
  are comparison baselines outside the scope.
"""
import argparse
import glob
import os

import numpy as np
import torch

from deformationpyramid_amd.config import load_config
from deformationpyramid_amd.loss import compute_flow_metrics
from deformationpyramid_amd.parallel import aggregate, shard_range
from deformationpyramid_amd.registration import Registration
from deformationpyramid_amd.synthetic import synthetic_pair
from deformationpyramid_amd.utils import AverageMeter, Logger, Timers, setup_seed


class FourDMatchPairs:
    """Minimal reader of the 4DMatch test split: yields (src, tgt, flow_gt, overlap) numpy/torch items
    with the ground truth built as in eval_nolearned.py:75-84."""

    def __init__(self, root, split, max_points=30000):
        # upstream keeps the raw glob order (_4dmatch.py:47); sorted here so that every rank of a sharded run sees the
        # same list (glob order is filesystem dependent) and the shards are a true partition
        self.files = sorted(glob.glob(os.path.join(root, split, "*/*.npz")))
        self.max_points = max_points

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        e = np.load(self.files[i])
        src, tgt = e["s_pc"].astype(np.float32), e["t_pc"].astype(np.float32)
        rot, trn, flow = e["rot"].astype(np.float32), e["trans"].astype(np.float32), e["s2t_flow"].astype(np.float32)
        corr = e["correspondences"]
        if src.shape[0] > self.max_points:                                 # _4dmatch.py:93-98
            idx = np.random.permutation(src.shape[0])[: self.max_points]
            remap = -np.ones(src.shape[0], dtype=np.int64)
            remap[idx] = np.arange(idx.size)
            src, flow = src[idx], flow[idx]
            corr = corr[remap[corr[:, 0]] >= 0]
            corr = np.stack([remap[corr[:, 0]], corr[:, 1]], 1)
        if tgt.shape[0] > self.max_points:
            tgt = tgt[np.random.permutation(tgt.shape[0])[: self.max_points]]
        warped = (rot @ (src + flow).T + trn.reshape(3, 1)).T
        flow_gt = torch.from_numpy((warped - src).astype(np.float32))
        overlap = np.zeros(src.shape[0], dtype=bool)
        overlap[corr[:, 0]] = True
        return torch.from_numpy(src), torch.from_numpy(tgt), flow_gt, torch.from_numpy(overlap)


class SyntheticDepthPairs:
    """Depth-image pairs for the embedded-deformation (N-ICP) baseline, which starts from depth maps
    (eval_nolearned.py:113-118): synthetic_depth_pair(i) written as 16-bit PNGs; the "sampled" clouds are every 9th valid
    pixel, the ground-truth flow is the motion of the same pixel's point, overlap = valid in both frames."""

    def __init__(self, n, workdir):
        self.n, self.dir = n, workdir

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        from PIL import Image
        from deformationpyramid_amd.geometry import depth_2_pc
        from deformationpyramid_amd.synthetic import synthetic_depth_pair
        d0, d1, K = synthetic_depth_pair(i)
        paths = [os.path.join(self.dir, f"pair{i}_{k}.png") for k in "st"]
        Image.fromarray(d0).save(paths[0])
        Image.fromarray(d1).save(paths[1])
        p0 = depth_2_pc(d0 / 1000.0, K).transpose(1, 2, 0)
        p1 = depth_2_pc(d1 / 1000.0, K).transpose(1, 2, 0)
        m0 = d0 > 0
        src = torch.from_numpy(p0[m0]).float()[::9].contiguous()
        tgt = torch.from_numpy(p1[d1 > 0]).float()[::9].contiguous()
        flow_gt = torch.from_numpy((p1 - p0)[m0]).float()[::9].contiguous()
        overlap = torch.from_numpy((d1 > 0)[m0])[::9].contiguous()
        return src, tgt, flow_gt, overlap, paths, K


class SyntheticPairs:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return synthetic_pair(i)


def dist_setup():
    """-> (world, rank, local_rank, backend).  Under torchrun one process per GPU; NDP_BENCH_BACKEND=gloo keeps every
    rank on cuda:0 and aggregates over gloo (rehearsal on a one-GPU box)."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    backend = os.environ.get("NDP_BENCH_BACKEND", "nccl")
    local_rank = 0 if backend == "gloo" else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    return world, rank, local_rank, backend


def reduce_meters(meters, n_items, world, local_rank, backend):
    """Mean of the per-pair metrics over ALL ranks (what upstream's AverageMeter reports on one GPU): one SUM
    all-reduce of the per-metric sums and the pair count."""
    if meters is None:                                                      # a rank without pairs still joins the all-reduce
        z = compute_flow_metrics(torch.zeros(2, 3), torch.ones(2, 3), overlap=torch.tensor([True, False]))
        meters = {k: AverageMeter() for k in z}
    keys = list(meters.keys())
    if world > 1:
        local = torch.tensor([m.sum for m in meters.values()] + [float(n_items)], dtype=torch.float64)
        tot, _ = aggregate(local, 0.0, torch.device("cpu") if backend == "gloo" else torch.device("cuda", local_rank))
        return keys, {k: float(tot[i] / tot[-1]) for i, k in enumerate(keys)}
    return keys, {k: m.avg for k, m in meters.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default="config/NDP.yaml", help="Path to the config file.")
    ap.add_argument("--visualize", action="store_true", help="(upstream flag; mayavi is out of scope here)")
    ap.add_argument("--batched", action="store_true", help="register all pairs through register_batch")
    ap.add_argument("--slots", type=int, default=64)
    ap.add_argument("--synthetic", type=int, default=32, help="pairs to generate when the dataset is absent")
    args = ap.parse_args()
    world, rank, local_rank, backend = dist_setup()
    setup_seed(rank)                                                        # once per process, as upstream (seed 0 on one GPU)
    config = load_config(args.config, make_dirs=rank == 0, device=local_rank)
    if config.deformation_model not in ("NDP", "NSFP", "Nerfies", "ED"):
        raise KeyError(config.deformation_model)
    model = Registration(config)
    timer = Timers()
    for benchmark in ["4DMatch-F", "4DLoMatch-F"]:
        config.split["test"] = benchmark
        root = os.path.join(config.data_root, benchmark)
        if os.path.isdir(root):
            data = FourDMatchPairs(config.data_root, benchmark)
        elif config.deformation_model == "ED":
            import tempfile
            print(f"[{benchmark}] {root} not found: using {args.synthetic} synthetic depth pairs")
            data = SyntheticDepthPairs(args.synthetic, tempfile.mkdtemp(prefix="ndp_depth_"))
        else:
            print(f"[{benchmark}] {root} not found: using {args.synthetic} synthetic pairs")
            data = SyntheticPairs(args.synthetic)
        lo, hi = shard_range(len(data), rank, world)                        # this rank's pairs
        items = [data[i] for i in range(lo, hi)]
        n_total = len(data)
        if config.deformation_model == "ED":                                # eval_nolearned.py:113-127 (N-ICP: needs the depth maps)
            if len(items) and len(items[0]) < 6:
                raise KeyError("the ED baseline needs depth images: the 4DMatch reader here does not provide them")
            flows, kept = [], []
            for src, tgt, flow_gt, overlap, depth_paths, cam_intrin in items:
                model.load_pcds(src, tgt)
                timer.tic("graph construction")
                model.load_raw_pcds_from_depth(depth_paths[0], depth_paths[1], cam_intrin, landmarks=None)
                timer.toc("graph construction")
                timer.tic("registration")
                warped, point_mask = model.register(visualize=args.visualize)
                torch.cuda.synchronize()
                timer.toc("registration")
                pm = point_mask.cpu()
                flows.append((warped - model.src_pcd[point_mask]).cpu())
                kept.append((src, tgt, flow_gt[pm], overlap[pm]))
            items = kept
        elif config.deformation_model in ("NSFP", "Nerfies"):               # eval_nolearned.py:97-110
            flows = []
            for src, tgt, _, _ in items:
                model.load_pcds(src, tgt)
                timer.tic("registration")
                warped, smpl_ind = model.register(visualize=args.visualize)
                torch.cuda.synchronize()
                timer.toc("registration")
                flows.append((warped - model.src_pcd).cpu())
        elif args.batched:
            timer.tic("registration")
            results = model.register_batch([(s, t) for s, t, _, _ in items], slots=args.slots)
            torch.cuda.synchronize()
            timer.toc("registration")
            flows = [w.cpu() - s for (w, _), (s, _, _, _) in zip(results, items)]
        else:
            flows = []
            for src, tgt, _, _ in items:
                model.load_pcds(src, tgt)
                timer.tic("registration")
                warped, iter_cnt, timer = model.register(visualize=args.visualize, timer=timer)
                timer.toc("registration")
                flows.append((warped - model.src_pcd).cpu())
        meters = None
        for flow, (_, _, flow_gt, overlap) in zip(flows, items):
            info = compute_flow_metrics(flow, flow_gt, overlap=overlap)
            if meters is None:
                meters = {k: AverageMeter() for k in info}
            for k, v in info.items():
                meters[k].update(v)
        keys, avgs = reduce_meters(meters, len(items), world, local_rank, backend)
        if rank == 0:
            message = f"{n_total}/{n_total}: " + "".join(f"{k}: {avgs[k]:.3f}\t" for k in keys)
            Logger(os.path.join(config.snapshot_dir, benchmark + ".log")).write(message + "\n")
            print("score on ", benchmark, "\n", message)
        if not os.path.isdir(root):
            break                                                           # one synthetic benchmark is enough
    if rank == 0:
        print("time cost average")
        for line in timer.get_strings():
            print(line)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
