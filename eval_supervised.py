#!/usr/bin/env python3
"""Drop-in counterpart of the reference's eval_supervised.py (/root/reference/eval_supervised.py:40-190) for the
LNDP path (landmark-guided deformation pyramid) on the MI355X engine.

    python eval_supervised.py --config config/LNDP.yaml [--batched] [--synthetic N] [--landmarks DIR]

Upstream predicts the landmark correspondences with the Lepard matcher (`Landmark_Model.inference`, :102), which
is outside this repository's scope.  Here they are *precomputed*: `--landmarks DIR` holds one
`<pair stem>.npz` per 4DMatch pair with `ldmk_s [K,3]`, `ldmk_t [K,3]` (un-centred coordinates, what
`ldmk_model.inference` returns); without it -- or without the dataset -- seeded synthetic pairs get K = 500 landmarks
`(src[idx], GT-warped src[idx] + N(0, 0.005^2))`.  Everything after the matcher
is upstream's flow: GT scene flow and overlap mask (:111-124), `load_pcds(src, tgt, landmarks=(ldmk_s, ldmk_t))`,
`register()`, `compute_flow_metrics`, `AverageMeter`, timers.
"""
import argparse
import os

import numpy as np
import torch

from deformationpyramid_amd.config import load_config
from deformationpyramid_amd.loss import compute_flow_metrics
from deformationpyramid_amd.registration import Registration
from deformationpyramid_amd.synthetic import synthetic_landmarks, synthetic_pair
from deformationpyramid_amd.utils import AverageMeter, Logger, Timers, setup_seed
from deformationpyramid_amd.parallel import shard_range
from eval_nolearned import FourDMatchPairs, dist_setup, reduce_meters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default="config/LNDP.yaml", help="Path to the config file.")
    ap.add_argument("--visualize", action="store_true", help="(upstream flag; mayavi is out of scope here)")
    ap.add_argument("--batched", action="store_true", help="register all pairs through register_batch")
    ap.add_argument("--slots", type=int, default=64)
    ap.add_argument("--synthetic", type=int, default=32, help="pairs to generate when the dataset is absent")
    ap.add_argument("--landmarks", type=str, default="", help="directory of precomputed <stem>.npz (ldmk_s, ldmk_t)")
    ap.add_argument("--K", type=int, default=500, help="synthetic landmarks per pair")
    args = ap.parse_args()
    world, rank, local_rank, backend = dist_setup()              # torchrun: pairs sharded over ranks (BASELINE config 5)
    setup_seed(rank)
    config = load_config(args.config, make_dirs=rank == 0, device=local_rank)
    if config.deformation_model != "NDP":
        raise KeyError(config.deformation_model)
    model = Registration(config)
    timer = Timers()
    for benchmark in ["4DMatch-F", "4DLoMatch-F"]:
        config.split["test"] = benchmark
        root = os.path.join(config.data_root, benchmark)
        items = []
        if os.path.isdir(root) and args.landmarks:
            data = FourDMatchPairs(config.data_root, benchmark)
            n_total = len(data)
            for i in range(*shard_range(n_total, rank, world)):
                src, tgt, flow_gt, overlap = data[i]
                stem = os.path.splitext(os.path.basename(data.files[i]))[0]
                lm = np.load(os.path.join(args.landmarks, stem + ".npz"))
                ldmk = (torch.from_numpy(lm["ldmk_s"]).float(), torch.from_numpy(lm["ldmk_t"]).float())
                items.append((src, tgt, flow_gt, overlap, ldmk))
        else:
            if rank == 0:
                print(f"[{benchmark}] dataset or --landmarks missing: {args.synthetic} synthetic pairs, K = {args.K} landmarks")
            n_total = args.synthetic
            for p in range(*shard_range(n_total, rank, world)):
                src, tgt, flow_gt, overlap = synthetic_pair(p)
                items.append((src, tgt, flow_gt, overlap, synthetic_landmarks(p, src, flow_gt, k=args.K)))
        if args.batched:
            timer.tic("registration")
            results = model.register_batch([(s, t, l) for s, t, _, _, l in items], slots=args.slots)
            torch.cuda.synchronize()
            timer.toc("registration")
            flows = [w.cpu() - s for (w, _), (s, _, _, _, _) in zip(results, items)]
        else:
            flows = []
            for src, tgt, _, _, ldmk in items:
                model.load_pcds(src, tgt, landmarks=ldmk)
                timer.tic("registration")
                warped, iter_cnt, timer = model.register(visualize=args.visualize, timer=timer)
                timer.toc("registration")
                flows.append((warped - model.src_pcd).cpu())
        meters = None
        for flow, (_, _, flow_gt, overlap, _) in zip(flows, items):
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
        if not (os.path.isdir(root) and args.landmarks):
            break
    if rank == 0:
        print("time cost average")
        for line in timer.get_strings():
            print(line)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
