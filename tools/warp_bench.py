"""Final all-point warp (registration.py:253-258) of 32 clouds x 8192 points x 9 levels: fp32-MFMA kernel vs the fp16-split kernel.
    python tools/warp_bench.py [n_jobs] [points]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deformationpyramid_amd import ops
from deformationpyramid_amd.nets import Deformation_Pyramid

nj = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dev = torch.device("cuda:0")
torch.manual_seed(0)
pyr = Deformation_Pyramid(depth=3, width=128, device="cpu", k0=-8, m=9, rotation_format="axis_angle", motion="SE3")
store = pyr.store.to(dev)
jobs = [(store, (torch.rand(n, 3, device=dev) - 0.5), None, None) for _ in range(nj)]
for split in (False, True):
    for _ in range(3):
        ops.pyramid_fwd_batch(pyr.descs[0], 9, -8, jobs, split=split)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.pyramid_fwd_batch(pyr.descs[0], 9, -8, jobs, split=split)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    fl = 617472.0 * n * nj
    print(f"{'fp16-split k_pyramid_fwd8' if split else 'fp32-MFMA  k_pyramid_fwd '}: {ms:.3f} ms per launch of {nj} clouds, {1e3 * ms / nj:.1f} us per cloud, {fl / ms / 1e9:.1f} TFLOP/s")
