import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from deformationpyramid_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
for (S, T, scale) in ((32, 32, 1.0), (64, 32, 1.0), (96, 32, 1.0), (128, 32, 1.0), (128, 64, 1.0), (32, 256, 1.0)):
    x = ((torch.rand(S, 3, generator=g) - 0.5) * scale).to(dev); y = ((torch.rand(T, 3, generator=g) - 0.5) * scale).to(dev)
    a = [t.cpu() for t in ops.chamfer_nn(x, y)]
    b = [t.cpu() for t in ops.chamfer_nn_onepass(x, y, matrix=True)]
    bad = [(int((a[k] != b[k]).sum())) for k in range(4)]
    print(S, T, "mismatches d2x idx_x d2y idx_y:", bad)
    if bad[1]:
        print("  bad sources:", (a[1] != b[1]).nonzero().flatten().tolist()[:40])
        i = int((a[1] != b[1]).nonzero()[0]); print("  first bad source", i, "ref", a[0][i].item(), a[1][i].item(), "got", b[0][i].item(), b[1][i].item())
    if bad[3]:
        print("  bad targets:", (a[3] != b[3]).nonzero().flatten().tolist()[:40])
        j = int((a[3] != b[3]).nonzero()[0]); print("  first bad target", j, "ref", a[2][j].item(), a[3][j].item(), "got", b[2][j].item(), b[3][j].item())
