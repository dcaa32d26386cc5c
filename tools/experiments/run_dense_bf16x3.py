"""Build and run the bf16-split layer experiment (tools/experiments/dense_bf16x3.hip) on cuda:0.
    python tools/experiments/run_dense_bf16x3.py [n_tiles] [reps]
The layer is applied 1, 2 and 4 times on the LDS-resident tile (the level kernels keep the activations in LDS between layers:
with one application both variants are bound by the HBM round trip of the tile).
Prints the time of the product's fp32-MFMA layer kernel and of the bf16 x 3 variant on the same activations, and the error
of both against a float64 evaluation."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
out = os.path.join(ROOT, "gpurun_out", "libexp_dense.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                       "-I" + os.path.join(ROOT, "include"), "-o", out, os.path.join(ROOT, "tools", "experiments", "dense_bf16x3.hip")],
                      stderr=subprocess.DEVNULL)
L = ctypes.CDLL(out)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
n = n_tiles * 64
for tag, scale_h, scale_w in (("activations ~U[0,1), Xavier-size weights", 1.0, 0.15), ("gradient-size inputs (1e-6)", 1e-6, 0.15)):
    h = (torch.rand(n, 128, generator=g) * scale_h).to(dev)
    W = ((torch.rand(128, 128, generator=g) - 0.5) * 2 * scale_w).to(dev)
    b = ((torch.rand(128, generator=g) - 0.5) * 0.1 * scale_h).to(dev)
    print(f"{tag}: n = {n} points")
    for layers in (1, 2, 4):
        y32 = torch.empty(n, 128, device=dev); y16 = torch.empty(n, 128, device=dev); y8 = torch.empty(n, 128, device=dev)
        ms = (ctypes.c_float * 3)()
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        rc = L.exp_dense_run(p(W), p(b), p(h), p(y32), p(y16), p(y8), n_tiles, layers, reps, ms)
        torch.cuda.synchronize()
        assert rc == 0, rc
        ref = h[:65536].double()
        for _ in range(layers):
            ref = torch.relu(ref @ W.double().T + b.double())
        s = ref.abs().max().item()
        e32 = (y32[:65536].double() - ref).abs().max().item() / s
        e16 = (y16[:65536].double() - ref).abs().max().item() / s
        e8 = (y8[:65536].double() - ref).abs().max().item() / s
        flop = 2.0 * n * 128 * 128 * layers
        print(f"  {layers} layer(s): fp32 MFMA {ms[0]:.4f} ms {flop / ms[0] / 1e9:6.1f} TFLOP/s err {e32:.2e} | bf16 x 3 {ms[1]:.4f} ms "
              f"{flop / ms[1] / 1e9:6.1f} TFLOP/s err {e16:.2e} | ratio {ms[0] / ms[1]:.2f} | 8 waves x 16 cols {ms[2]:.4f} ms err {e8:.2e} ratio {ms[0] / ms[2]:.2f}")

# ---- weight gradient: dW = dz^T h over all tiles (per-workgroup partials summed here)
dz = ((torch.rand(n, 128, generator=g) - 0.5) * 2e-4).to(dev)
h = torch.rand(n, 128, generator=g).to(dev)
h = torch.where(torch.rand(n, 128, generator=g).to(dev) < 0.5, torch.zeros_like(h), h)        # post-ReLU sparsity
g32 = torch.zeros(512, 128, 128, device=dev); g16 = torch.zeros(256, 128, 128, device=dev)
ref = dz.double().T @ h.double()
s = ref.abs().max().item()
flop = 2.0 * n * 128 * 128
for rep in (1, 2, 4):
    ms = (ctypes.c_float * 2)()
    rc = L.exp_wgrad_run(p(dz), p(h), p(g32), p(g16), n_tiles, rep, reps, ms)
    torch.cuda.synchronize()
    assert rc == 0, rc
    e32 = (g32.double().sum(0) / rep - ref).abs().max().item() / s
    e16 = (g16.double().sum(0) / rep - ref).abs().max().item() / s
    print(f"weight gradient dz^T h x {rep} on the resident tile, n = {n}: fp32 MFMA outer product {ms[0]:.4f} ms {rep * flop / ms[0] / 1e9:6.1f} TFLOP/s "
          f"err {e32:.2e} | bf16 x 3 with transposing LDS reads {ms[1]:.4f} ms {rep * flop / ms[1] / 1e9:6.1f} TFLOP/s err {e16:.2e} | ratio {ms[0] / ms[1]:.2f}")
