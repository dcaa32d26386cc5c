"""What justifies `dtype: f32` for the bf16-split level kernels (engine gemm_mode 7).

Every kernel of one tick is run on its own (ndp_engine_run_stages) in both arithmetic configurations -- the fp32-MFMA
kernels, bitwise the oracle's fma chain, and the kernels that form the 128x128 contractions from three-way bf16 splits
on the bf16 MFMA -- and each output (activations, head outputs, weight / bias gradients, the data gradient dz1) is
compared with a FLOAT64 evaluation of the same kernel on the SAME inputs (the kernel's own input buffers, cast up).
The split path's maximum error must not exceed 1.5x the fp32 chain's: it is fp32 arithmetic in everything but the
summation order.  (/root/reference/model/nets.py:111-140 forward; its autograd for the gradients.)
"""
import numpy as np
import pytest
import torch

from tests._helpers import VARIANTS, seeded_pyramid, scale_heads

pytestmark = pytest.mark.gpu
K0 = -8
RATIO = 1.5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from deformationpyramid_amd import _native
    _native.lib()
    return torch.device("cuda:0")


def _run_tick_by_stages(dev, tag, gemm_mode, S, T, level, head_scale):
    """One tick of a one-pair engine at `level`, stage by stage; returns every buffer a kernel consumed or produced (CPU, fp32)."""
    from deformationpyramid_amd.engine import BatchedEngine, OptConfig
    m = level + 1
    pyr = seeded_pyramid(11, m=m, **VARIANTS[tag])
    for lvl in range(m):
        scale_heads(pyr, lvl, head_scale)                   # head outputs of O(0.01): rotations / translations that matter
    d = pyr.descs[0]
    cfg = OptConfig(m=m, iters=2, early_stop=False)
    eng = BatchedEngine(d, cfg, 1, n_cap=S, t_cap=T, device=dev, gemm_mode=gemm_mode, nn_mode=1)
    g = torch.Generator().manual_seed(3)
    src = (torch.rand(S, 3, generator=g) - 0.5).contiguous()
    tgt = ((torch.rand(T, 3, generator=g) - 0.5) * 1.05 + 0.02).contiguous()
    eng.load(0, src, 0, S, None, tgt, pyr.store)
    for _ in range(2 * level):                               # iters = 2 per level: arrive at `level` with trained lower levels
        eng.run_ticks(1)
    st = eng.read_states()[0]
    assert st.level == level
    P = d.param_count
    out = {"desc": d, "n": S, "params": eng.params[0, level, :P].cpu().clone()}
    eng.run_stages(0, 2)                                      # forward, nearest neighbours, loss / dL/dx'
    torch.cuda.synchronize()
    out["act_fwd"] = eng.act[0].cpu().clone()                 # h0, h1, h2
    out["heads"] = eng.heads[0].cpu().clone()
    out["dO"] = eng.dO[0].cpu().clone()
    eng.run_stages(3, 3)                                      # bwd2
    torch.cuda.synchronize()
    out["dz1"] = eng.act[0, 2].cpu().clone()
    out["g_bwd2"] = eng.gpart[0].double().sum(0)[:P].cpu().clone()
    eng.run_stages(4, 4)                                      # bwd1
    torch.cuda.synchronize()
    out["g_all"] = eng.gpart[0].double().sum(0)[:P].cpu().clone()
    eng.run_stages(5, 5)
    torch.cuda.synchronize()
    return out


def _f64_reference(r):
    """float64 evaluation of each kernel from that kernel's own inputs."""
    d, n, P = r["desc"], r["n"], r["params"].double()
    W = d.width
    nh = d.n_heads
    W0 = P[d.off_W(0):d.off_W(0) + W * 6].view(W, 6); b0 = P[d.off_b(0):d.off_b(0) + W]
    W1 = P[d.off_W(1):d.off_W(1) + W * W].view(W, W); b1 = P[d.off_b(1):d.off_b(1) + W]
    W2 = P[d.off_W(2):d.off_W(2) + W * W].view(W, W); b2 = P[d.off_b(2):d.off_b(2) + W]
    Wh = P[d.off_Wh:d.off_Wh + nh * W].view(nh, W); bh = P[d.off_bh:d.off_bh + nh]
    pe = r["heads"][:n, 16:22].double()                       # the forward's own layer-0 input
    ref = {}
    # ---- forward: pe -> h0, h1, h2, scaled head outputs
    h0 = torch.relu(pe @ W0.T + b0); h1 = torch.relu(h0 @ W1.T + b1); h2 = torch.relu(h1 @ W2.T + b2)
    ref["h0"], ref["h1"], ref["h2"] = h0, h1, h2
    ref["heads"] = d.mlp_scale * (h2 @ Wh.T + bh)
    # ---- bwd2 from ITS inputs: dO, h2, h1 as the forward kernel left them
    dO = r["dO"][:n, :nh].double()
    g_h0, g_h1, g_h2 = (r["act_fwd"][k, :n].double() for k in range(3))
    dz2 = (dO @ Wh) * (g_h2 > 0)
    ref["dWh"], ref["dbh"] = dO.T @ g_h2, dO.sum(0)
    ref["dW2"], ref["db2"] = dz2.T @ g_h1, dz2.sum(0)
    ref["dz1"] = (dz2 @ W2) * (g_h1 > 0)
    # ---- bwd1 from ITS inputs: dz1 as bwd2 left it, h0, pe
    dz1 = r["dz1"][:n].double()
    ref["dW1"], ref["db1"] = dz1.T @ g_h0, dz1.sum(0)
    dz0 = (dz1 @ W1) * (g_h0 > 0)
    ref["dW0"], ref["db0"] = dz0.T @ pe, dz0.sum(0)
    return ref


def _kernel_outputs(r):
    d, n, W, nh = r["desc"], r["n"], r["desc"].width, r["desc"].n_heads
    ga, g2 = r["g_all"], r["g_bwd2"]
    return {
        "h0": r["act_fwd"][0, :n].double(), "h1": r["act_fwd"][1, :n].double(), "h2": r["act_fwd"][2, :n].double(),
        "heads": r["heads"][:n, :nh].double(),
        "dWh": g2[d.off_Wh:d.off_Wh + nh * W].view(nh, W), "dbh": g2[d.off_bh:d.off_bh + nh],
        "dW2": g2[d.off_W(2):d.off_W(2) + W * W].view(W, W), "db2": g2[d.off_b(2):d.off_b(2) + W],
        "dz1": r["dz1"][:n].double(),
        "dW1": ga[d.off_W(1):d.off_W(1) + W * W].view(W, W), "db1": ga[d.off_b(1):d.off_b(1) + W],
        "dW0": ga[d.off_W(0):d.off_W(0) + W * 6].view(W, 6), "db0": ga[d.off_b(0):d.off_b(0) + W],
    }


def _errors(r):
    ref, got = _f64_reference(r), _kernel_outputs(r)
    return {k: float((got[k] - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-300)) for k in ref}


@pytest.mark.parametrize("tag,level", [("se3aa", 0), ("se3aa", 3), ("sim3eu", 1), ("sflow", 2)])
def test_split_kernels_are_as_close_to_float64_as_the_fp32_chain(dev, tag, level):
    S, T = 2000, 2000
    e_chain = _errors(_run_tick_by_stages(dev, tag, 0, S, T, level, 20.0))
    e_split = _errors(_run_tick_by_stages(dev, tag, 7, S, T, level, 20.0))
    report = {k: (e_chain[k], e_split[k]) for k in e_chain}
    for k, (ec, es) in report.items():
        # relative to the tensor's own scale both sit at a few fp32 ulps of a 128- (or 2000-) term sum
        assert ec < 5e-6, (k, ec, report)
        assert es <= RATIO * ec + 2e-8, (k, "split / chain error ratio", es / max(ec, 1e-30), report)
    # outputs that do not pass through a split contraction are the same arithmetic in both configurations
    if level == 0:                                            # (at level 0 both runs see the very same inputs)
        assert e_split["h0"] == e_chain["h0"]
