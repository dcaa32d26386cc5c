"""Nerfies comparison baseline (second half): the torch-CPU restatement against golden F14 captured
from the reference, and the HIP path against both."""
import os

import numpy as np
import pytest
import torch

from oracle import nerfies_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["input.0.weight", "input.0.bias"] + [f"mlp.pts_linears.{i}.{k}" for i in range(6) for k in ("weight", "bias")] + \
        ["w_branch.weight", "w_branch.bias", "v_branch.weight", "v_branch.bias"]


def ref_flat(seed):
    """The reference's initial parameters in the product's flat layout: torch.nn.Linear modules constructed on the CPU
    generator in Nerfies_Deformation's order (nets.py:195-201)."""
    from deformationpyramid_amd.nerfies import Nerfies_Deformation
    torch.manual_seed(seed)
    return Nerfies_Deformation(max_iter=5000)


def grads_by_name(net, gflat):
    return {k: g for (k, _), g in zip(net.named_parameters(), net.split_like(gflat))}


def test_init_replays_the_reference_rng_stream(golden):
    g = golden("F14_nerfies")
    net = ref_flat(23)
    assert [k for k, _ in net.named_parameters()] == list(g["names"]) == NAMES
    for k, v in net.named_parameters():
        a = v.numpy().astype(np.float64)
        assert abs(a.sum() - float(g[f"init.{k}.sum"])) < 1e-9 + 1e-6 * float(g[f"init.{k}.abs"]), k
        np.testing.assert_array_equal(v.numpy().reshape(-1)[:8], g[f"init.{k}.head"])


@pytest.mark.parametrize("it", [0, 700, 2999])
def test_oracle_forward_jacobian_regulariser_and_gradients(golden, it):
    g = golden("F14_nerfies")
    net = ref_flat(23)
    flat = net.flat[:R.P_COUNT].clone().requires_grad_(True)
    x, y = torch.from_numpy(g["fb.x"]), torch.from_numpy(g["fb.y"])
    warped, J, pe = R.forward(flat, x, it, 5000)
    np.testing.assert_allclose(pe.numpy(), g[f"fb{it}.pe"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(warped.detach().numpy(), g[f"fb{it}.warped"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(J.numpy(), g[f"fb{it}.J"], rtol=0, atol=2e-4)
    reg = R.regularization(J)
    assert abs(reg.item() - float(g[f"fb{it}.reg"])) < 1e-3 * float(g[f"fb{it}.reg"]) + 1e-9
    cd = R.chamfer_l1(warped, y)
    assert abs(cd.item() - float(g[f"fb{it}.cd"])) < 2e-6 * float(g[f"fb{it}.cd"])
    assert int(g[f"fb{it}.J_requires_grad"]) == 0 and not J.requires_grad      # upstream: the regulariser carries no gradient
    (cd + 0.001 * reg).backward()
    for k, gr in grads_by_name(net, flat.grad).items():
        ref = g[f"fb{it}.grad.{k}"].reshape(-1)
        got = gr.numpy().reshape(-1)
        got = got if got.size <= 4992 else got[::37]
        assert np.abs(got - ref).max() < 2e-4 * (np.abs(ref).max() + 1e-12) + 1e-9, k
        assert abs(gr.numpy().astype(np.float64).sum() - float(g[f"fb{it}.gsum.{k}"])) < 1e-3 * float(g[f"fb{it}.gabs.{k}"]) + 1e-9, k


def test_oracle_end_to_end_follows_the_reference_trace(golden):
    g = golden("F14_nerfies")
    src, tgt = torch.from_numpy(g["e2e.src"]), torch.from_numpy(g["e2e.tgt"])
    torch.manual_seed(int(g["e2e.seed"]))
    from deformationpyramid_amd.nerfies import Nerfies_Deformation
    net = Nerfies_Deformation(max_iter=40)
    src_c, tgt_c = src - src.mean(0, keepdim=True), tgt - tgt.mean(0, keepdim=True)
    s = src_c[torch.randperm(src.shape[0])[:256]]
    t = tgt_c[torch.randperm(tgt.shape[0])[:256]]
    flat, last, trace = R.optimize(net.flat[:R.P_COUNT], s, t, iters=40)
    cd = np.array([c for c, _ in trace])
    ref = g["e2e.cd_trace"]
    assert len(cd) == len(ref)
    assert abs(cd[0] - ref[0]) < 2e-6 * ref[0]
    assert np.abs(cd[:8] - ref[:8]).max() < 2e-3 * ref[:8].max()              # Adam on unscaled heads: chaos sets in early
    reg = np.array([r for _, r in trace])
    assert abs(reg[0] - g["e2e.reg_trace"][0]) < 2e-3 * g["e2e.reg_trace"][0]


# ------------------------------------------------------------------------------------------------ HIP path (GPU)
@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("it", [0, 700, 2999])
def test_hip_forward_jacobian_regulariser_and_gradients(dev, golden, it):
    """ndp_nerfies_fwd / ndp_nerfies_bwd against the reference's golden (and hence the oracle): warp, per-point Jacobian
    from the forward tangent rows, regulariser value, and the parameter gradients of cd + 0.001 reg."""
    from deformationpyramid_amd import nerfies as NF
    from deformationpyramid_amd import ops
    g = golden("F14_nerfies")
    net = ref_flat(23).to(dev)
    x, y = torch.from_numpy(g["fb.x"]).to(dev), torch.from_numpy(g["fb.y"]).to(dev)
    warped, J, reg, saved = NF.nerfies_fwd(net.flat, x, it, 5000, save=True)
    np.testing.assert_allclose(saved[1][:x.shape[0], :39].cpu().numpy(), g[f"fb{it}.pe"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(warped.cpu().numpy(), g[f"fb{it}.warped"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(J.cpu().numpy(), g[f"fb{it}.J"], rtol=0, atol=5e-4)
    assert abs(reg.item() - float(g[f"fb{it}.reg"])) < 5e-3 * float(g[f"fb{it}.reg"]) + 1e-9
    w2, J2, reg2 = NF.nerfies_fwd(net.flat, x, it, 5000)                      # inference form (two ping-pong planes)
    assert torch.equal(w2, warped) and torch.equal(J2, J)
    cd, gx, _ = ops.chamfer_l1(warped, y, 1e9)
    assert abs(cd.item() - float(g[f"fb{it}.cd"])) < 1e-5 * float(g[f"fb{it}.cd"])
    grads = NF.nerfies_bwd(net.flat, x, saved, gx).cpu()
    for k, gr in grads_by_name(net, grads).items():
        ref = g[f"fb{it}.grad.{k}"].reshape(-1)
        got = gr.numpy().reshape(-1)
        got = got if got.size <= 4992 else got[::37]
        assert np.abs(got - ref).max() < 1e-3 * (np.abs(ref).max() + 1e-12) + 1e-9, k
        assert abs(gr.numpy().astype(np.float64).sum() - float(g[f"fb{it}.gsum.{k}"])) < 2e-3 * float(g[f"fb{it}.gabs.{k}"]) + 1e-9, k


@pytest.mark.gpu
def test_hip_register_follows_the_reference_trace(dev, golden):
    """config.deformation_model = Nerfies through Registration.register(): (warped, None) like upstream, the evaluated
    (cd, reg) pairs follow the reference's trace before trajectory chaos sets in."""
    from deformationpyramid_amd.config import Config
    from deformationpyramid_amd.registration import Registration
    g = golden("F14_nerfies")
    cfg = Config(dict(deformation_model="Nerfies", device=0, iters=40, lr=0.01, max_break_count=70,
                      break_threshold_ratio=0.001, samples=256))
    torch.manual_seed(int(g["e2e.seed"]))
    model = Registration(cfg)
    model.load_pcds(g["e2e.src"], g["e2e.tgt"])
    warped, second = model.register()
    assert second is None and warped.shape == (g["e2e.src"].shape[0], 3) and torch.isfinite(warped).all()
    tr = model.last_nerfies["trace"]
    ref_cd, ref_reg = g["e2e.cd_trace"], g["e2e.reg_trace"]
    assert len(tr) == len(ref_cd)
    assert abs(tr[0][0] - ref_cd[0]) < 1e-5 * ref_cd[0] and abs(tr[0][1] - ref_reg[0]) < 5e-3 * ref_reg[0]
    assert np.abs(np.array([c for c, _ in tr[:8]]) - ref_cd[:8]).max() < 5e-3 * ref_cd[:8].max()
    assert np.abs(warped.cpu().numpy() - g["e2e.warped"]).mean() < 0.05
