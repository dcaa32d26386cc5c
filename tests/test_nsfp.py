"""NSFP baseline: the oracle against goldens captured from the reference (CPU), the HIP path
against the oracle and the goldens (GPU).  Reference: model/nets.py:256-292, model/registration.py:470-540."""
import os

import numpy as np
import pytest
import torch

from oracle import ndp_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(seed, scale=1.0):
    from deformationpyramid_amd.nsfp import Neural_Prior
    torch.manual_seed(seed)
    m = Neural_Prior()
    if scale != 1.0:
        for k, v in m.named_parameters():
            if k.endswith("weight"):
                v.mul_(scale)
    return m


def _slice_like(golden_arr, full):
    return full if golden_arr.size == full.size else full.reshape(-1)[::37]


# ------------------------------------------------------------------------------------------ CPU: oracle vs reference
def test_F12_init_consumes_the_rng_like_the_reference(golden):
    g = golden("F12_nsfp")
    m = _model(21)
    names = [k for k, _ in m.named_parameters()]
    assert names == list(g["names"])
    for k, v in m.named_parameters():
        a = v.numpy()
        np.testing.assert_array_equal(a.reshape(-1)[:8], g[f"init.{k}.head"])
        assert abs(a.astype(np.float64).sum() - g[f"init.{k}.sum"]) < 1e-9
        assert abs(np.abs(a.astype(np.float64)).sum() - g[f"init.{k}.abs"]) < 1e-9
    # the permutations that follow continue the same stream (registration.py:494-495)
    from deformationpyramid_amd import nsfp
    assert nsfp.PARAM_COUNT == O.NSFP_P == 116483


def test_F12_oracle_forward_and_gradients_match_the_reference(golden):
    g = golden("F12_nsfp")
    m = _model(21, float(g["fb.scale"]))
    p = m.flat.numpy()
    x, y = g["fb.x"], g["fb.y"]
    out = O.nsfp_fwd(p, x, nthreads=4)
    assert np.abs((out - x) - g["fb.flow"]).max() < 2e-6
    r = O.chamfer(out, y, trunc=1e9)
    assert abs(r["loss"] - g["fb.loss"]) < 2e-6 * g["fb.loss"]
    grads = O.nsfp_bwd(p, x, r["gx"])
    from deformationpyramid_amd import nsfp
    for l, (o, i) in enumerate(nsfp.layer_shapes(), start=1):
        for kind, off, size in (("weight", nsfp.off_W(l), o * i), ("bias", nsfp.off_b(l), o)):
            name = f"layer{l}.{kind}"
            ref = g[f"fb.grad.{name}"].reshape(-1)
            got = _slice_like(ref, grads[off:off + size])
            assert np.abs(got - ref).max() < 2e-4 * np.abs(ref).max() + 1e-9, name
            assert abs(grads[off:off + size].astype(np.float64).sum() - g[f"fb.gsum.{name}"]) < 1e-3 * g[f"fb.gabs.{name}"] + 1e-9


def test_F12_oracle_optimisation_follows_the_reference_trace(golden):
    g = golden("F12_nsfp")
    from deformationpyramid_amd.config import Config
    src, tgt = torch.from_numpy(g["e2e.src"]), torch.from_numpy(g["e2e.tgt"])
    torch.manual_seed(int(g["e2e.seed"]))
    from deformationpyramid_amd.nsfp import Neural_Prior
    m = Neural_Prior()
    sc, tc = src - src.mean(0, keepdim=True), tgt - tgt.mean(0, keepdim=True)
    ps, pt = torch.randperm(src.shape[0]), torch.randperm(tgt.shape[0])
    s, t = sc[ps[:256]].numpy(), tc[pt[:256]].numpy()
    r = O.nsfp_optimize(m.flat.numpy(), s, t, iters=60, max_break_count=70, nthreads=4)
    ref = g["e2e.loss_trace"]
    assert r["steps"] == 60 and len(ref) == 60
    got = r["loss_trace"]
    assert abs(got[0] - ref[0]) < 1e-6 * ref[0]                       # same init, same samples
    assert np.abs(got[:10] - ref[:10]).max() < 2e-3 * ref[0]          # chaotic trajectory: tight early ...
    assert abs(got[-1] - ref[-1]) < 0.1 * ref[-1]                     # ... metric-level at the end
    warped_all = O.nsfp_fwd(r["params"], sc.numpy(), nthreads=4) + tgt.mean(0, keepdim=True).numpy()
    # mid-optimisation snapshot of a chaotic trajectory (the traces agree to 1e-7 for nine steps, then drift apart):
    # mean |difference| stays well below the mean flow magnitude (0.055)
    assert np.abs(warped_all - g["e2e.warped"]).mean() < 0.04


# ------------------------------------------------------------------------------------------ GPU: HIP vs oracle
@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 64, 200, 1000])
def test_nsfp_forward_matches_oracle_and_reference(dev, golden, n):
    from deformationpyramid_amd import nsfp
    g = golden("F12_nsfp")
    m = _model(21, float(g["fb.scale"]))
    gen = torch.Generator().manual_seed(n)
    x = torch.from_numpy(g["fb.x"]) if n == 200 else torch.rand(n, 3, generator=gen) - 0.5
    want = O.nsfp_fwd(m.flat.numpy(), x.numpy(), nthreads=4)
    got, act = nsfp.nsfp_fwd(m.flat.to(dev), x.to(dev), save=True)
    assert np.abs(got.cpu().numpy() - want).max() < 2e-6
    got2 = nsfp.nsfp_fwd(m.flat.to(dev), x.to(dev))                    # inference path (ping-pong scratch)
    assert torch.equal(got, got2)
    if n == 200:
        assert np.abs((got.cpu().numpy() - x.numpy()) - g["fb.flow"]).max() < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("n,n_part", [(200, 2), (1000, 7), (2000, None)])
def test_nsfp_gradients_match_oracle_and_reference(dev, golden, n, n_part):
    from deformationpyramid_amd import nsfp, ops
    g = golden("F12_nsfp")
    m = _model(21, float(g["fb.scale"]))
    gen = torch.Generator().manual_seed(100 + n)
    if n == 200:
        x, y = torch.from_numpy(g["fb.x"]), torch.from_numpy(g["fb.y"])
    else:
        x = torch.rand(n, 3, generator=gen) - 0.5
        y = (torch.rand(n - 13, 3, generator=gen) - 0.5) * 1.1
    p = m.flat.to(dev)
    out, act = nsfp.nsfp_fwd(p, x.to(dev), save=True)
    loss, gx, _ = ops.chamfer_l1(out, y.to(dev), 1e9)
    grads = nsfp.nsfp_bwd(p, x.to(dev), act, gx, n_part=n_part).cpu().numpy()
    ro = O.chamfer(O.nsfp_fwd(m.flat.numpy(), x.numpy(), nthreads=4), y.numpy(), trunc=1e9)
    want = O.nsfp_bwd(m.flat.numpy(), x.numpy(), ro["gx"])
    for l, (o, i) in enumerate(nsfp.layer_shapes(), start=1):
        for kind, off, size in (("weight", nsfp.off_W(l), o * i), ("bias", nsfp.off_b(l), o)):
            a, b = grads[off:off + size], want[off:off + size]
            assert np.abs(a - b).max() < 2e-4 * np.abs(b).max() + 1e-9, (l, kind)        # fp32 summation-order class
            if n == 200:
                ref = g[f"fb.grad.layer{l}.{kind}"].reshape(-1)
                assert np.abs(_slice_like(ref, a) - ref).max() < 3e-4 * np.abs(ref).max() + 1e-9, (l, kind)


@pytest.mark.gpu
def test_nsfp_register_end_to_end_against_the_reference(dev, golden):
    """Registration(config NSFP).register() -> (warped, None), the reference's own end-to-end run as the yardstick."""
    from deformationpyramid_amd.config import Config
    from deformationpyramid_amd.loss import compute_flow_metrics
    from deformationpyramid_amd.registration import Registration
    g = golden("F12_nsfp")
    cfg = Config(dict(deformation_model="NSFP", device=0, gpu_mode=True, iters=60, lr=0.01, max_break_count=70,
                      break_threshold_ratio=0.001, samples=256))
    src, tgt = torch.from_numpy(g["e2e.src"]), torch.from_numpy(g["e2e.tgt"])
    torch.manual_seed(int(g["e2e.seed"]))
    model = Registration(cfg)
    model.load_pcds(src, tgt)
    warped, smpl = model.register()
    assert smpl is None and warped.shape == src.shape and warped.is_cuda
    assert model.last_nsfp["iters"] == 60
    ref_trace = g["e2e.loss_trace"]
    assert abs(model.last_nsfp["loss"] - ref_trace[-1]) < 0.1 * ref_trace[-1]
    assert (warped.cpu() - torch.from_numpy(g["e2e.warped"])).abs().mean().item() < 0.04      # see the oracle test
    mt = compute_flow_metrics(warped.cpu() - src, torch.from_numpy(g["e2e.flow_gt"]), torch.from_numpy(g["e2e.overlap"]))
    ref = dict(zip(g["e2e.metric_keys"], g["e2e.metric_vals"]))
    assert abs(mt["full-epe"] - ref["full-epe"]) < 0.2 * ref["full-epe"]


@pytest.mark.gpu
def test_nsfp_short_run_tracks_the_oracle_step_by_step(dev):
    from deformationpyramid_amd import nsfp, ops
    m = _model(5)
    gen = torch.Generator().manual_seed(8)
    s = torch.rand(500, 3, generator=gen) - 0.5
    t = (torch.rand(470, 3, generator=gen) - 0.5) * 1.05 + 0.02
    r = O.nsfp_optimize(m.flat.numpy(), s.numpy(), t.numpy(), iters=8, early_stop=False, nthreads=4)
    p = m.flat.to(dev)
    mm, vv = torch.zeros(nsfp.PARAM_COUNT, device=dev), torch.zeros(nsfp.PARAM_COUNT, device=dev)
    losses = []
    for it in range(8):
        out, act = nsfp.nsfp_fwd(p, s.to(dev), save=True)
        loss, gx, _ = ops.chamfer_l1(out, t.to(dev), 1e9)
        losses.append(loss.item())
        ops.adam_step(p[:nsfp.PARAM_COUNT], nsfp.nsfp_bwd(p, s.to(dev), act, gx), mm, vv, it + 1)
    assert np.abs(np.array(losses) - r["loss_trace"][:8]).max() < 1e-4 * r["loss_trace"][0]
