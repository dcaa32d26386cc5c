"""GPU parity tests: the HIP path (through the C ABI of libndp_hip.so) against the CPU oracle on the
same seeded inputs, and against the golden vectors captured from the reference.

Bar (BASELINE.json north_star): integer/index results bit-exact; floating point within 1e-4 on
warped coordinates -- tolerances are written at each assert.
"""
import os

import numpy as np
import pytest
import torch

from tests._helpers import VARIANTS, engine_modes, registration_modes, seeded_pyramid, scale_heads, rel_err, flat_from_named

pytestmark = pytest.mark.gpu
K0 = -8


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from deformationpyramid_amd import _native
    _native.lib()            # must load: no fallback
    return torch.device("cuda:0")


def O():
    from oracle import ndp_oracle
    return ndp_oracle


def cdesc(d):
    return O().make_desc(d.width, d.n_hidden, d.motion, d.rotfmt, d.nonrigidity, d.mlp_scale)


def cloud(n, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(n, 3, generator=g) - 0.5) * scale).contiguous()


# ------------------------------------------------------------------------------- level forward
@pytest.mark.parametrize("tag", list(VARIANTS))
@pytest.mark.parametrize("n", [1, 63, 256, 1000])
def test_level_fwd_matches_oracle(dev, tag, n):
    from deformationpyramid_amd import ops
    pyr = seeded_pyramid(5, **VARIANTS[tag])
    x = cloud(n, 17)
    for lvl in (0, 3, 8):
        scale_heads(pyr, lvl, 30.0)
        d = pyr.descs[lvl]
        p = pyr.store[lvl].clone()
        ref = O().level_fwd(cdesc(d), p[:d.param_count].numpy(), lvl, K0, x.numpy())
        got = ops.level_fwd(d, p.to(dev), lvl, K0, x.to(dev)).cpu().numpy()
        # warped coordinates, abs; the 6D / quaternion heads normalise their raw outputs, which amplifies the
        # summation-order round-off of the head dot products (still 10x inside the 1e-4 budget)
        tol = 1e-5 if ("6d" in tag or "quat" in tag) else 2e-6
        assert np.abs(got - ref).max() < tol, (tag, lvl, n)


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_level_fwd_golden_from_reference(dev, golden, tag):
    from deformationpyramid_amd import ops
    g = golden("F2_layer_forward")
    pyr = seeded_pyramid(int(g["seed"]), **VARIANTS[tag])
    levels = (0, 4, 8) if tag in ("se3aa", "sim3eu") else (4,)
    x = torch.from_numpy(g["x"]).to(dev)
    for lvl in levels:
        scale_heads(pyr, lvl, float(g["head_scale"]))
    for lvl in levels:
        d = pyr.descs[lvl]
        got = ops.level_fwd(d, pyr.store[lvl].to(dev), lvl, K0, x).cpu().numpy()
        assert np.abs(got - g[f"{tag}.L{lvl}.out"]).max() < 1e-5
    full = ops.pyramid_fwd(pyr.descs[0], 9, K0, pyr.store.to(dev), x).cpu().numpy()
    assert np.abs(full - g[f"{tag}.full_out"]).max() < 1e-4          # north_star tolerance on warped coords


# ------------------------------------------------------------------------------ level backward
@pytest.mark.parametrize("tag", list(VARIANTS))
@pytest.mark.parametrize("n,n_part", [(64, 1), (1000, 3), (1000, 16), (2000, 8)])
def test_level_bwd_matches_oracle(dev, tag, n, n_part):
    from deformationpyramid_amd import ops
    pyr = seeded_pyramid(6, **VARIANTS[tag])
    lvl = 4
    scale_heads(pyr, lvl, 30.0)
    d = pyr.descs[lvl]
    x = cloud(n, 23)
    gsrc = cloud(n, 29, scale=2.0)
    p = pyr.store[lvl].clone()
    ref = O().level_bwd(cdesc(d), p[:d.param_count].numpy(), lvl, K0, x.numpy(), gsrc.numpy(), nthreads=4)
    out, act, heads = ops.level_fwd(d, p.to(dev), lvl, K0, x.to(dev), save=True)
    got = ops.level_bwd(d, p.to(dev), lvl, K0, x.to(dev), act, heads, gsrc.to(dev), n_part=n_part).cpu().numpy()
    for name, off, shape in d.named_slices():
        sz = int(np.prod(shape))
        e = rel_err(got[off:off + sz], ref[off:off + sz])
        assert e < 1e-4, (tag, name, e)                                # fp32 summation-order class


def test_level_bwd_golden_from_reference(dev, golden):
    from deformationpyramid_amd import ops
    g = golden("F2_layer_forward")
    x = torch.from_numpy(g["x"]).to(dev)
    coef = torch.linspace(-1.0, 1.0, 256 * 3).reshape(256, 3).to(dev)
    for tag in ("se3aa", "sim3eu", "sflow"):
        pyr = seeded_pyramid(int(g["seed"]), **VARIANTS[tag])
        lvl = 4
        scale_heads(pyr, lvl, float(g["head_scale"]))
        d = pyr.descs[lvl]
        p = pyr.store[lvl].to(dev)
        _, act, heads = ops.level_fwd(d, p, lvl, K0, x, save=True)
        got = ops.level_bwd(d, p, lvl, K0, x, act, heads, coef).cpu().numpy()
        for name, off, shape in d.named_slices():
            ref = g[f"{tag}.L{lvl}.grad.{name}"]
            e = rel_err(got[off:off + ref.size].reshape(ref.shape), ref)
            assert e < 2e-4, (tag, name, e)


# ------------------------------------------------------------------------------------ Chamfer
@pytest.mark.parametrize("S,T", [(300, 257), (1, 5), (2000, 2000), (1025, 3000)])
def test_chamfer_nn_bit_exact(dev, S, T):
    from deformationpyramid_amd import ops
    x, y = cloud(S, 31), cloud(T, 37, scale=1.1)
    r = O().chamfer(x.numpy(), y.numpy(), nthreads=4)
    d2x, ix, d2y, iy = [t.cpu().numpy() for t in ops.chamfer_nn(x.to(dev), y.to(dev))]
    np.testing.assert_array_equal(ix, r["idx_x"])
    np.testing.assert_array_equal(iy, r["idx_y"])
    np.testing.assert_array_equal(d2x, r["d2x"])                       # same fma chain -> same bits
    np.testing.assert_array_equal(d2y, r["d2y"])


def test_chamfer_nn_ties_take_lowest_index(dev):
    from deformationpyramid_amd import ops
    y = cloud(700, 41)
    y = torch.cat([y, y, y[:100]]).contiguous()                         # every reference point duplicated
    x = cloud(333, 43)
    _, ix, _, iy = ops.chamfer_nn(x.to(dev), y.to(dev))
    r = O().chamfer(x.numpy(), y.numpy())
    np.testing.assert_array_equal(ix.cpu().numpy(), r["idx_x"])
    assert ix.max().item() < 700                                        # never a later duplicate
    np.testing.assert_array_equal(iy.cpu().numpy(), r["idx_y"])


@pytest.mark.parametrize("trunc", [1e9, 0.01])
def test_chamfer_loss_and_grad(dev, golden, trunc):
    from deformationpyramid_amd import ops
    g = golden("F3_chamfer")
    tag = "full" if trunc > 1 else "trunc"
    x, y = torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["y"]).to(dev)
    loss, gx, nn = ops.chamfer_l1(x, y, trunc)
    np.testing.assert_array_equal(nn[1].cpu().numpy(), g[f"{tag}.idx_x"])
    np.testing.assert_array_equal(nn[3].cpu().numpy(), g[f"{tag}.idx_y"])
    assert abs(loss.item() - float(g[f"{tag}.loss"])) < 2e-6 * float(g[f"{tag}.loss"])
    ref = g[f"{tag}.grad_x"]
    assert np.abs(gx.cpu().numpy() - ref).max() < 2e-6 * np.abs(ref).max()
    r = O().chamfer(g["x"], g["y"], trunc=trunc)
    assert np.abs(gx.cpu().numpy() - r["gx"]).max() < 1e-7 * np.abs(ref).max() + 1e-12


def test_chamfer_full_size_properties(dev):
    """8192 x 8192 (BASELINE stress size): symmetry and idempotence properties, no oracle needed."""
    from deformationpyramid_amd import ops
    x, y = cloud(8192, 51).to(dev), cloud(8192, 53).to(dev)
    d2x, ix, d2y, iy = ops.chamfer_nn(x, y)
    d2x_b, ix_b, d2y_b, iy_b = ops.chamfer_nn(y, x)                      # swap roles
    assert torch.equal(d2x, d2y_b) and torch.equal(ix, iy_b) and torch.equal(d2y, d2x_b) and torch.equal(iy, ix_b)
    # the reported distance is the distance to the reported index, and nothing is closer
    rec = ((x - y[ix.long()]) ** 2).sum(-1)
    assert torch.allclose(rec, d2x, rtol=1e-5, atol=0)
    full = torch.cdist(x[:512].double(), y.double()) ** 2
    assert torch.all(full.min(dim=1).values.float() >= d2x[:512] * (1 - 1e-5))
    # self-match: a cloud against itself gives zero distances and identity indices
    d0, i0, _, _ = ops.chamfer_nn(x, x.clone())
    assert torch.all(d0 == 0) and torch.equal(i0.long(), torch.arange(8192, device=dev))


def test_landmark_and_adam_bit_exact(dev):
    from deformationpyramid_amd import ops
    x, t = cloud(200, 61), cloud(200, 67)
    L, gx = ops.landmark_mse(x.to(dev), t.to(dev))
    Lr, gr = O().landmark(x.numpy(), t.numpy())
    assert abs(L.item() - float(Lr)) < 1e-6 * float(Lr)
    np.testing.assert_array_equal(gx.cpu().numpy(), gr)
    P = 34694
    g = torch.Generator().manual_seed(71)
    p = torch.randn(P, generator=g); gr_ = torch.randn(P, generator=g) * 1e-3
    m = torch.zeros(P); v = torch.zeros(P)
    pd, md, vd = p.to(dev), m.to(dev), v.to(dev)
    pn, mn, vn = p.numpy().copy(), m.numpy().copy(), v.numpy().copy()
    for step in range(1, 6):
        gi = gr_ * step
        ops.adam_step(pd, gi.to(dev), md, vd, step)
        O().adam(pn, gi.numpy(), mn, vn, step)
    np.testing.assert_array_equal(pd.cpu().numpy(), pn)                 # op-for-op the same arithmetic
    np.testing.assert_array_equal(vd.cpu().numpy(), vn)


# ----------------------------------------------------------------------------- batched engine
def _engine_vs_oracle(dev, tag, K, S, T, m, iters, early_stop, w_cd, trunc, B=3, seed=7, G=None, ratio=0.001, nn_mode=None, gemm_mode=None, arith=None):
    from deformationpyramid_amd.engine import BatchedEngine, OptConfig
    kw = VARIANTS[tag]
    cfg = OptConfig(m=m, iters=iters, early_stop=early_stop, w_cd=w_cd, trunc=trunc, break_threshold_ratio=ratio)
    eng = None
    refs = []
    modes = engine_modes(arith, K + S, nn_mode) if arith is not None else dict(nn_mode=nn_mode, gemm_mode=gemm_mode)
    for b in range(B):
        pyr = seeded_pyramid(seed + b, m=m, **kw)
        d = pyr.descs[0]
        if eng is None:
            eng = BatchedEngine(d, cfg, B, n_cap=K + S, t_cap=max(T, 1), device=dev, G=G, **modes)
        # slots of different sizes: slot b drops 7*b samples and 3*b targets
        Kb, Sb, Tb = K, max(S - 7 * b, 0), max(T - 3 * b, 0)
        src = cloud(Kb + Sb, 100 + b)
        c, s_ = np.cos(0.2), np.sin(0.2)
        Rz = torch.tensor([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
        tgt = (cloud(Tb, 200 + b) @ Rz.T + torch.tensor([0.03, -0.02, 0.01])).contiguous() if Tb else None
        lt = ((src[:Kb] + 0.04 * torch.sin(4 * src[:Kb])) @ Rz.T).contiguous() if Kb else None
        eng.load(b, src, Kb, Sb, lt, tgt, pyr.store)
        params_all = np.concatenate([pyr.store[i, :d.param_count].numpy() for i in range(m)])
        refs.append(O().optimize([cdesc(d)] * m, params_all, src.numpy(), Kb, Sb,
                                 lt.numpy() if Kb else None, tgt.numpy() if Tb else None, k0=K0, iters=iters,
                                 w_cd=w_cd, trunc=trunc, early_stop=early_stop, nthreads=4, ratio=ratio))
    states = eng.run_until_done(chunk=8)
    return eng, states, refs


@pytest.mark.parametrize("tag", ["se3aa", "sim3eu", "sflow"])
def test_engine_fixed_work_matches_oracle(dev, tag, arith):
    """Early stop off, 6 iterations x 2 levels: same number of steps, parameters and points agree."""
    eng, states, refs = _engine_vs_oracle(dev, tag, K=0, S=300, T=280, m=2, iters=6, early_stop=False, w_cd=1.0, trunc=1e9, arith=arith)
    P = eng.P
    for b, (st, ref) in enumerate(zip(states, refs)):
        assert st.level == 2 and list(st.evals_per_level[:2]) == [6, 6] and st.total_steps == 12
        assert abs(st.loss - ref["loss_trace"][-1]) < 1e-4 * abs(ref["loss_trace"][-1])
        got = eng.params[b, :, :P].cpu().numpy().reshape(-1)
        # Adam's first steps move every weight by ~lr regardless of |g|: weights whose gradient is
        # round-off noise may differ by O(lr); compare the bulk and the warped points.
        diff = np.abs(got - ref["params_all"])
        assert np.mean(diff < 1e-4) > 0.97, np.mean(diff < 1e-4)
        pts = eng.final_points(b, st).cpu().numpy()
        assert np.abs(pts - ref["pts"]).max() < 1e-4                     # north_star: warped coordinates


def test_engine_early_stop_counts_match_oracle(dev, arith):
    eng, states, refs = _engine_vs_oracle(dev, "se3aa", K=0, S=256, T=256, m=3, iters=60, early_stop=True,
                                          w_cd=1.0, trunc=1e9, ratio=0.01, arith=arith)
    for b, (st, ref) in enumerate(zip(states, refs)):
        assert st.level == 3
        assert list(st.evals_per_level[:3]) == list(ref["iters_per_level"]), (b, list(st.evals_per_level[:3]), ref["iters_per_level"])
        pts = eng.final_points(b, st).cpu().numpy()
        assert np.abs(pts - ref["pts"]).max() < 5e-4


def test_engine_landmark_only_and_mixed(dev, arith):
    eng, states, refs = _engine_vs_oracle(dev, "se3aa", K=150, S=0, T=0, m=2, iters=5, early_stop=False, w_cd=0.0, trunc=0.25, arith=arith)
    for b, (st, ref) in enumerate(zip(states, refs)):
        assert st.total_steps == 10
        assert abs(st.loss - ref["loss_trace"][-1]) < 1e-4 * abs(ref["loss_trace"][-1])
        assert np.abs(eng.final_points(b, st).cpu().numpy() - ref["pts"]).max() < 1e-4
    eng, states, refs = _engine_vs_oracle(dev, "se3aa", K=70, S=200, T=222, m=2, iters=4, early_stop=False, w_cd=0.5, trunc=0.05, arith=arith)
    for b, (st, ref) in enumerate(zip(states, refs)):
        assert abs(st.loss - ref["loss_trace"][-1]) < 1e-4 * abs(ref["loss_trace"][-1])
        assert np.abs(eng.final_points(b, st).cpu().numpy() - ref["pts"]).max() < 1e-4


def test_F9c_engine_mixed_objective_against_the_reference(dev, golden, arith):
    """The mixed landmark + truncated-Chamfer objective (registration.py:189-197) through the HIP engine, against the
    trace captured from the reference: loss of evaluations 1..8 and the parameters after three Adam steps."""
    from deformationpyramid_amd.engine import BatchedEngine, OptConfig
    g = golden("F9c_mixed")
    K, S, T = g["src_ldmk"].shape[0], g["s_sample"].shape[0], g["t_sample"].shape[0]
    pts = torch.from_numpy(np.concatenate([g["src_ldmk"], g["s_sample"]]))
    ref = g["losses"]
    pyr = seeded_pyramid(int(g["seed"]), **VARIANTS["se3aa"])
    d = pyr.descs[0]
    for n_eval in (1, 4, 8):
        cfg = OptConfig(m=1, iters=n_eval, early_stop=False, w_cd=float(g["w_cd"]), trunc=float(g["trunc"]))
        eng = BatchedEngine(d, cfg, 1, n_cap=K + S, t_cap=T, device=dev, **engine_modes(arith, K + S))
        eng.load(0, pts, K, S, torch.from_numpy(g["tgt_ldmk"]), torch.from_numpy(g["t_sample"]), pyr.store[:1])
        st = eng.run_until_done(chunk=n_eval)[0]
        assert st.total_evals == n_eval
        assert abs(st.loss - ref[n_eval - 1]) < (2e-6 if n_eval == 1 else 1e-3) * ref[n_eval - 1], (n_eval, st.loss, ref[n_eval - 1])
    cfg = OptConfig(m=1, iters=3, early_stop=False, w_cd=float(g["w_cd"]), trunc=float(g["trunc"]))
    eng = BatchedEngine(d, cfg, 1, n_cap=K + S, t_cap=T, device=dev, **engine_modes(arith, K + S))
    eng.load(0, pts, K, S, torch.from_numpy(g["tgt_ldmk"]), torch.from_numpy(g["t_sample"]), pyr.store[:1])
    eng.run_until_done(chunk=3)
    got = eng.params[0, 0, :d.param_count].cpu().numpy()
    for name, off, shape in d.named_slices():
        ref3, r0 = g[f"step3.{name}"], g[f"grad0.{name}"]
        mask = np.abs(r0) > 1e-3 * np.abs(r0).max()
        assert np.abs(got[off:off + ref3.size].reshape(ref3.shape) - ref3)[mask].max() < 2e-4, name


def test_F13_shape_transfer_on_the_engine_against_the_reference(dev, golden, arith):
    """BASELINE config 4 against reference data: Sim3 / euler, 6000 + 6000 seeded mesh vertices, ten iterations of level 0
    (loss of evaluations 1 and 10), then all 24 856 source vertices through the nine levels (1e-4 on coordinates)."""
    from deformationpyramid_amd import ops
    from deformationpyramid_amd.engine import BatchedEngine, OptConfig
    g = golden("F13_shape_transfer")
    pyr = seeded_pyramid(0, **VARIANTS["sim3eu"])
    d = pyr.descs[0]
    S, T = g["s_sample"].shape[0], g["t_sample"].shape[0]
    ref = g["losses"]
    for n_eval in (1, 10):
        eng = BatchedEngine(d, OptConfig(m=1, iters=n_eval, early_stop=False), 1, n_cap=S, t_cap=T, device=dev, **engine_modes(arith, S))
        eng.load(0, torch.from_numpy(g["s_sample"]), 0, S, None, torch.from_numpy(g["t_sample"]), pyr.store[:1])
        st = eng.run_until_done(chunk=n_eval)[0]
        assert abs(st.loss - ref[n_eval - 1]) < (2e-6 if n_eval == 1 else 2e-3) * ref[n_eval - 1], (n_eval, st.loss, ref[n_eval - 1])
    store = pyr.store.clone()
    named = {name: g[f"final.{name}"] for name, _, _ in d.named_slices()}
    store[0, :d.param_count] = torch.from_numpy(flat_from_named(d, named))
    got = ops.pyramid_fwd(d, 9, K0, store.to(dev).contiguous(), torch.from_numpy(np.ascontiguousarray(g["mesh_vert"])).to(dev)).cpu().numpy()
    assert got.shape == (24856, 3) and np.abs(got - g["warped_vert"]).max() < 1e-4


def test_flow_metrics_on_the_device_match_the_reference_golden(dev, golden):
    """compute_flow_metrics on GPU tensors (k_flow_metrics: one launch, 15 sums) against golden F8 captured from the
    reference, against the torch-CPU path, and the NaN of an empty subset (loss.py:461)."""
    from deformationpyramid_amd.loss import compute_flow_metrics
    g = golden("F8_metrics")
    flow, gt, ov = torch.from_numpy(g["flow"]), torch.from_numpy(g["flow_gt"]), torch.from_numpy(g["overlap"])
    got = compute_flow_metrics(flow.to(dev), gt.to(dev), ov.to(dev))
    cpu = compute_flow_metrics(flow, gt, ov)
    for k, v in zip(g["keys"], g["vals"]):
        assert abs(got[str(k)] - float(v)) < 1e-4 * max(1.0, abs(float(v))), k
        assert abs(got[str(k)] - cpu[str(k)]) < 1e-4 * max(1.0, abs(cpu[str(k)])), k
    only_full = compute_flow_metrics(flow.to(dev), gt.to(dev))
    assert list(only_full) == ["full-epe", "full-AccS", "full-AccR", "full-outlier"] and only_full["full-epe"] == got["full-epe"]
    empty = compute_flow_metrics(flow.to(dev), gt.to(dev), torch.ones_like(ov).to(dev))
    assert np.isnan(empty["occ-epe"]) and empty["vis-epe"] == got["full-epe"]


def test_chamfer_point_reduction_sum(dev):
    """point_reduction="sum" (loss.py:233-235): the two directions' sums without the division by their point counts --
    value and gradient against a plain torch statement of the same formula."""
    from deformationpyramid_amd.loss import compute_truncated_chamfer_distance
    x, y = cloud(300, 5), cloud(257, 6) * 1.1 + 0.02
    xr = x.clone().requires_grad_(True)
    dx = ((xr[:, None] - y[None]) ** 2).sum(-1)
    ref = (dx.min(1)[0].sqrt().sum() + dx.min(0)[0].sqrt().sum())
    ref.backward()
    xg = x.to(dev).requires_grad_(True)
    got = compute_truncated_chamfer_distance(xg[None], y.to(dev)[None], trunc=1e9, point_reduction="sum")
    got.backward()
    assert abs(got.item() - ref.item()) < 1e-5 * ref.item()
    assert (xg.grad.cpu() - xr.grad).abs().max().item() < 1e-4
    mean = compute_truncated_chamfer_distance(x.to(dev)[None], y.to(dev)[None], trunc=1e9)
    assert mean.item() < got.item() / 200


def test_engine_is_deterministic_and_G_independent_in_loss(dev, arith):
    e1, s1, _ = _engine_vs_oracle(dev, "se3aa", K=0, S=500, T=400, m=2, iters=5, early_stop=False, w_cd=1.0, trunc=1e9, B=2, G=2, arith=arith)
    e2, s2, _ = _engine_vs_oracle(dev, "se3aa", K=0, S=500, T=400, m=2, iters=5, early_stop=False, w_cd=1.0, trunc=1e9, B=2, G=2, arith=arith)
    assert torch.equal(e1.params, e2.params)                              # bit-reproducible run to run
    e3, s3, _ = _engine_vs_oracle(dev, "se3aa", K=0, S=500, T=400, m=2, iters=5, early_stop=False, w_cd=1.0, trunc=1e9, B=2, G=8, arith=arith)
    assert abs(s1[0].loss - s3[0].loss) < 1e-5 * abs(s1[0].loss)


def test_adam_behind_the_backward_is_bitwise_the_update_stage(dev):
    """G = 1 (the 256-slot engines of bench.py): the fused backward steps the two 128 x 128 matrices behind its tile loop and the compact
    update kernel the rest; gemm_mode bit 1024 keeps the whole step in k_eng_update.  Through Adam steps, early stops and level hand-overs
    (fresh moments), with pairs finishing at different ticks: parameters, moments, points and states bit for bit."""
    runs = []
    for mode in (7, 7 | 1024):
        eng, states, _ = _engine_vs_oracle(dev, "se3aa", K=0, S=300, T=280, m=3, iters=40, early_stop=True, w_cd=1.0, trunc=1e9, B=3, G=1,
                                           ratio=0.01, gemm_mode=mode, nn_mode=2)
        runs.append((eng, states))
    (e0, s0), (e1, s1) = runs
    assert [(s.level, s.total_steps, list(s.evals_per_level[:3])) for s in s0] == [(s.level, s.total_steps, list(s.evals_per_level[:3])) for s in s1]
    assert any(s.total_steps < 3 * 40 for s in s0)                        # an early stop happened: the ADVANCE path (moments zeroed, no step) ran
    for name in ("params", "adam_m", "adam_v", "pts"):
        assert torch.equal(getattr(e0, name), getattr(e1, name)), name


@pytest.mark.parametrize("G", [3, 4, 9, 10, 16, 32, 33])
def test_update_folds_any_number_of_partials(dev, G):
    """The update kernel requests the G gradient partials in batches (<= 3: one by one, 4..9: eight at a clamped index, >= 10: 32 at a
    clamped index, more than 33 in two batches) and adds them in index order: every branch against the oracle -- parameters after 2 x 4 Adam
    steps and the warped points (north_star: 1e-4) -- on a pair with fewer tiles than workgroups (idle partials are zero)."""
    eng, states, refs = _engine_vs_oracle(dev, "se3aa", K=0, S=700, T=650, m=2, iters=4, early_stop=False, w_cd=1.0, trunc=1e9, B=2, G=G,
                                          arith="split")
    P = eng.P
    for b, (st, ref) in enumerate(zip(states, refs)):
        assert st.level == 2 and st.total_steps == 8
        assert abs(st.loss - ref["loss_trace"][-1]) < 1e-4 * abs(ref["loss_trace"][-1])
        diff = np.abs(eng.params[b, :, :P].cpu().numpy().reshape(-1) - ref["params_all"])
        assert np.mean(diff < 1e-4) > 0.97, np.mean(diff < 1e-4)
        assert np.abs(eng.final_points(b, st).cpu().numpy() - ref["pts"]).max() < 1e-4


# ------------------------------------------------------------------- nonrigidity gate + BCE (w_reg > 0)
@pytest.mark.parametrize("tag", ["se3aa", "sim3quat", "sflow"])
def test_gate_forward_backward_matches_oracle_and_reference(dev, golden, tag):
    from deformationpyramid_amd import ops
    g = golden("F11_nonrigidity")
    pyr = seeded_pyramid(int(g["seed"]), m=6, nonrigidity_est=True, **VARIANTS[tag])
    lvl = 4
    scale_heads(pyr, lvl, float(g["head_scale"]))
    d = pyr.descs[lvl]
    assert d.nonrigidity and not pyr.descs[0].nonrigidity
    p = pyr.store[lvl].to(dev)
    x = torch.from_numpy(g["x"]).to(dev)
    out, act, heads, nr = ops.level_fwd(d, p, lvl, K0, x, save=True, want_nonrig=True)
    tol = 1e-5 if "quat" in tag else 2e-6
    assert np.abs(out.cpu().numpy() - g[f"{tag}.out"]).max() < tol                  # vs the reference
    assert np.abs(nr.cpu().numpy() - g[f"{tag}.nonrig"]).max() < 1e-6
    ro, rn = O().level_fwd(cdesc(d), pyr.store[lvl, :d.param_count].numpy(), lvl, K0, g["x"], want_nonrig=True)
    assert np.abs(out.cpu().numpy() - ro).max() < tol and np.abs(nr.cpu().numpy() - rn).max() < 1e-6
    coef = torch.linspace(-1.0, 1.0, 256 * 3).reshape(256, 3).to(dev)
    c2 = torch.linspace(0.5, -0.25, 256).to(dev)
    got = ops.level_bwd(d, p, lvl, K0, x, act, heads, coef, g_nr=c2).cpu().numpy()
    for name, off, shape in d.named_slices():
        ref = g[f"{tag}.grad.{name}"]
        e = rel_err(got[off:off + ref.size].reshape(ref.shape), ref)
        assert e < 2e-4, (tag, name, e)
    full = ops.pyramid_fwd(d, 6, K0, pyr.store.to(dev), x).cpu().numpy()
    assert np.abs(full - g[f"{tag}.full_out"]).max() < 1e-4


def test_engine_with_bce_regulariser_matches_oracle(dev, golden, arith):
    from deformationpyramid_amd.engine import BatchedEngine, OptConfig
    from deformationpyramid_amd.layout import LayerDesc
    g = golden("F11_nonrigidity")
    m, iters, w_reg = 3, 5, 0.5
    gated = LayerDesc(nonrigidity=True)
    cfg = OptConfig(m=m, iters=iters, early_stop=False, w_reg=w_reg)
    eng = BatchedEngine(gated, cfg, 2, n_cap=300, t_cap=280, device=dev, **engine_modes(arith, 300))
    refs = []
    for b in range(2):
        pyr = seeded_pyramid(int(g["it.seed"]) + b, m=m, nonrigidity_est=True, **VARIANTS["se3aa"])
        x = g["it.x"][: 300 - 11 * b]
        y = g["it.y"][: 280 - 5 * b]
        eng.load(b, torch.from_numpy(x), 0, x.shape[0], None, torch.from_numpy(y), pyr.store)
        descs = [cdesc(dd) for dd in pyr.descs]
        pa = np.concatenate([pyr.store[i, :dd.param_count].numpy() for i, dd in enumerate(pyr.descs)])
        refs.append(O().optimize(descs, pa, x, 0, x.shape[0], None, y, iters=iters, early_stop=False, w_reg=w_reg, nthreads=4))
    states = eng.run_until_done(chunk=8)
    for b, (st, ref) in enumerate(zip(states, refs)):
        assert st.level == m and st.total_steps == m * iters
        assert abs(st.loss - ref["loss_trace"][-1]) < 1e-4 * abs(ref["loss_trace"][-1]), (st.loss, ref["loss_trace"][-1])
        assert np.abs(eng.final_points(b, st).cpu().numpy() - ref["pts"]).max() < 1e-4


def test_register_with_w_reg_runs_end_to_end(dev, golden, arith):
    import os
    from deformationpyramid_amd.config import Config, load_config
    from deformationpyramid_amd.registration import Registration
    from deformationpyramid_amd.synthetic import synthetic_pair
    g = golden("F11_nonrigidity")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c = Config(load_config(os.path.join(root, "config", "NDP.yaml"), device=0), samples=256, w_reg=0.3, m=5)
    src, tgt, _, _ = synthetic_pair(11, n_total=2048)
    torch.manual_seed(4)
    model = Registration(c, **registration_modes(arith))
    model.load_pcds(src.numpy(), tgt.numpy())
    warped, cnt, _ = model.register()
    counts = np.array([cnt[l] for l in range(5)])
    ref = g["e2e.iters_per_level"]
    assert abs(int(counts[0]) - int(ref[0])) <= 3, (counts, ref)
    assert np.abs(warped.cpu().numpy() - g["e2e.warped"]).mean() < 0.08               # chaos bar, see test_registration_gpu


# ------------------------------------------------------------------- BASELINE.json configs 4 and 5 as parity cases
def test_config4_shape_transfer_sizes_sim3_euler(dev, arith):
    """shape_transfer.py: Sim3 / euler, ALL 6000 surface samples per cloud, then a 24 856-vertex inference warp."""
    from deformationpyramid_amd import ops
    eng, states, refs = _engine_vs_oracle(dev, "sim3eu", K=0, S=6000, T=6000, m=2, iters=3, early_stop=False,
                                          w_cd=1.0, trunc=1e9, B=1, arith=arith)
    st, ref = states[0], refs[0]
    assert st.total_steps == 6
    assert abs(st.loss - ref["loss_trace"][-1]) < 1e-4 * abs(ref["loss_trace"][-1])
    assert np.abs(eng.final_points(0, st).cpu().numpy() - ref["pts"]).max() < 1e-4
    verts = cloud(24856, 77)
    d = eng.desc
    got = ops.pyramid_fwd(d, 2, K0, eng.params[0], verts.to(dev)).cpu().numpy()
    want = O().pyramid_fwd([cdesc(d)] * 2, K0, eng.params[0, :, :eng.P].cpu().numpy().reshape(-1), verts.numpy(), nthreads=4)
    assert np.abs(got - want).max() < 1e-5


def test_config5_lndp_sizes(dev, arith):
    """LNDP.yaml: K = 500 precomputed landmark correspondences, m = 10 levels, no Chamfer (w_cd = 0)."""
    eng, states, refs = _engine_vs_oracle(dev, "se3aa", K=500, S=0, T=0, m=10, iters=3, early_stop=False,
                                          w_cd=0.0, trunc=0.25, B=2, arith=arith)
    for b, (st, ref) in enumerate(zip(states, refs)):
        assert st.level == 10 and st.total_steps == 30
        assert abs(st.loss - ref["loss_trace"][-1]) < 1e-4 * abs(ref["loss_trace"][-1])
        assert np.abs(eng.final_points(b, st).cpu().numpy() - ref["pts"]).max() < 1e-4


# ------------------------------------------------------------------------------ fused pair preparation / final warp
@pytest.mark.parametrize("tag", ["se3aa", "sim3eu", "se36d", "sflow"])
def test_pyramid_fwd_batch_equals_level_chain_bitwise(dev, tag):
    """The single-launch pyramid (points carried in LDS between levels) must reproduce the level-by-level kernel
    bit for bit, for several clouds of different sizes in one launch, with the centring shifts folded in."""
    from deformationpyramid_amd import ops
    m = 5
    pyrs = [seeded_pyramid(40 + j, m=m, **VARIANTS[tag]) for j in range(3)]
    for pyr in pyrs:
        for lvl in range(m):
            scale_heads(pyr, lvl, 20.0)
    sizes = [1, 777, 2048]
    jobs, want = [], []
    for j, (pyr, n) in enumerate(zip(pyrs, sizes)):
        x = (cloud(n, 60 + j) + 0.3).to(dev)
        s_in = torch.tensor([0.31, 0.29, 0.33, 0.0], device=dev)
        s_out = torch.tensor([-1.5, 2.0, 0.25, 0.0], device=dev)
        store = pyr.store.to(dev)
        jobs.append((store, x, s_in if j != 1 else None, s_out if j != 1 else None))
        cur = x - s_in[:3] if j != 1 else x
        for lvl in range(m):
            cur = ops.level_fwd(pyr.descs[lvl], store[lvl], lvl, K0, cur.contiguous())
        want.append(cur + s_out[:3] if j != 1 else cur)
    outs = ops.pyramid_fwd_batch(pyrs[0].descs[0], m, K0, jobs)
    for got, ref in zip(outs, want):
        assert torch.equal(got, ref)
    # and the single-cloud ABI entry goes through the same kernel
    one = ops.pyramid_fwd(pyrs[1].descs[0], m, K0, jobs[1][0], jobs[1][1])
    assert torch.equal(one, want[1])


@pytest.mark.parametrize("tag", ["se3aa", "sim3eu", "se36d", "sflow"])
def test_pyramid_fwd_batch_split_matches_the_fp32_kernel_and_the_oracle(dev, tag):
    """The final warp in the engine's split arithmetic (k_pyramid_fwd8: 256 points per workgroup through all levels in LDS)
    against the fp32-MFMA kernel on the same jobs -- 1e-5 on warped coordinates -- and against the oracle; clouds that end inside a
    tile, inside a workgroup's four tiles, and that span several workgroups; shifts folded in; 40 jobs = two launches."""
    from deformationpyramid_amd import ops
    m = 9
    pyrs = [seeded_pyramid(70 + j, m=m, **VARIANTS[tag]) for j in range(4)]
    for pyr in pyrs:
        for lvl in range(m):
            scale_heads(pyr, lvl, 10.0)
    sizes = [1, 65, 256, 257, 777, 2048, 8192] + [300 + 7 * j for j in range(33)]
    jobs = []
    for j, n in enumerate(sizes):
        x = (cloud(n, 160 + j) + 0.3).to(dev)
        s_in = torch.tensor([0.31, 0.29, 0.33, 0.0], device=dev) if j % 2 == 0 else None
        s_out = torch.tensor([-1.5, 2.0, 0.25, 0.0], device=dev) if j % 3 == 0 else None
        jobs.append((pyrs[j % 4].store.to(dev), x, s_in, s_out))
    ref = ops.pyramid_fwd_batch(pyrs[0].descs[0], m, K0, jobs)
    got = ops.pyramid_fwd_batch(pyrs[0].descs[0], m, K0, jobs, split=True)
    for a, b in zip(got, ref):
        assert a.shape == b.shape and torch.isfinite(a).all()
    # ndp_pyramid_fwd_batch_split_tiles: the tiles per workgroup change the launch geometry only -- 8 (what register_batch passes), 1
    # and 5 give the bits of the default 4
    for tiles in (8, 1, 5):
        alt = ops.pyramid_fwd_batch(pyrs[0].descs[0], m, K0, jobs, split=True, tiles=tiles)
        assert all(torch.equal(a, b) for a, b in zip(alt, got)), tiles
    err = torch.cat([(a - b).abs().max(dim=1).values for a, b in zip(got, ref)])
    if "6d" in tag:
        # 6D rotations are a Gram-Schmidt of two RAW head vectors: where a point's two vectors come out nearly parallel the map is
        # ill-conditioned and ANY two fp32 evaluations differ (the fp32 kernel is just as far from the oracle, checked below): the
        # bulk agrees like the other formats, the tail is bounded
        assert err.median().item() < 5e-6 and torch.quantile(err, 0.999).item() < 2e-4 and err.max().item() < 2e-3, \
            (tag, err.median().item(), torch.quantile(err, 0.999).item(), err.max().item())
    else:
        assert err.max().item() < 1e-5, (tag, err.max().item())
    d = pyrs[2].descs[0]
    pa = np.concatenate([pyrs[2].store[i, :d.param_count].numpy() for i in range(m)])
    want = O().pyramid_fwd([cdesc(d)] * m, K0, pa, (jobs[6][1].cpu() - torch.tensor([0.31, 0.29, 0.33])).numpy(), nthreads=8) + np.array([-1.5, 2.0, 0.25], dtype=np.float32)
    e_split = np.abs(got[6].cpu().numpy() - want).max(axis=1)                 # 8192 points x 9 levels against the oracle
    e_fp32 = np.abs(ref[6].cpu().numpy() - want).max(axis=1)
    if "6d" in tag:                                                            # same conditioning class as the fp32 kernel
        assert np.quantile(e_split, 0.999) < 2.0 * np.quantile(e_fp32, 0.999) + 1e-6, (np.quantile(e_split, 0.999), np.quantile(e_fp32, 0.999))
        assert np.median(e_split) < 1e-5
    else:
        assert e_split.max() < 1e-4                                            # north_star tolerance


def test_pyramid_fwd_batch_more_jobs_than_one_launch_and_gated_levels(dev, golden):
    from deformationpyramid_amd import ops
    m = 3
    pyr = seeded_pyramid(5, m=m, nonrigidity_est=True, **VARIANTS["se3aa"])
    store = pyr.store.to(dev)
    xs = [cloud(64 + 3 * j, 300 + j).to(dev) for j in range(40)]          # > NDP_MAX_WARP_JOBS
    outs = ops.pyramid_fwd_batch(pyr.descs[1], m, K0, [(store, x, None, None) for x in xs])
    descs = [cdesc(dd) for dd in pyr.descs]
    pa = np.concatenate([pyr.store[i, :dd.param_count].numpy() for i, dd in enumerate(pyr.descs)])
    for j in (0, 17, 39):
        ref = O().pyramid_fwd(descs, K0, pa, xs[j].cpu().numpy(), nthreads=2)
        assert np.abs(outs[j].cpu().numpy() - ref).max() < 1e-5


def test_pair_means_is_the_correctly_rounded_mean(dev):
    from deformationpyramid_amd import ops
    g = torch.Generator().manual_seed(3)
    src = torch.rand(8192, 3, generator=g) * 4 - 1
    tgt = torch.rand(5001, 3, generator=g) * 0.1 + 7
    got = ops.pair_means(src.to(dev), tgt.to(dev)).cpu().numpy()
    assert np.array_equal(got[:3], src.double().mean(0).float().numpy())
    assert np.array_equal(got[4:7], tgt.double().mean(0).float().numpy())
    assert got[3] == 0 and got[7] == 0


def test_engine_load_computes_the_pair_means_of_its_jobs_in_one_launch(dev):
    """A load job with n_src / n_tgt > 0 has its two cloud means computed BY the load call (k_pair_means_jobs: one launch for the
    whole group, what register_batch relies on since round 5): the bits of ndp_pair_means, left in the job's buffer for the final warp,
    and the slot centred with them -- next to a job of the same call that brings its means along."""
    from deformationpyramid_amd import ops
    from deformationpyramid_amd.engine import BatchedEngine, OptConfig
    m = 2
    pyr = seeded_pyramid(4, m=m, **VARIANTS["se3aa"])
    eng = BatchedEngine(pyr.descs[0], OptConfig(m=m, iters=4, early_stop=False), 3, n_cap=256, t_cap=256, device=dev)
    eng.park_all()
    g = torch.Generator().manual_seed(2)
    store = torch.zeros(m, eng.p_stride, device=dev)
    store[:, :eng.P] = pyr.store[:, :eng.P].to(dev)
    jobs, clouds = [], []
    for b, (ns, nt) in enumerate([(8192, 5001), (700, 4097), (1234, 999)]):
        src = (torch.rand(ns, 3, generator=g) * 3 - 1).to(dev)
        tgt = (torch.rand(nt, 3, generator=g) * 0.2 + 5).to(dev)
        S, T = min(200, ns), min(180, nt)
        ps = torch.randperm(ns, generator=g)[:S].to(torch.int32).to(dev)
        pt = torch.randperm(nt, generator=g)[:T].to(torch.int32).to(dev)
        means = torch.full((8,), float("nan"), device=dev) if b != 1 else ops.pair_means(src, tgt)
        jobs.append(dict(slot=b, params=store, K=0, S=S, T=T, src=src, tgt=tgt, perm_s=ps, perm_t=pt, means=means,
                         n_src=ns if b != 1 else 0, n_tgt=nt if b != 1 else 0))
        clouds.append((src, tgt, ps, pt, S, T, means))
    eng.load_jobs(jobs)
    torch.cuda.synchronize()
    for b, (src, tgt, ps, pt, S, T, means) in enumerate(clouds):
        want = ops.pair_means(src, tgt)
        assert torch.equal(means, want), b
        assert torch.equal(eng.pts[b, 0, :S], src[ps.long()] - want[:3]) and torch.equal(eng.tgt[b, :T], tgt[pt.long()] - want[4:7])


def test_engine_load_jobs_centres_samples_and_resets_the_slot(dev):
    """k_eng_load against the torch statement of registration.py:150-164 (bit-exact: one subtraction per value)."""
    from deformationpyramid_amd import ops
    from deformationpyramid_amd.engine import BatchedEngine, OptConfig
    from deformationpyramid_amd import _native as N
    m = 2
    pyr = seeded_pyramid(9, m=m, **VARIANTS["se3aa"])
    d = pyr.descs[0]
    eng = BatchedEngine(d, OptConfig(m=m, iters=4, early_stop=False), 3, n_cap=320, t_cap=256, device=dev)
    eng.park_all()
    eng.adam_m.fill_(3.0); eng.adam_v.fill_(4.0); eng.pts.fill_(9.0)
    g = torch.Generator().manual_seed(1)
    jobs, want = [], []
    for b, (K, S, T) in enumerate([(0, 300, 250), (20, 190, 256), (7, 0, 0)]):
        src = (torch.rand(1000, 3, generator=g) + 0.5).to(dev)
        tgt = (torch.rand(900, 3, generator=g) - 2.0).to(dev)
        ps = torch.randperm(1000, generator=g)[:max(S, 1)].to(torch.int32).to(dev)
        pt = torch.randperm(900, generator=g)[:max(T, 1)].to(torch.int32).to(dev)
        ls = torch.rand(max(K, 1), 3, generator=g).to(dev)
        lt = torch.rand(max(K, 1), 3, generator=g).to(dev)
        means = ops.pair_means(src, tgt)
        store = torch.zeros(m, eng.p_stride, device=dev)
        store[:, :eng.P] = pyr.store[:, :eng.P].to(dev) + b
        jobs.append(dict(slot=2 - b, params=store, K=K, S=S, T=T, src=src, tgt=tgt, perm_s=ps, perm_t=pt,
                         ldmk_s=ls if K else None, ldmk_t=lt if K else None, means=means))
        pts = torch.cat([ls[:K] - means[:3], src[ps[:S].long()] - means[:3]]) if K + S else None
        want.append((2 - b, K, S, T, pts, lt[:K] - means[4:7], tgt[pt[:T].long()] - means[4:7], store))
    eng.load_jobs(jobs)
    torch.cuda.synchronize()
    states = eng.read_states()
    for slot, K, S, T, pts, lt, tg, store in want:
        n = K + S
        assert torch.equal(eng.pts[slot, 0, :n], pts)
        assert torch.count_nonzero(eng.pts[slot, 0, n:]) == 0
        assert torch.equal(eng.ldmk_t[slot, :K], lt) and torch.equal(eng.tgt[slot, :T], tg)
        assert torch.equal(eng.params[slot], store)
        assert torch.count_nonzero(eng.adam_m[slot]) == 0 and torch.count_nonzero(eng.adam_v[slot]) == 0
        assert eng.geom[slot].tolist() == [K, S, T, 0]
        st = states[slot]
        assert (st.level, st.iter, st.break_counter, st.adam_t, st.cur, st.total_steps) == (0, 0, 0, 0, 0, 0)
        assert st.loss_prev == 1e6
    eng.park(1)
    torch.cuda.synchronize()
    assert eng.read_states()[1].level == m and eng.read_states()[0].level == 0
    # invalid jobs are refused by the C ABI, not silently clipped
    bad = dict(jobs[0]); bad["S"] = 400
    with pytest.raises((ValueError, N.NdpError)):
        eng.load_jobs([bad])


# ------------------------------------------------------------------------------ 1-NN: adversarial layouts
def _nn_case(name):
    g = torch.Generator().manual_seed(hash(name) % 1000)
    u = lambda n, s=1.0: (torch.rand(n, 3, generator=g) - 0.5) * s
    if name == "clusters":               # two tight blobs far apart: most cells empty, queries sit between them
        y = torch.cat([u(900, 0.02) + 1.0, u(1100, 0.02) - 1.0])
        x = torch.cat([u(500, 3.0), u(500, 0.02) + 1.0])
    elif name == "far_queries":          # every query outside the references' bounding box, at all distances
        y = u(2000)
        x = u(1500) * torch.tensor([0.2, 0.2, 0.2]) + torch.tensor([5.0, -3.0, 40.0])
    elif name == "plane":                # flat cloud: one grid axis collapses to a single cell
        y = u(2000); y[:, 2] = 0.25
        x = u(1800); x[:, 2] = x[:, 2] * 0.01
    elif name == "line":                 # two axes collapse
        y = u(1500); y[:, 1:] = 0.0
        x = u(700)
    elif name == "lattice_ties":         # integer lattice: massive exact ties in d2, lowest index must win
        k = torch.arange(12, dtype=torch.float32)
        y = torch.stack(torch.meshgrid(k, k, k, indexing="ij"), -1).reshape(-1, 3) * 0.125
        y = y[torch.randperm(y.shape[0], generator=g)].contiguous()
        x = (torch.stack(torch.meshgrid(k, k, k, indexing="ij"), -1).reshape(-1, 3)[:1500] + 0.5) * 0.125
    elif name == "single_ref":
        y = u(1); x = u(300)
    elif name == "identical":            # every reference is the same point (zero extent)
        y = torch.full((600, 3), 0.3); x = u(500)
    elif name == "skewed":               # 95 % of the references in one corner cell
        y = torch.cat([u(1900, 0.01) - 0.49, u(100)])
        x = u(2000)
    elif name == "large":
        y = u(9000); x = u(700, 1.2)
    elif name == "many_sources":         # more sources than one LDS pass of the matrix kernel holds (2048): 2 full passes + a ragged one
        y = u(777, 1.1); x = u(5000)
    elif name == "cross_pass_ties":      # every source three times, 1500 apart: the copies of a target's nearest source sit in different
        b = u(1500)                      # passes at the SAME exact distance -- the lowest index must win
        x = torch.cat([b, b, b]); y = u(600, 0.9)
    return x.contiguous(), y.contiguous()


@pytest.mark.parametrize("name", ["clusters", "far_queries", "plane", "line", "lattice_ties", "single_ref", "identical",
                                  "skewed", "large"])
def test_chamfer_nn_is_exact_on_adversarial_layouts(dev, name):
    from deformationpyramid_amd import ops
    x, y = _nn_case(name)
    r = O().chamfer(x.numpy(), y.numpy(), want_grad=False, nthreads=8)
    d2x, ix, d2y, iy = [t.cpu().numpy() for t in ops.chamfer_nn(x.to(dev), y.to(dev))]
    np.testing.assert_array_equal(d2x, r["d2x"])
    np.testing.assert_array_equal(d2y, r["d2y"])
    np.testing.assert_array_equal(ix, r["idx_x"])
    np.testing.assert_array_equal(iy, r["idx_y"])


@pytest.mark.parametrize("name", ["clusters", "far_queries", "plane", "line", "lattice_ties", "single_ref", "identical",
                                  "skewed", "large", "ragged", "near_ties", "many_sources", "cross_pass_ties"])
@pytest.mark.parametrize("matrix", [False, True])
def test_onepass_nn_is_exact_on_adversarial_layouts(dev, name, matrix):
    """The engine's one-pass kernels (every distance evaluated once, column minima by cross-lane butterflies, second-stage
    fold) must give the bits of the two-pass brute force: d2 and lowest index, both directions."""
    from deformationpyramid_amd import ops
    if name == "near_ties":                      # a shell of targets around a tight cluster of sources, radii 1e-7 apart, far from the
        g = torch.Generator().manual_seed(5)     # origin: thousands of candidates that differ in the last bits of d2
        d = torch.randn(1500, 3, generator=g); d = d / d.norm(dim=1, keepdim=True)
        y = (torch.tensor([3.0, -2.0, 1.0]) + d * (0.25 + 1e-7 * torch.arange(1500)[:, None])).contiguous()
        x = (torch.tensor([3.0, -2.0, 1.0]) + (torch.rand(900, 3, generator=g) - 0.5) * 1e-3).contiguous()
        cases = [(x, y), (y[:1100].contiguous(), x)]
    elif name == "ragged":                       # sizes that end inside a wave / a 16-target sub-chunk / a 512-target chunk
        g = torch.Generator().manual_seed(77)
        x, y = torch.rand(1, 3, generator=g), torch.rand(1, 3, generator=g)
        cases = [(x, y)] + [(torch.rand(s, 3, generator=g) - 0.5, torch.rand(t, 3, generator=g) - 0.5)
                            for s, t in ((65, 17), (129, 513), (700, 1025), (2000, 1490), (513, 2047))]
    else:
        cases = [_nn_case(name)]
    for x, y in cases:
        r = O().chamfer(x.numpy(), y.numpy(), want_grad=False, nthreads=8)
        d2x, ix, d2y, iy = [t.cpu().numpy() for t in ops.chamfer_nn_onepass(x.to(dev), y.to(dev), matrix=matrix)]
        np.testing.assert_array_equal(d2x, r["d2x"])
        np.testing.assert_array_equal(d2y, r["d2y"])
        np.testing.assert_array_equal(ix, r["idx_x"])
        np.testing.assert_array_equal(iy, r["idx_y"])


@pytest.mark.parametrize("arith,nn_mode,G", [("bitwise", 0, 4), ("bitwise", 1, 4), ("bitwise", 2, 4),
                                             ("split", 2, 2), ("split", 0, 2), ("split", 1, 2)])
def test_engine_matches_oracle_at_the_bench_cloud_sizes(dev, arith, nn_mode, G):
    """The bench's cloud sizes in a SMALL engine: S = T = 2000 samples, 8 resident pairs of slightly different sizes, 3 iterations x
    2 levels, G workgroups per pair as a 128-slot engine gets them (4 four-wave workgroups with the fp32-MFMA level kernels,
    2 eight-wave workgroups with the split ones; the slot geometry bench.py runs since round 4 -- 256 slots, G = 1 -- is the next
    test's) -- with each nearest-neighbour kernel (0 one-pass on the vector pipe, 1
    latency shape, 2 one-pass on the matrix pipe): identical step counts, loss and warped samples within the per-step budget."""
    eng, states, refs = _engine_vs_oracle(dev, "se3aa", K=0, S=2000, T=2000, m=2, iters=3, early_stop=False,
                                          w_cd=1.0, trunc=1e9, B=8, G=G, nn_mode=nn_mode, arith=arith)
    assert eng.G == G and eng.c_engine.nn_mode == nn_mode and eng.c_engine.gemm_mode == (7 if arith == "split" else 0)
    for b, (st, ref) in enumerate(zip(states, refs)):
        assert st.level == 2 and st.total_steps == 6
        assert abs(st.loss - ref["loss_trace"][-1]) < 1e-4 * abs(ref["loss_trace"][-1])
        assert np.abs(eng.final_points(b, st).cpu().numpy() - ref["pts"]).max() < 1e-4


def test_engine_matches_oracle_at_the_slot_geometry_bench_runs(dev, arith):
    """The engine exactly as bench.py builds it since round 4: B = 256 resident slots per engine, hence G = 1 (ONE level-kernel
    workgroup per pair, 32 tiles per gradient accumulator) on the split path (G = 2 four-wave workgroups on the bitwise one), the
    512-target matrix-pipe nearest-neighbour kernel with its every-second-slot row partials, and the XCD-aware block -> pair
    placement of the loss and nearest-neighbour stages.  S = T = 2000; eight distinct pairs of slightly different sizes, each
    replicated 32 times across the slots (slot s holds pair s % 8): slots 0..7 against the oracle at the usual budget, and every
    replica BIT-identical to its original -- a slot's result must not depend on which workgroup / XCD served it."""
    from deformationpyramid_amd.engine import BatchedEngine, OptConfig
    B, NP, S, T, m, iters = 256, 8, 2000, 2000, 2, 3
    cfg = OptConfig(m=m, iters=iters, early_stop=False, w_cd=1.0, trunc=1e9)
    eng, refs, loads = None, [], []
    for b in range(NP):
        pyr = seeded_pyramid(7 + b, m=m, **VARIANTS["se3aa"])
        d = pyr.descs[0]
        if eng is None:
            eng = BatchedEngine(d, cfg, B, n_cap=S, t_cap=T, device=dev, **engine_modes(arith, S))
        Sb, Tb = S - 7 * b, T - 3 * b
        src = cloud(Sb, 100 + b)
        c, s_ = np.cos(0.2), np.sin(0.2)
        Rz = torch.tensor([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
        tgt = (cloud(Tb, 200 + b) @ Rz.T + torch.tensor([0.03, -0.02, 0.01])).contiguous()
        loads.append((src, Sb, tgt, pyr.store))
        params_all = np.concatenate([pyr.store[i, :d.param_count].numpy() for i in range(m)])
        refs.append(O().optimize([cdesc(d)] * m, params_all, src.numpy(), 0, Sb, None, tgt.numpy(), k0=K0, iters=iters,
                                 w_cd=1.0, trunc=1e9, early_stop=False, nthreads=4, ratio=0.001))
    for slot in range(B):
        src, Sb, tgt, store = loads[slot % NP]
        eng.load(slot, src, 0, Sb, None, tgt, store)
    if arith == "split":
        assert eng.G == 1 and eng.c_engine.gemm_mode == 7 and eng.c_engine.nn_mode == 2
    else:
        assert eng.G == 2 and eng.c_engine.gemm_mode == 0
    states = eng.run_until_done(chunk=8)
    for b in range(NP):
        st, ref = states[b], refs[b]
        assert st.level == 2 and st.total_steps == 6
        assert abs(st.loss - ref["loss_trace"][-1]) < 1e-4 * abs(ref["loss_trace"][-1])
        assert np.abs(eng.final_points(b, st).cpu().numpy() - ref["pts"]).max() < 1e-4
    P = eng.P
    for slot in range(NP, B):
        o = slot % NP
        assert states[slot].loss == states[o].loss and states[slot].total_steps == states[o].total_steps, slot
        assert torch.equal(eng.params[slot, :, :P], eng.params[o, :, :P]), slot
        assert torch.equal(eng.final_points(slot, states[slot]), eng.final_points(o, states[o])), slot
        assert torch.equal(eng.d2y[slot], eng.d2y[o]) and torch.equal(eng.idx_y[slot], eng.idx_y[o]), slot


@pytest.mark.parametrize("G", [1, 4])
def test_pipelined_forward_across_chunk_boundaries(dev, G):
    """The split forward pipelines its workgroup's tiles (layer 0 of the next tile and the heads of the previous one ride in the
    layers' MFMA gaps) and computes the encodings in chunks of eight tiles: 11 tiles per pair as ONE workgroup (a chunk of eight and a
    chunk of three: the tile behind the boundary takes layer 0 in the open, the heads of the last tile of a chunk ride in the next
    chunk's first layer 1) and as four workgroups of 3, 3, 3 and 2 tiles -- same steps, loss and points as the oracle, and both
    partitions within round-off of each other (the forward's results do not depend on the partition: bit-identical points)."""
    eng, states, refs = _engine_vs_oracle(dev, "se3aa", K=0, S=700, T=650, m=2, iters=4, early_stop=False, w_cd=1.0, trunc=1e9,
                                          B=3, G=G, arith="split")
    assert eng.G == G and eng.c_engine.gemm_mode == 7
    for b, (st, ref) in enumerate(zip(states, refs)):
        assert st.level == 2 and st.total_steps == 8
        assert abs(st.loss - ref["loss_trace"][-1]) < 1e-4 * abs(ref["loss_trace"][-1])
        assert np.abs(eng.final_points(b, st).cpu().numpy() - ref["pts"]).max() < 1e-4
    # one forward of the same state under both partitions: the activations a tile leaves do not depend on which workgroup ran it
    outs = []
    for g2 in (1, 4, 11):
        e2, _, _ = _engine_vs_oracle(dev, "se3aa", K=0, S=700, T=650, m=2, iters=1, early_stop=False, w_cd=1.0, trunc=1e9, B=1, G=g2, arith="split")
        outs.append((e2.act[0, 1:, :704].clone(), e2.heads[0, :700].clone()))
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1])


def test_engine_nn_shapes_are_bit_identical(dev):
    """The three nearest-neighbour shapes of the engine must produce the same bits (losses, parameters) tick for tick."""
    runs = []
    for mode in (0, 1, 2):
        eng, states, _ = _engine_vs_oracle(dev, "se3aa", K=0, S=700, T=650, m=2, iters=4, early_stop=False, w_cd=1.0,
                                           trunc=0.05, B=3, nn_mode=mode)
        runs.append((eng.params.clone(), [s.loss for s in states], eng.d2y.clone(), eng.idx_y.clone()))
    for other in runs[1:]:
        assert runs[0][1] == other[1]
        assert torch.equal(runs[0][0], other[0]) and torch.equal(runs[0][2], other[2]) and torch.equal(runs[0][3], other[3])


@pytest.mark.parametrize("gemm_mode", [1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("tag", ["se3aa", "sim3eu", "sflow"])
def test_engine_on_fp16_splits_stays_inside_the_parity_budget(dev, tag, gemm_mode):
    """Opt-in gemm_mode mask (1 forward, 2 bwd1, 4 bwd2: their 128x128 contractions from two-way fp16 splits on the fp16 MFMA): the
    SAME tolerances against the oracle as the default kernels -- loss to 1e-4 relative, warped points to 1e-4, the bulk of the
    parameters after 12 Adam steps -- and, for the forward, activations within 2e-6 of the default forward's (both are ~5e-7 of the
    output scale away from a float64 evaluation)."""
    eng, states, refs = _engine_vs_oracle(dev, tag, K=0, S=300, T=280, m=2, iters=6, early_stop=False, w_cd=1.0, trunc=1e9, gemm_mode=gemm_mode)
    assert eng.c_engine.gemm_mode == gemm_mode
    P = eng.P
    for b, (st, ref) in enumerate(zip(states, refs)):
        assert st.level == 2 and list(st.evals_per_level[:2]) == [6, 6] and st.total_steps == 12
        assert abs(st.loss - ref["loss_trace"][-1]) < 1e-4 * abs(ref["loss_trace"][-1])
        assert np.abs(eng.final_points(b, st).cpu().numpy() - ref["pts"]).max() < 1e-4
        diff = np.abs(eng.params[b, :, :P].cpu().numpy().reshape(-1) - ref["params_all"])
        assert np.mean(diff < 1e-4) > 0.97, np.mean(diff < 1e-4)
    if not gemm_mode & 1:
        return
    # one tick of both forwards on the same state: activations side by side
    acts = []
    for mode in (0, 1):
        e2, _, _ = _engine_vs_oracle(dev, tag, K=0, S=300, T=280, m=2, iters=1, early_stop=False, w_cd=1.0, trunc=1e9, B=1, gemm_mode=mode)
        acts.append((e2.act.clone(), e2.heads.clone()))
    scale = acts[0][0].abs().max().item()
    assert (acts[0][0] - acts[1][0]).abs().max().item() < 2e-6 * max(scale, 1.0)
    assert (acts[0][1][..., :16] - acts[1][1][..., :16]).abs().max().item() < 1e-8


def test_config3_stress_samples_8192(dev, arith):
    """BASELINE config 3 (samples = 8192, Chamfer-bound): S = T = 8192 in the engine, two slots of different sizes."""
    eng, states, refs = _engine_vs_oracle(dev, "se3aa", K=0, S=8192, T=8192, m=2, iters=2, early_stop=False,
                                          w_cd=1.0, trunc=1e9, B=2, arith=arith)
    for b, (st, ref) in enumerate(zip(states, refs)):
        assert st.level == 2 and st.total_steps == 4
        assert abs(st.loss - ref["loss_trace"][-1]) < 1e-4 * abs(ref["loss_trace"][-1])
        assert np.abs(eng.final_points(b, st).cpu().numpy() - ref["pts"]).max() < 1e-4


def test_config2_fixed_work_450_iterations_per_pair(dev, arith):
    """early stop off, 50 iterations x 9 levels = exactly 450 Adam steps per pair; the
    loss trace of the whole run stays within the per-step budget of the oracle's (small clouds keep the oracle fast)."""
    eng, states, refs = _engine_vs_oracle(dev, "se3aa", K=0, S=256, T=240, m=9, iters=50, early_stop=False,
                                          w_cd=1.0, trunc=1e9, B=2, arith=arith)
    for b, (st, ref) in enumerate(zip(states, refs)):
        assert st.level == 9 and st.total_steps == 450 and st.total_evals == 450
        assert list(st.evals_per_level[:9]) == [50] * 9
        # 450 chaotic steps: the end point agrees at the trajectory-noise level, the loss within a few percent
        assert abs(st.loss - ref["loss_trace"][-1]) < 0.05 * abs(ref["loss_trace"][-1])
