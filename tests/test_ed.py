"""Embedded-deformation N-ICP baseline: the native graph builder (csrc/ndp_graph.cpp) against the
deformation graph the reference's own MVRegC build produced (golden F15), the torch-CPU restatement of the optimisation
loop against the reference's trace, and the HIP path against both."""
import os

import numpy as np
import pytest
import torch

from deformationpyramid_amd.config import Config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = dict(deformation_model="ED", iters=25, lr=0.02, max_break_count=30, break_threshold_ratio=0.01, w_ldmk=1, w_cd=1,
            w_arap=0.5, samples=600, max_triangle_distance=0.06, node_coverage=0.09, USE_ONLY_VALID_VERTICES=True,
            num_neighbors=8, ENFORCE_TOTAL_NUM_NEIGHBORS=False, SAMPLE_RANDOM_SHUFFLE=False,
            REMOVE_NODES_WITH_NOT_ENOUGH_NEIGHBORS=False)


def test_depth_to_mesh_matches_the_reference_build(golden):
    from deformationpyramid_amd.geometry import depth_to_mesh
    g = golden("F15_embedded_deformation")
    d0 = g["depth_src"]
    v, f, pix, pim = depth_to_mesh(d0.copy(), d0 > 0, g["K"], max_triangle_distance=0.06, depth_scale=1000.)
    assert [v.shape[0], f.shape[0]] == list(g["mesh.counts"])
    np.testing.assert_array_equal(f[:32], g["mesh.faces_head"])
    np.testing.assert_array_equal(pix[:32], g["mesh.vpix_head"])
    np.testing.assert_allclose(v.astype(np.float64).sum(0), g["mesh.vsum"], rtol=1e-12)
    assert int((f.astype(np.int64) * np.array([1, 3, 7])).sum()) == int(g["mesh.fhash"])


@pytest.mark.parametrize("tag,remove", [("g", False), ("gr", True)])
def test_deformation_graph_is_bit_identical_to_the_reference_build(golden, tag, remove):
    """Nodes, geodesic edges and weights, per-pixel anchors and skinning weights: every number the optimiser receives."""
    from deformationpyramid_amd.geometry import get_deformation_graph_from_depthmap
    g = golden("F15_embedded_deformation")
    cfg = Config(dict(BASE, REMOVE_NODES_WITH_NOT_ENOUGH_NEIGHBORS=remove, node_coverage=float(g[f"{tag}.coverage"])))
    data = get_deformation_graph_from_depthmap(g["depth_src"].copy(), g["K"], cfg)
    np.testing.assert_array_equal(data["graph_nodes"].numpy(), g[f"{tag}.graph_nodes"])
    np.testing.assert_array_equal(data["graph_edges"].numpy().astype(np.int32), g[f"{tag}.graph_edges"])
    np.testing.assert_array_equal(data["graph_edges_weights"].numpy(), g[f"{tag}.graph_edges_weights"])
    pa, pw = data["pixel_anchors"].numpy(), data["pixel_weights"].numpy()
    np.testing.assert_array_equal(pa[::7, ::5], g[f"{tag}.pixel_anchors_rows"])
    np.testing.assert_array_equal(pw[::7, ::5], g[f"{tag}.pixel_weights_rows"])
    assert int(pa.astype(np.int64).sum()) == int(g[f"{tag}.pixel_anchors_sum"])
    assert int((pa.astype(np.int64) * (1 + np.arange(pa.size).reshape(pa.shape) % 9973)).sum()) == int(g[f"{tag}.pixel_anchors_hash"])
    assert abs(pw.astype(np.float64).sum() - float(g[f"{tag}.pixel_weights_sum"])) < 1e-9
    assert int((pa.sum(-1) > -4).sum()) == int(g[f"{tag}.valid_pixels"])
    assert tuple(data["point_image"].shape) == g["depth_src"].shape + (3,)


def _problem(golden):
    """Graph + raw clouds of the F15 depth pair, as Registration.load_raw_pcds_from_depth builds them (registration.py:38-77)."""
    from deformationpyramid_amd.geometry import depth_2_pc, get_deformation_graph_from_depthmap
    g = golden("F15_embedded_deformation")
    data = get_deformation_graph_from_depthmap(g["depth_src"].copy(), g["K"], Config(BASE))
    valid = torch.sum(data["pixel_anchors"], dim=-1) > -4
    src_raw = data["point_image"][valid]
    anchors = data["pixel_anchors"][valid].long()
    weights = data["pixel_weights"][valid]
    tgt_depth = g["depth_tgt"] / 1000.
    tgt = depth_2_pc(tgt_depth, g["K"]).transpose(1, 2, 0)
    tgt_raw = torch.from_numpy(tgt[tgt_depth > 0]).float()
    return g, data, src_raw, tgt_raw, anchors, weights


def test_oracle_loop_follows_the_reference_trace(golden):
    from oracle import ed_ref as R
    g, data, src_raw, tgt_raw, anchors, weights = _problem(golden)
    torch.manual_seed(int(g["e2e.seed"]))
    phi, t, trace = R.optimize(src_raw, tgt_raw, anchors, weights, data["graph_nodes"], data["graph_edges"],
                               data["graph_edges_weights"], iters=25, samples=600)
    cd = np.array([c for c, _ in trace])
    ar = np.array([a for _, a in trace])
    assert len(cd) == len(g["e2e.cd_trace"]) == 25                       # the relative-change stop never fires upstream
    assert abs(cd[0] - g["e2e.cd_trace"][0]) < 2e-6 * g["e2e.cd_trace"][0] and abs(ar[0] - g["e2e.arap_trace"][0]) < 1e-12
    assert np.abs(cd - g["e2e.cd_trace"]).max() < 2e-3 * g["e2e.cd_trace"].max()
    assert np.abs(ar[1:] - g["e2e.arap_trace"][1:]).max() < 2e-2 * g["e2e.arap_trace"].max()


# ------------------------------------------------------------------------------------------------ HIP path (GPU)
@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_hip_warp_arap_and_node_gradients_match_the_oracle(dev, golden):
    """ndp_ed_warp / ndp_ed_arap / ndp_ed_grad against the torch-CPU restatement (autograd) at a non-trivial state:
    random small rotations and translations, the first 600 anchored points as the batch, a random upstream gradient."""
    from deformationpyramid_amd.ed import EDGraph
    from oracle import ed_ref as R
    g, data, src_raw, tgt_raw, anchors, weights = _problem(golden)
    n = data["graph_nodes"].shape[0]
    gen = torch.Generator().manual_seed(3)
    phi = (torch.rand(n, 3, generator=gen) - 0.5) * 0.3
    phi[::7] = 0.0                                                        # the Taylor branch of the axis-angle map
    t = (torch.rand(n, 3, generator=gen) - 0.5) * 0.05
    x, a, w = src_raw[:600].contiguous(), anchors[:600], weights[:600].contiguous()
    gy = torch.rand(600, 3, generator=gen) - 0.5
    phi_r, t_r = phi.clone().requires_grad_(True), t.clone().requires_grad_(True)
    Rm = R.axis_angle_to_matrix(phi_r)
    y_ref = R.ed_warp(x, a, w, data["graph_nodes"], Rm, t_r)
    arap_ref = R.arap(Rm, t_r, data["graph_nodes"], data["graph_edges"], data["graph_edges_weights"])
    ((y_ref * gy).sum() + 0.5 * arap_ref).backward()
    graph = EDGraph(data["graph_nodes"].to(dev), data["graph_edges"].to(dev), data["graph_edges_weights"].to(dev))
    params = torch.cat([phi.reshape(-1), t.reshape(-1)]).to(dev)
    y = graph.warp(params, x.to(dev), a.to(dev).to(torch.int32).contiguous(), w.to(dev))
    assert (y.cpu() - y_ref.detach()).abs().max().item() < 2e-6
    assert abs(graph.arap(params).item() - arap_ref.item()) < 1e-5 * arap_ref.item()
    gr = graph.grads(params, x.to(dev), a.to(dev).to(torch.int32).contiguous(), w.to(dev), gy.to(dev), 0.5).cpu()
    ref = torch.cat([phi_r.grad.reshape(-1), t_r.grad.reshape(-1)])
    assert (gr - ref).abs().max().item() < 1e-4 * ref.abs().max().item()


@pytest.mark.gpu
def test_hip_register_follows_the_reference_run(dev, golden, tmp_path):
    """config.deformation_model = ED through Registration.load_raw_pcds_from_depth + register(): the (cd, arap) trace of the
    reference's own run on the synthetic depth pair, its returned cloud and its validity mask."""
    from PIL import Image
    from deformationpyramid_amd.registration import Registration
    g = golden("F15_embedded_deformation")
    ps, pt = str(tmp_path / "s.png"), str(tmp_path / "t.png")
    Image.fromarray(g["depth_src"]).save(ps)
    Image.fromarray(g["depth_tgt"]).save(pt)
    torch.manual_seed(int(g["e2e.seed"]))
    model = Registration(Config(dict(BASE, device=0)))
    model.load_pcds(torch.from_numpy(g["e2e.src_pcd"]), torch.from_numpy(g["e2e.tgt_pcd"]))
    model.load_raw_pcds_from_depth(ps, pt, g["K"], landmarks=None)
    warped, valid_id = model.register()
    np.testing.assert_array_equal(valid_id.cpu().numpy(), g["e2e.valid_id"])
    tr = model.last_ed["trace"]
    assert len(tr) == 25
    cd, ar = np.array([c for c, _ in tr]), np.array([a for _, a in tr])
    assert abs(cd[0] - g["e2e.cd_trace"][0]) < 1e-5 * g["e2e.cd_trace"][0]
    assert np.abs(cd - g["e2e.cd_trace"]).max() < 5e-3 * g["e2e.cd_trace"].max()
    assert np.abs(ar[1:] - g["e2e.arap_trace"][1:]).max() < 3e-2 * g["e2e.arap_trace"].max()
    assert np.abs(warped.cpu().numpy() - g["e2e.warped"]).max() < 2e-3


def test_synthetic_depth_pair_matches_the_golden_generator(golden):
    from deformationpyramid_amd.synthetic import synthetic_depth_pair
    g = golden("F15_embedded_deformation")
    d0, d1, K = synthetic_depth_pair(0)
    np.testing.assert_array_equal(d0, g["depth_src"])
    np.testing.assert_array_equal(d1, g["depth_tgt"])
    np.testing.assert_array_equal(K, g["K"])


@pytest.mark.gpu
def test_eval_driver_runs_the_nicp_baseline(dev, tmp_path):
    """eval_nolearned.py --config config/baselines/NICP.yaml on synthetic depth pairs (30 iterations): graph construction,
    the loop and the metrics of the returned (masked) cloud -- well below the do-nothing error of these pairs."""
    import re
    import subprocess
    import sys
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "baselines", "NICP.yaml")).read().replace("!join [ node_coverage, *node_coverage]", "x"))
    cfg.update(iters=30, samples=600, exp_dir="x")
    path = tmp_path / "nicp.yaml"
    path.write_text(yaml.safe_dump(cfg))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "eval_nolearned.py"), "--config", str(path), "--synthetic", "2"],
                         capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"2/2: full-epe: ([0-9.]+)", out.stdout)
    assert m and float(m.group(1)) < 8.0, out.stdout[-800:]          # finite, sane metrics (the same-pixel flow of these pairs is not what N-ICP recovers)
