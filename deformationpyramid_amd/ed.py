"""Embedded-deformation N-ICP comparison baseline on the HIP path.

    Registration.load_raw_pcds_from_depth     /root/reference/model/registration.py:38-90
    Registration.optimize_Embeded_deformation /root/reference/model/registration.py:342-467
    ED_warp / arap_cost                       /root/reference/model/geometry.py:37-49, model/loss.py:261-285

The deformation graph comes from the native builder (geometry.py -> csrc/ndp_graph.cpp, bit-identical to the reference's
MVRegC); the loop is host driven, exactly like upstream: every iteration draws fresh Chamfer samples with two CPU randperms,
so there is nothing to keep resident -- it is a comparison baseline, not the hot path.  Warp, ARAP, Chamfer, the node
gradients and Adam run in libndp_hip.so (`ndp_ed_warp`, `ndp_ed_arap`, `ndp_ed_grad`, the NDP Chamfer / Adam operators).
Three upstream quirks are kept: -1 anchor / edge slots index the last node with weight 0; `loss_prev` is never updated
inside the loop (registration.py:428-433), so only `loss < 1e-5` or the iteration cap end it; and the final full-cloud warp
uses the rotations computed at the top of the last iteration (one Adam step behind phi) with the updated translations
(registration.py:376, 449-453).  No CPU fallback.
"""
import ctypes

import numpy as np
import torch

from . import _native as N
from . import ops
from .geometry import depth_2_pc, get_deformation_graph_from_depthmap, map_pixel_to_pcd, pc_2_uv

_p = ops._p


def imread(path):
    """Depth PNG -> uint16 array (skimage.io.imread upstream; PIL here)."""
    from PIL import Image
    return np.array(Image.open(path))


def load_raw_pcds_from_depth(reg, source_depth_path, tgt_depth_path, K, landmarks=None):
    """registration.py:38-90: deformation graph of the source depth map, raw source / target clouds, pixel -> cloud maps."""
    assert reg.deformation_model == "ED"
    dev = reg._dev()
    reg.intrinsics = K
    depth_image = imread(source_depth_path)
    data = get_deformation_graph_from_depthmap(depth_image, K, reg.config)
    reg.graph_nodes = data["graph_nodes"].to(dev).contiguous()
    reg.graph_edges = data["graph_edges"].to(dev)
    reg.graph_edges_weights = data["graph_edges_weights"].to(dev).contiguous()
    valid_pixels = torch.sum(data["pixel_anchors"], dim=-1) > -4
    reg.src_pcd_raw = data["point_image"][valid_pixels].to(dev).contiguous()
    reg.point_anchors = data["pixel_anchors"][valid_pixels].long().to(dev)
    reg.anchor_weight = data["pixel_weights"][valid_pixels].to(dev).contiguous()
    reg.anchor_loc = reg.graph_nodes[reg.point_anchors]
    reg.frame_point_len = [len(reg.src_pcd_raw)]
    reg.src_pix_2_pcd_map = [map_pixel_to_pcd(valid_pixels)]
    tgt_depth = imread(tgt_depth_path) / 1000.
    depth_mask = torch.from_numpy(tgt_depth > 0)
    tgt_pcd = depth_2_pc(tgt_depth, reg.intrinsics).transpose(1, 2, 0)
    reg.tgt_pcd_raw = torch.from_numpy(tgt_pcd[tgt_depth > 0]).float().to(dev).contiguous()
    reg.tgt_pix_2_pcd_map = map_pixel_to_pcd(depth_mask)
    if landmarks is not None:
        s_uv, t_uv = landmarks
        s_id = reg.src_pix_2_pcd_map[-1][s_uv[:, 1], s_uv[:, 0]]
        t_id = reg.tgt_pix_2_pcd_map[t_uv[:, 1], t_uv[:, 0]]
        valid_id = (s_id > -1) * (t_id > -1)
        reg.landmarks = (s_id[valid_id], t_id[valid_id])
    else:
        reg.landmarks = None


class EDGraph:
    """Device-side operators of one graph: y = warp(x), arap(), grads(gy)."""

    def __init__(self, nodes, edges, edge_w):
        self.dev = nodes.device
        self.nodes = nodes.float().contiguous()
        self.n = nodes.shape[0]
        self.K = edges.shape[1]
        self.edges = edges.to(torch.int32).contiguous()
        self.edge_w = edge_w.float().contiguous()
        self.R = torch.empty(self.n, 9, device=self.dev, dtype=torch.float32)
        self.reg = torch.empty(1, device=self.dev, dtype=torch.float32)

    def warp(self, params, x, anchors, weights, keep_R=False):
        """params: flat [phi (3n) | t (3n)]; x [S,3]; anchors [S,6] int32; weights [S,6].  keep_R: do not recompute the node
        rotations from phi -- warp with the R of the previous call and the CURRENT translations."""
        y = torch.empty_like(x)
        N.check(N.lib().ndp_ed_warp(_p(x), x.shape[0], _p(anchors), _p(weights), _p(self.nodes), self.n, None if keep_R else _p(params),
                                    _p(params[3 * self.n:]), _p(self.R), _p(y), N.stream_ptr(self.dev)), "ndp_ed_warp")
        return y

    def arap(self, params):
        N.check(N.lib().ndp_ed_arap(_p(self.nodes), self.n, _p(self.R), _p(params[3 * self.n:]), _p(self.edges), _p(self.edge_w),
                                    self.K, _p(self.reg), N.stream_ptr(self.dev)), "ndp_ed_arap")
        return self.reg

    def grads(self, params, x, anchors, weights, gy, w_arap):
        g = torch.empty(6 * self.n, device=self.dev, dtype=torch.float32)
        N.check(N.lib().ndp_ed_grad(_p(x), x.shape[0], _p(anchors), _p(weights), _p(gy), _p(self.nodes), self.n, _p(self.R),
                                    _p(params[3 * self.n:]), _p(params), _p(self.edges), _p(self.edge_w), self.K, float(w_arap),
                                    _p(g), N.stream_ptr(self.dev)), "ndp_ed_grad")
        return g


def optimize_Embeded_deformation(reg, visualize=False):
    """registration.py:342-467 -> (warped sampled cloud, valid_id)."""
    if visualize:
        raise NotImplementedError("mayavi visualisation is outside the hot path")
    config = reg.config
    dev = reg._dev()
    graph = EDGraph(reg.graph_nodes, reg.graph_edges, reg.graph_edges_weights)
    n = graph.n
    params = torch.zeros(6 * n, device=dev, dtype=torch.float32)                  # phi = 0, t = 0            (:353-360)
    m, v = torch.zeros_like(params), torch.zeros_like(params)
    anchors_all = reg.point_anchors.to(torch.int32).contiguous()
    weights_all = reg.anchor_weight
    lr, samples = float(config.lr), int(config.samples)
    trace, steps = [], 0
    for i in range(config.iters):                                                  # :378
        src = torch.randperm(reg.src_pcd_raw.shape[0])                             # :385-386 (CPU RNG)
        tgt = torch.randperm(reg.tgt_pcd_raw.shape[0])
        si, ti = src[:samples].to(dev), tgt[:samples].to(dev)
        s_sample = reg.src_pcd_raw[si]
        t_sample = reg.tgt_pcd_raw[ti].contiguous()
        anchors, weights = anchors_all[si], weights_all[si]
        n_l = 0
        if reg.landmarks:                                                          # :398-414: landmark rows go first
            s_l, t_l = reg.landmarks
            s_l, t_l = s_l.to(dev), t_l.to(dev)
            n_l = s_l.shape[0]
            t_ldmk = reg.tgt_pcd_raw[t_l]
            s_sample = torch.cat([reg.src_pcd_raw[s_l], s_sample])
            anchors = torch.cat([anchors_all[s_l], anchors])
            weights = torch.cat([weights_all[s_l], weights])
        s_sample, anchors, weights = s_sample.contiguous(), anchors.contiguous(), weights.contiguous()
        warped = graph.warp(params, s_sample, anchors, weights)                    # :418
        cd, gx, _ = ops.chamfer_l1(warped, t_sample, 1e10)                         # :421
        reg_v = graph.arap(params)                                                 # :424
        L = cd.item() * config.w_cd + reg_v.item() * config.w_arap
        gy = gx if float(config.w_cd) == 1.0 else gx * float(config.w_cd)
        if n_l:                                                                    # :427-429 landmark_cost on the first rows
            l_loss, l_g = ops.landmark_mse(warped[:n_l].contiguous(), t_ldmk.contiguous())
            L += float(config.w_ldmk) * l_loss.item()
            gy[:n_l] += float(config.w_ldmk) * l_g
        trace.append((cd.item(), reg_v.item()))
        if L < 1e-5:                                                               # :435 (loss_prev is never updated upstream:
            break                                                                  #       the relative-change counter cannot fire)
        grads = graph.grads(params, s_sample, anchors, weights, gy.contiguous(), float(config.w_arap))
        steps += 1
        ops.adam_step(params, grads, m, v, steps, lr=lr)
        lr = lr * 0.999                                                            # ExponentialLR(gamma=0.999).step()   (:367, :445)
    reg.last_ed = dict(iters=steps, trace=trace)
    # :449-453 -- upstream warps with the R computed at the top of the LAST loop iteration (:376), i.e. one Adam step behind phi,
    # together with the updated translations; kept (the loop's last warp() left exactly that R in graph.R)
    warped_pcd = graph.warp(params, reg.src_pcd_raw, anchors_all, weights_all, keep_R=len(trace) > 0)
    s_uv = pc_2_uv(reg.src_pcd, reg.intrinsics)                                    # :459-464: motion of the dataset's sampled cloud
    pix_map = reg.src_pix_2_pcd_map[-1].to(dev)
    s_id = pix_map[s_uv[:, 1], s_uv[:, 0]]
    valid_id = s_id > -1
    return warped_pcd[s_id[valid_id]], valid_id
