"""Embedded-deformation graph of the N-ICP comparison baseline: the host side of
/root/reference/model/geometry.py:80-380 -- `depth_2_pc`, `depth_to_mesh`, `get_deformation_graph_from_depthmap`,
`map_pixel_to_pcd`, `pc_2_uv` -- on the native graph builder of libndp_host.so (csrc/ndp_graph.cpp) instead of the
reference's `MVRegC` extension.  Plain numpy + C++; not on the hot path (it runs once per pair, before the optimisation).
"""
import ctypes

import numpy as np
import torch

from . import _native as N

GRAPH_K = 6


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def depth_2_pc(depth, intrin):
    """[H,W] metric depth, 3x3 intrinsics -> [3,H,W] camera-space points (geometry.py:100-117; float64 like upstream)."""
    fx, cx, fy, cy = intrin[0, 0], intrin[0, 2], intrin[1, 1], intrin[1, 2]
    height, width = depth.shape
    u = np.arange(width) * np.ones([height, width])
    v = np.transpose(np.arange(height) * np.ones([width, height]))
    return np.stack([(u - cx) * depth / fx, (v - cy) * depth / fy, depth])


def map_pixel_to_pcd(valid_pix_mask):
    """pixel -> index into the cloud of valid pixels, -1 elsewhere (geometry.py:54-63)."""
    image_size = valid_pix_mask.shape
    pix_2_pcd_map = torch.cumsum(valid_pix_mask.view(-1), dim=0).view(image_size).long() - 1
    pix_2_pcd_map[~valid_pix_mask] = -1
    return pix_2_pcd_map


def pc_2_uv(pcd, intrin):
    """geometry.py:81-93"""
    X, Y, Z = pcd[:, 0], pcd[:, 1], pcd[:, 2]
    fx, cx, fy, cy = intrin[0, 0], intrin[0, 2], intrin[1, 1], intrin[1, 2]
    u = (fx * X / Z + cx).to(torch.long)
    v = (fy * Y / Z + cy).to(torch.long)
    return torch.stack([u, v], -1)


def depth_to_mesh(depth_image, mask_image, intrin, depth_scale=1000., max_triangle_distance=0.04):
    """geometry.py:120-148 -> (vertices [V,3] f32, faces [F,3] i32, vertex_pixels [V,2] i32 (x, y), point_image [3,H,W] f32)."""
    H, W = depth_image.shape
    mask = (mask_image > 0).astype(depth_image.dtype)
    point_image = np.ascontiguousarray(depth_2_pc((depth_image * mask) / depth_scale, intrin).astype(np.float32))
    vertices = np.empty((H * W, 3), np.float32)
    pixels = np.empty((H * W, 2), np.int32)
    faces = np.empty((2 * H * W, 3), np.int32)
    nv, nf = ctypes.c_int(), ctypes.c_int()
    rc = N.host_lib().ndp_depth_to_mesh(_p(point_image), H, W, float(max_triangle_distance), _p(vertices), _p(pixels), _p(faces),
                                        ctypes.byref(nv), ctypes.byref(nf))
    if rc:
        raise N.NdpError("ndp_depth_to_mesh failed")
    return vertices[:nv.value].copy(), faces[:nf.value].copy(), pixels[:nv.value].copy(), point_image


def get_deformation_graph_from_depthmap(depth_image, intrin, config):
    """geometry.py:155-380: mesh from the depth map, greedy node sampling by `node_coverage`, geodesic edges with
    exp(-d^2 / 2 c^2) weights, optional removal of poorly connected nodes, per-pixel anchors (GRAPH_K = 6) and skinning
    weights.  Returns the dict upstream returns (graph_clusters is not computed: nothing on the path reads it)."""
    L = N.host_lib()
    H, W = depth_image.shape
    vertices, faces, vertex_pixels, point_image = depth_to_mesh(depth_image, depth_image > 0, intrin,
                                                                max_triangle_distance=config.max_triangle_distance, depth_scale=1000.)
    nv, nf = vertices.shape[0], faces.shape[0]
    assert nv > 0 and nf > 0
    non_eroded = np.zeros(nv, np.uint8)
    L.ndp_erode_mesh(nv, _p(faces), nf, 0, 0, _p(non_eroded))                                   # erode_mesh(vertices, faces, 0, 0)
    if config.SAMPLE_RANDOM_SHUFFLE:
        raise N.NdpError("SAMPLE_RANDOM_SHUFFLE: upstream shuffles with std::random_device (not reproducible); NICP.yaml has False")
    node_indices = np.empty(nv, np.int32)
    n_node = L.ndp_sample_nodes(_p(vertices), nv, _p(non_eroded), float(config.node_coverage),
                                int(bool(config.USE_ONLY_VALID_VERTICES)), _p(node_indices))
    node_indices = node_indices[:n_node].copy()
    node_coords = vertices[node_indices]
    K = int(config.num_neighbors)
    graph_edges = np.empty((n_node, K), np.int32)
    graph_edges_weights = np.empty((n_node, K), np.float32)
    graph_edges_distances = np.empty((n_node, K), np.float32)
    visible = np.ones(nv, np.uint8)                                                            # geometry.py:150 passes all-ones
    geo = L.ndp_edges_geodesic(_p(vertices), nv, _p(visible), _p(faces), nf, _p(node_indices), n_node, K, float(config.node_coverage),
                               int(bool(config.USE_ONLY_VALID_VERTICES)), int(bool(config.ENFORCE_TOTAL_NUM_NEIGHBORS)),
                               _p(graph_edges), _p(graph_edges_weights), _p(graph_edges_distances))
    if not geo:
        raise N.NdpError("ndp_edges_geodesic failed")
    try:
        valid_nodes = np.ones(n_node, np.uint8)
        if config.REMOVE_NODES_WITH_NOT_ENOUGH_NEIGHBORS:
            L.ndp_node_cleanup(_p(graph_edges), n_node, K, _p(valid_nodes))
        pixel_anchors = np.empty((H, W, GRAPH_K), np.int32)
        pixel_weights = np.empty((H, W, GRAPH_K), np.float32)
        L.ndp_pixel_anchors(ctypes.c_void_p(geo), _p(valid_nodes), _p(vertex_pixels), nv, H, W, float(config.node_coverage),
                            _p(pixel_anchors), _p(pixel_weights))
    finally:
        L.ndp_geodesic_free(ctypes.c_void_p(geo))
    keep = valid_nodes.astype(bool)
    if not keep.all():                                                                          # geometry.py:262-330: re-number the survivors
        remap = -np.ones(n_node + 1, np.int64)
        remap[:n_node][keep] = np.arange(int(keep.sum()))
        for n in range(n_node):
            ids, ws, ds = graph_edges[n].copy(), graph_edges_weights[n].copy(), graph_edges_distances[n].copy()
            ok = np.array([(i == -1) or keep[i] for i in ids])
            graph_edges[n], graph_edges_weights[n], graph_edges_distances[n] = -1, 0.0, 0.0
            m = int(ok.sum())
            graph_edges[n, :m] = remap[ids[ok]]
            graph_edges_weights[n, :m], graph_edges_distances[n, :m] = ws[ok], ds[ok]
            s = graph_edges_weights[n].sum()
            if keep[n] and not s > 0:
                raise N.NdpError("a surviving node lost all its neighbours")
            if s > 0:
                graph_edges_weights[n] /= s
        pixel_anchors = remap[pixel_anchors].astype(np.int32)
        node_coords, graph_edges = node_coords[keep], graph_edges[keep]
        graph_edges_weights = graph_edges_weights[keep]
    if node_coords.shape[0] == 0:
        raise N.NdpError("the deformation graph has no nodes")
    return {
        "graph_nodes": torch.from_numpy(np.ascontiguousarray(node_coords)),
        "graph_edges": torch.from_numpy(graph_edges).long(),
        "graph_edges_weights": torch.from_numpy(graph_edges_weights),
        "graph_clusters": None,
        "pixel_anchors": torch.from_numpy(pixel_anchors),
        "pixel_weights": torch.from_numpy(pixel_weights),
        "point_image": torch.from_numpy(point_image).permute(1, 2, 0),
    }
