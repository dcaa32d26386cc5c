// ndp_graph.cpp -- host-side (plain C++, no GPU, no Eigen, no pybind) construction of the embedded-deformation graph of
// the N-ICP comparison baseline.  It restates what the reference's `MVRegC` extension does for
// get_deformation_graph_from_depthmap (/root/reference/model/geometry.py:155-380):
//
//   depth image -> triangle mesh            cxx/cpu/image_proc.cpp:58-196   (depthToMesh)
//   eroded-vertex mask                      cxx/cpu/graph_proc.cpp:16-78    (erode_mesh)
//   greedy node sampling by coverage        cxx/cpu/graph_proc.cpp:81-139   (sample_nodes, no shuffle: NICP.yaml)
//   geodesic node edges + weights           cxx/cpu/graph_proc.cpp:161-310  (compute_edges_geodesic)
//   removal of nodes with <= 1 neighbour    cxx/cpu/graph_proc.cpp:409-458  (node_and_edge_clean_up)
//   per-pixel anchors + skinning weights    cxx/cpu/graph_proc.cpp:504-641  (compute_pixel_anchors_geodesic)
//
// Same visiting orders as upstream wherever the result depends on them (pixels row-major, vertex neighbours in ascending
// index, a binary min-heap on the geodesic distance, anchors taken in ascending NODE id -- upstream sorts a set by
// distance and then iterates the id-ordered map, which is what the data it hands to the optimiser look like).  The
// node-to-vertex distance matrix (nodes x vertices floats upstream: hundreds of MB for a VGA depth map) is kept sparse:
// each vertex remembers the (node, distance) pairs that reached it.
// Bit-exactness against the reference's own build of MVRegC is asserted by tests/test_ed.py (golden F15).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <queue>
#include <set>
#include <utility>
#include <vector>

namespace {
constexpr int GRAPH_K = 6;                     // anchors per pixel (cxx/cpu/graph_proc.h:8)

struct V3 {
    float x, y, z;
};
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float sq(V3 a) { return a.x * a.x + (a.y * a.y + a.z * a.z); }               // Eigen's unrolled reduction of a 3-vector: x*x + (y*y + z*z)
inline float norm(V3 a) { return std::sqrt(sq(a)); }
inline float anchor_weight(float dist, float coverage) { return std::exp(-(dist * dist) / (2.f * coverage * coverage)); }

struct HeapLess {                              // min-heap on the distance (graph_proc.cpp:145-149)
    bool operator()(const std::pair<int, float> &l, const std::pair<int, float> &r) const { return l.second > r.second; }
};
}  // namespace

extern "C" {

// point_image [3][H][W] (X, Y, Z planes; Z <= 0: invalid).  Outputs sized by the caller for the worst case:
// vertices [H*W][3], vertex_pixels [H*W][2] (x, y), faces [2*H*W][3].  Returns the counts through n_vert / n_face.
int ndp_depth_to_mesh(const float *pim, int H, int W, float max_edge, float *vertices, int *vertex_pixels, int *faces,
                      int *n_vert, int *n_face) {
    if (!pim || !vertices || !vertex_pixels || !faces || !n_vert || !n_face || H < 2 || W < 2) return -1;
    const size_t plane = (size_t)H * W;
    std::vector<int> pix2v(plane, -1);
    int nv = 0, nf = 0;
    auto obs = [&](int y, int x) { const size_t i = (size_t)y * W + x; return V3{pim[i], pim[plane + i], pim[2 * plane + i]}; };
    auto vertex_of = [&](int y, int x, V3 p) {
        int &slot = pix2v[(size_t)y * W + x];
        if (slot == -1) {
            slot = nv;
            vertices[3 * nv] = p.x; vertices[3 * nv + 1] = p.y; vertices[3 * nv + 2] = p.z;
            vertex_pixels[2 * nv] = x; vertex_pixels[2 * nv + 1] = y;
            ++nv;
        }
        return slot;
    };
    for (int y = 0; y < H - 1; ++y)
        for (int x = 0; x < W - 1; ++x) {
            const V3 p00 = obs(y, x), p01 = obs(y + 1, x), p10 = obs(y, x + 1), p11 = obs(y + 1, x + 1);
            const bool v00 = p00.z > 0, v01 = p01.z > 0, v10 = p10.z > 0, v11 = p11.z > 0;
            if (v00 && v01 && v10 && norm(sub(p00, p01)) <= max_edge && norm(sub(p00, p10)) <= max_edge && norm(sub(p01, p10)) <= max_edge) {
                // (look the three slots up before any is created: creation order is 00, 01, 10)
                const int a = vertex_of(y, x, p00), b = vertex_of(y + 1, x, p01), c = vertex_of(y, x + 1, p10);
                faces[3 * nf] = a; faces[3 * nf + 1] = b; faces[3 * nf + 2] = c;
                ++nf;
            }
            if (v01 && v10 && v11 && norm(sub(p10, p01)) <= max_edge && norm(sub(p10, p11)) <= max_edge && norm(sub(p01, p11)) <= max_edge) {
                const int a = vertex_of(y + 1, x + 1, p11), b = vertex_of(y, x + 1, p10), c = vertex_of(y + 1, x, p01);
                faces[3 * nf] = a; faces[3 * nf + 1] = b; faces[3 * nf + 2] = c;
                ++nf;
            }
        }
    *n_vert = nv;
    *n_face = nf;
    return 0;
}

// mask [n_vert] bytes: 1 for vertices of faces that survive n_iter rounds of "drop faces touching a vertex with fewer than
// min_neighbors incident faces"
int ndp_erode_mesh(int n_vert, const int *faces, int n_face, int n_iter, int min_neighbors, unsigned char *mask) {
    if (n_vert < 0 || n_face < 0 || (n_face && !faces) || (n_vert && !mask)) return -1;
    std::vector<int> live(n_face);
    for (int i = 0; i < n_face; ++i) live[i] = i;
    for (int it = 0; it < n_iter; ++it) {
        std::vector<int> cnt(n_vert, 0);
        for (int f : live) { cnt[faces[3 * f]]++; cnt[faces[3 * f + 1]]++; cnt[faces[3 * f + 2]]++; }
        std::vector<int> keep;
        keep.reserve(live.size());
        for (int f : live)
            if (cnt[faces[3 * f]] >= min_neighbors && cnt[faces[3 * f + 1]] >= min_neighbors && cnt[faces[3 * f + 2]] >= min_neighbors) keep.push_back(f);
        live.swap(keep);
    }
    std::memset(mask, 0, (size_t)n_vert);
    for (int f : live) { mask[faces[3 * f]] = 1; mask[faces[3 * f + 1]] = 1; mask[faces[3 * f + 2]] = 1; }
    return 0;
}

// greedy coverage sampling in vertex order: a vertex becomes a node unless an earlier node lies within `coverage`.
// node_index [<= n_vert]; returns the node count.
int ndp_sample_nodes(const float *vertices, int n_vert, const unsigned char *valid, float coverage, int only_valid, int *node_index) {
    if (n_vert < 0 || (n_vert && (!vertices || !node_index))) return -1;
    const float c2 = coverage * coverage;
    std::vector<V3> nodes;
    int n = 0;
    for (int v = 0; v < n_vert; ++v) {
        if (only_valid && valid && !valid[v]) continue;
        const V3 p{vertices[3 * v], vertices[3 * v + 1], vertices[3 * v + 2]};
        bool is_node = true;
        for (const V3 &q : nodes)
            if (sq(sub(p, q)) <= c2) { is_node = false; break; }
        if (is_node) { nodes.push_back(p); node_index[n++] = v; }
    }
    return n;
}

// Opaque result of the geodesic pass: for every vertex the (node, distance) pairs of the nodes whose front reached it.
struct ndp_geodesic {
    std::vector<std::vector<std::pair<int, float>>> reach;       // per vertex, in ascending node id (nodes are processed in order)
};

// Dijkstra from every node over the mesh edges, fronts cut at 2 * coverage (unless enforce_total): the first
// max_neighbors OTHER nodes met become the node's edges (ids, normalised weights exp(-d^2 / 2 c^2), distances).
// edges / weights / dists: [n_node][max_neighbors], pre-filled by this function with -1 / 0 / 0.
ndp_geodesic *ndp_edges_geodesic(const float *vertices, int n_vert, const unsigned char *valid, const int *faces, int n_face,
                                 const int *node_index, int n_node, int max_neighbors, float coverage, int only_valid,
                                 int enforce_total, int *edges, float *weights, float *dists) {
    if (!vertices || !faces || !node_index || !edges || !weights || !dists) return nullptr;
    const float max_influence = 2.f * coverage;
    std::vector<std::set<int>> nb(n_vert);
    for (int f = 0; f < n_face; ++f)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k)
                if (faces[3 * f + j] != faces[3 * f + k]) nb[faces[3 * f + j]].insert(faces[3 * f + k]);
    std::vector<int> v2n(n_vert, -1);
    for (int n = 0; n < n_node; ++n)
        if (node_index[n] >= 0) v2n[node_index[n]] = n;
    for (int i = 0; i < n_node * max_neighbors; ++i) { edges[i] = -1; weights[i] = 0.f; dists[i] = 0.f; }
    auto *geo = new ndp_geodesic;
    geo->reach.resize(n_vert);
    std::vector<int> stamp(n_vert, -1);                            // visited-by-node marker
    for (int n = 0; n < n_node; ++n) {
        if (node_index[n] < 0) continue;
        std::priority_queue<std::pair<int, float>, std::vector<std::pair<int, float>>, HeapLess> heap;
        heap.push({node_index[n], 0.f});
        std::vector<int> ids;
        std::vector<float> ws, ds;
        while (!heap.empty()) {
            const auto top = heap.top();
            heap.pop();
            const int v = top.first;
            const float d = top.second;
            if (stamp[v] == n) continue;
            const int other = v2n[v];
            if (other >= 0 && other != n) {
                ids.push_back(other); ws.push_back(anchor_weight(d, coverage)); ds.push_back(d);
                if ((int)ids.size() >= max_neighbors) break;
            }
            geo->reach[v].push_back({n, d});
            stamp[v] = n;
            const V3 p{vertices[3 * v], vertices[3 * v + 1], vertices[3 * v + 2]};
            for (int u : nb[v]) {
                if (only_valid && valid && !valid[u]) continue;
                const V3 q{vertices[3 * u], vertices[3 * u + 1], vertices[3 * u + 2]};
                const float du = d + norm(sub(p, q));
                if (enforce_total || du <= max_influence) heap.push({u, du});
            }
        }
        float wsum = 0.f;
        for (size_t i = 0; i < ids.size(); ++i) { edges[n * max_neighbors + i] = ids[i]; wsum += ws[i]; }
        for (size_t i = 0; i < ids.size(); ++i) {
            weights[n * max_neighbors + i] = wsum > 0 ? ws[i] / wsum : ws[i] / (float)ids.size();
            dists[n * max_neighbors + i] = ds[i];
        }
    }
    return geo;
}
void ndp_geodesic_free(ndp_geodesic *g) { delete g; }

// iteratively drop nodes that keep at most one live neighbour (valid [n_node] bytes, updated in place)
int ndp_node_cleanup(const int *edges, int n_node, int max_neighbors, unsigned char *valid) {
    if (!edges || !valid) return -1;
    std::vector<char> removed(n_node, 0);
    for (;;) {
        int newly = 0;
        for (int n = 0; n < n_node; ++n) {
            if (!valid[n]) continue;
            int cnt = 0;
            for (int i = 0; i < max_neighbors; ++i) {
                const int e = edges[n * max_neighbors + i];
                if (e == -1) break;
                if (removed[e]) continue;
                ++cnt;
            }
            if (cnt <= 1) { valid[n] = 0; removed[n] = 1; ++newly; }
        }
        if (!newly) break;
    }
    return 0;
}

// anchors [H][W][6] (-1 padded), weights [H][W][6]: for every mesh vertex the first six VALID nodes (ascending id) whose
// geodesic front reached it, with normalised skinning weights
int ndp_pixel_anchors(const ndp_geodesic *geo, const unsigned char *valid_node, const int *vertex_pixels, int n_vert, int H, int W,
                      float coverage, int *anchors, float *weights) {
    if (!geo || !valid_node || !vertex_pixels || !anchors || !weights) return -1;
    const size_t n = (size_t)H * W * GRAPH_K;
    for (size_t i = 0; i < n; ++i) { anchors[i] = -1; weights[i] = 0.f; }
    for (int v = 0; v < n_vert; ++v) {
        const int u = vertex_pixels[2 * v], y = vertex_pixels[2 * v + 1];
        int ids[GRAPH_K];
        float w[GRAPH_K];
        int cnt = 0;
        float wsum = 0.f;
        for (const auto &nd : geo->reach[v]) {
            if (!valid_node[nd.first]) continue;
            ids[cnt] = nd.first;
            w[cnt] = anchor_weight(nd.second, coverage);
            wsum += w[cnt];
            if (++cnt == GRAPH_K) break;
        }
        for (int i = 0; i < cnt; ++i) {
            const size_t o = ((size_t)y * W + u) * GRAPH_K + i;
            anchors[o] = ids[i];
            weights[o] = wsum > 0 ? w[i] / wsum : 1.f / (float)cnt;
        }
    }
    return 0;
}

}  // extern "C"
