"""ASCII-PLY reader / writer and area-weighted surface sampler for the shape-transfer driver.

Stands in for the three open3d calls of the reference's shape_transfer.py (`read_triangle_mesh`,
`sample_points_uniformly`, vertex replacement + `write_triangle_mesh`; /root/reference/shape_transfer.py:69-81,
162-170) -- open3d is not in this image.  Plain numpy; not on the hot path.
"""
import numpy as np


def read_ply_ascii(path):
    """-> (vertices [V,3] float32, faces [F,3] int64).  Polygons with more than 3 vertices are fan-triangulated."""
    with open(path, "r") as f:
        if f.readline().strip() != "ply":
            raise ValueError("not a PLY file")
        fmt = f.readline().split()
        if fmt[:2] != ["format", "ascii"]:
            raise ValueError("only ASCII PLY is supported")
        n_vert = n_face = 0
        vprops, current = [], None
        for line in f:
            tok = line.split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "element":
                current = tok[1]
                if current == "vertex":
                    n_vert = int(tok[2])
                elif current == "face":
                    n_face = int(tok[2])
            elif tok[0] == "property" and current == "vertex":
                vprops.append(tok[-1])
            elif tok[0] == "end_header":
                break
        ix, iy, iz = vprops.index("x"), vprops.index("y"), vprops.index("z")
        verts = np.empty((n_vert, 3), dtype=np.float32)
        for i in range(n_vert):
            tok = f.readline().split()
            verts[i] = (float(tok[ix]), float(tok[iy]), float(tok[iz]))
        faces = []
        for _ in range(n_face):
            tok = f.readline().split()
            k = int(tok[0])
            idx = [int(t) for t in tok[1:1 + k]]
            for j in range(1, k - 1):
                faces.append((idx[0], idx[j], idx[j + 1]))
    return verts, np.asarray(faces, dtype=np.int64).reshape(-1, 3)


def write_ply_ascii(path, verts, faces):
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\n")
        f.write(f"element vertex {len(verts)}\nproperty float x\nproperty float y\nproperty float z\n")
        f.write(f"element face {len(faces)}\nproperty list uchar uint vertex_indices\nend_header\n")
        for v in verts:
            f.write(f"{v[0]:.6f} {v[1]:.6f} {v[2]:.6f}\n")
        for t in faces:
            f.write(f"3 {t[0]} {t[1]} {t[2]}\n")


def sample_surface(verts, faces, n, rng):
    """n points uniform in area over the triangle soup (what o3d's sample_points_uniformly draws; the random
    stream is numpy's, so the points differ from open3d's for the same seed)."""
    a, b, c = verts[faces[:, 0]].astype(np.float64), verts[faces[:, 1]].astype(np.float64), verts[faces[:, 2]].astype(np.float64)
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    tri = rng.choice(len(faces), size=n, p=area / area.sum())
    r1, r2 = np.sqrt(rng.random(n)), rng.random(n)
    w0, w1, w2 = 1.0 - r1, r1 * (1.0 - r2), r1 * r2
    pts = w0[:, None] * a[tri] + w1[:, None] * b[tri] + w2[:, None] * c[tri]
    return pts.astype(np.float32)
