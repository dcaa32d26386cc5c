"""Flat parameter layout of one NDP level -- Python mirror of include/ndp_types.h.

The reference keeps each level as an nn.Module with 10-14 small tensors
(/root/reference/model/nets.py:67-109).  Here a level is ONE contiguous float32 block
(so a whole pyramid, its gradients and its Adam moments are three flat HBM arrays):

    [ W0 (W*6) | b0 (W) | W1 (W*W) | b1 (W) | ... | Wh (NH*W) | bh (NH) ]

Head rows follow the reference's module registration order: rot rows, Sim3 scale row,
3 translation rows, nonrigidity row.
"""
import ctypes
from dataclasses import dataclass

MOTIONS = {"SE3": 0, "Sim3": 1, "sflow": 2}
ROTFMTS = {"axis_angle": 0, "euler": 1, "quaternion": 2, "6D": 3}


class CLayerDesc(ctypes.Structure):
    """struct ndp_layer_desc (include/ndp_types.h)."""
    _fields_ = [("width", ctypes.c_int), ("n_hidden", ctypes.c_int), ("motion", ctypes.c_int),
                ("rotfmt", ctypes.c_int), ("nonrigidity", ctypes.c_int), ("mlp_scale", ctypes.c_float)]


@dataclass(frozen=True)
class LayerDesc:
    width: int = 128
    n_hidden: int = 2          # depth - 1
    motion: str = "SE3"
    rotfmt: str = "axis_angle"
    nonrigidity: bool = False
    mlp_scale: float = 0.001   # nets.py:107

    def __post_init__(self):
        # nets.py:17
        assert self.motion in MOTIONS, f"motion must be one of {list(MOTIONS)}"
        if self.motion != "sflow" and self.rotfmt not in ROTFMTS:
            raise KeyError(self.rotfmt)

    # --- head bookkeeping (ndp_types.h: ndp_n_rot / ndp_n_heads / ndp_head_row_*) ---
    @property
    def n_rot(self):
        if self.motion == "sflow":
            return 0
        return {"quaternion": 4, "6D": 6}.get(self.rotfmt, 3)

    @property
    def n_heads(self):
        return self.n_rot + (1 if self.motion == "Sim3" else 0) + 3 + (1 if self.nonrigidity else 0)

    @property
    def row_scale(self):
        return self.n_rot

    @property
    def row_trn(self):
        return self.n_rot + (1 if self.motion == "Sim3" else 0)

    @property
    def row_nr(self):
        return self.row_trn + 3

    # --- offsets in floats ---
    def off_W(self, i):
        """i = 0: input layer [W,6]; i >= 1: hidden layer i [W,W]."""
        W = self.width
        return 0 if i == 0 else W * 7 + (i - 1) * (W * W + W)

    def off_b(self, i):
        W = self.width
        return W * 6 if i == 0 else self.off_W(i) + W * W

    @property
    def off_Wh(self):
        return self.off_W(self.n_hidden + 1)

    @property
    def off_bh(self):
        return self.off_Wh + self.n_heads * self.width

    @property
    def param_count(self):
        return self.off_bh + self.n_heads

    def c_struct(self):
        rot = ROTFMTS.get(self.rotfmt, 0)
        return CLayerDesc(self.width, self.n_hidden, MOTIONS[self.motion], rot,
                          1 if self.nonrigidity else 0, self.mlp_scale)

    def named_slices(self):
        """Reference parameter name (nets.py module names) -> (offset, shape) in the flat block,
        in the reference's registration order (input, mlp, rot, s, trn, nr)."""
        W = self.width
        out = [("input.0.weight", self.off_W(0), (W, 6)), ("input.0.bias", self.off_b(0), (W,))]
        for i in range(1, self.n_hidden + 1):
            out.append((f"mlp.pts_linears.{i - 1}.weight", self.off_W(i), (W, W)))
            out.append((f"mlp.pts_linears.{i - 1}.bias", self.off_b(i), (W,)))

        def head(name, row, rows):
            out.append((f"{name}.weight", self.off_Wh + row * W, (rows, W)))
            out.append((f"{name}.bias", self.off_bh + row, (rows,)))

        if self.n_rot:
            head("rot_brach", 0, self.n_rot)        # (sic) reference spelling, nets.py:85
        if self.motion == "Sim3":
            head("s_branch", self.row_scale, 1)
        head("trn_branch", self.row_trn, 3)
        if self.nonrigidity:
            head("nr_branch", self.row_nr, 1)
        return out
