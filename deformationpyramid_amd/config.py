"""YAML config loading with the reference's conventions (/root/reference/eval_nolearned.py:17-20,
33-40,50-53): a `!join` constructor, attribute access, `snapshot_dir` and `device` injected."""
import os

import torch
import yaml


class Config(dict):
    """Minimal attribute-accessible dict (stands in for easydict, which this image lacks)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, Config):
            v = Config(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = __setitem__


class _Loader(yaml.SafeLoader):
    pass


def _join(loader, node):
    return "_".join(str(i) for i in loader.construct_sequence(node))


_Loader.add_constructor("!join", _join)


def load_config(path, device=None, make_dirs=False):
    with open(path, "r") as f:
        cfg = Config(yaml.load(f, Loader=_Loader))
    if "folder" in cfg and "exp_dir" in cfg:
        cfg["snapshot_dir"] = "snapshot/%s/%s" % (cfg["folder"], cfg["exp_dir"])
        if make_dirs:
            os.makedirs(cfg["snapshot_dir"], exist_ok=True)
    if device is not None:
        cfg["device"] = device
    elif cfg.get("gpu_mode", False) and torch.cuda.is_available():
        cfg["device"] = torch.cuda.current_device()       # an int, as upstream
    else:
        cfg["device"] = torch.device("cpu")
    return cfg
