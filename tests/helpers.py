"""Shared test helpers: fixture loading and tolerance reporting."""
import os

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def load_cfg(name):
    return yaml.safe_load(open(os.path.join(ROOT, "cfgs", name)))


def load_npz(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def tiny_state(z, prefix="sd/"):
    """state_dict (name -> tensor) stored in the tiny fixture."""
    sd = {}
    for k in z.files:
        if k.startswith(prefix):
            sd[k[len(prefix):]] = torch.from_numpy(z[k].copy())
    return sd


def rel_err(a, b):
    a = torch.as_tensor(a).detach().to(torch.float64)
    b = torch.as_tensor(b).detach().to(torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def assert_close(a, b, rtol, name="", atol=0.0):
    """max|a-b| <= rtol * max|b| + atol  (SURVEY section 8c policy: tolerance scaled by the tensor's magnitude).
    atol is for tensors that are mathematically zero (e.g. the gradient of a conv bias feeding a BatchNorm)."""
    a = torch.as_tensor(a).detach().to(torch.float64)
    b = torch.as_tensor(b).detach().to(torch.float64)
    assert a.shape == b.shape, f"{name}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = float((a - b).abs().max()) if a.numel() else 0.0
    bound = rtol * float(b.abs().max() if b.numel() else 0.0) + atol
    assert err <= bound, f"{name}: max error {err:.3e} > {bound:.3e} (rtol {rtol:.1e}, atol {atol:.1e})"
    return err
