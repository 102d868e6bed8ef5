"""Host-side operator layer: thin wrappers that allocate outputs with torch and enqueue the HIP kernels of
libhydranet_hip.so, plus the torch.autograd.Function objects that give them a backward.

This is the same plug-in point the reference uses for its one hand-written fwd/bwd op (SwishImplementation,
model/net/common.py:11-22): torch.autograd.Function.forward/backward.  Tensors here are NHWC bf16 ("channels last"):
shape [N, H, W, C], unit channel stride, possibly a channel-slice view of a wider buffer (row stride = stride(2)).
There is no eager / CPU fallback in this module: every op goes through lib().call and raises if the library is missing.
"""
from __future__ import annotations

import sys
import types

from . import core, backbone, neck, seg, heads, losses, xstage      # noqa: F401
from .core import *          # noqa: F401,F403
from .backbone import *      # noqa: F401,F403
from .neck import *          # noqa: F401,F403
from .seg import *           # noqa: F401,F403
from .heads import *         # noqa: F401,F403
from .losses import *        # noqa: F401,F403
from .xstage import *        # noqa: F401,F403

_SUBMODULES = (core, backbone, neck, seg, heads, losses, xstage)


class _OpsPackage(types.ModuleType):
    """One namespace over the section modules (round 4: ops.py split along its section banners).  The policy switches (DEFER_WGRAD,
    EPILOGUE_STATS, SEG_FWD_PHASE, ...) are module globals read by the functions of the section that defines them; tests and tools set
    them on this package (`K.DEFER_WGRAD = False`), so an assignment here is forwarded to every section that holds a binding of that name."""

    def __setattr__(self, name, value):
        for m in _SUBMODULES:
            if name in m.__dict__:
                m.__dict__[name] = value
        super().__setattr__(name, value)


sys.modules[__name__].__class__ = _OpsPackage
