"""ctypes binding of libhydranet_hip.so (C ABI declared in include/hydranet_hip.h).

The argument types of every entry point are parsed from the header itself, so the header is the single source of truth for
the boundary.  There is no fallback: if the shared library is missing or a symbol cannot be resolved, importing / calling
raises -- the product path never runs without the HIP kernels.
"""
from __future__ import annotations

import ctypes
import os
import re
import subprocess
from typing import Dict, List, Tuple

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
HEADER = os.path.join(_ROOT, "include", "hydranet_hip.h")
CSRC = os.path.join(_PKG, "csrc")
# HN_TUNING=1 (tools/ only): a second build with -DHN_TUNING -- the kernels' ablation bits, in-kernel stamps and the never-shipped template
# instantiations behind hn_debug_knob / hn_debug_* -- as libhydranet_hip_tuning.so.  The product library is built without any of it.
TUNING = os.environ.get("HN_TUNING") == "1"
# HN_TUNING=ab (tools/ab_*.sh): the PRODUCT library, but the policy switches below and HN_LIB_AB are read from the environment for same-box
# A/B runs.  Without HN_TUNING the package reads no HN_* variable at all: every policy is the constant written in the source.
AB = TUNING or os.environ.get("HN_TUNING") == "ab"


def policy(name: str, default: str) -> str:
    """value of a tuning / A-B policy switch: the environment's only under HN_TUNING=1 | ab (tools/), else `default`"""
    return os.environ.get(name, default) if AB else default

SO_PATH = os.path.join(_PKG, "libhydranet_hip_tuning.so" if TUNING else "libhydranet_hip.so")
SOURCES = ["hn_gemm.hip", "hn_norm.hip", "hn_fused.hip", "hn_stencil.hip", "hn_loss.hip", "hn_post.hip", "hn_xstage.hip"]

_ERR = {1: "bad argument", 2: "kernel launch failure", 3: "unsupported shape"}


class HipKernelError(RuntimeError):
    pass


def _ctype(decl: str):
    decl = decl.strip()
    if "*" in decl or "hipStream_t" in decl:
        return ctypes.c_void_p
    words = decl.replace("const", " ").split()
    base = words[0]
    return {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double}[base]


TUNING_HEADER = os.path.join(_ROOT, "include", "hydranet_hip_tuning.h")


def parse_header(path: str = HEADER) -> Dict[str, Tuple[object, List[object], bool]]:
    """name -> (restype, argtypes, takes_stream).  The tuning build also binds include/hydranet_hip_tuning.h (hn_debug_*)."""
    txt = open(path).read()
    if TUNING and path == HEADER:
        txt += open(TUNING_HEADER).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|long)\s+(hn_\w+)\s*\((.*?)\)\s*;", txt, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        plist = [p for p in (q.strip() for q in params.replace("\n", " ").split(",")) if p and p != "void"]
        args = [_ctype(p) for p in plist]
        out[name] = (ctypes.c_int if ret == "int" else ctypes.c_long, args, bool(plist) and "hipStream_t" in plist[-1])
    return out


def sources() -> List[str]:
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into one shared library, in-tree."""
    srcs = sources()
    deps = srcs + [os.path.join(CSRC, "hn_common.h")]
    if not force and os.path.exists(SO_PATH) and all(os.path.getmtime(SO_PATH) >= os.path.getmtime(d) for d in deps):
        return SO_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(CSRC, os.path.basename(s).replace(".hip", ".tuning.o" if TUNING else ".o"))
        objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17"] + (["-DHN_TUNING"] if TUNING else []) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        outp = p.communicate()[0].decode()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), outp))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO_PATH] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stdout.decode()))
    return SO_PATH


class _Lib:
    def __init__(self):
        if not os.path.exists(SO_PATH):
            raise HipKernelError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU / eager fallback for the HydraNet hot path)")
        # HN_LIB_AB (only under HN_TUNING=ab, tools/): same-box A/B measurements of two builds (an alternate in-tree .so with the same ABI)
        self._dll = ctypes.CDLL(policy("HN_LIB_AB", "") or SO_PATH)
        self._sig = parse_header()
        self._fn = {}
        for name, (ret, args, has_stream) in self._sig.items():
            f = getattr(self._dll, name)          # AttributeError here = header/library mismatch: fail loudly
            f.restype = ret
            f.argtypes = args
            self._fn[name] = (f, has_stream)

    def symbols(self):
        return sorted(self._sig)

    def raw(self, name):
        return self._fn[name][0]

    def call(self, name, *args):
        """Invoke an int-returning entry point on torch's current HIP stream; raise on a non-zero status."""
        import torch
        f, has_stream = self._fn[name]
        if has_stream:
            args = args + (torch.cuda.current_stream().cuda_stream,)
        rc = f(*args)
        if rc != 0:
            raise HipKernelError(f"{name} failed: {_ERR.get(rc, rc)} (args={args})")

    def query(self, name, *args):
        """Invoke a host-side planning helper that returns a value (no stream, no status)."""
        if name not in self._fn and name.startswith("hn_debug_"):
            raise HipKernelError(f"{name} exists only in the tuning build of the library: set HN_TUNING=1 before importing the package "
                                 "(tools/ only; the product library has no tuning hooks)")
        return self._fn[name][0](*args)


_LIB = None


def lib() -> _Lib:
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB
