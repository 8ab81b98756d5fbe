"""Import harness for the upstream reference (``/root/reference``, read-only).

Used ONLY in the build container by ``tests/golden/make_golden.py`` (to emit the committed
fixtures) and by the optional ``reference_live`` CPU tests (skipped when the reference tree is
absent, as on the GPU box).  Nothing here copies reference source: it installs empty stub
modules for the reference's missing third-party imports, makes ``.cuda()`` the identity, and
plugs the CPU oracle in as ``hashencoder.backend._backend`` so the reference's own Python
(``code/hashencoder/hashgrid.py``, ``code/model/*.py``, ``code/utils/{rend_util,general}.py``)
runs on CPU tensors.  (SURVEY.md 8c lists why each shim is needed.)
"""
import importlib
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("NICER_REFERENCE_ROOT", "/root/reference")
REF_CODE = os.path.join(REF_ROOT, "code")


def available():
    return os.path.isdir(os.path.join(REF_CODE, "hashencoder"))


class Conf(dict):
    """dict-backed stand-in for the pyhocon ConfigTree accessors the reference calls."""

    def _get(self, key, default):
        cur = self
        for part in key.split("."):
            if not isinstance(cur, dict) or part not in cur:
                if default is _MISSING:
                    raise KeyError(key)
                return default
            cur = cur[part]
        return cur

    def get_int(self, k, default=None): return int(self._get(k, _MISSING if default is None else default))
    def get_float(self, k, default=None): return float(self._get(k, _MISSING if default is None else default))
    def get_bool(self, k, default=None): return bool(self._get(k, _MISSING if default is None else default))
    def get_string(self, k, default=None): return str(self._get(k, _MISSING if default is None else default))
    def get_list(self, k, default=None): return list(self._get(k, _MISSING if default is None else default))

    def get_config(self, k, default=None):
        v = self._get(k, _MISSING if default is None else default)
        return Conf(v)


_MISSING = object()
_installed = False


def install(backend):
    """Install the shims; ``backend`` becomes ``hashencoder.backend._backend``."""
    global _installed
    if not available():
        raise RuntimeError("reference tree not present")
    if not _installed:
        def stub(name, **attrs):
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
            return m

        stub("cachetools", cached=lambda *a, **k: (lambda f: f))
        for name in ("cv2", "imageio", "trimesh", "lpips", "open3d"):
            stub(name)
        sk = stub("skimage"); sk.measure = stub("skimage.measure"); sk.metrics = stub("skimage.metrics")
        tv = stub("torchvision"); tv.transforms = stub("torchvision.transforms", ToPILImage=object)
        stub("pytorch_msssim", SSIM=object)
        stub("easydict", EasyDict=dict)
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
        if REF_CODE not in sys.path:
            sys.path.insert(0, REF_CODE)
        # the reference's "hashencoder" package imports .backend (a CUDA JIT build) at import
        # time: pre-seed that submodule so the package's own hashgrid.py loads untouched.
        pkg = types.ModuleType("hashencoder")
        pkg.__path__ = [os.path.join(REF_CODE, "hashencoder")]
        sys.modules["hashencoder"] = pkg
        _installed = True
    be = types.ModuleType("hashencoder.backend")
    be._backend = backend
    sys.modules["hashencoder.backend"] = be
    if "hashencoder.hashgrid" in sys.modules:
        sys.modules["hashencoder.hashgrid"]._backend = backend
    return importlib.import_module("hashencoder.hashgrid")


def import_ref(name):
    """import_ref('model.network') etc. (after install())."""
    return importlib.import_module(name)
