"""Import machinery for tests/golden/make_golden.py: makes the reference's own Python
(/root/reference/cbench) importable in THIS container so golden vectors can be generated from it.

Nothing from the reference is copied: its modules are imported from where they lie.  Missing
third-party packages are replaced by (a) functional restatements for compressai
(oracle/compressai_restated.py), (b) inert dummies for packages the hot path never executes
(pytorch_lightning, torchvision, pytorch_msssim, entmax, tensorboard, thop, ...), and the native
extensions cbench.ans / cbench.rans are the reference's own sources compiled into oracle/_ref.
This file is only ever used here (the reference does not exist on the GPU box).
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

REF = os.environ.get("CBENCH_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

_DUMMY_ROOTS = ("pytorch_lightning", "torchvision", "pytorch_msssim", "entmax", "tensorboard", "thop", "survae", "lpips",
                "zstandard", "oss2", "timm", "einops_exts", "torchmetrics", "compressai")


class _Anything:
    """Class usable as a base class / decorator / callable placeholder."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


class _DummyModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (_Anything,), {})


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _DUMMY_ROOTS and fullname not in sys.modules:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _DummyModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    sys.dont_write_bytecode = True
    if not any(isinstance(f, _Finder) for f in sys.meta_path):
        sys.meta_path.append(_Finder())
    # functional compressai pieces
    from oracle import compressai_restated as cr
    from oracle import rans_oracle as ro

    def mod(name, **attrs):
        m = _DummyModule(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("compressai")
    mod("compressai.layers", GDN=cr.GDN, MaskedConv2d=cr.MaskedConv2d)
    mod("compressai.models")
    mod("compressai.models.utils", conv=cr.conv, deconv=cr.deconv, update_registered_buffers=cr.update_registered_buffers)
    mod("compressai.entropy_models", EntropyBottleneck=cr.EntropyBottleneck, GaussianConditional=cr.GaussianConditional)
    mod("compressai.ops")
    mod("compressai.ops.bound_ops", LowerBound=cr.LowerBound, LowerBoundFunction=cr.LowerBoundFunction)
    mod("compressai.ops.parametrizers", NonNegativeParametrizer=cr.NonNegativeParametrizer)
    mod("compressai.ans", BufferedRansEncoder=cr.BufferedRansEncoder, RansEncoder=cr.RansEncoder, RansDecoder=cr.RansDecoder)
    # tensorboard writer is imported at module import time by the reference
    tb = mod("torch.utils.tensorboard")
    tb.SummaryWriter = _Anything
    mod("torch.utils.tensorboard.writer", SummaryWriter=_Anything)
    # the reference's native extensions, built from its own sources by oracle/Makefile
    ans, rans = ro.load_ref()
    if ans is None:
        raise RuntimeError("oracle/_ref is missing: run `make -C oracle ref` (needs /root/reference)")
    import cbench  # noqa: F401  (the reference package)
    sys.modules["cbench.ans"] = ans
    sys.modules["cbench.rans"] = rans
    cbench.ans, cbench.rans = ans, rans
    return cbench
