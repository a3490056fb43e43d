"""ctypes front-end of oracle/rans64_oracle.c -- CPU ORACLE, test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The classes mirror the reference's ``cbench.ans`` objects (csrc/ans/rans64.hpp:127-149)
closely enough that parity tests can drive oracle, oracle/_ref and the HIP path with the
same calls.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile the C restatement (and oracle/_ref when /root/reference is present)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle_rans.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.orc_rans64_encode.restype = ctypes.c_int64
        L.orc_rans64_encode_ar.restype = ctypes.c_int64
        L.orc_rans64_decode.restype = ctypes.c_int
        L.orc_rans64_decode_ar.restype = ctypes.c_int
        L.orc_rans64_encode_arop.restype = ctypes.c_int64
        L.orc_rans64_decode_arop.restype = ctypes.c_int
        L.orc_pmf_to_quantized_cdf.restype = ctypes.c_int
        L.orc_tables_from_freqs.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _i32(a):
    return np.ascontiguousarray(np.asarray(a).astype(np.int32))


def pmf_to_quantized_cdf(pmf, precision=16):
    pmf = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32))
    out = np.zeros(pmf.size + 1, dtype=np.int32)
    rc = lib().orc_pmf_to_quantized_cdf(_p(pmf), ctypes.c_int(pmf.size), ctypes.c_int(precision), _p(out))
    if rc:
        raise ValueError("pmf_to_quantized_cdf: no bin can donate mass")
    return out.tolist()


class _Base:
    def __init__(self, freq_precision=16, bypass_coding=True, bypass_precision=4):
        self.prec, self.bypass, self.bprec = int(freq_precision), int(bool(bypass_coding)), int(bypass_precision)
        self.cdfs = None
        self.ar = None

    def init_params(self, freqs, num_symbols, offsets):
        freqs, nsym = _i32(freqs), _i32(num_symbols).reshape(-1)
        if freqs.ndim != 2 or freqs.shape[0] != nsym.size:
            raise ValueError("freqs should be 2-dimensional with shape (num_symbols.size(), >num_symbols.max())")
        stride = int(nsym.max()) + 2
        self.cdfs = np.zeros((freqs.shape[0], stride), dtype=np.int32)
        self.sizes = np.zeros(freqs.shape[0], dtype=np.int32)
        lib().orc_tables_from_freqs(_p(freqs), ctypes.c_int(freqs.shape[0]), ctypes.c_int(freqs.shape[1]), _p(nsym),
                                    ctypes.c_int(self.prec), _p(self.cdfs), ctypes.c_int(stride), _p(self.sizes))
        self.offsets = _i32(offsets).reshape(-1)

    def init_cdf_params(self, cdfs, cdfs_sizes, offsets):
        cdfs, sizes = _i32(cdfs), _i32(cdfs_sizes).reshape(-1)
        if cdfs.ndim != 2 or cdfs.shape[0] != sizes.size:
            raise ValueError("cdfs should be 2-dimensional with shape (cdfs_sizes.size(), >cdfs_sizes.max())")
        self.cdfs, self.sizes, self.offsets = cdfs, sizes, _i32(offsets).reshape(-1)

    def init_ar_params(self, ar_table, ar_offsets):
        tab = _i32(ar_table)
        order = tab.ndim - 2
        if order not in (1, 2):
            raise ValueError("Too many dimensions!")
        self.ar = (tab, order)

    def init_custom_ar_ops(self, ops):
        """ANSBase::init_custom_ar_ops (ans_interface.hpp:40-48): a list of ar_limited_scaled_add_linear_op."""
        if len(ops):
            self.arops = np.ascontiguousarray([list(o.weight) + [0.0] * (3 - len(o.weight)) + [o.bias, o.scale, o.min, o.max] for o in ops],
                                              dtype=np.float32)

    def get_cdfs(self):
        return self.cdfs[:, : int(self.sizes.max())].copy()

    def _aropargs(self, ar_indexes, ar_offsets, n):
        if ar_offsets is None:
            raise ValueError("ar_offsets is required for ar coding!")
        off = _i32(ar_offsets).reshape(-1, n)
        if off.shape[0] > 3:
            raise ValueError("Too many dimensions!")
        rows = [np.ascontiguousarray(r) for r in off]
        ai = _i32(ar_indexes).reshape(-1) if ar_indexes is not None else None
        keep = [rows, ai, self.arops]
        return keep, (_p(self.arops), ctypes.c_int(off.shape[0]), _p(ai) if ai is not None else None, _p(rows[0]),
                      _p(rows[1]) if len(rows) > 1 else None, _p(rows[2]) if len(rows) > 2 else None)

    def _targs(self):
        return (_p(self.cdfs), ctypes.c_int(self.cdfs.shape[1]), _p(self.sizes), _p(self.offsets),
                ctypes.c_int(self.prec), ctypes.c_int(self.bypass), ctypes.c_int(self.bprec))

    def _arargs(self, ar_indexes, ar_offsets, n):
        tab, order = self.ar
        if ar_offsets is None:
            raise ValueError("ar_offsets is required for ar coding!")
        ar_offsets = _i32(ar_offsets).reshape(order, n)
        keep = [tab, ar_offsets]
        ai = None
        if ar_indexes is not None:
            ai = _i32(ar_indexes).reshape(-1)
            keep.append(ai)
        o1 = _p(ar_offsets[1]) if order == 2 else None
        return keep, (_p(tab), ctypes.c_int(order), ctypes.c_int(tab.shape[1]), ctypes.c_int(tab.shape[2]),
                      _p(ai) if ai is not None else None, _p(ar_offsets[0]), o1)


class Rans64Encoder(_Base):
    def encode_with_indexes(self, symbols, indexes, ar_indexes=None, ar_offsets=None, cache=0):
        if self.cdfs is None:
            raise ValueError("ANS not initialized!")
        sym, idx = _i32(symbols).reshape(-1), _i32(indexes).reshape(-1)
        n = idx.size
        cap = 2 * n + 8
        out = np.empty(cap, dtype=np.uint32)
        if getattr(self, "arops", None) is not None:
            keep, a = self._aropargs(ar_indexes, ar_offsets, n)
            nw = lib().orc_rans64_encode_arop(*self._targs(), *a, _p(sym), _p(idx), ctypes.c_int64(n), _p(out), ctypes.c_int64(cap))
        elif self.ar is not None:
            keep, a = self._arargs(ar_indexes, ar_offsets, n)
            nw = lib().orc_rans64_encode_ar(*self._targs(), *a, _p(sym), _p(idx), ctypes.c_int64(n), _p(out),
                                            ctypes.c_int64(cap))
        else:
            nw = lib().orc_rans64_encode(*self._targs(), _p(sym), _p(idx), ctypes.c_int64(n), _p(out),
                                         ctypes.c_int64(cap))
        if nw < 0:
            raise RuntimeError("oracle encoder overflow")
        return out[cap - nw:].tobytes()


class Rans64Decoder(_Base):
    def decode_with_indexes(self, encoded, indexes, ar_indexes=None, ar_offsets=None):
        if self.cdfs is None:
            raise ValueError("ANS not initialized!")
        idx = _i32(indexes)
        words = np.frombuffer(encoded, dtype=np.uint32).copy()
        out = np.empty(idx.shape, dtype=np.int32)
        if getattr(self, "arops", None) is not None:
            keep, a = self._aropargs(ar_indexes, ar_offsets, idx.size)
            lib().orc_rans64_decode_arop(*self._targs(), *a, _p(words), _p(idx), ctypes.c_int64(idx.size), _p(out))
            return out
        if self.ar is not None:
            keep, a = self._arargs(ar_indexes, ar_offsets, idx.size)
            lib().orc_rans64_decode_ar(*self._targs(), *a, _p(words), _p(idx), ctypes.c_int64(idx.size), _p(out))
            return out
        st, pos = ctypes.c_uint64(0), ctypes.c_int64(-1)
        lib().orc_rans64_decode(*self._targs(), _p(words), _p(idx), ctypes.c_int64(idx.size), _p(out),
                                ctypes.byref(st), ctypes.byref(pos))
        return out

    def set_stream(self, encoded):
        self._words = np.frombuffer(encoded, dtype=np.uint32).copy()
        self._st, self._pos = ctypes.c_uint64(0), ctypes.c_int64(-1)

    def decode_stream(self, indexes, ar_indexes=None, ar_offsets=None):
        if self.cdfs is None:
            raise ValueError("ANS not initialized!")
        idx = _i32(indexes)
        out = np.empty(idx.shape, dtype=np.int32)
        lib().orc_rans64_decode(*self._targs(), _p(self._words), _p(idx), ctypes.c_int64(idx.size), _p(out),
                                ctypes.byref(self._st), ctypes.byref(self._pos))
        return out


def load_ref():
    """Return (ans, rans) modules of oracle/_ref -- the reference's own extensions compiled
    by oracle/Makefile -- or (None, None) when that build is absent."""
    import importlib.util
    import sysconfig
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    mods = []
    for name in ("ans", "rans"):
        path = os.path.join(_HERE, "_ref", name + ext)
        if not os.path.exists(path):
            return None, None
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods.append(m)
    return tuple(mods)


class ar_limited_scaled_add_linear_op:
    """csrc/ans/ar_funcs.hpp:58-87 (the only op type init_custom_ar_ops accepts, ans_interface.hpp:40)."""

    def __init__(self, weight, bias, scale, min, max):
        self.weight, self.bias, self.scale, self.min, self.max = [float(w) for w in weight], float(bias), float(scale), float(min), float(max)
