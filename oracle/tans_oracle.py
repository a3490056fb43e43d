"""ctypes front-end of oracle/tans_oracle.c -- CPU ORACLE, test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.  The classes mirror the
reference's ``cbench.ans.TansEncoder`` / ``TansDecoder`` (csrc/ans/tans.hpp:78-157) closely enough that parity tests can
drive the oracle, oracle/_ref and the HIP path with the same calls.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_ERRORS = {-1: "Error (generic)", -2: "Destination buffer is too small", -3: "Src size is incorrect", -4: "bad argument"}


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle_tans.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle_tans.so"])
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _i32(a):
    return np.ascontiguousarray(np.asarray(a).astype(np.int32))


def normalize(freqs, table_log):
    f = _i32(freqs).reshape(-1)
    norm, rle = np.zeros(f.size, np.int16), ctypes.c_int(0)
    rc = lib().tans_oracle_normalize(_p(f), f.size, int(table_log), _p(norm), ctypes.byref(rle))
    if rc:
        raise ValueError(_ERRORS.get(rc, str(rc)))
    return norm, bool(rle.value)


def tables(freqs, table_log):
    f = _i32(freqs).reshape(-1)
    size = 1 << table_log
    out = dict(next_state=np.zeros(size, np.uint16), delta_bits=np.zeros(f.size, np.uint32), delta_state=np.zeros(f.size, np.int32),
               d_base=np.zeros(size, np.uint32), d_symbol=np.zeros(size, np.uint16), d_bits=np.zeros(size, np.uint16))
    rc = lib().tans_oracle_tables(_p(f), f.size, int(table_log), *[_p(out[k]) for k in
                                  ("next_state", "delta_bits", "delta_state", "d_base", "d_symbol", "d_bits")])
    if rc:
        raise ValueError(_ERRORS.get(rc, str(rc)))
    return out


class _Base:
    def __init__(self, table_log=11, max_symbol_value=255, bypass_coding=False, bypass_precision=4):
        self.L, self.bypass, self.bprec = int(table_log), int(bool(bypass_coding)), int(bypass_precision)
        self.freqs = None
        self.ar = None

    def init_params(self, freqs, num_symbols, offsets):
        self.freqs, self.nsym, self.offsets = _i32(freqs), _i32(num_symbols).reshape(-1), _i32(offsets).reshape(-1)

    def init_ar_params(self, ar_table, ar_offsets):
        tab = _i32(ar_table)
        self.ar = (tab, tab.ndim - 2, tab.shape[2])

    def _common(self):
        if self.freqs is None:
            raise ValueError("ANS not initialized!")
        return (_p(self.freqs), self.freqs.shape[0], self.freqs.shape[1], _p(self.nsym), _p(self.offsets), self.L, self.bypass,
                self.bprec)

    def _ar(self, ar_indexes, ar_offsets, n):
        if self.ar is None:
            return (None, 0, 0, None, None, None), ()
        tab, order, s1 = self.ar
        off = _i32(ar_offsets).reshape(order, n)
        ai = _i32(ar_indexes).reshape(-1) if ar_indexes is not None else None
        o1 = np.ascontiguousarray(off[1]) if order == 2 else None
        o0 = np.ascontiguousarray(off[0])
        return (_p(tab), order, s1, _p(ai), _p(o0), _p(o1)), (tab, ai, o0, o1)


class TansEncoder(_Base):
    def encode_with_indexes(self, symbols, indexes, ar_indexes=None, ar_offsets=None, cache=0, capacity_syms=-1):
        s, ix = _i32(symbols).reshape(-1), _i32(indexes).reshape(-1)
        n = ix.size
        ar, keep = self._ar(ar_indexes, ar_offsets, n)
        cap = (n * 11 + 4) * self.L // 8 + 16
        out = np.zeros(cap, np.uint8)
        out_len, nsyms = ctypes.c_int64(0), ctypes.c_int64(0)
        rc = lib().tans_oracle_encode(*self._common(), *ar, _p(s), _p(ix), ctypes.c_int64(n), ctypes.c_int64(capacity_syms),
                                      _p(out), ctypes.c_int64(cap), ctypes.byref(out_len), ctypes.byref(nsyms))
        if rc:
            raise ValueError(_ERRORS.get(rc, str(rc)))
        self.coded_symbols = nsyms.value
        return out[: out_len.value].tobytes()


class TansDecoder(_Base):
    def decode_with_indexes(self, encoded, indexes, ar_indexes=None, ar_offsets=None):
        ix = _i32(indexes)
        n = ix.size
        ar, keep = self._ar(ar_indexes, ar_offsets, n)
        buf = np.frombuffer(bytes(encoded), np.uint8)
        out = np.zeros(ix.shape, np.int32)
        rc = lib().tans_oracle_decode(*self._common(), *ar, _p(buf), ctypes.c_int64(buf.size), _p(ix), ctypes.c_int64(n), _p(out))
        if rc:
            raise ValueError(_ERRORS.get(rc, str(rc)))
        return out
