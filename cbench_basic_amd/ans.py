"""Drop-in for the reference's ``cbench.ans`` extension (csrc/ans/lib.cpp:10-33,
csrc/ans/rans64.hpp:127-149) backed by the HIP kernels in libbasic_hip.so.

Same classes, constructor defaults, method names, argument meaning and error behaviour:
arrays are converted to C-contiguous int32 like pybind11's ``py::array_t<int32_t>``
(forcecast), ``ValueError`` is raised where the reference raises ``py::value_error``.
The coding itself runs on the GPU (one wavefront per stream); there is no CPU path.
"""
import ctypes

import numpy as np

from . import _lib


def _i32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


def pmf_to_quantized_cdf(pmf, precision=16):
    """csrc/ans/rans64.cpp:69-126 -- returns a list of ints of length len(pmf)+1."""
    pmf = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32)).reshape(-1)
    out = np.zeros(pmf.size + 1, dtype=np.int32)
    _lib.check(_lib.lib().basic_pmf_to_quantized_cdf(pmf.ctypes.data, pmf.size, int(precision), out.ctypes.data))
    return out.tolist()


class ar_op_default:
    """csrc/ans/lib.cpp:17-18 (``ar_op<int32>``): an op that does nothing; kept for import compatibility."""


class ar_linear_op(ar_op_default):
    """csrc/ans/ar_funcs.hpp:29-53, lib.cpp:20-22: callable on a list of ints; NOT accepted by init_custom_ar_ops in the
    reference either (its binding takes ar_limited_scaled_add_linear_op only, ans_interface.hpp:40)."""

    def __init__(self, weight, bias, scale):
        self.weight, self.bias, self.scale = [float(w) for w in weight], float(bias), float(scale)

    def __call__(self, values):
        ret = np.float32(0)
        for v, w in zip(values, self.weight):    # (sic) the bias is added once per element, ar_funcs.hpp:47
            ret = np.float32(ret + np.float32(np.float32(np.float32(v) * np.float32(w)) + np.float32(self.bias)))
        return int(np.float32(ret * np.float32(self.scale)))


class ar_limited_scaled_add_linear_op(ar_op_default):
    """csrc/ans/ar_funcs.hpp:58-87, lib.cpp:23-25: index' = index + (round(clamp(floor(index / scale) + w . v + bias, min, max))
    - floor(index / scale)) * scale, in float32.  ``__call__`` evaluates it on the host (tests, table design); the coders
    evaluate it on the GPU."""

    def __init__(self, weight, bias, scale, min, max):
        self.weight, self.bias, self.scale, self.min, self.max = [float(w) for w in weight], float(bias), float(scale), float(min), float(max)

    def __call__(self, values):
        f = np.float32
        base = f(values[0])
        unscaled = f(np.floor(base / f(self.scale)))
        adder = f(0)
        for v, w in zip(values[1:], self.weight):
            adder = f(adder + f(f(v) * f(w)))
        adder = f(adder + f(self.bias))
        lim = f(unscaled + adder)
        lim = lim if lim < f(self.max) else f(self.max)
        lim = f(self.min) if f(self.min) > lim else lim
        r = f(np.copysign(np.floor(np.abs(lim) + f(0.5)), lim))   # std::round: halves away from zero
        return int(f(base + f(f(r - unscaled) * f(self.scale))))


class _Rans64Base:
    def __init__(self, freq_precision=16, bypass_coding=True, bypass_precision=4):
        self._freq_precision = int(freq_precision)
        self._bypass_coding = bool(bypass_coding)
        self._bypass_precision = int(bypass_precision)
        self._tables = None
        self._ar_order = 0

    def __del__(self):
        self._free()

    def _free(self):
        t, self._tables = getattr(self, "_tables", None), None
        if t:
            try:
                _lib.lib().basic_rans_tables_destroy(t)
            except Exception:
                pass

    def __reduce__(self):  # the reference's pybind11 objects are not picklable either
        raise TypeError("cannot pickle '%s' object" % type(self).__name__)

    def init_params(self, freqs, num_symbols, offsets):
        freqs, nsym, offsets = _i32(freqs), _i32(num_symbols).reshape(-1), _i32(offsets).reshape(-1)
        if freqs.ndim != 2 or freqs.shape[0] != nsym.size:
            raise ValueError("freqs should be 2-dimensional with shape (num_symbols.size(), >num_symbols.max())")
        if offsets.size < nsym.size:
            raise ValueError("offsets should have one entry per distribution")
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().basic_rans_tables_from_freqs(
            freqs.ctypes.data, freqs.shape[0], freqs.shape[1], nsym.ctypes.data, offsets.ctypes.data,
            self._freq_precision, int(self._bypass_coding), self._bypass_precision, ctypes.byref(h)))
        self._free()
        self._tables = h
        self._host_sizes, self._host_offsets = (nsym + 2).copy(), offsets[: nsym.size].copy()   # _cdfs_sizes = nsym + 2 (rans64.cpp:128-159)

    def init_cdf_params(self, cdfs, cdfs_sizes, offsets):
        cdfs, sizes, offsets = _i32(cdfs), _i32(cdfs_sizes).reshape(-1), _i32(offsets).reshape(-1)
        if cdfs.ndim != 2 or cdfs.shape[0] != sizes.size:
            raise ValueError("cdfs should be 2-dimensional with shape (cdfs_sizes.size(), >cdfs_sizes.max())")
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().basic_rans_tables_from_cdfs(
            cdfs.ctypes.data, cdfs.shape[0], cdfs.shape[1], sizes.ctypes.data, offsets.ctypes.data,
            self._freq_precision, int(self._bypass_coding), self._bypass_precision, ctypes.byref(h)))
        self._free()
        self._tables = h
        self._host_sizes, self._host_offsets = sizes.copy(), offsets[: sizes.size].copy()

    def create_ar_ptrs(self, indexes, ar_offsets):
        """ANSBase::create_ar_ptrs (csrc/ans/ans_interface.cpp:34-73; bound on both coders, rans64.hpp:136,147): for every
        autoregressive offset vector (one non-positive entry per data dimension after the batch dimension) the flat position
        of each element's AR neighbour, or -1 where the neighbour would fall outside a dimension.  Host-side index
        arithmetic only."""
        ix = np.asarray(indexes)
        n, nd = ix.size, ix.ndim
        strides = [st // ix.itemsize for st in ix.strides]
        out = []
        for vec in ar_offsets:
            vec = [int(v) for v in vec]
            if any(v > 0 for v in vec):
                raise ValueError("ar_offset should be non-positive!")
            reach = (ix.shape[0] - 1) * strides[0]
            limits = []          # (steps back, stride of the dimension, stride of the dimension before it)
            for j in range(nd - 1):
                cur = vec[j] if j < len(vec) else 0
                if cur < 0:
                    limits.append((-cur, strides[j], strides[j + 1]))
                reach += (ix.shape[j + 1] - 1 + cur) * strides[j + 1]
            back = n - 1 - reach
            k = np.arange(n, dtype=np.int64)
            ok = np.ones(n, dtype=bool)
            for steps, outer, inner in limits:
                # the reference's test AS WRITTEN (:59-62) compares k % (inner stride) with the OUTER stride, which is never
                # smaller: with any negative offset every entry comes out -1 (pinned by tests/golden/rans_cache_kat.npz)
                ok &= (k % inner) >= outer
            out.append(np.where(ok, k - back, -1).tolist())
        return out

    def init_ar_params(self, ar_table, ar_offsets):
        """ANSBase::init_ar_params, csrc/ans/ans_interface.cpp:75-137."""
        tab = _i32(ar_table)
        ar_offsets = np.asarray(ar_offsets)
        order = tab.ndim - 2
        if ar_offsets.ndim != 3 or ar_offsets.shape[1] != order or ar_offsets.shape[0] != tab.shape[0]:
            raise ValueError("ar_offset should be 3-dimensional with shape (ar_tables_size, ar_order, <=data_dims)")
        if order <= 0:
            raise ValueError("ar_tables should be at least 3-dimensional with shape (ar_tables_size, index_dim, *ar_order_dims)")
        if order > 2:
            raise ValueError("Too many dimensions!")
        if self._tables is None:
            raise ValueError("ANS not initialized!")
        _lib.check(_lib.lib().basic_rans_tables_set_ar(self._tables, tab.ctypes.data, tab.shape[0], tab.shape[1], order, tab.shape[2]))
        self._ar_order = order
        self._ar_host = ("table", tab.copy())

    def init_custom_ar_ops(self, ops):
        """ANSBase::init_custom_ar_ops (csrc/ans/ans_interface.hpp:40-48): a list of ``ar_limited_scaled_add_linear_op`` (the
        only type the reference's binding accepts); the table row of an element becomes op(index, previous raw symbols)."""
        ops = list(ops)
        if not ops:
            return
        if self._tables is None:
            raise ValueError("ANS not initialized!")
        for o in ops:
            if not isinstance(o, ar_limited_scaled_add_linear_op):
                raise TypeError("init_custom_ar_ops(): incompatible function arguments (a list of ar_limited_scaled_add_linear_op)")
            if len(o.weight) > 3:
                raise ValueError("Too many dimensions!")
        arr = np.ascontiguousarray([list(o.weight) + [0.0] * (3 - len(o.weight)) + [o.bias, o.scale, o.min, o.max] for o in ops],
                                   dtype=np.float32)
        _lib.check(_lib.lib().basic_rans_tables_set_ar_ops(self._tables, arr.ctypes.data, arr.shape[0]))
        self._ar_order = -1   # custom ops: the arity is the number of ar_offsets rows of a call
        self._ar_host = ("ops", arr.copy())

    def get_cdfs(self):
        if self._tables is None:
            return np.zeros((0,), dtype=np.int32)
        rows, mx = ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.lib().basic_rans_tables_info(self._tables, ctypes.byref(rows), ctypes.byref(mx)))
        out = np.zeros((rows.value, mx.value), dtype=np.int32)
        _lib.check(_lib.lib().basic_rans_tables_get_cdfs(self._tables, out.ctypes.data, mx.value))
        return out

    def _ar_rows(self, symbols, indexes, ar_indexes, ar_offsets):
        """ar_update_index (csrc/ans/ans_interface.hpp:58-104) for every element of a call, on the host: the table row an
        element is coded with once its autoregressive neighbours -- earlier elements of the SAME call, ``off[j][i]`` positions
        back (0 = none) -- are taken into account.  Table mode: row = ar_table[ar_index][index][v0 + 1]([v1 + 1]) with 0 for a
        missing neighbour; custom ops: the limited scaled-add op on the raw neighbour symbols in float32, every product and
        sum rounded on its own.  Indices are clamped like the device kernels clamp them (the reference reads out of range).
        Used by the cached encode path, whose symbols are resolved when they are cached (rans64.cpp:258-263,343)."""
        n = indexes.size
        if ar_offsets is None:
            raise ValueError("ar_offsets is required for ar coding!")
        off = _i32(ar_offsets).reshape(-1, n) if self._ar_order < 0 else _i32(ar_offsets).reshape(self._ar_order, n)
        if off.shape[0] > 3:
            raise ValueError("Too many dimensions!")
        a = _i32(ar_indexes).reshape(-1) if ar_indexes is not None else np.zeros(n, dtype=np.int32)
        pos = np.arange(n, dtype=np.int64)
        kind, data = self._ar_host
        if kind == "table":
            k, rows, s1 = data.shape[:3]

            def nb(j):   # GET_AR_VALUE_DEFAULT: neighbour symbol + 1, 0 when there is none
                o = off[j].astype(np.int64)
                return np.clip(np.where(o > 0, symbols[np.clip(pos - o, 0, n - 1)].astype(np.int64) + 1, 0), 0, s1 - 1)
            aa, rr = np.clip(a, 0, k - 1), np.clip(indexes, 0, rows - 1)
            out = data[aa, rr, nb(0)] if data.ndim == 3 else data[aa, rr, nb(0), nb(1)]
            return np.ascontiguousarray(out, dtype=np.int32)
        f = np.float32
        op = data[np.clip(a, 0, data.shape[0] - 1)]          # [n, 7]: w0 w1 w2 bias scale min max

        def raw(j):    # GET_AR_VALUE_NORMAL: the raw neighbour symbol, 0 when there is none
            o = off[j].astype(np.int64)
            return np.where(o > 0, symbols[np.clip(pos - o, 0, n - 1)], 0).astype(f)
        base = indexes.astype(f)
        unscaled = np.floor((base / op[:, 4]).astype(f)).astype(f)
        adder = (f(0) + (raw(0) * op[:, 0]).astype(f)).astype(f)
        for j in range(1, off.shape[0]):
            adder = (adder + (raw(j) * op[:, j]).astype(f)).astype(f)
        adder = (adder + op[:, 3]).astype(f)
        lim = np.maximum(op[:, 5], np.minimum((unscaled + adder).astype(f), op[:, 6])).astype(f)
        rnd = (np.trunc(lim) + (np.abs(lim - np.trunc(lim)) >= f(0.5)) * np.sign(lim)).astype(f)   # roundf: halves away from zero
        step = ((rnd - unscaled).astype(f) * op[:, 4]).astype(f)
        return np.ascontiguousarray((base + step).astype(f).astype(np.int32))

    def _ar_args(self, ar_indexes, ar_offsets, n):
        """(ar_indexes, off0, off1, off2 addresses, keep-alive) of a call."""
        if not self._ar_order:
            return None, None, None, None, ()
        if ar_offsets is None:
            raise ValueError("ar_offsets is required for ar coding!")
        off = _i32(ar_offsets).reshape(-1, n) if self._ar_order < 0 else _i32(ar_offsets).reshape(self._ar_order, n)
        if off.shape[0] > 3:
            raise ValueError("Too many dimensions!")
        ai = _i32(ar_indexes).reshape(-1) if ar_indexes is not None else None
        rows = [off[i].ctypes.data if i < off.shape[0] else None for i in range(3)]
        return (ai.ctypes.data if ai is not None else None), rows[0], rows[1], rows[2], (off, ai)


class Rans64Encoder(_Rans64Base):
    def __init__(self, freq_precision=16, bypass_coding=True, bypass_precision=4):
        super().__init__(freq_precision, bypass_coding, bypass_precision)
        self._cache = []

    def encode_with_indexes(self, symbols, indexes, ar_indexes=None, ar_offsets=None, cache=0):
        """csrc/ans/rans64.cpp:203-361.  With ``cache`` truthy the call only buffers and returns b""."""
        if self._tables is None:
            raise ValueError("ANS not initialized!")
        symbols, indexes = _i32(symbols).reshape(-1), _i32(indexes).reshape(-1)
        if cache:
            # reference semantics (rans64.cpp:332,343): symbols are appended in REVERSE order and
            # flush() codes the buffer front-to-back, i.e. last call's symbols are decoded first.  An AR call's rows are
            # resolved NOW, from this call's own symbols (:258-263): the cache holds (symbol, final row) pairs
            rows = self._ar_rows(symbols, indexes, ar_indexes, ar_offsets) if self._ar_order else indexes.copy()
            self._cache.append((symbols.copy(), rows))
            return b""
        n = indexes.size
        ai, o0, o1, o2, keep = self._ar_args(ar_indexes, ar_offsets, n)
        cap = _lib.lib().basic_rans_encode_bound(n)
        out = np.empty(cap, dtype=np.uint8)
        out_len = ctypes.c_int64()
        _lib.check(_lib.lib().basic_rans_encode_host_ex(self._tables, symbols.ctypes.data, indexes.ctypes.data, n, ai, o0, o1, o2,
                                                        out.ctypes.data, cap, ctypes.byref(out_len)))
        return out[: out_len.value].tobytes()

    def flush(self):
        """csrc/ans/rans64.cpp:363-386."""
        if self._tables is None:
            raise ValueError("ANS not initialized!")
        # _syms = [rev(call1), rev(call2), ...] coded front-to-back == one encode call over
        # concat(callN, ..., call1) coded back-to-front.
        chunks, self._cache = self._cache[::-1], []
        if chunks:
            symbols = np.concatenate([c[0] for c in chunks])
            indexes = np.concatenate([c[1] for c in chunks])
        else:
            symbols = indexes = np.zeros(0, dtype=np.int32)
        if not self._ar_order:
            return self.encode_with_indexes(symbols, indexes)
        # an AR table set: the cached rows are final, flush() codes them as they are (rans64.cpp:363-386 knows no AR)
        n = indexes.size
        symbols, indexes = np.ascontiguousarray(symbols), np.ascontiguousarray(indexes)
        cap = _lib.lib().basic_rans_encode_bound(n)
        out = np.empty(cap, dtype=np.uint8)
        out_len = ctypes.c_int64()
        _lib.check(_lib.lib().basic_rans_encode_host_rows(self._tables, symbols.ctypes.data, indexes.ctypes.data, n, out.ctypes.data, cap,
                                                          ctypes.byref(out_len)))
        return out[: out_len.value].tobytes()

    def peek_cache(self):
        """rans64.hpp:78-86: the cached rANS symbols as int32 [m, 3] rows (start, range, bypass flag) in the order flush()
        codes them: per cached call its elements last to first, an escaped element's bypass digits -- count nibble(s), then
        the value's nibbles -- in REVERSE before the element's own (sentinel) symbol (rans64.cpp:292-344).  Host-side
        bookkeeping on the table set's integer CDFs (a debugging aid: nothing is coded here)."""
        if self._tables is None:
            raise ValueError("ANS not initialized!")
        cdfs, sizes, offs = self.get_cdfs(), self._host_sizes, self._host_offsets
        bp, mb = self._bypass_precision, (1 << self._bypass_precision) - 1
        rows = []
        for symbols, indexes in self._cache:
            for i in range(symbols.size - 1, -1, -1):
                r = int(indexes[i])
                maxv, v, raw = int(sizes[r]) - 2, int(symbols[i]) - int(offs[r]), 0
                if self._bypass_coding:
                    if v < 0:
                        raw, v = -2 * v - 1, maxv
                    elif v >= maxv:
                        raw, v = 2 * (v - maxv), maxv
                if not 0 <= v < int(sizes[r]) - 1:
                    raise ValueError("symbol outside its table and bypass coding is off")
                if self._bypass_coding and v == maxv:
                    digits, nb = [], 0
                    while (raw >> (nb * bp)) != 0:
                        nb += 1
                    val = nb
                    while val >= mb:
                        digits.append((mb, 0, 1))
                        val -= mb
                    digits.append((val, val + 1, 1))
                    digits.extend((((raw >> (j * bp)) & mb), ((raw >> (j * bp)) & mb) + 1, 1) for j in range(nb))
                    rows.extend(reversed(digits))
                rows.append((int(cdfs[r, v]) & 0xFFFF, (int(cdfs[r, v + 1]) - int(cdfs[r, v])) & 0xFFFF, 0))
        return np.array(rows, dtype=np.int32).reshape(-1, 3)


class Rans64Decoder(_Rans64Base):
    def __init__(self, freq_precision=16, bypass_coding=True, bypass_precision=4):
        super().__init__(freq_precision, bypass_coding, bypass_precision)
        self._stream = None

    def _close_stream(self):
        s, self._stream = getattr(self, "_stream", None), None
        if s:
            try:
                _lib.lib().basic_rans_stream_close(s)
            except Exception:
                pass

    def __del__(self):
        self._close_stream()
        super().__del__()

    def _free(self):
        # an open stream points INTO the table set (basic_rans_stream.t): replacing or releasing the tables ends it
        # (the reference would go on with the new tables; here set_stream() has to be called again)
        self._close_stream()
        super()._free()

    def decode_with_indexes(self, encoded, indexes, ar_indexes=None, ar_offsets=None):
        """csrc/ans/rans64.cpp:389-499 -- int32 array shaped like ``indexes``."""
        if self._tables is None:
            raise ValueError("ANS not initialized!")
        indexes = _i32(indexes)
        n = indexes.size
        out = np.empty(indexes.shape, dtype=np.int32)
        buf = np.frombuffer(bytes(encoded), dtype=np.uint8)
        ai, o0, o1, o2, keep = self._ar_args(ar_indexes, ar_offsets, n)
        _lib.check(_lib.lib().basic_rans_decode_host_ex(self._tables, buf.ctypes.data, buf.size, indexes.ctypes.data, n, ai, o0, o1, o2,
                                                        out.ctypes.data))
        return out

    def set_stream(self, stream):
        """csrc/ans/rans64.hpp:104-111."""
        if self._tables is None:
            raise ValueError("ANS not initialized!")
        buf = np.frombuffer(bytes(stream), dtype=np.uint8)
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().basic_rans_stream_open(self._tables, buf.ctypes.data, buf.size, ctypes.byref(h)))
        self._close_stream()
        self._stream = h

    def decode_stream(self, indexes, ar_indexes=None, ar_offsets=None):
        """csrc/ans/rans64.cpp:501-598 (AR arguments are ignored there too, :529,:537)."""
        if self._tables is None:
            raise ValueError("ANS not initialized!")
        if self._stream is None:
            raise ValueError("set_stream must be called before decode_stream")
        indexes = _i32(indexes)
        out = np.empty(indexes.shape, dtype=np.int32)
        _lib.check(_lib.lib().basic_rans_stream_decode(self._stream, indexes.ctypes.data, indexes.size, out.ctypes.data))
        return out


class _TansBase:
    """TansBase, csrc/ans/tans.hpp:37-71 (constructor defaults: FSE_DEFAULT_TABLELOG = 11, FSE_MAX_SYMBOL_VALUE = 255,
    fse.h:590,610)."""

    def __init__(self, table_log=11, max_symbol_value=255, bypass_coding=False, bypass_precision=4):
        self._table_log = int(table_log)
        self._max_symbol_value = int(max_symbol_value)
        self._bypass_coding = bool(bypass_coding)
        self._bypass_precision = int(bypass_precision)
        self._tables = None
        self._ar_order = 0

    def __del__(self):
        self._free()

    def _free(self):
        t, self._tables = getattr(self, "_tables", None), None
        if t:
            try:
                _lib.lib().basic_tans_tables_destroy(t)
            except Exception:
                pass

    def __reduce__(self):
        raise TypeError("cannot pickle '%s' object" % type(self).__name__)

    def init_params(self, freqs, num_symbols, offsets):
        """tans.cpp:368-383: freqs [rows][>= max(num_symbols)]; the counts are normalised to 2^table_log per row."""
        freqs, nsym, offsets = _i32(freqs), _i32(num_symbols).reshape(-1), _i32(offsets).reshape(-1)
        if freqs.ndim != 2 or freqs.shape[0] != nsym.size:
            raise ValueError("freqs should be 2-dimensional with shape (num_symbols.size(), >num_symbols.max())")
        if offsets.size < nsym.size:
            raise ValueError("offsets should have one entry per distribution")
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().basic_tans_tables_create(
            freqs.ctypes.data, freqs.shape[0], freqs.shape[1], nsym.ctypes.data, offsets.ctypes.data, self._table_log,
            self._max_symbol_value, int(self._bypass_coding), self._bypass_precision, ctypes.byref(h)))
        self._free()
        self._tables = h
        self._ar_order = 0

    def init_ar_params(self, ar_table, ar_offsets):
        """ANSBase::init_ar_params, csrc/ans/ans_interface.cpp:75-137."""
        tab = _i32(ar_table)
        ar_offsets = np.asarray(ar_offsets)
        order = tab.ndim - 2
        if ar_offsets.ndim != 3 or ar_offsets.shape[1] != order or ar_offsets.shape[0] != tab.shape[0]:
            raise ValueError("ar_offset should be 3-dimensional with shape (ar_tables_size, ar_order, <=data_dims)")
        if order <= 0:
            raise ValueError("ar_tables should be at least 3-dimensional with shape (ar_tables_size, index_dim, *ar_order_dims)")
        if order > 2:
            raise ValueError("Too many dimensions!")
        if self._tables is None:
            raise ValueError("ANS not initialized!")
        _lib.check(_lib.lib().basic_tans_tables_set_ar(self._tables, tab.ctypes.data, tab.shape[0], tab.shape[1], order, tab.shape[2]))
        self._ar_order = order
        self._ar_host = ("table", tab.copy())

    def get_table_row(self, row):
        """(next_state, delta_bits, delta_state, decode entries) of one distribution -- not part of the reference's bound
        surface; the table-level parity tests read it."""
        if self._tables is None:
            raise ValueError("ANS not initialized!")
        size = 1 << self._table_log
        nxt, dec = np.zeros(size, np.uint16), np.zeros(size, np.uint32)
        db, ds = np.zeros(65536, np.uint32), np.zeros(65536, np.int32)
        _lib.check(_lib.lib().basic_tans_tables_get_row(self._tables, int(row), nxt.ctypes.data, db.ctypes.data, ds.ctypes.data,
                                                        dec.ctypes.data))
        return nxt, db, ds, dec

    _ar_args = _Rans64Base._ar_args
    _ar_rows = _Rans64Base._ar_rows


class TansEncoder(_TansBase):
    def __init__(self, table_log=11, max_symbol_value=255, bypass_coding=False, bypass_precision=4):
        super().__init__(table_log, max_symbol_value, bypass_coding, bypass_precision)
        self._cache = []

    def _encode(self, symbols, indexes, ar_indexes, ar_offsets, capacity_syms):
        n = indexes.size
        ai, o0, o1, _, keep = self._ar_args(ar_indexes, ar_offsets, n)
        L = _lib.lib()
        cap = 4 * int(L.basic_tans_encode_bound_words(self._tables, n)) + 8
        out = np.empty(cap, dtype=np.uint8)
        out_len, coded = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(L.basic_tans_encode_host(self._tables, symbols.ctypes.data, indexes.ctypes.data, n, ai, o0, o1, int(capacity_syms),
                                            out.ctypes.data, cap, ctypes.byref(out_len), ctypes.byref(coded)))
        return out[: out_len.value].tobytes(), coded.value

    def encode_with_indexes(self, symbols, indexes, ar_indexes=None, ar_offsets=None, cache=0):
        """csrc/ans/tans.cpp:527-680.  Like the reference: ``ValueError`` when the output budget ``len(indexes) * table_log
        / 8`` is 8 bytes or less, and ``b""`` when the stream does not fit that budget.  ``cache`` truthy only buffers."""
        if self._tables is None:
            raise ValueError("ANS not initialized!")
        symbols, indexes = _i32(symbols).reshape(-1), _i32(indexes).reshape(-1)
        if cache:
            if self._ar_order:
                raise NotImplementedError("cached AR encoding")
            self._cache.append((symbols.copy(), indexes.copy()))
            return b""
        return self._encode(symbols, indexes, ar_indexes, ar_offsets, -1)[0]

    def flush(self):
        """csrc/ans/tans.cpp:682-713 (not bound by PYBIND11_TANS_CLASSES, so unreachable from Python in the reference;
        kept for the C++ semantics): the cached symbols -- each call stored in reverse -- are coded front to back, i.e. as
        ONE encode over concat(last call, ..., first call), with the output budget counted in cached symbols incl. bypass
        digits (:686)."""
        if self._tables is None:
            raise ValueError("ANS not initialized!")
        chunks, self._cache = self._cache[::-1], []
        if chunks:
            symbols = np.concatenate([c[0] for c in chunks])
            indexes = np.concatenate([c[1] for c in chunks])
        else:
            symbols = indexes = np.zeros(0, dtype=np.int32)
        # the budget depends on the coded symbol count, known after a first pass with a budget that cannot fail
        _, coded = self._encode(symbols, indexes, None, None, 1 << 40)
        return self._encode(symbols, indexes, None, None, coded)[0]


class TansDecoder(_TansBase):
    def decode_with_indexes(self, encoded, indexes, ar_indexes=None, ar_offsets=None):
        """csrc/ans/tans.cpp:715-815 -- int32 array shaped like ``indexes``."""
        if self._tables is None:
            raise ValueError("ANS not initialized!")
        indexes = _i32(indexes)
        n = indexes.size
        out = np.empty(indexes.shape, dtype=np.int32)
        buf = np.frombuffer(bytes(encoded), dtype=np.uint8)
        ai, o0, o1, _, keep = self._ar_args(ar_indexes, ar_offsets, n)
        _lib.check(_lib.lib().basic_tans_decode_host(self._tables, buf.ctypes.data, buf.size, indexes.ctypes.data, n, ai, o0, o1,
                                                     out.ctypes.data))
        return out

    def set_stream(self, stream):
        raise NotImplementedError("TansDecoder.decode_stream is an empty stub in the reference (tans.cpp:817-840)")

    def decode_stream(self, indexes, ar_indexes=None, ar_offsets=None):
        raise NotImplementedError("TansDecoder.decode_stream is an empty stub in the reference (tans.cpp:817-840)")
