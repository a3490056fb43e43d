"""Drop-in for the reference's ``cbench.rans`` extension (csrc/rans/rans_interface.cpp:550-576), the in-tree fork of
``compressai.ans``: BufferedRansEncoder / RansEncoder / RansDecoder / pmf_to_quantized_cdf(_np), backed by the HIP rANS
of libbasic_hip.so through ``basic_rans_tables_from_cdfs`` (include/basic_hip.h).

Same call signatures as the pybind11 module: tables travel with every call (lists, or arrays for the ``_np`` variants),
precision is fixed at 16 bits, bypass coding is always on with 4-bit nibbles (rans_interface.cpp:50-53).  The bitstream
is the one ``cbench.ans.Rans64Encoder(16, True, 4)`` writes for the same CDF rows (pinned by tests/golden/rans_kat.npz
``fork.*``, bytes of the reference's compiled module).  No CPU path: coding runs on the GPU.
"""
import hashlib

import numpy as np

from . import _lib
from .ans import Rans64Decoder, Rans64Encoder, _i32

precision = 16
bypass_precision = 4


def _tables_key(cdfs, sizes, offsets):
    h = hashlib.blake2b(digest_size=16)
    for a in (cdfs, sizes, offsets):
        h.update(np.asarray(a.shape, np.int64).tobytes())
        h.update(a.tobytes())
    return h.digest()


def _as_table_arrays(cdfs, cdfs_sizes, offsets):
    """lists (ragged rows allowed) or arrays -> (int32 [rows, width], int32 [rows], int32 [rows])."""
    sizes = _i32(cdfs_sizes).reshape(-1)
    offsets = _i32(offsets).reshape(-1)
    if isinstance(cdfs, np.ndarray):
        arr = _i32(cdfs)
        if arr.ndim != 2 or arr.shape[0] != sizes.size:
            raise ValueError("cdfs should be 2-dimensional with shape (cdfs_sizes.size, cdfs_sizes)")
    else:
        rows = [np.asarray(r, dtype=np.int64).reshape(-1) for r in cdfs]
        if len(rows) != sizes.size:
            raise ValueError("cdfs should be 2-dimensional with shape (cdfs_sizes.size, cdfs_sizes)")
        width = max([r.size for r in rows] + [1])
        arr = np.zeros((len(rows), width), dtype=np.int32)
        for i, r in enumerate(rows):
            arr[i, : r.size] = r
    if sizes.size and int(sizes.max()) > arr.shape[1]:
        raise ValueError("cdfs rows are shorter than cdfs_sizes")
    return arr, sizes, offsets


class _TableCache:
    """Coder objects keyed by table content: the reference rebuilds nothing per call, but here a table set has a device
    image, so identical tables passed again (every compress() of a codec) reuse it."""

    def __init__(self, cls, limit=8):
        self._cls, self._limit, self._items = cls, limit, {}

    def get(self, cdfs, sizes, offsets):
        key = _tables_key(cdfs, sizes, offsets)
        coder = self._items.pop(key, None)
        if coder is None:
            coder = self._cls(precision, True, bypass_precision)
            coder.init_cdf_params(cdfs, sizes, offsets)
            while len(self._items) >= self._limit:
                self._items.pop(next(iter(self._items)))
        self._items[key] = coder
        return coder


class BufferedRansEncoder:
    """rans_interface.cpp:109-225: encode_with_indexes() appends to a symbol buffer, flush() codes the buffer back to
    front (so it decodes front to back) and empties it."""

    def __init__(self):
        self._segments = []          # (symbols, indexes, cdfs, sizes, offsets)
        self._coders = _TableCache(Rans64Encoder)

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        symbols, indexes = _i32(symbols).reshape(-1), _i32(indexes).reshape(-1)
        if symbols.size != indexes.size:
            raise ValueError("symbols and indexes differ in size")
        self._segments.append((symbols.copy(), indexes.copy()) + _as_table_arrays(cdfs, cdfs_sizes, offsets))

    encode_with_indexes_np = encode_with_indexes

    def flush(self):
        segs, self._segments = self._segments, []
        if not segs:
            cd = np.array([[0, 1 << precision, 0]], np.int32)
            return self._coders.get(cd, np.array([2], np.int32), np.array([0], np.int32)).encode_with_indexes(
                np.zeros(0, np.int32), np.zeros(0, np.int32))
        keys = [_tables_key(*s[2:]) for s in segs]
        if all(k == keys[0] for k in keys):
            cdfs, sizes, offsets = segs[0][2:]
            idx = [s[1] for s in segs]
        else:  # calls with different tables: one table set of all rows, indexes shifted per call
            width = max(s[2].shape[1] for s in segs)
            cdfs = np.concatenate([np.pad(s[2], ((0, 0), (0, width - s[2].shape[1]))) for s in segs])
            sizes = np.concatenate([s[3] for s in segs])
            offsets = np.concatenate([s[4][: s[3].size] for s in segs])
            base = np.cumsum([0] + [s[3].size for s in segs[:-1]])
            idx = [s[1] + int(b) for s, b in zip(segs, base)]
        coder = self._coders.get(cdfs, sizes, offsets)
        return coder.encode_with_indexes(np.concatenate([s[0] for s in segs]), np.concatenate(idx))


class RansEncoder:
    """rans_interface.cpp:227-252: one-shot BufferedRansEncoder."""

    def __init__(self):
        self._buffered = BufferedRansEncoder()

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        self._buffered.encode_with_indexes(symbols, indexes, cdfs, cdfs_sizes, offsets)
        return self._buffered.flush()

    encode_with_indexes_np = encode_with_indexes


class RansDecoder:
    """rans_interface.cpp:254-447."""

    def __init__(self):
        self._coders = _TableCache(Rans64Decoder)
        self._stream = None
        self._stream_coder = None

    def decode_with_indexes(self, encoded, indexes, cdfs, cdfs_sizes, offsets):
        coder = self._coders.get(*_as_table_arrays(cdfs, cdfs_sizes, offsets))
        return coder.decode_with_indexes(encoded, _i32(indexes).reshape(-1)).tolist()

    def decode_with_indexes_np(self, encoded, indexes, cdfs, cdfs_sizes, offsets):
        coder = self._coders.get(*_as_table_arrays(cdfs, cdfs_sizes, offsets))
        return coder.decode_with_indexes(encoded, _i32(indexes).reshape(-1))   # flat, like the reference (:352-353)

    def set_stream(self, encoded):
        self._stream, self._stream_coder = bytes(encoded), None

    def _stream_decode(self, indexes, cdfs, cdfs_sizes, offsets):
        if self._stream is None:
            raise ValueError("set_stream must be called before decode_stream")
        coder = self._coders.get(*_as_table_arrays(cdfs, cdfs_sizes, offsets))
        if self._stream_coder is not coder:
            if self._stream_coder is not None:
                raise NotImplementedError("decode_stream with tables that change inside one stream")
            coder.set_stream(self._stream)
            self._stream_coder = coder
        return coder.decode_stream(_i32(indexes).reshape(-1))

    def decode_stream(self, indexes, cdfs, cdfs_sizes, offsets):
        return self._stream_decode(indexes, cdfs, cdfs_sizes, offsets).tolist()

    def decode_stream_np(self, indexes, cdfs, cdfs_sizes, offsets):
        return self._stream_decode(indexes, cdfs, cdfs_sizes, offsets)


def pmf_to_quantized_cdf(pmf, precision=precision):
    """rans_interface.cpp:450-519 -- list of len(pmf)+1 ints; std::domain_error -> ValueError."""
    p = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32)).reshape(-1)
    if p.size and (not np.all(np.isfinite(p)) or np.any(p < 0)):
        bad = p[(~np.isfinite(p)) | (p < 0)][0]
        raise ValueError(f"Invalid `pmf`, non-finite or negative element found: {bad}")
    if p.size == 0 or float(np.round(p.astype(np.float64) * (1 << int(precision))).sum()) == 0:
        raise ValueError("Invalid `pmf`: at least one element must have a non-zero probability.")
    out = np.zeros(p.size + 1, dtype=np.int32)
    _lib.check(_lib.lib().basic_pmf_to_quantized_cdf(p.ctypes.data, p.size, int(precision), out.ctypes.data))
    return out.tolist()


def pmf_to_quantized_cdf_np(pmf, precision=precision):
    """rans_interface.cpp:521-546: 1-D -> uint32 [n+1]; batched over the leading dimensions -> uint32 [batch, n+1]."""
    p = np.asarray(pmf, dtype=np.float32)
    if p.ndim == 1:
        return np.asarray(pmf_to_quantized_cdf(p, precision), dtype=np.uint32)
    rows = p.reshape(-1, p.shape[-1])
    return np.stack([np.asarray(pmf_to_quantized_cdf(r, precision), dtype=np.uint32) for r in rows])
