"""Stream framing helpers -- same wire format as cbench/utils/bytes_ops.py:19-70.

merge_bytes: every segment is preceded by its length as a native-endian unsigned integer of
``num_bytes_length`` bytes, except that when ``num_segments`` is given the LAST segment is
written bare (its length is implied by the end of the buffer).
"""
import struct
from typing import List, Optional, Sequence, Tuple

_FMT = {1: "B", 2: "H", 4: "I", 8: "L"}


def _fmt(num_bytes):
    try:
        return _FMT[num_bytes]
    except KeyError:
        raise ValueError("")


def merge_bytes(data: Sequence[bytes], num_bytes_length=4, num_segments: Optional[int] = None) -> bytes:
    fmt = _fmt(num_bytes_length)
    parts = []
    for i, seg in enumerate(data):
        if num_segments is not None:
            assert i < num_segments, "Number of segments exceed predefined {}".format(num_segments)
        if num_segments is None or i < num_segments - 1:
            parts.append(struct.pack(fmt, len(seg)))
        parts.append(seg)
    return b"".join(parts)


def split_merged_bytes(data: bytes, num_bytes_length=4, num_segments: Optional[int] = None) -> List[bytes]:
    fmt = _fmt(num_bytes_length)
    out, cur, total = [], 0, len(data)
    while cur < total:
        if num_segments is not None and len(out) >= num_segments - 1:
            out.append(data[cur:])
            cur = total
        else:
            (n,) = struct.unpack(fmt, data[cur:cur + num_bytes_length])
            cur += num_bytes_length
            out.append(data[cur:cur + n])
            cur += n
    if num_segments is not None:
        out.extend([b""] * (num_segments - len(out)))  # empty trailing segments
    return out


def encode_shape(shape: Tuple[int]) -> bytes:
    assert len(shape) < (1 << 8)
    for d in shape:
        assert d < (1 << 16)
    return struct.pack("B", len(shape)) + b"".join(struct.pack("<H", d) for d in shape)


def decode_shape(byte_string: bytes):
    n = byte_string[0]
    dims = list(struct.unpack("<%dH" % n, byte_string[1:1 + 2 * n]))
    return dims, 1 + 2 * n


def split_merged_views(data, num_bytes_length=4, num_segments: Optional[int] = None) -> List[memoryview]:
    """split_merged_bytes without copying the segments (memoryviews into ``data``): the multi-megabyte latent stream of a
    batch is handed to the decoder in place."""
    fmt = _fmt(num_bytes_length)
    view = memoryview(data)
    out, cur, total = [], 0, len(view)
    while cur < total:
        if num_segments is not None and len(out) >= num_segments - 1:
            out.append(view[cur:])
            cur = total
        else:
            (n,) = struct.unpack_from(fmt, view, cur)
            cur += num_bytes_length
            out.append(view[cur:cur + n])
            cur += n
    if num_segments is not None:
        out.extend([memoryview(b"")] * (num_segments - len(out)))
    return out


def merge_bodies(bodies, num_bytes_length=4) -> bytes:
    """merge_bytes(bodies, num_segments=len(bodies)) for bodies that are either ``bytes`` or objects with ``nbytes()`` /
    ``write_into(address, capacity)`` (PendingBody): the result is allocated once and every body is written into it in
    place -- the same bytes as merge_bytes would produce from the bodies' ``result()``."""
    import ctypes
    fmt = _fmt(num_bytes_length)
    sizes = [len(b) if isinstance(b, (bytes, bytearray, memoryview)) else b.nbytes() for b in bodies]
    total = sum(sizes) + num_bytes_length * (len(bodies) - 1)
    api = ctypes.pythonapi
    api.PyBytes_FromStringAndSize.restype, api.PyBytes_FromStringAndSize.argtypes = ctypes.py_object, [ctypes.c_char_p, ctypes.c_ssize_t]
    api.PyBytes_AsString.restype, api.PyBytes_AsString.argtypes = ctypes.c_void_p, [ctypes.py_object]
    out = api.PyBytes_FromStringAndSize(None, total)     # fresh, unshared bytes object: filling it in place is legal
    base = api.PyBytes_AsString(out)
    cur = 0
    for i, (b, n) in enumerate(zip(bodies, sizes)):
        if i < len(bodies) - 1:
            ctypes.memmove(base + cur, struct.pack(fmt, n), num_bytes_length)
            cur += num_bytes_length
        if isinstance(b, (bytes, bytearray, memoryview)):
            if n:
                ctypes.memmove(base + cur, bytes(b) if isinstance(b, memoryview) else b, n)
        else:
            written = b.write_into(base + cur, n)
            assert written == n, (written, n)
        cur += n
    assert cur == total
    return out
