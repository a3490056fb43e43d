"""Shared driver of the table-ANS parity tests: runs any (TansEncoder, TansDecoder) pair -- the CPU oracle, the HIP drop-in,
oracle/_ref -- over the reference-generated known answers in tests/golden/tans_kat.npz (tests/golden/make_golden.py::tans_kats)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    return np.load(os.path.join(G, "tans_kat.npz"), allow_pickle=False)


def check_known_answers(mod):
    z = load()
    nonempty = 0
    for name in z["names"]:
        L, byp, bprec = (int(v) for v in z[f"{name}.cfg"])
        enc, dec = mod.TansEncoder(L, 255, bool(byp), bprec), mod.TansDecoder(L, 255, bool(byp), bprec)
        enc.init_params(z[f"{name}.freqs"], z[f"{name}.nsym"], z[f"{name}.offsets"])
        dec.init_params(z[f"{name}.freqs"], z[f"{name}.nsym"], z[f"{name}.offsets"])
        kw = {}
        if f"{name}.ar_table" in z.files:
            enc.init_ar_params(z[f"{name}.ar_table"], z[f"{name}.ar_cfg"])
            dec.init_ar_params(z[f"{name}.ar_table"], z[f"{name}.ar_cfg"])
            kw = dict(ar_indexes=z[f"{name}.ar_indexes"], ar_offsets=z[f"{name}.ar_offsets"])
        sym, idx, ref = z[f"{name}.symbols"], z[f"{name}.indexes"], z[f"{name}.bytes"].tobytes()
        if int(z[f"{name}.error"]):
            with pytest.raises(ValueError):   # the reference: "Destination buffer is too small" (bitstream.h:192)
                enc.encode_with_indexes(sym, idx, **kw)
            continue
        assert enc.encode_with_indexes(sym, idx, **kw) == ref, name   # b"" where the reference's budget overflows
        if ref:
            nonempty += 1
            assert np.array_equal(dec.decode_with_indexes(ref, idx, **kw), z[f"{name}.decoded"]), name
    assert nonempty >= 7
    return nonempty


def random_case(rng, trial):
    L = int(rng.integers(9, 13))
    nd, ns = int(rng.integers(1, 8)), int(rng.integers(2, min(200, (1 << L) // 2)))
    if trial % 3 == 0:
        freqs = np.maximum((rng.random((nd, ns)) ** 6 * 5000).astype(np.int32), 1)
    else:
        freqs = rng.integers(1, 1024, (nd, ns)).astype(np.int32)
    nsym = rng.integers(2, ns + 1, nd).astype(np.int32)
    off = rng.integers(-5, 5, nd).astype(np.int32)
    n = int(rng.integers(6, 3000))
    idx = rng.integers(0, nd, n).astype(np.int32)
    byp = bool(trial % 2)
    if byp:
        sym = (off[idx] + rng.integers(-3, 1 << 30, n) % (nsym[idx] + 6)).astype(np.int32)
        sym[::13] = rng.integers(-70000, 70000, sym[::13].size)
    else:
        sym = (off[idx] + rng.integers(0, 1 << 30, n) % nsym[idx]).astype(np.int32)
    return L, freqs, nsym, off, byp, sym, idx


def run(mod, case):
    """(error text or None, bytes or None, decoded or None) of one case through one implementation."""
    L, freqs, nsym, off, byp, sym, idx = case
    try:
        enc = mod.TansEncoder(L, 255, byp, 4)
        enc.init_params(freqs, nsym, off)
        data = enc.encode_with_indexes(sym, idx)
    except ValueError as e:
        return str(e), None, None
    back = None
    if data:
        dec = mod.TansDecoder(L, 255, byp, 4)
        dec.init_params(freqs, nsym, off)
        back = dec.decode_with_indexes(data, idx)
    return None, data, back
