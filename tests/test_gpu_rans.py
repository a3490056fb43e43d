"""GPU parity: HIP rANS (through the cbench.ans-compatible API over the C ABI) vs the CPU oracle.
Bit-exact: identical bytes out of the encoder, identical symbols out of the decoder."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(rng, nd, ns, ragged=True):
    freqs = rng.integers(1, 1024, (nd, ns)).astype(np.int32)
    nsym = (rng.integers(2, ns + 1, nd) if ragged else np.full(nd, ns)).astype(np.int32)
    off = rng.integers(-5, 5, nd).astype(np.int32)
    return freqs, nsym, off


@pytest.mark.parametrize("seed", range(24))
def test_encode_decode_matches_oracle(oracle, seed):
    from cbench_basic_amd import ans
    rng = np.random.default_rng(seed)
    nd, ns = int(rng.integers(1, 12)), int(rng.integers(2, 300))
    freqs, nsym, off = _params(rng, nd, ns)
    byp = bool(seed % 2)
    prec = 16 if seed % 3 == 0 else int(rng.integers(12, 17))
    n = [0, 1, 63, 64, 65, 4097][seed % 6] if seed < 6 else int(rng.integers(100, 20000))
    idx = rng.integers(0, nd, n).astype(np.int32)
    if byp:
        sym = rng.integers(-40, ns + 40, n).astype(np.int32)
        if seed % 4 == 1:
            sym[::7] = rng.integers(-100000, 100000, sym[::7].size)
    else:
        sym = (off[idx] + rng.integers(0, 1 << 30, n) % nsym[idx]).astype(np.int32)
    eo, eg = oracle.Rans64Encoder(prec, byp, 4), ans.Rans64Encoder(prec, byp, 4)
    eo.init_params(freqs, nsym, off)
    eg.init_params(freqs, nsym, off)
    assert np.array_equal(eo.get_cdfs(), eg.get_cdfs())
    bo, bg = eo.encode_with_indexes(sym, idx), eg.encode_with_indexes(sym, idx)
    assert bo == bg
    dg = ans.Rans64Decoder(prec, byp, 4)
    dg.init_params(freqs, nsym, off)
    assert np.array_equal(dg.decode_with_indexes(bg, idx), sym)
    if n > 3:
        dg.set_stream(bg)
        k = n // 3
        a, b = dg.decode_stream(idx[:k]), dg.decode_stream(idx[k:])
        assert np.array_equal(np.concatenate([a, b]), sym)


def test_reference_test_shape_roundtrip(oracle):
    """Shape of the reference's own tests/ans_test.py:17-43 (8 dists x 512 symbols, bypass),
    at a size the oracle finishes in seconds."""
    from cbench_basic_amd import ans
    rng = np.random.default_rng(123)
    nd, ns, byp_n = 8, 512, 32
    freqs = rng.integers(1, 1024, (nd, ns))
    nfreqs = np.zeros(nd) + ns  # float arrays, like the reference test (forcecast)
    offsets = np.zeros(nd)
    enc, dec = ans.Rans64Encoder(bypass_coding=True), ans.Rans64Decoder(bypass_coding=True)
    enc.init_params(freqs, nfreqs, offsets)
    dec.init_params(freqs, nfreqs, offsets)
    shape = (100, 3, 32, 32)
    data = rng.integers(0, ns + byp_n, shape)  # int64, like the reference test
    indexes = rng.integers(0, nd, shape)
    bs = enc.encode_with_indexes(data, indexes)
    eo = oracle.Rans64Encoder(16, True, 4)
    eo.init_params(freqs, nfreqs, offsets)
    assert bs == eo.encode_with_indexes(data, indexes)
    out = dec.decode_with_indexes(bs, indexes)
    assert out.shape == shape and out.dtype == np.int32
    assert np.array_equal(out, data)


def test_large_table_two_level_search(oracle):
    """Rows wider than 64 entries exercise the 64-ary coarse level of the decoder search
    (the Gaussian PGM table has rows up to 2219 entries)."""
    from cbench_basic_amd import ans
    rng = np.random.default_rng(5)
    nd, ns = 4, 2217
    freqs = rng.integers(1, 50, (nd, ns)).astype(np.int32)
    nsym = np.array([2217, 65, 64, 63], np.int32)
    off = np.array([-1108, -32, 0, 5], np.int32)
    n = 30000
    idx = rng.integers(0, nd, n).astype(np.int32)
    sym = (off[idx] + rng.integers(0, 1 << 30, n) % nsym[idx]).astype(np.int32)
    sym[::97] += 5000
    eo, eg = oracle.Rans64Encoder(16, True, 4), ans.Rans64Encoder(16, True, 4)
    eo.init_params(freqs, nsym, off)
    eg.init_params(freqs, nsym, off)
    b = eg.encode_with_indexes(sym, idx)
    assert b == eo.encode_with_indexes(sym, idx)
    dg = ans.Rans64Decoder(16, True, 4)
    dg.init_params(freqs, nsym, off)
    assert np.array_equal(dg.decode_with_indexes(b, idx), sym)


def test_ar_table_mode(oracle):
    """AR index remap, reference tests/ans_test.py:45-77 at reduced size."""
    from cbench_basic_amd import ans
    rng = np.random.default_rng(9)
    nd, ns = 8, 64
    freqs = rng.integers(1, 1024, (nd, ns)).astype(np.int32)
    nsym, off = np.full(nd, ns, np.int32), np.zeros(nd, np.int32)
    shape = (5, 3, 8, 8)
    n = int(np.prod(shape))
    for order in (1, 2):
        tab = rng.integers(0, nd, [1, nd] + [ns + 1] * order).astype(np.int32)
        data = rng.integers(0, ns, shape).astype(np.int32)
        idx = rng.integers(0, nd, shape).astype(np.int32)
        # per-element back distances (what cbench.utils.ar_utils.create_ar_offsets produces)
        ar_off = np.zeros((order,) + shape, np.int32)
        w = np.arange(shape[3])[None, None, None, :]
        h = np.arange(shape[2])[None, None, :, None]
        ar_off[0] = np.where(w > 0, 1, 0) if order == 1 else np.where(h > 0, shape[3], 0)
        if order == 2:
            ar_off[1] = np.where(w > 0, 1, 0)
        dims = [[[0, 0, -1]]] if order == 1 else [[[0, -1, 0], [0, 0, -1]]]
        eo, eg = oracle.Rans64Encoder(16, False, 4), ans.Rans64Encoder(16, False, 4)
        for e in (eo, eg):
            e.init_params(freqs, nsym, off)
            e.init_ar_params(tab, dims)
        bo = eo.encode_with_indexes(data, idx, ar_indexes=np.zeros_like(idx), ar_offsets=ar_off)
        bg = eg.encode_with_indexes(data, idx, ar_indexes=np.zeros_like(idx), ar_offsets=ar_off)
        assert bo == bg
        dg = ans.Rans64Decoder(16, False, 4)
        dg.init_params(freqs, nsym, off)
        dg.init_ar_params(tab, dims)
        assert np.array_equal(dg.decode_with_indexes(bg, idx, ar_indexes=np.zeros_like(idx), ar_offsets=ar_off), data)


def test_errors():
    from cbench_basic_amd import ans
    e = ans.Rans64Encoder()
    with pytest.raises(ValueError, match="ANS not initialized"):
        e.encode_with_indexes(np.zeros(4, np.int32), np.zeros(4, np.int32))
    with pytest.raises(ValueError):
        e.init_params(np.ones((3,), np.int32), np.ones(3, np.int32), np.zeros(3, np.int32))


def test_batched_streams_device(oracle):
    """The hot-path entry points: many independent streams in one launch, device pointers."""
    import torch
    from cbench_basic_amd.nn.kernels import RansTables
    rng = np.random.default_rng(3)
    nd, ns = 64, 64
    freqs = rng.integers(1, 1024, (nd, ns)).astype(np.int32)
    nsym, off = np.full(nd, ns, np.int32), np.full(nd, -32, np.int32)
    T = RansTables(freqs=freqs, nsym=nsym, offsets=off)
    eo = oracle.Rans64Encoder(16, True, 4)
    eo.init_params(freqs, nsym, off)
    lens = [0, 1, 5000, 49152, 777, 64, 128, 49152]
    seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    tot = int(seg[-1])
    idx = rng.integers(0, nd, tot).astype(np.int32)
    sym = (rng.integers(-36, 36, tot)).astype(np.int32)
    slot = max(lens) * 3 + 4
    words, nwords = T.encode_batch(torch.from_numpy(sym).cuda(), torch.from_numpy(idx).cuda(), torch.from_numpy(seg).cuda(), slot)
    words, nwords = words.cpu().numpy().view(np.uint32), nwords.cpu().numpy()
    streams = []
    for i, L in enumerate(lens):
        b = words[i, slot - nwords[i]:].tobytes()
        assert b == eo.encode_with_indexes(sym[seg[i]:seg[i + 1]], idx[seg[i]:seg[i + 1]]), i
        streams.append(np.frombuffer(b, np.uint32))
    woff = np.concatenate([[0], np.cumsum([s.size for s in streams])]).astype(np.int64)
    allw = np.concatenate(streams).view(np.int32)
    out, state, pos = T.decode_batch(torch.from_numpy(allw).cuda(), torch.from_numpy(woff).cuda(), torch.from_numpy(idx).cuda(),
                                     torch.from_numpy(seg).cuda())
    assert np.array_equal(out.cpu().numpy(), sym)


@pytest.mark.parametrize("seed", range(72))
def test_batched_streams_fuzz(oracle, seed):
    """Random tables (narrow and wide rows, precision 12..16, bypass on/off) and ragged stream lengths through the
    batched device entry points -- the lane-parallel fast encoder / decoder whenever the table allows them -- against
    the oracle stream by stream; includes bypass-heavy data that overflows the reference's n+2-word slot."""
    import torch
    from cbench_basic_amd.nn.kernels import RansTables
    rng = np.random.default_rng(500 + seed)
    nd = int(rng.integers(1, 40))
    wide = seed % 3 == 0
    ns = int(rng.integers(70, 1500)) if wide else int(rng.integers(2, 64))
    byp = bool(seed % 2)
    prec = 16 if seed % 4 == 0 else int(rng.integers(12, 17))
    freqs = rng.integers(1, 200, (nd, ns)).astype(np.int32)
    nsym = rng.integers(max(2, ns // 2), ns + 1, nd).astype(np.int32)
    nsym[0] = ns
    off = rng.integers(-40, 5, nd).astype(np.int32)
    T = RansTables(freqs=freqs, nsym=nsym, offsets=off, precision=prec, bypass=byp, bypass_precision=4)
    eo = oracle.Rans64Encoder(prec, byp, 4)
    eo.init_params(freqs, nsym, off)
    lens = [int(v) for v in rng.choice([0, 1, 2, 63, 64, 65, 127, 128, 129, 1000, 5000], size=int(rng.integers(1, 9)))]
    seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    tot = int(seg[-1])
    idx = rng.integers(0, nd, tot).astype(np.int32)
    if byp:
        sym = (off[idx] + rng.integers(-3, nsym[idx] + 3)).astype(np.int32)
        if seed % 6 == 1:
            sym[::3] = rng.integers(-10 ** 6, 10 ** 6, sym[::3].size)  # long escape codes
    else:
        sym = (off[idx] + rng.integers(0, 1 << 30, tot) % nsym[idx]).astype(np.int32)
    slot = max(lens) * 3 + 4
    words, nwords = T.encode_batch(torch.from_numpy(sym).cuda(), torch.from_numpy(idx).cuda(), torch.from_numpy(seg).cuda(), slot)
    words, nwords = words.cpu().numpy().view(np.uint32), nwords.cpu().numpy()
    streams = []
    for i, L in enumerate(lens):
        b = words[i, slot - nwords[i]:].tobytes()
        assert b == eo.encode_with_indexes(sym[seg[i]:seg[i + 1]], idx[seg[i]:seg[i + 1]]), (i, L)
        streams.append(np.frombuffer(b, np.uint32))
    woff = np.concatenate([[0], np.cumsum([s.size for s in streams])]).astype(np.int64)
    allw = np.concatenate(streams).view(np.int32)
    out, _, _ = T.decode_batch(torch.from_numpy(allw).cuda(), torch.from_numpy(woff).cuda(), torch.from_numpy(idx).cuda(),
                               torch.from_numpy(seg).cuda())
    assert np.array_equal(out.cpu().numpy(), sym)
    # equal-length convenience path with the reference's n+2 slot and its overflow fallback
    if tot >= 64:
        n = 32
        k = tot // n
        strs = T.encode_batch_to_bytes(torch.from_numpy(sym[: k * n]).cuda(), torch.from_numpy(idx[: k * n]).cuda(), n)
        for i in (0, k - 1):
            assert strs[i] == eo.encode_with_indexes(sym[i * n:(i + 1) * n], idx[i * n:(i + 1) * n])
        back = T.decode_batch_from_bytes(strs, torch.from_numpy(idx[: k * n]).cuda(), n)
        assert np.array_equal(back.cpu().numpy(), sym[: k * n])


@pytest.mark.parametrize("prec,every", [(16, 1), (16, 2), (14, 1), (16, 5)])
def test_bypass_heavy_chunks(oracle, prec, every):
    """Chunks of 64 symbols that read far more stream words than symbols: every `every`-th value is a bypass value of up to
    31 bits (eight payload nibbles + count + sentinel), so a chunk of the wave decoder (csrc/wave_decoder.h) runs through its
    64-word register and the blocks behind it and has to line its words up again after every one of them; the values between
    take the common path at precision 16 and the generic one at 14."""
    import torch
    from cbench_basic_amd.nn.kernels import RansTables
    rng = np.random.default_rng(prec * 10 + every)
    nd, ns = 5, 40
    freqs = rng.integers(1, 300, (nd, ns)).astype(np.int32)
    nsym, off = np.full(nd, ns, np.int32), np.full(nd, -20, np.int32)
    T = RansTables(freqs=freqs, nsym=nsym, offsets=off, precision=prec, bypass=True, bypass_precision=4)
    eo = oracle.Rans64Encoder(prec, True, 4)
    eo.init_params(freqs, nsym, off)
    lens = [64, 640, 1000, 4096, 63]
    seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    tot = int(seg[-1])
    idx = rng.integers(0, nd, tot).astype(np.int32)
    sym = rng.integers(-20, 18, tot).astype(np.int32)
    big = rng.integers(1 << 20, (1 << 30) - 1, sym[::every].size) * rng.choice([-1, 1], sym[::every].size)
    sym[::every] = big.astype(np.int32)
    slot = max(lens) * 3 + 4
    words, nwords = T.encode_batch(torch.from_numpy(sym).cuda(), torch.from_numpy(idx).cuda(), torch.from_numpy(seg).cuda(), slot)
    words, nwords = words.cpu().numpy().view(np.uint32), nwords.cpu().numpy()
    streams = []
    for i, L in enumerate(lens):
        b = words[i, slot - nwords[i]:].tobytes()
        assert b == eo.encode_with_indexes(sym[seg[i]:seg[i + 1]], idx[seg[i]:seg[i + 1]]), (i, L)
        streams.append(np.frombuffer(b, np.uint32))
    assert streams[3].size > 4096 * (1.2 if every == 1 else 0.2)   # more words than the one-per-symbol the common path can read
    woff = np.concatenate([[0], np.cumsum([s.size for s in streams])]).astype(np.int64)
    allw = np.concatenate(streams).view(np.int32)
    out, _, _ = T.decode_batch(torch.from_numpy(allw).cuda(), torch.from_numpy(woff).cuda(), torch.from_numpy(idx).cuda(),
                               torch.from_numpy(seg).cuda())
    assert np.array_equal(out.cpu().numpy(), sym)
