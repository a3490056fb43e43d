"""The reference's own outputs fed DIRECTLY to the HIP path (no oracle in between): known-answer vectors of the
reference's compiled cbench.ans / cbench.rans (tests/golden/rans_kat.npz) through the cbench.ans / cbench.rans drop-ins,
and the reference's TopoGroupDynamicMaskConv2d outputs (tests/golden/masked_conv.npz) through the masked-conv kernel."""
import hashlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_ans_known_answers_on_hip():
    from cbench_basic_amd import ans
    z = load("rans_kat.npz")
    for name in z["names"]:
        prec, byp, bprec = (int(v) for v in z[f"{name}.cfg"])
        enc, dec = ans.Rans64Encoder(prec, bool(byp), bprec), ans.Rans64Decoder(prec, bool(byp), bprec)
        enc.init_params(z[f"{name}.freqs"], z[f"{name}.nsym"], z[f"{name}.offsets"])
        dec.init_params(z[f"{name}.freqs"], z[f"{name}.nsym"], z[f"{name}.offsets"])
        cd = enc.get_cdfs()
        assert np.array_equal(cd[:, : z[f"{name}.cdfs"].shape[1]], z[f"{name}.cdfs"][:, : cd.shape[1]]), name
        ref = z[f"{name}.bytes"].tobytes()
        assert enc.encode_with_indexes(z[f"{name}.symbols"], z[f"{name}.indexes"]) == ref, name
        assert np.array_equal(dec.decode_with_indexes(ref, z[f"{name}.indexes"]), z[f"{name}.symbols"]), name
        n = z[f"{name}.indexes"].size
        if n > 3:    # streamed decode of the reference's bytes in two calls
            dec.set_stream(ref)
            a, b = dec.decode_stream(z[f"{name}.indexes"][: n // 3]), dec.decode_stream(z[f"{name}.indexes"][n // 3:])
            assert np.array_equal(np.concatenate([a, b]), z[f"{name}.symbols"]), name
    assert ans.pmf_to_quantized_cdf(z["pmf_cdf.in"], 16) == z["pmf_cdf.out"].tolist()


def test_ans_survey_large_vector_on_hip():
    """seed-0 recipe, 49,152 symbols -> 46,528 bytes with the reference's sha256."""
    from cbench_basic_amd import ans
    z = load("rans_kat.npz")
    np.random.seed(0)
    freqs = np.random.randint(1, 1024, (64, 64)).astype(np.int32)
    enc = ans.Rans64Encoder(16, True, 4)
    enc.init_params(freqs, np.full(64, 64, np.int32), np.zeros(64, np.int32))
    data = np.random.randint(-3, 67, (1, 192, 16, 16)).astype(np.int32)
    idx = np.random.randint(0, 64, (1, 192, 16, 16)).astype(np.int32)
    b = enc.encode_with_indexes(data, idx)
    assert len(b) == 46528 == int(z["survey_large.nbytes"][0])
    assert hashlib.sha256(b).digest() == z["survey_large.sha256"].tobytes()
    dec = ans.Rans64Decoder(16, True, 4)
    dec.init_params(freqs, np.full(64, 64, np.int32), np.zeros(64, np.int32))
    out = dec.decode_with_indexes(b, idx)
    assert out.shape == idx.shape and np.array_equal(out, data)


def test_rans_fork_module_known_answers_on_hip():
    """cbench.rans drop-in (rans_interface.cpp:550-576) on the bytes of the reference's compiled cbench.rans."""
    from cbench_basic_amd import rans
    z = load("rans_kat.npz")
    sizes = z["fork.sizes"].tolist()
    cdfs = [row[:n].tolist() for row, n in zip(z["fork.cdfs"], sizes)]
    offs, sym, idx = z["fork.offsets"].tolist(), z["fork.symbols"].tolist(), z["fork.indexes"].tolist()
    ref = z["fork.bytes"].tobytes()
    assert rans.RansEncoder().encode_with_indexes(sym, idx, cdfs, sizes, offs) == ref
    assert rans.RansEncoder().encode_with_indexes_np(z["fork.symbols"], z["fork.indexes"], z["fork.cdfs"], z["fork.sizes"], z["fork.offsets"]) == ref
    buf = rans.BufferedRansEncoder()          # two calls, one flush: same stream
    buf.encode_with_indexes(sym[:123], idx[:123], cdfs, sizes, offs)
    buf.encode_with_indexes_np(z["fork.symbols"][123:], z["fork.indexes"][123:], z["fork.cdfs"], z["fork.sizes"], z["fork.offsets"])
    assert buf.flush() == ref
    dec = rans.RansDecoder()
    assert dec.decode_with_indexes(ref, idx, cdfs, sizes, offs) == sym
    out = dec.decode_with_indexes_np(ref, z["fork.indexes"], z["fork.cdfs"], z["fork.sizes"], z["fork.offsets"])
    assert isinstance(out, np.ndarray) and out.dtype == np.int32 and np.array_equal(out, z["fork.symbols"])
    dec.set_stream(ref)
    a = dec.decode_stream(idx[:200], cdfs, sizes, offs)
    b = dec.decode_stream_np(z["fork.indexes"][200:], z["fork.cdfs"], z["fork.sizes"], z["fork.offsets"])
    assert a + b.tolist() == sym
    assert rans.pmf_to_quantized_cdf([.1, .2, .7], 16) == [0, 6554, 19661, 65536]
    got = rans.pmf_to_quantized_cdf_np(np.array([[.1, .2, .7], [.5, .25, .25]], np.float32), 16)
    assert got.dtype == np.uint32 and got.tolist() == [[0, 6554, 19661, 65536], [0, 32768, 49152, 65536]]
    with pytest.raises(ValueError):
        rans.pmf_to_quantized_cdf([0.5, float("nan")], 16)
    with pytest.raises(ValueError):
        rans.pmf_to_quantized_cdf([0.5, -0.1], 16)


def test_rans_fork_tables_changing_between_buffered_calls():
    """BufferedRansEncoder converts each call's symbols with THAT call's tables (rans_interface.cpp:109-174); the drop-in
    must give the stream of the equivalent single table set."""
    from cbench_basic_amd import ans, rans
    rng = np.random.default_rng(5)
    t1 = [rans.pmf_to_quantized_cdf((p / p.sum()).tolist() + [1e-6], 16) for p in rng.random((3, 9)).astype(np.float32)]
    t2 = [rans.pmf_to_quantized_cdf((p / p.sum()).tolist() + [1e-6], 16) for p in rng.random((2, 17)).astype(np.float32)]
    s1, i1 = rng.integers(-3, 12, 300).tolist(), rng.integers(0, 3, 300).tolist()
    s2, i2 = rng.integers(-9, 9, 200).tolist(), rng.integers(0, 2, 200).tolist()
    buf = rans.BufferedRansEncoder()
    buf.encode_with_indexes(s1, i1, t1, [len(c) for c in t1], [0, 0, 0])
    buf.encode_with_indexes(s2, i2, t2, [len(c) for c in t2], [-8, -8])
    data = buf.flush()
    width = max(len(c) for c in t1 + t2)
    cd = np.zeros((5, width), np.int32)
    for r, c in enumerate(t1 + t2):
        cd[r, : len(c)] = c
    dec = ans.Rans64Decoder(16, True, 4)
    dec.init_cdf_params(cd, [len(c) for c in t1 + t2], [0, 0, 0, -8, -8])
    out = dec.decode_with_indexes(data, np.array(i1 + [i + 3 for i in i2]))
    assert out.tolist() == s1 + s2


def test_symbol_cache_peek_and_flush_vs_reference():
    """encode_with_indexes(..., cache=True) / peek_cache / flush (rans64.cpp:237-386, rans64.hpp:78-86) against the
    reference's compiled Rans64Encoder: the (start, range, bypass) rows after every cached call and the flushed bytes."""
    from cbench_basic_amd import ans
    z = load("rans_cache_kat.npz")
    for k in (str(n) for n in z["names"]):
        enc = ans.Rans64Encoder(16, bool(int(z[f"{k}.bypass"])), 4)
        enc.init_params(z[f"{k}.freqs"], z[f"{k}.nsym"], z[f"{k}.offsets"])
        for j in range(int(z[f"{k}.ncalls"])):
            assert enc.encode_with_indexes(z[f"{k}.sym{j}"], z[f"{k}.idx{j}"], cache=True) == b""
            assert np.array_equal(enc.peek_cache(), z[f"{k}.peek{j}"]), (k, j)
        assert enc.flush() == z[f"{k}.flush"].tobytes(), k
        assert enc.peek_cache().shape == (0, 3) and z[f"{k}.peek_after"].shape == (0, 3)
    # cached encoding on autoregressive table sets (rans64.cpp:237-361): remap tables of order 1 / 2 and the custom op
    for k in (str(n) for n in z["arnames"]):
        enc = ans.Rans64Encoder(16, True, 4)
        enc.init_params(z[f"{k}.freqs"], z[f"{k}.nsym"], z[f"{k}.offsets"])
        order = int(z[f"{k}.order"])
        if str(z[f"{k}.mode"]) == "table":
            tab = z[f"{k}.ar_table"]
            enc.init_ar_params(tab, np.zeros((tab.shape[0], order, 1), np.int32))
        else:
            enc.init_custom_ar_ops([ans.ar_limited_scaled_add_linear_op([float(v) for v in o[:order]], float(o[3]), float(o[4]), float(o[5]), float(o[6]))
                                    for o in z[f"{k}.ops"]])
        for j in range(int(z[f"{k}.ncalls"])):
            assert enc.encode_with_indexes(z[f"{k}.sym{j}"], z[f"{k}.idx{j}"], z[f"{k}.ai{j}"], z[f"{k}.aoff{j}"], True) == b""
            assert np.array_equal(enc.peek_cache(), z[f"{k}.peek{j}"]), (k, j)
        assert enc.flush() == z[f"{k}.flush"].tobytes(), k
        assert enc.peek_cache().shape == (0, 3)


def test_masked_conv_reference_outputs_on_hip():
    """TopoGroupDynamicMaskConv2d.forward outputs of the reference (masked_conv.py:102-228; 5x5 / 3x3 / 1x1 kernels,
    1-4 channel groups, allow_same_topogroup_conv, channel_group_mask) reproduced by masked_conv_pos_kernel."""
    from cbench_basic_amd.nn import kernels as K
    z = load("masked_conv.npz")
    for k in z["keys"]:
        cin, cout, ks, gi, same, use_mask = (int(v) for v in z[f"{k}.cfg"])
        x, w, b = (torch.from_numpy(z[f"{k}.{n}"]) for n in ("x", "weight", "bias"))
        topo = torch.from_numpy(z[f"{k}.topo"])[0]                  # [gi, H, W]
        sel = ([True] * (gi // 2) + [False] * (gi - gi // 2)) if use_mask else [True] * gi
        topo_out = topo[torch.tensor(sel)]
        B, _, H, W = x.shape
        plan = K.MaskedConvPlan(w, b, gi, int(topo_out.shape[0]), bool(same))
        out = torch.zeros(B, cout, H, W).cuda()
        pos = torch.arange(B * H * W, dtype=torch.int32).cuda()
        plan(x.cuda(), topo.int().cuda(), topo_out.int().cuda(), pos, out)
        torch.cuda.synchronize()
        ref = torch.from_numpy(z[f"{k}.y"])
        err = float((out.cpu() - ref).abs().max())
        assert err <= 1e-4 * max(1.0, float(ref.abs().max())), (str(k), err)


def test_tans_known_answers_on_hip():
    """cbench.ans.TansEncoder / TansDecoder drop-ins (csrc/tans.hip) on the bytes of the reference's compiled classes:
    10 random table sets (table_log 5..12, with / without bypass coding, skewed counts with rare symbols), the reference's
    two budget rules (ValueError / empty result) and the order-1 / order-2 AR remap."""
    from cbench_basic_amd import ans
    import tans_cases
    tans_cases.check_known_answers(ans)


def test_tans_tables_and_random_streams_vs_oracle_on_hip():
    from cbench_basic_amd import ans
    from oracle import tans_oracle
    import tans_cases
    rng = np.random.default_rng(5)
    coded = 0
    for trial in range(40):
        case = tans_cases.random_case(rng, trial)
        L, freqs, nsym, off, byp, sym, idx = case
        eo, bo, do = tans_cases.run(tans_oracle, case)
        eh, bh, dh = tans_cases.run(ans, case)
        assert (eo is None) == (eh is None) and bo == bh, trial
        if bo:
            coded += 1
            assert np.array_equal(do, dh), trial
        if eo is None and trial < 12:    # table level: every row of the device images == the oracle's tables
            enc = ans.TansEncoder(L, 255, byp, 4)
            enc.init_params(freqs, nsym, off)
            for r in range(freqs.shape[0]):
                t = tans_oracle.tables(freqs[r, : nsym[r]], L)
                nxt, db, ds, dec = enc.get_table_row(r)
                assert np.array_equal(nxt, t["next_state"]), (trial, r)
                live = freqs[r, : nsym[r]] > 0
                assert np.array_equal(db[: nsym[r]][live], t["delta_bits"][live]) and np.array_equal(ds[: nsym[r]][live], t["delta_state"][live])
                assert np.array_equal(dec & 0xFFF, t["d_base"]) and np.array_equal((dec >> 12) & 0xF, t["d_bits"]) \
                    and np.array_equal(dec >> 16, t["d_symbol"]), (trial, r)
    assert coded >= 20


def test_tans_batched_device_streams_on_hip():
    """basic_tans_encode_batch_dev / decode_batch_dev: 37 ragged streams in one launch each == the one-stream drop-in."""
    import ctypes
    from cbench_basic_amd import _lib, ans
    rng = np.random.default_rng(9)
    L, nd, ns = 11, 6, 90
    freqs = rng.integers(1, 1024, (nd, ns)).astype(np.int32)
    nsym, off = np.full(nd, ns, np.int32), rng.integers(-4, 4, nd).astype(np.int32)
    enc, dec = ans.TansEncoder(L, 255, True, 4), ans.TansDecoder(L, 255, True, 4)
    enc.init_params(freqs, nsym, off)
    dec.init_params(freqs, nsym, off)
    lens = rng.integers(0, 700, 37)
    lens[3] = 0
    seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    n = int(seg[-1])
    idx = rng.integers(0, nd, n).astype(np.int32)
    sym = (off[idx] + rng.integers(-2, ns + 3, n)).astype(np.int32)
    sym[::17] = rng.integers(-3000, 3000, sym[::17].size)
    dev = torch.device("cuda")
    d_sym, d_idx, d_seg = (torch.from_numpy(a).to(dev) for a in (sym, idx, seg))
    lib = _lib.lib()
    slot = int(lib.basic_tans_encode_bound_words(enc._tables, int(lens.max())))
    d_words = torch.zeros(37 * slot, dtype=torch.int32, device=dev)
    d_info = torch.zeros(37 * 2, dtype=torch.int64, device=dev)
    _lib.check(lib.basic_tans_encode_batch_dev(enc._tables, d_sym.data_ptr(), d_idx.data_ptr(), d_seg.data_ptr(), 37, d_words.data_ptr(),
                                               slot, d_info.data_ptr(), None))
    torch.cuda.synchronize()
    info = d_info.cpu().numpy().reshape(37, 2)
    words = d_words.cpu().numpy().view(np.uint8).reshape(37, slot * 4)
    streams = [words[i, : (info[i, 0] + 7) // 8].tobytes() for i in range(37)]
    for i in (0, 3, 11, 36):
        s, e = seg[i], seg[i + 1]
        big = ans.TansEncoder(L, 255, True, 4)
        big.init_params(freqs, nsym, off)
        one, coded = big._encode(sym[s:e], idx[s:e], None, None, 1 << 40)   # no budget rule: the raw stream
        assert streams[i] == one and coded == info[i, 1], i
    blob = np.frombuffer(b"".join(streams), np.uint8)
    boff = np.concatenate([[0], np.cumsum([len(b) for b in streams])]).astype(np.int64)
    d_blob, d_boff = torch.from_numpy(blob.copy()).to(dev), torch.from_numpy(boff).to(dev)
    d_out = torch.zeros(n, dtype=torch.int32, device=dev)
    d_status = torch.full((37,), -1, dtype=torch.int32, device=dev)
    _lib.check(lib.basic_tans_decode_batch_dev(dec._tables, d_blob.data_ptr(), d_boff.data_ptr(), d_idx.data_ptr(), d_seg.data_ptr(), 37,
                                               d_out.data_ptr(), d_status.data_ptr(), None))
    torch.cuda.synchronize()
    assert (d_status.cpu().numpy() == 0).all()
    assert np.array_equal(d_out.cpu().numpy(), sym)


def test_custom_ar_ops_on_hip():
    """cbench.ans.init_custom_ar_ops / ar_limited_scaled_add_linear_op / ar_linear_op (lib.cpp:17-25, ar_funcs.hpp:29-87) on the
    bytes and call results of the reference's compiled module."""
    from cbench_basic_amd import ans
    from test_oracle_golden import _ar_ops_case
    z = load("ar_ops_kat.npz")
    for name in z["names"]:
        _ar_ops_case(z, str(name), ans)
    lim = ans.ar_limited_scaled_add_linear_op([0.37, -0.21], 0.4, 2.0, 0.0, 7.0)
    lin = ans.ar_linear_op([1.0, 0.5, -0.25], 0.125, 2.0)
    assert [lim(v.tolist()) for v in z["call.vectors"]] == z["call.limited"].tolist()
    assert [lin(v.tolist()) for v in z["call.vectors"]] == z["call.linear"].tolist()
    enc = ans.Rans64Encoder(16, True, 4)
    enc.init_params(z["o1.freqs"], z["o1.nsym"], z["o1.offsets"])
    with pytest.raises(TypeError):
        enc.init_custom_ar_ops([lin])


@pytest.mark.parametrize("nd", [64, 4])
def test_tans_full_size_round_trip_property(nd):
    """BASELINE configs[4] size through the batched tANS entry points: 256 image streams x 49,152 symbols over 64 distributions
    (table_log 12, bypass coding on; tables in L2) and over 4 (tables copied to LDS by every workgroup), encode -> decode == input;
    stream 0 equals the one-stream drop-in (which is pinned to the reference by the known answers).  Prints the kernel rates."""
    import time
    from cbench_basic_amd import _lib, ans
    rng = np.random.default_rng(77)
    L, ns, S, n = 12, 64, 256, 49152
    freqs = rng.integers(1, 1024, (nd, ns)).astype(np.int32)
    nsym, off = np.full(nd, ns, np.int32), np.zeros(nd, np.int32)
    enc, dec = ans.TansEncoder(L, 255, True, 4), ans.TansDecoder(L, 255, True, 4)
    enc.init_params(freqs, nsym, off)
    dec.init_params(freqs, nsym, off)
    dev = torch.device("cuda")
    sym = torch.from_numpy(rng.integers(-2, ns + 2, (S, n)).astype(np.int32)).to(dev)
    idx = torch.from_numpy(rng.integers(0, nd, (S, n)).astype(np.int32)).to(dev)
    seg = torch.arange(0, (S + 1) * n, n, dtype=torch.int64, device=dev)
    lib = _lib.lib()
    slot = int(lib.basic_tans_encode_bound_words(enc._tables, n))
    words = torch.zeros(S * slot, dtype=torch.int32, device=dev)
    info = torch.zeros(S * 2, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    _lib.check(lib.basic_tans_encode_batch_dev(enc._tables, sym.data_ptr(), idx.data_ptr(), seg.data_ptr(), S, words.data_ptr(), slot,
                                               info.data_ptr(), None))
    torch.cuda.synchronize()
    t_enc = time.time() - t0
    bits = info.cpu().numpy().reshape(S, 2)[:, 0]
    assert (bits > 0).all()
    nbytes = (bits + 7) // 8
    host = words.cpu().numpy().view(np.uint8).reshape(S, slot * 4)
    streams = [host[i, : nbytes[i]].tobytes() for i in range(S)]
    one, _ = enc._encode(sym[0].cpu().numpy(), idx[0].cpu().numpy(), None, None, 1 << 40)
    assert streams[0] == one
    blob = torch.from_numpy(np.frombuffer(b"".join(streams), np.uint8).copy()).to(dev)
    boff = torch.from_numpy(np.concatenate([[0], np.cumsum(nbytes)]).astype(np.int64)).to(dev)
    out = torch.zeros_like(sym)
    status = torch.full((S,), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    _lib.check(lib.basic_tans_decode_batch_dev(dec._tables, blob.data_ptr(), boff.data_ptr(), idx.data_ptr(), seg.data_ptr(), S,
                                               out.data_ptr(), status.data_ptr(), None))
    torch.cuda.synchronize()
    t_dec = time.time() - t0
    assert int(status.abs().sum()) == 0 and torch.equal(out, sym)
    print(f"tANS, {nd} distributions, 256 streams x 49,152 symbols: encode {t_enc * 1e3:.1f} ms ({t_enc / n * 1e9:.0f} ns per symbol and stream), "
          f"decode {t_dec * 1e3:.1f} ms ({t_dec / n * 1e9:.0f} ns), {nbytes.sum() / S / n * 8:.2f} bits per symbol")
