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
