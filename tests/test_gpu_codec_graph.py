"""GPU parity of the WHOLE codec (GeneralCodec + LatentGraphicalANSEntropyCoder on the HIP path) against
tests/golden/codec_graph.npz = the reference's own GeneralCodec / LatentGraphicalANSEntropyCoder / slimmable models run
end to end on CPU (make_golden.py::codec_graph): three topo-group graphs and the BaSIC graph at eight controller
settings covering all five widths.  Bytes must be IDENTICAL to the reference's (the only tolerated difference would be
an fp32 rounding tie, and then the case must be listed in KNOWN_TIES with the flipped symbol)."""
import math

import numpy as np
import pytest
import torch

import codec_cases as cc

pytestmark = pytest.mark.gpu

KNOWN_TIES = {}   # record prefix -> description of the flipped element; empty = every stream is byte-identical


@pytest.mark.parametrize("k", ["t0", "t1", "t2", "b0", "h0", "h1", "h2", "g0"])   # g0: two inference edges into z, averaged (multi-edge aggregation)
def test_codec_bytes_latents_reconstruction_vs_reference(k):
    z = cc.load()
    codec, _ = cc.build_codec(z, k)
    codec = codec.cuda()
    codec.update_state()
    ec = codec.entropy_coder
    x = cc.case_input(z, k)
    identical = 0
    recs = cc.records(z, k)
    for rec, level in recs:
        if level is not None:
            codec.set_complex_level(level)
        ref_bytes = z[f"{rec}.bytes"].tobytes()
        data = codec.compress(x)          # host tensor in: the upload is the codec's (general_codec.py:46-47)
        # latents of the inference pass
        node = ec._node_generate_process(**ec._get_default_node_dict(force_add_default_dynamic_nodes=True))
        lat = ec._inference_process({"x": x.cuda(), **node})
        for name in ("y", "z"):
            ref = torch.from_numpy(z[f"{rec}.{name}"])
            err = float((lat[name].cpu() - ref).abs().max())
            assert err <= 1e-4 * max(1.0, float(ref.abs().max())), (rec, name, err)
        same = data == ref_bytes
        identical += int(same)
        if not same:
            assert rec in KNOWN_TIES, f"{rec}: {len(data)} B vs reference {len(ref_bytes)} B - stream differs and no tie is recorded"
        # the REFERENCE's stream decodes on the GPU to the reference's reconstruction
        xhat = codec.decompress(ref_bytes).cpu()
        ref = torch.from_numpy(z[f"{rec}.xhat"])
        assert float((xhat - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), rec
        if same:
            assert torch.equal(codec.decompress(data).cpu(), xhat), rec
        # forward(): rate estimate and distortion metrics of the reference's eval forward
        codec.reset_all_cache() if hasattr(codec, "reset_all_cache") else None
        xf = codec(x)
        m = cc.metrics(z, rec)
        got = float(ec.get_raw_cache("metric_dict")["prior_entropy"])
        assert abs(got - m["prior_entropy"]) <= 2e-3 * m["prior_entropy"], (rec, got, m["prior_entropy"])
        bpd = float(ec.get_raw_cache("metric_dict")["estimated_bpd"])
        assert abs(bpd - m["estimated_bpd"]) <= 2e-3 * m["estimated_bpd"], (rec, bpd, m["estimated_bpd"])
        if f"{rec}.xfwd_shape" in z:    # forward() returns the input-sized image, decompress() the un-cropped synthesis output
            assert list(xf.shape) == [int(v) for v in z[f"{rec}.xfwd_shape"]], rec
        ref = ref[..., :xf.shape[-2], :xf.shape[-1]]
        assert float((xf.cpu() - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())) + float(z[f"{rec}.xfwd_minus_xhat_max"]), rec
    print(f"{k}: {identical}/{len(recs)} streams byte-identical to the reference's")
    assert identical == len(recs) - sum(1 for r, _ in recs if r in KNOWN_TIES)


@pytest.mark.parametrize("k", ["h0", "h1", "h2"])
def test_hyperprior_graph_module_and_fused_paths_vs_reference(k):
    """The plain hyperprior graph (what bench.py times and basic_hp_encode_images / basic_hp_decode_images serve) against the
    REFERENCE's own run of GeneralCodec(LatentGraphicalANSEntropyCoder(y=CompressAIGaussianConditionalCoder, z=CompressAI
    EntropyBottleneckPriorCoder)) (configs/lossy_graph_scalable_exp_hp.py:182-215): the integers both coders hand to the native
    encoder, the GaussianConditional tables, and the bytes -- through the module-by-module path AND the fused C entry points,
    for a host and a device batch; the reference's bytes decode to the reference's reconstruction on both paths."""
    from cbench_basic_amd.nn import kernels as K
    z = cc.load()
    codec, _ = cc.build_codec(z, k)
    codec = codec.cuda()
    codec.update_state()
    ec = codec.entropy_coder
    x = cc.case_input(z, k)
    B = x.shape[0]
    ref_bytes = z[f"{k}.bytes"].tobytes()
    zc, yc = ec.latent_node_entropy_coders["z"], ec.latent_node_entropy_coders["y"]
    # tables of the y-coder == the reference's gaussian_conditional buffers
    cdf, length, offset = yc.gaussian_conditional.host_tables()
    assert np.array_equal(length, z[f"{k}.gc_cdf_length"]) and np.array_equal(offset, z[f"{k}.gc_offset"])
    import hashlib
    assert hashlib.sha256(cdf.tobytes()).hexdigest() == str(z[f"{k}.gc_cdf_sha256"])
    # integers: the HIP quantise / index kernels on the HIP transforms' latents
    node = ec._node_generate_process(**ec._get_default_node_dict(force_add_default_dynamic_nodes=True))
    lat = ec._inference_process({"x": x.cuda(), **node})
    zc._ready()
    yc._ready()
    zsym, zidx, zhat = K.eb_quantize_index(lat["z"], zc._medians_dev)
    assert np.array_equal(zsym.cpu().numpy().reshape(B, -1), z[f"{k}.z_symbols"]), k
    assert np.array_equal(zidx.cpu().numpy().reshape(B, -1), z[f"{k}.z_indexes"]), k
    scales = ec.latent_generative_modules["z_y"](zhat)[..., :lat["y"].shape[-2], :lat["y"].shape[-1]].contiguous()
    ysym, yidx, _ = K.gc_quantize_index(lat["y"], scales, yc._scale_table_dev, yc.scale_bound)
    assert np.array_equal(yidx.cpu().numpy().reshape(B, -1), z[f"{k}.indexes"]), k
    assert np.array_equal(ysym.cpu().numpy().reshape(B, -1), z[f"{k}.symbols"]), k
    # bytes, both paths, both input placements
    xref = torch.from_numpy(z[f"{k}.xhat"])
    outs = {}
    for fused in (False, True):
        ec.use_fused_session = fused
        assert (ec._fused_session({}, None) is not None) == fused
        for inp in (x, x.cuda()):
            assert codec.compress(inp) == ref_bytes, (k, fused, inp.device)
        xhat = codec.decompress(ref_bytes)
        assert xhat.shape == xref.shape, (k, fused, xhat.shape)
        assert float((xhat.cpu() - xref).abs().max()) <= 1e-4 * max(1.0, float(xref.abs().max())), (k, fused)
        outs[fused] = xhat
    assert torch.equal(outs[False], outs[True])
    assert ec.profiler.count["encode_fused"] >= 2 and ec.profiler.count["decode_fused"] >= 1


def test_complexity_metrics_of_levels_match_reference_flops():
    """get_current_complex_metrics / the slimmable models' operation counters at the fixture's eight controller settings
    equal the reference models' get_current_flops() after the same forward."""
    z = cc.load()
    codec, _ = cc.build_codec(z, "b0")
    codec = codec.cuda()
    codec.update_state()
    ec = codec.entropy_coder
    x = cc.case_input(z, "b0")
    for rec, level in cc.records(z, "b0"):
        if f"{rec}.flops" not in z:
            pytest.skip("fixture has no FLOPs")
        codec.set_complex_level(level)
        codec(x)
        mods = list(ec.latent_inference_modules.values()) + list(ec.latent_generative_modules.values())
        got = sum(float(m.get_current_flops()) for m in mods)
        assert got == float(z[f"{rec}.flops"]), (rec, got, float(z[f"{rec}.flops"]))
