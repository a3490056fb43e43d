#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF, in this container:
  * the reference's native rANS (cbench/csrc/ans, cbench/csrc/rans compiled to oracle/_ref), and
  * the reference's own Python (cbench.modules..., imported through tests/golden/ref_import.py).
Only data is written (inputs and expected outputs as .npz); no reference source is copied.
Run:  python tests/golden/make_golden.py     (needs /root/reference; never runs on the GPU box)
"""
import hashlib
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

cbench = ref_import.install()
from cbench import ans as ref_ans, rans as ref_rans  # noqa: E402
from cbench.utils.bytes_ops import merge_bytes, split_merged_bytes, encode_shape  # noqa: E402
from cbench.nn.layers.masked_conv import TopoGroupDynamicMaskConv2d, TopoGroupDynamicMaskConv2dContextModel  # noqa: E402
from cbench.modules.prior_model.prior_coder.pgm_coder import GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder  # noqa: E402
from cbench.modules.prior_model.prior_coder.compressai_coder import get_scale_table  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def b2a(b):
    return np.frombuffer(b, dtype=np.uint8).copy()


# ---------------------------------------------------------------- 1. native rANS known-answer vectors
def rans_kats():
    out = {}
    cases = []
    # SURVEY 8c vectors
    cases.append(("tiny_nobypass", np.array([[3, 1]]), np.array([2]), np.array([0]), 16, False, np.array([0, 1, 0, 0]), np.zeros(4)))
    cases.append(("tiny_bypass", np.array([[3, 1]]), np.array([2]), np.array([0]), 16, True, np.array([0, 1, 5, -2]), np.zeros(4)))
    rng = np.random.default_rng(2024)
    for i, (nd, ns, n, prec, byp) in enumerate([(8, 512, 3000, 16, True), (5, 40, 777, 12, True), (3, 9, 64, 16, False),
                                                (64, 64, 4096, 16, True), (2, 2217, 2000, 16, True), (1, 2, 1, 16, True)]):
        freqs = rng.integers(1, 1024, (nd, ns))
        nsym = rng.integers(2, ns + 1, nd)
        off = rng.integers(-6, 6, nd)
        idx = rng.integers(0, nd, n)
        if byp:
            sym = rng.integers(-30, ns + 30, n)
            sym[::11] = rng.integers(-70000, 70000, sym[::11].size)
        else:
            sym = off[idx] + rng.integers(0, 1 << 30, n) % nsym[idx]
        cases.append((f"rand{i}", freqs, nsym, off, prec, byp, sym, idx))
    names = []
    for name, freqs, nsym, off, prec, byp, sym, idx in cases:
        f, n_, o = (np.asarray(a).astype(np.int32) for a in (freqs, nsym, off))
        s, ix = np.asarray(sym).astype(np.int32), np.asarray(idx).astype(np.int32)
        enc = ref_ans.Rans64Encoder(prec, byp, 4)
        enc.init_params(f, n_, o)
        data = enc.encode_with_indexes(s, ix)
        dec = ref_ans.Rans64Decoder(prec, byp, 4)
        dec.init_params(f, n_, o)
        assert np.array_equal(dec.decode_with_indexes(data, ix), s)
        cd = enc.get_cdfs()
        for r in range(cd.shape[0]):
            cd[r, n_[r] + 2:] = 0  # get_cdfs leaves the padding uninitialised
        out.update({f"{name}.freqs": f, f"{name}.nsym": n_, f"{name}.offsets": o, f"{name}.cfg": np.array([prec, int(byp), 4]),
                    f"{name}.symbols": s, f"{name}.indexes": ix, f"{name}.bytes": b2a(data), f"{name}.cdfs": cd})
        names.append(name)
    # SURVEY 8c large vector: only its digest is kept (inputs are a seeded recipe)
    np.random.seed(0)
    freqs = np.random.randint(1, 1024, (64, 64)).astype(np.int32)
    enc = ref_ans.Rans64Encoder(16, True, 4)
    enc.init_params(freqs, np.full(64, 64, np.int32), np.zeros(64, np.int32))
    data = np.random.randint(-3, 67, (1, 192, 16, 16)).astype(np.int32)
    idx = np.random.randint(0, 64, (1, 192, 16, 16)).astype(np.int32)
    b = enc.encode_with_indexes(data, idx)
    out["survey_large.sha256"] = np.frombuffer(hashlib.sha256(b).digest(), np.uint8)
    out["survey_large.nbytes"] = np.array([len(b)])
    out["pmf_cdf.in"] = np.array([.1, .2, .7], np.float32)
    out["pmf_cdf.out"] = np.array(ref_ans.pmf_to_quantized_cdf([.1, .2, .7], 16), np.int32)
    # CompressAI-fork module (cbench.rans): same bitstream from CDF lists
    cdfs = [ref_rans.pmf_to_quantized_cdf(list(p / p.sum()) + [1e-6], 16) for p in rng.random((6, 30)).astype(np.float32)]
    sizes = [len(c) for c in cdfs]
    offs = [-15] * 6
    sym = rng.integers(-25, 25, 500).tolist()
    idx = rng.integers(0, 6, 500).tolist()
    fb = ref_rans.RansEncoder().encode_with_indexes(sym, idx, cdfs, sizes, offs)
    assert ref_rans.RansDecoder().decode_with_indexes(fb, idx, cdfs, sizes, offs) == sym
    out.update({"fork.cdfs": np.array(cdfs, np.int32), "fork.sizes": np.array(sizes, np.int32), "fork.offsets": np.array(offs, np.int32),
                "fork.symbols": np.array(sym, np.int32), "fork.indexes": np.array(idx, np.int32), "fork.bytes": b2a(fb)})
    out["names"] = np.array(names)
    save("rans_kat.npz", **out)


# ---------------------------------------------------------------- 2. Gaussian PGM tables
def gauss_tables():
    c = GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder(in_channels=192)
    c.update_state()
    f, n, o = c._get_ans_params()
    cd = c.ans_encoder.get_cdfs()
    for r in range(cd.shape[0]):
        cd[r, n[r] + 2:] = 0
    st = get_scale_table().numpy()
    sel = c._select_best_indexes(torch.stack([torch.zeros(8), torch.tensor([0.0, 0.11, 0.117, 0.5, 5.0, 100.0, 256.0, 1000.0])], -1)
                                 .reshape(1, 16, 1, 1))
    save("gauss_pgm_tables.npz", freqs=f, nsym=n, offsets=o, cdfs=cd, scale_table=st,
         freqs_sha256=np.frombuffer(hashlib.sha256(f.tobytes()).digest(), np.uint8),
         select_scales=np.array([0.0, 0.11, 0.117, 0.5, 5.0, 100.0, 256.0, 1000.0], np.float32),
         select_indexes=sel.reshape(-1).numpy().astype(np.int32))


# ---------------------------------------------------------------- 3. topo-group maps
def topo_maps():
    out, keys = {}, []
    for method, G, C in [("none", 1, 16), ("checkerboard", 1, 16), ("raster2x2", 2, 16), ("channelwise", 4, 16),
                         ("channelwise-checkerboard", 2, 16), ("scanline", 1, 16), ("zigzag", 1, 16), ("elic", 1, 128),
                         ("half-checkerboard", 1, 16), ("quarter-checkerboard", 1, 16), ("interlace-checkerboard", 2, 16),
                         ("channelwise-scanline", 2, 16), ("channelwise-g10", 1, 160), ("halfinv-checkerboard", 1, 16)]:
        c = GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder(in_channels=C, channel_groups=G, default_topo_group_method=method)
        for (h, w) in [(4, 4), (5, 7)]:
            pgm = c._get_default_pgm((1, c.channel_groups, h, w))
            k = f"{method}|{G}|{C}|{h}x{w}"
            out[k] = pgm.numpy().astype(np.int32)
            keys.append(k)
    out["keys"] = np.array(keys)
    save("topo_maps.npz", **out)


# ---------------------------------------------------------------- 4. masked convolution
def masked_conv():
    out, keys = {}, []
    g = torch.Generator().manual_seed(11)
    for i, (cin, cout, k, gi, same, use_mask) in enumerate([(8, 16, 5, 1, False, False), (8, 16, 5, 2, False, False),
                                                            (16, 24, 1, 4, True, True), (12, 12, 3, 3, True, False)]):
        conv = TopoGroupDynamicMaskConv2d(cin, cout, k, padding=k // 2, allow_same_topogroup_conv=same)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.2)
            conv.bias.copy_(torch.randn(conv.bias.shape, generator=g) * 0.1)
        x = torch.randn(2, cin, 5, 6, generator=g)
        topo = torch.randint(-1 if same else 0, 4, (1, gi, 5, 6), generator=g)
        mask = ([True] * (gi // 2) + [False] * (gi - gi // 2)) if use_mask else None
        with torch.no_grad():
            y = conv(x, topo, channel_group_mask=mask)
        out.update({f"c{i}.weight": conv.weight.detach().numpy(), f"c{i}.bias": conv.bias.detach().numpy(), f"c{i}.x": x.numpy(),
                    f"c{i}.topo": topo.numpy().astype(np.int32), f"c{i}.y": y.numpy(),
                    f"c{i}.cfg": np.array([cin, cout, k, gi, int(same), int(use_mask)])})
        keys.append(f"c{i}")
    out["keys"] = np.array(keys)
    save("masked_conv.npz", **out)


# ---------------------------------------------------------------- 5. full AR coder runs
def ar_coder():
    out, keys = {}, []
    cfgs = [("none", 16, 1, False, False, (1, 6, 6)), ("checkerboard", 16, 1, True, False, (1, 6, 6)),
            ("channelwise", 16, 2, False, False, (1, 5, 7)), ("raster2x2", 16, 1, False, False, (1, 6, 6)),
            ("scanline", 16, 1, False, True, (1, 4, 4)), ("elic", 128, 1, False, False, (1, 4, 4)),
            ("checkerboard", 16, 1, False, False, (2, 4, 6)), ("channelwise-checkerboard", 16, 2, True, False, (1, 4, 4))]
    for i, (method, C, G, expand, ctxm, (B, H, W)) in enumerate(cfgs):
        torch.manual_seed(100 + i)
        kw = dict(in_channels=C, channel_groups=G, default_topo_group_method=method, param_merger_expand_bottleneck=expand)
        if ctxm:
            kw["topo_group_context_model"] = TopoGroupDynamicMaskConv2dContextModel(in_channels=C, out_channels=2 * C)
        coder = GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder(**kw).eval()
        # weights by RECIPE (kept out of the fixture): torch.manual_seed(100+i), then for every parameter in
        # named_parameters() order  p = randn(shape) * (0.05 if p.dim() > 1 else 0.02)
        pnames, pshapes = [], []
        torch.manual_seed(100 + i)  # re-seed AFTER construction (module init consumes the generator)
        with torch.no_grad():
            for name, p in coder.named_parameters():
                p.copy_(torch.randn(p.shape) * (0.05 if p.dim() > 1 else 0.02))
                pnames.append(name)
                pshapes.append(",".join(str(d) for d in p.shape))
        coder.update_state()
        gen = torch.Generator().manual_seed(200 + i)
        y = torch.randn(B, C, H, W, generator=gen) * 3
        prior = torch.cat([torch.randn(B, C, H, W, generator=gen), torch.rand(B, C, H, W, generator=gen) * 4 + 0.2], 1)
        # interleave so that channel 2c = mean, 2c+1 = scale ("split_interleave")
        prior = prior.reshape(B, 2, C, H, W).transpose(1, 2).reshape(B, 2 * C, H, W).contiguous()
        captured = {}
        orig = coder.ans_encoder

        class Spy:
            def encode_with_indexes(self, data, indexes, **k):
                captured["symbols"], captured["indexes"] = np.array(data), np.array(indexes)
                return orig.encode_with_indexes(data, indexes, **k)
        coder.ans_encoder = Spy()
        with torch.no_grad():
            data = coder.encode(y, prior=prior)
            yhat = coder.decode(data, prior=prior)
            yfwd = coder(y, prior=prior)
        k = f"a{i}"
        out[f"{k}.pnames"] = np.array(pnames)
        out[f"{k}.pshapes"] = np.array(pshapes)
        out[f"{k}.wsum"] = np.array([float(sum(p.double().sum() for p in coder.parameters()))])  # recipe checksum
        out.update({f"{k}.y": y.numpy(), f"{k}.prior": prior.numpy(), f"{k}.bytes": b2a(data), f"{k}.symbols": captured["symbols"].astype(np.int32),
                    f"{k}.indexes": captured["indexes"].astype(np.int32), f"{k}.yhat": yhat.numpy(), f"{k}.yfwd": yfwd.numpy(),
                    f"{k}.cfg": np.array([C, G, int(expand), int(ctxm), B, H, W]), f"{k}.method": np.array(method)})
        keys.append(k)
        print(f"  {k} {method}: {len(data)} bytes, max|yhat-y| {float((yhat - y).abs().max()):.3f}")
    out["keys"] = np.array(keys)
    save("ar_coder.npz", **out)


# ---------------------------------------------------------------- 6. framing
def framing():
    segs = [b"abc", b"", b"\x00\x01\x02\x03\x04", b"z" * 300]
    out = {"segs": np.array([len(s) for s in segs]), "raw": b2a(b"".join(segs))}
    out["merged_all"] = b2a(merge_bytes(segs))
    out["merged_n4"] = b2a(merge_bytes(segs, num_segments=4))
    out["merged_2"] = b2a(merge_bytes(segs[:2], num_segments=2))
    assert split_merged_bytes(merge_bytes(segs, num_segments=4), num_segments=4) == segs
    out["shape_bytes"] = b2a(encode_shape((1, 192, 16, 16)))
    save("framing.npz", **out)


# ---------------------------------------------------------------- 7. benchmark harness: BD metrics, metrics.csv
def harness():
    import csv, io, tempfile
    from cbench.benchmark.metrics.bj_delta import bj_delta
    rng = np.random.default_rng(11)
    out = {}
    for case in range(4):
        n1, n2 = 4 + case, 5 + (case % 2)
        r1 = np.sort(rng.uniform(0.1, 2.0, n1)); p1 = 28 + 6 * np.log(r1 + 1) + rng.normal(0, 0.05, n1)
        r2 = np.sort(rng.uniform(0.15, 2.2, n2)); p2 = 27.5 + 6.3 * np.log(r2 + 1) + rng.normal(0, 0.05, n2)
        out[f"c{case}_r1"], out[f"c{case}_p1"], out[f"c{case}_r2"], out[f"c{case}_p2"] = r1, p1, r2, p2
        out[f"c{case}_bd_psnr"] = np.float64(bj_delta(r1, p1, r2, p2, mode=0))
        out[f"c{case}_bd_rate"] = np.float64(bj_delta(r1, p1, r2, p2, mode=1))
    # save_metrics (cbench/benchmark/base.py:54-112) on a 2-row metric list with names: the CSV text is the fixture
    from cbench.benchmark.base import BaseBenchmark
    class _B(BaseBenchmark):
        def __init__(self, d):
            self.output_dir = d
            import logging
            self.logger = logging.getLogger("golden")
        def run_benchmark(self, *a, **k):
            pass
        def collect_metrics(self, *a, **k):
            return None
    with tempfile.TemporaryDirectory() as d:
        b = _B(d)
        rows = [dict(compression_ratio=0.0125, compressed_length=9830.5, psnr=31.25), dict(compression_ratio=0.02, compressed_length=15728.0, psnr=33.5, FLOPs=1.5e9)]
        try:
            b.save_metrics(metric_file=os.path.join(d, "metrics_2d.csv"), metric_data=rows, names=["sclevel0_vrlevel0", "sclevel1_vrlevel0"])
            out["csv_2d"] = b2a(open(os.path.join(d, "metrics_2d.csv"), "rb").read())
        except Exception as e:  # the writer needs helpers this import environment may not provide
            print("save_metrics not runnable here:", repr(e))
    save("harness.npz", **out)

# ---------------------------------------------------------------- 8. greedy complexity-level search (latent_graph.py:1397-1640)
def complexity_search():
    """The reference's own post_training_process run on synthetic (FLOPs, loss) tables: the dataset evaluation
    (_test_dataset_complexity_performance) is replaced by a table lookup, everything else (target FLOPs, candidate order,
    tie breaking, level insertion, metric cache) is the reference's code.  Fixture = tables + selected levels + metric cache."""
    import itertools
    import logging
    from cbench.modules.entropy_coder.latent_graph import LatentGraphicalANSEntropyCoder
    from cbench.nn.layers.param_generator import IndexSelectParameterGeneratorWrapper, NNParameterGenerator
    logging.disable(logging.CRITICAL)

    def node(n):
        return IndexSelectParameterGeneratorWrapper(
            batched_generator=NNParameterGenerator(shape=(n, 1, 1, n), init_method="value",
                                                   init_value=torch.eye(n).flip(-1).unsqueeze(1).unsqueeze(1), fix_params=True),
            fix_for_inference=True)

    cases = [  # (controller sizes, num_levels, custom constraint fractions of [min,max], table seed, loss noise)
        (dict(pa=3, pb=4, pc=2), 5, None, 5, 0.5),
        (dict(pgmxy=5, pgmyz=5, pgmzy=5, pgmyx=5), 8, None, 1, 0.05),
        (dict(pgmxy=5, pgmyz=5, pgmzy=5, pgmyx=5), 8, None, 2, 3.0),
        (dict(pa=4, pb=3), 4, None, 3, 0.0),           # ties in loss: the later candidate wins
        (dict(pa=3, pb=3, pc=3), None, [0.9, 0.55, 0.3, 0.05], 4, 0.2),  # custom FLOPs constraints
    ]
    out = {"ncases": np.int64(len(cases))}
    for ci, (sizes, nlev, custom, seed, noise) in enumerate(cases):
        names, shape = list(sizes), tuple(sizes.values())
        rng = np.random.default_rng(seed)
        w = rng.uniform(1, 12, len(shape)).round(1)
        flops, loss = np.zeros(shape), np.zeros(shape)
        for t in itertools.product(*[range(n) for n in shape]):
            c = float(sum(w[k] * (shape[k] - 1 - t[k]) for k in range(len(shape))))
            flops[t] = 10 + c
            loss[t] = 100.0 / (1 + c) + (rng.uniform(0, noise) if noise > 0 else 0.0)
        if noise == 0.0:
            loss = np.round(loss, 0)  # many exact ties
        tmax, tmin = tuple(0 for _ in shape), tuple(n - 1 for n in shape)
        loss[tmax], loss[tmin] = loss.min() - 1, loss.max() + 1
        kw = dict(complexity_level_greedy_search_num_levels=nlev)
        if custom is not None:
            lo, hi = flops[tmin], flops[tmax]
            kw = dict(complexity_level_greedy_search_custom_constraint=[float(lo + f * (hi - lo)) for f in custom])
        ec = LatentGraphicalANSEntropyCoder(
            node_generator_dict={k: node(n) for k, n in sizes.items()},
            latent_node_inference_topo_order=["x"], latent_node_generative_topo_order=["x"],
            complexity_level_greedy_search=True, complexity_level_greedy_search_dataset=[],
            complexity_level_greedy_search_dataset_cached=True, complexity_level_controller_nodes=names, **kw)

        def fake(dataset, *a, performance_method=None, complexity_method=None, **params):
            t = tuple(int(sizes[k] - 1 - params[k].reshape(-1).argmax().item()) for k in names)
            return float(flops[t]), float(loss[t])
        ec._test_dataset_complexity_performance = fake
        ec.update_state = lambda *a, **k: None
        ec.post_training_process()
        levels = []
        for m in ec._complexity_param_all_levels:
            d = m()
            levels.append([int(sizes[k] - 1 - d[k].reshape(-1).argmax().item()) for k in names])
        out[f"c{ci}_names"] = np.array(names)
        out[f"c{ci}_sizes"] = np.array(shape)
        out[f"c{ci}_num_levels"] = np.int64(-1 if nlev is None else nlev)
        out[f"c{ci}_constraint"] = np.array(kw.get("complexity_level_greedy_search_custom_constraint", []), dtype=np.float64)
        out[f"c{ci}_flops"], out[f"c{ci}_loss"] = flops, loss
        out[f"c{ci}_levels"] = np.array(levels, dtype=np.int64)
        out[f"c{ci}_metric_names"] = np.array(list(ec.complexity_metric_list))
        out[f"c{ci}_metric_cache"] = ec._complexity_metric_list_cache.double().numpy()
        print(ci, levels)
    # FLOP counters of the reference's slimmable transforms (slimmable_layers.py:186-206,284-293 through
    # pgm_layers.py:781-845): total_ops per width level on small zero inputs
    from cbench.nn.layers import pgm_layers as P
    W = [48, 72, 96, 144, 192]
    mods = dict(
        g_a=(P.HyperpriorAnalysisSlimmableConv2dPGMModel(in_channels=3, out_channels=192, mid_channels_list=W), (2, 3, 64, 64)),
        h_a=(P.MeanScaleHyperpriorHyperAnalysisSlimmableConv2dPGMModel(in_channels=192, out_channels=192, mid_channels_list=W), (2, 192, 4, 4)),
        h_s=(P.MeanScaleHyperpriorHyperSynthesisSlimmableConv2dPGMModel(in_channels=192, out_channels=384, mid_channels_list=W), (2, 192, 1, 1)),
        g_s=(P.HyperpriorSynthesisSlimmableConv2dPGMModel(in_channels=192, out_channels=3, mid_channels_list=W), (2, 192, 4, 4)))
    out["ops_widths"] = np.array(W)
    for k, (m, shape) in mods.items():
        m.eval()
        row = []
        for lvl in range(len(W)):
            with torch.no_grad():
                m(torch.zeros(shape), pgm=torch.eye(len(W))[lvl].reshape(1, 1, 1, len(W)))
            row.append(float(m.get_current_flops()))
        out[f"ops_{k}_shape"], out[f"ops_{k}"] = np.array(shape), np.array(row, dtype=np.float64)
    logging.disable(logging.NOTSET)
    save("complexity_search.npz", **out)

# ---------------------------------------------------------------- 9. AR coder with supplied / learned topo groups, combined coder
def ar_coder_pgm():
    """encode(..., pgm=...) with integer maps and logits (tiled / trimmed to the latent, pgm_coder.py:1340-1380), the
    eval path of a topo_group_predictor (its cached output, :1551) and CombinedNNTrainablePGMPriorCoder (:632-715)."""
    from cbench.modules.prior_model.prior_coder.pgm_coder import CombinedNNTrainablePGMPriorCoder

    class FixedPredictor(torch.nn.Module):
        def __init__(self, value):
            super().__init__()
            self.register_buffer("value", value)

        def forward(self, *a, **k):
            return self.value

    def seeded(coder, seed):
        names, shapes = [], []
        torch.manual_seed(seed)
        with torch.no_grad():
            for name, p in coder.named_parameters():
                p.copy_(torch.randn(p.shape) * (0.05 if p.dim() > 1 else 0.02))
                names.append(name)
                shapes.append(",".join(str(d) for d in p.shape))
        return names, shapes

    def inputs(seed, B, C, H, W):
        gen = torch.Generator().manual_seed(seed)
        y = torch.randn(B, C, H, W, generator=gen) * 3
        prior = torch.cat([torch.randn(B, C, H, W, generator=gen), torch.rand(B, C, H, W, generator=gen) * 4 + 0.2], 1)
        return y, prior.reshape(B, 2, C, H, W).transpose(1, 2).reshape(B, 2 * C, H, W).contiguous()

    def spy(coder, captured):
        orig = coder.ans_encoder

        class Spy:
            def encode_with_indexes(self, data, indexes, **k):
                captured["symbols"], captured["indexes"] = np.array(data), np.array(indexes)
                return orig.encode_with_indexes(data, indexes, **k)
        coder.ans_encoder = Spy()

    g = torch.Generator().manual_seed(77)
    cases = [  # (C, G, ctx model, (B,H,W), pgm tensor or None, predictor tensor or None)
        (16, 2, False, (1, 6, 6), torch.randint(0, 4, (1, 2, 2, 2), generator=g), None),          # int patch, tiled
        (16, 1, False, (1, 5, 7), torch.randn(1, 4, 2, 2, generator=g), None),                    # logits, ragged tiling
        (16, 1, True, (1, 4, 6), torch.randint(0, 6, (1, 1, 8, 8), generator=g), None),           # larger map, trimmed
        (16, 2, False, (2, 4, 4), None, torch.randn(1, 2 * 3, 2, 2, generator=g)),                # predictor cache, B=2
        (16, 4, False, (1, 4, 8), None, torch.randint(0, 8, (1, 4, 2, 2), generator=g)),          # integer predictor cache
    ]
    # per-sample topo groups: a pgm with the BATCH's leading dimension (pgm_coder.py:1340-1380 keep it; the masks of a group then
    # differ from image to image, :885-890, and the masked convolutions take per-sample maps, masked_conv.py:119-173) -- full-size
    # integer maps, and logits patches that are tiled, with a context model.  (Own generator: the draws above stay as they were.)
    g2 = torch.Generator().manual_seed(78)
    cases += [
        (16, 2, False, (2, 4, 6), torch.stack([torch.randint(0, 5, (2, 4, 6), generator=g2), torch.randint(0, 3, (2, 4, 6), generator=g2)]), None),
        (16, 1, True, (3, 4, 4), torch.randn(3, 1 * 3, 2, 2, generator=g2), None),
    ]
    out, keys = {}, []
    for i, (C, G, ctxm, (B, H, W), pgm, pred) in enumerate(cases):
        kw = dict(in_channels=C, channel_groups=G)
        if ctxm:
            kw["topo_group_context_model"] = TopoGroupDynamicMaskConv2dContextModel(in_channels=C, out_channels=2 * C)
        if pred is not None:
            kw["topo_group_predictor"] = FixedPredictor(pred)
        coder = GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder(**kw).eval()
        names, shapes = seeded(coder, 300 + i)
        coder.update_state()
        y, prior = inputs(400 + i, B, C, H, W)
        cap = {}
        spy(coder, cap)
        with torch.no_grad():
            data = coder.encode(y, prior=prior, pgm=pgm)
            yhat = coder.decode(data, prior=prior, pgm=pgm)
        k = f"p{i}"
        src = pgm if pgm is not None else pred
        out.update({f"{k}.pnames": np.array(names), f"{k}.pshapes": np.array(shapes),
                    f"{k}.wsum": np.array([float(sum(p.double().sum() for p in coder.parameters()))]),
                    f"{k}.y": y.numpy(), f"{k}.prior": prior.numpy(), f"{k}.bytes": b2a(data),
                    f"{k}.symbols": cap["symbols"].astype(np.int32), f"{k}.indexes": cap["indexes"].astype(np.int32),
                    f"{k}.yhat": yhat.numpy(), f"{k}.cfg": np.array([C, G, int(ctxm), B, H, W, int(pgm is None)]),
                    f"{k}.pgm": src.numpy()})
        keys.append(k)
        print(f"  {k}: {len(data)} bytes, groups {int(cap['indexes'].size)}, max|yhat-y| {float((yhat - y).abs().max()):.3f}")
    out["keys"] = np.array(keys)

    # combined coder: sub-coder 0 = scanline + context model, sub-coder 1 = G=2 with an integer predictor cache
    C, B, H, W = 16, 1, 4, 4
    pred = torch.randint(0, 4, (1, 2, 2, 2), generator=g)
    subs = [GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder(
                in_channels=C, default_topo_group_method="scanline",
                topo_group_context_model=TopoGroupDynamicMaskConv2dContextModel(in_channels=C, out_channels=2 * C)),
            GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder(in_channels=C, channel_groups=2, topo_group_predictor=FixedPredictor(pred))]
    comb = CombinedNNTrainablePGMPriorCoder(subs).eval()
    names, shapes = seeded(comb, 390)
    comb.update_state()
    y, prior = inputs(490, B, C, H, W)
    out.update({"comb.pnames": np.array(names), "comb.pshapes": np.array(shapes),
                "comb.wsum": np.array([float(sum(p.double().sum() for p in comb.parameters()))]),
                "comb.y": y.numpy(), "comb.prior": prior.numpy(), "comb.pred": pred.numpy()})
    for sel in (0, 1):
        bw = torch.eye(2)[sel]
        with torch.no_grad():
            data = comb.encode(y, prior=prior, blend_weight=bw)
            yhat = comb.decode(data, prior=prior, blend_weight=bw)
        out[f"comb.bytes{sel}"], out[f"comb.yhat{sel}"] = b2a(data), yhat.numpy()
        print(f"  combined sel {sel}: {len(data)} bytes")
    save("ar_coder_pgm.npz", **out)

# ---------------------------------------------------------------- 10. joint-AR raster implementation (use_joint_ar_model_impl)
def ar_coder_joint():
    """pgm_coder.py:1975-2070: raster-scan coding with the plain 1x1 entropy_parameters network on cat(prior, ctx) and
    "chunk" parameters (scales first, then means)."""
    out, keys = {}, []
    for i, (C, B, H, W) in enumerate([(16, 1, 4, 5), (16, 2, 3, 3), (32, 1, 2, 6)]):
        coder = GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder(in_channels=C, use_joint_ar_model_impl=True).eval()
        names, shapes = [], []
        torch.manual_seed(500 + i)
        with torch.no_grad():
            for name, p in coder.named_parameters():
                p.copy_(torch.randn(p.shape) * (0.05 if p.dim() > 1 else 0.02))
                names.append(name)
                shapes.append(",".join(str(d) for d in p.shape))
        coder.update_state()
        gen = torch.Generator().manual_seed(600 + i)
        y = torch.randn(B, C, H, W, generator=gen) * 3
        # "chunk" layout with inverse_mean_scale: first C channels scales, last C channels means; the prior feeds the
        # entropy_parameters network, so any values do -- kept positive-ish in the scale half for realism
        prior = torch.cat([torch.rand(B, C, H, W, generator=gen) * 4 + 0.2, torch.randn(B, C, H, W, generator=gen)], 1)
        captured = {}
        orig = coder.ans_encoder

        class Spy:
            def encode_with_indexes(self, data, indexes, **k):
                captured["symbols"], captured["indexes"] = np.array(data), np.array(indexes)
                return orig.encode_with_indexes(data, indexes, **k)
        coder.ans_encoder = Spy()
        with torch.no_grad():
            data = coder.encode(y, prior=prior)
            yhat = coder.decode(data, prior=prior)
        k = f"j{i}"
        out.update({f"{k}.pnames": np.array(names), f"{k}.pshapes": np.array(shapes),
                    f"{k}.wsum": np.array([float(sum(p.detach().double().sum() for p in coder.parameters()))]),
                    f"{k}.y": y.numpy(), f"{k}.prior": prior.numpy(), f"{k}.bytes": b2a(data),
                    f"{k}.symbols": captured["symbols"].astype(np.int32).reshape(-1), f"{k}.indexes": captured["indexes"].astype(np.int32).reshape(-1),
                    f"{k}.yhat": yhat.numpy(), f"{k}.cfg": np.array([C, B, H, W])})
        keys.append(k)
        print(f"  {k}: {len(data)} bytes, max|yhat-y| {float((yhat - y).abs().max()):.3f}, params {names}")
    out["keys"] = np.array(keys)
    save("ar_coder_joint.npz", **out)


# ---------------------------------------------------------------- 11. whole codec graphs (GeneralCodec + LatentGraphicalANSEntropyCoder)
from recipe import named_seed_weights, recipe_input  # noqa: E402  (tests/golden/recipe.py)


def _basic_graph(y_extra=None):
    """The tiny BaSIC slimmable graph of the codec_graph fixture (presets/lossy_latent_graph_scalable_ar_models.py:73-197):
    (codec, entropy coder, levels, controller names, touched parameter list, calibration, widths, M)."""
    from cbench.codecs.general_codec import GeneralCodec
    from cbench.modules.entropy_coder.latent_graph import LatentGraphicalANSEntropyCoder
    from cbench.modules.prior_model.prior_coder.compressai_coder import CompressAIEntropyBottleneckPriorCoder
    from cbench.nn.layers import pgm_layers as P
    from cbench.nn.layers.param_generator import IndexSelectParameterGeneratorWrapper, NNParameterGenerator
    Wd, M = [4, 6, 8, 12, 16], 16
    n = len(Wd)

    def slim_node():
        return IndexSelectParameterGeneratorWrapper(
            batched_generator=NNParameterGenerator(shape=(n, 1, 1, n), init_method="value",
                                                   init_value=torch.eye(n).flip(-1).unsqueeze(1).unsqueeze(1), fix_params=True),
            fix_for_inference=True)
    ctl = ["pgmxy", "pgmyz", "pgmzy", "pgmyx"]
    levels = [dict(zip(ctl, t)) for t in [(0, 0, 0, 0), (1, 1, 1, 1), (2, 2, 2, 2), (3, 3, 3, 3), (4, 4, 4, 4), (0, 2, 1, 4), (3, 0, 4, 1), (2, 4, 0, 3)]]
    ec = LatentGraphicalANSEntropyCoder(
        node_generator_dict={c: slim_node() for c in ["pgmxy", "pgmyx", "pgmyz", "pgmzy"]},
        use_lossy_compression=True, lossy_compression_lambda_rd=145.2225,
        latent_node_inference_topo_order=["x", "y", "z"], latent_node_generative_topo_order=["z", "y", "x"],
        latent_node_entropy_coder_dict=dict(
            y=GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder(
                in_channels=M, default_topo_group_method="scanline",
                topo_group_context_model=TopoGroupDynamicMaskConv2dContextModel(in_channels=M, out_channels=2 * M), **(y_extra or {})),
            z=CompressAIEntropyBottleneckPriorCoder(entropy_bottleneck_channels=M, use_inner_aux_opt=True)),
        latent_inference_dict=dict(
            x_y=P.HyperpriorAnalysisSlimmableConv2dPGMModel(in_channels=3, out_channels=M, mid_channels_list=Wd),
            y_z=P.MeanScaleHyperpriorHyperAnalysisSlimmableConv2dPGMModel(in_channels=M, out_channels=M, mid_channels_list=Wd)),
        latent_generative_dict=dict(
            z_y=P.MeanScaleHyperpriorHyperSynthesisSlimmableConv2dPGMModel(in_channels=M, out_channels=2 * M, mid_channels_list=Wd),
            y_x=P.HyperpriorSynthesisSlimmableConv2dPGMModel(in_channels=M, out_channels=3, mid_channels_list=Wd)),
        latent_inference_input_mapping=dict(x_y={"pgmxy": "pgm"}, y_z={"pgmyz": "pgm"}),
        latent_generative_input_mapping=dict(y_x={"pgmyx": "pgm"}, z_y={"pgmzy": "pgm"}, y={"z": "prior"}),
        complexity_level_greedy_search=True, complexity_level_greedy_search_custom_params=levels,
        complexity_level_greedy_search_custom_constraint=[float(i) for i in range(len(levels))],
        complexity_level_controller_nodes=ctl)
    codec = GeneralCodec(entropy_coder=ec).eval()
    def last_conv(mod, prefix, first=False):
        names = [nm for nm, p_ in mod.named_parameters() if p_.dim() == 4]
        return prefix + (names[0] if first else names[-1])
    pre, gen = "entropy_coder.latent_inference_modules.", "entropy_coder.latent_generative_modules."
    w_hs = last_conv(ec.latent_generative_modules["z_y"], gen + "z_y.")
    calib = [(last_conv(ec.latent_inference_modules["x_y"], pre + "x_y."), 8.0, 0.0),
             ("entropy_coder.latent_node_entropy_coders.y.topo_group_context_model.param_merger_out.3.bias", 1.0, 1.2),
             (last_conv(ec.latent_inference_modules["y_z"], pre + "y_z."), 6.0, 0.0),
             (w_hs[:-len("weight")] + "bias", 1.0, 1.2),
             (last_conv(ec.latent_generative_modules["y_x"], gen + "y_x.", first=True), 0.05, 0.0)]
    touched = named_seed_weights(codec, 790, calib)
    with torch.no_grad():
        ec._complexity_param_valid.fill_(True)   # as after post_training_process / a loaded checkpoint (latent_graph.py:673-675)
    codec.update_state()
    return codec, ec, levels, ctl, touched, calib, Wd, M


def codec_graph():
    """The reference's GeneralCodec(entropy_coder=LatentGraphicalANSEntropyCoder(...)) (general_codec.py:44-130,
    latent_graph.py:306,1232-1295) run end to end on CPU at tiny channel counts:
      t0  topo-group graph, checkerboard + expand-bottleneck merger   (configs/lossy_latent_graph_topogroup.py:203-244)
      t1  topo-group graph, channel-wise G=2
      t2  topo-group graph, scanline, batch of 2 (one stream for the batch: the reference's layout)
      h0-h2  plain hyperprior graph (configs/lossy_graph_scalable_exp_hp.py:182-215): batch 1, batch 3, and a batch of 2 whose
          size is not a multiple of 64 (prior crop); also the z / y (symbols, indexes) of every native encode call and the
          GaussianConditional table buffers
      b0  BaSIC slimmable graph (presets/lossy_latent_graph_scalable_ar_models.py:73-197), widths [4,6,8,12,16], eight
          complexity levels = eight fixed controller index tuples (complexity_level_greedy_search_custom_params), every
          uniform width plus three mixed ones
    Recorded per case / level: compress() bytes, the y-coder's (symbols, indexes) as handed to the native encoder, the
    latents y and z, x-hat of decompress(), the forward() metrics (prior_entropy, estimated_bpd, mse, per-node entropies), and
    the complete state_dict key / shape list.  compressai's EntropyBottleneck / GDN / conv arithmetic underneath is
    oracle/compressai_restated.py (compressai is absent here): these fixtures pin the reference's OWN graph code, not that."""
    import logging
    from cbench.codecs.general_codec import GeneralCodec
    from cbench.modules.entropy_coder.latent_graph import LatentGraphicalANSEntropyCoder, LossyDummyEntropyCoder
    from cbench.modules.prior_model.prior_coder.compressai_coder import CompressAIEntropyBottleneckPriorCoder
    from cbench.nn.models.google import (HyperpriorAnalysisModel, HyperpriorSynthesisModel, HyperpriorHyperAnalysisModel,
                                         HyperpriorHyperSynthesisModel)
    from cbench.nn.layers import pgm_layers as P
    from cbench.nn.layers.param_generator import IndexSelectParameterGeneratorWrapper, NNParameterGenerator
    logging.disable(logging.CRITICAL)
    out, keys = {}, []

    def spy_y(coder, log):
        orig = coder.ans_encoder

        class Spy:
            def encode_with_indexes(self, data, indexes, **k):
                log.append((np.array(data).astype(np.int32).reshape(-1), np.array(indexes).astype(np.int32).reshape(-1)))
                return orig.encode_with_indexes(data, indexes, **k)
        coder.ans_encoder = Spy()

    def record(k, codec, x, log):
        ec = codec.entropy_coder
        del log[:]
        with torch.no_grad():
            data = codec.compress(x)
            sym, idx = log[-1]
            xhat = codec.decompress(data)
            node = ec._node_generate_process(**ec._get_default_node_dict(force_add_default_dynamic_nodes=True))
            lat = ec._inference_process({"x": x, **node})
            codec.reset_all_cache()
            xfwd = codec(x)
            met = {n.split("metric_dict/entropy_coder/")[-1]: float(v) for n, v in codec.get_cache("metric_dict").items()}
        out.update({f"{k}.bytes": b2a(data), f"{k}.symbols": sym, f"{k}.indexes": idx, f"{k}.xhat": xhat.numpy(),
                    f"{k}.y": lat["y"].numpy(), f"{k}.z": lat["z"].numpy(), f"{k}.xfwd_minus_xhat_max": np.float64((xfwd - xhat).abs().max()),
                    f"{k}.metric_names": np.array(list(met)), f"{k}.metric_values": np.array(list(met.values()), np.float64)})
        print(f"  {k}: {len(data)} bytes, {sym.size} y symbols, mse {met.get('mse', float('nan')):.4f}, y std {float(lat['y'].std()):.2f}, "
              f"prior_entropy {met.get('prior_entropy', float('nan')):.2f}")

    def finish(k, codec, touched, seed, xseed, calib):
        sd = codec.state_dict()
        out.update({f"{k}.xseed": np.int64(xseed), f"{k}.seed": np.int64(seed),
                    f"{k}.calib_names": np.array([c[0] for c in calib]), f"{k}.calib_mul": np.array([c[1] for c in calib]),
                    f"{k}.calib_add_odd": np.array([c[2] for c in calib]),
                    f"{k}.pnames": np.array([n for n, _ in touched]), f"{k}.pshapes": np.array([s for _, s in touched]),
                    f"{k}.wsum": np.array([float(sum(codec.state_dict()[n].double().sum() for n, _ in touched))]),
                    f"{k}.sd_keys": np.array(list(sd)), f"{k}.sd_shapes": np.array([",".join(str(d) for d in v.shape) for v in sd.values()])})
        keys.append(k)

    # ---- topo-group graphs
    N, M = 8, 16
    for ci, (method, G, expand, ctxm, (B, H, W)) in enumerate([("checkerboard", 1, True, False, (1, 64, 128)), ("channelwise", 2, False, False, (1, 64, 64)),
                                                                ("scanline", 1, False, False, (2, 64, 64))]):
        k = f"t{ci}"
        ykw = dict(in_channels=M, channel_groups=G, default_topo_group_method=method, param_merger_expand_bottleneck=expand)
        ec = LatentGraphicalANSEntropyCoder(
            latent_node_inference_topo_order=["x", "y", "z"], latent_node_generative_topo_order=["z", "y", "x"],
            latent_node_entropy_coder_dict=dict(x=LossyDummyEntropyCoder(lambda_rd=145.2225),
                                                y=GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder(**ykw),
                                                z=CompressAIEntropyBottleneckPriorCoder(entropy_bottleneck_channels=N, use_inner_aux_opt=True)),
            latent_inference_dict=dict(x_y=HyperpriorAnalysisModel(N=N, M=M), y_z=HyperpriorHyperAnalysisModel(N=N, M=M)),
            latent_generative_dict=dict(z_y=HyperpriorHyperSynthesisModel(N=N, M=2 * M), y_x=HyperpriorSynthesisModel(N=N, M=M)))
        codec = GeneralCodec(entropy_coder=ec).eval()
        seed = 700 + ci
        pre = "entropy_coder.latent_inference_modules."
        gen = "entropy_coder.latent_generative_modules."
        # spread the latents over the scale table (y std ~ 2, predicted scales ~ 1) and keep x-hat O(1)
        calib = [(pre + "x_y.model.6.weight", 8.0, 0.0), (pre + "y_z.model.4.weight", 6.0, 0.0),
                 (gen + "z_y.model.4.bias", 1.0, 1.2), (gen + "y_x.model.0.weight", 0.05, 0.0),
                 ("entropy_coder.latent_node_entropy_coders.y.param_merger.4.bias", 1.0, 1.2)]
        touched = named_seed_weights(codec, seed, calib)
        codec.update_state()
        log = []
        spy_y(ec.latent_node_entropy_coders["y"], log)
        x = recipe_input(800 + ci, (B, 3, H, W))
        record(k, codec, x, log)
        out[f"{k}.cfg"] = np.array([N, M, G, int(expand), int(ctxm), B, H, W])
        out[f"{k}.method"] = np.array(method)
        finish(k, codec, touched, seed, 800 + ci, calib)

    # ---- plain hyperprior graph (configs/lossy_graph_scalable_exp_hp.py:182-215): y = CompressAIGaussianConditionalCoder,
    #      z = CompressAIEntropyBottleneckPriorCoder, Hyperprior*Model transforms -- the graph bench.py and basic_hp_* run
    from cbench.modules.prior_model.prior_coder.compressai_coder import CompressAIGaussianConditionalCoder
    from oracle import compressai_restated as cr

    def spy_native(log):
        """Log every (symbols, indexes) the coders hand to compressai.ans (EntropyModel.compress, one call per batch item:
        first the z items, then the y items)."""
        orig = cr.RansEncoder.encode_with_indexes

        def spy(self, symbols, indexes, *a, **k):
            log.append((np.array(symbols, np.int32).reshape(-1), np.array(indexes, np.int32).reshape(-1)))
            return orig(self, symbols, indexes, *a, **k)
        cr.RansEncoder.encode_with_indexes = spy
        return lambda: setattr(cr.RansEncoder, "encode_with_indexes", orig)

    for ci, (B, H, W) in enumerate([(1, 64, 64), (3, 64, 96), (2, 72, 100)]):   # the last: not a multiple of 64 (prior crop path)
        k = f"h{ci}"
        ec = LatentGraphicalANSEntropyCoder(
            latent_node_inference_topo_order=["x", "y", "z"], latent_node_generative_topo_order=["z", "y", "x"],
            latent_node_entropy_coder_dict=dict(x=LossyDummyEntropyCoder(lambda_rd=145.2225),
                                                y=CompressAIGaussianConditionalCoder(),
                                                z=CompressAIEntropyBottleneckPriorCoder(entropy_bottleneck_channels=N, use_inner_aux_opt=True)),
            latent_inference_dict=dict(x_y=HyperpriorAnalysisModel(N=N, M=M), y_z=HyperpriorHyperAnalysisModel(N=N, M=M)),
            latent_generative_dict=dict(z_y=HyperpriorHyperSynthesisModel(N=N, M=M), y_x=HyperpriorSynthesisModel(N=N, M=M)))
        codec = GeneralCodec(entropy_coder=ec).eval()
        seed = 720 + ci
        pre = "entropy_coder.latent_inference_modules."
        gen = "entropy_coder.latent_generative_modules."
        # y std ~ 2; predicted scales (ReLU output of h_s) spread over the lower half of the scale table; x-hat O(1)
        calib = [(pre + "x_y.model.6.weight", 8.0, 0.0), (pre + "y_z.model.4.weight", 6.0, 0.0),
                 (gen + "z_y.model.4.bias", 1.0, 1.5), (gen + "y_x.model.0.weight", 0.05, 0.0)]
        touched = named_seed_weights(codec, seed, calib)
        codec.update_state()
        x = recipe_input(820 + ci, (B, 3, H, W))
        log = []
        undo = spy_native(log)
        with torch.no_grad():
            data = codec.compress(x)
        undo()
        assert len(log) == 2 * B
        zs, ys = log[:B], log[B:]
        with torch.no_grad():
            xhat = codec.decompress(data)
            node = ec._node_generate_process(**ec._get_default_node_dict(force_add_default_dynamic_nodes=True))
            lat = ec._inference_process({"x": x, **node})
            codec.reset_all_cache()
            xfwd = codec(x)
            met = {n.split("metric_dict/entropy_coder/")[-1]: float(v) for n, v in codec.get_cache("metric_dict").items()}
        gc = ec.latent_node_entropy_coders["y"].gaussian_conditional
        out.update({f"{k}.bytes": b2a(data), f"{k}.symbols": np.stack([s for s, _ in ys]), f"{k}.indexes": np.stack([i for _, i in ys]),
                    f"{k}.z_symbols": np.stack([s for s, _ in zs]), f"{k}.z_indexes": np.stack([i for _, i in zs]),
                    f"{k}.xhat": xhat.numpy(), f"{k}.y": lat["y"].numpy(), f"{k}.z": lat["z"].numpy(),
                    # decompress() returns the un-cropped synthesis output, forward() the input-sized one: compare the overlap
                    f"{k}.xfwd_minus_xhat_max": np.float64((xfwd - xhat[..., :xfwd.shape[-2], :xfwd.shape[-1]]).abs().max()),
                    f"{k}.xfwd_shape": np.array(xfwd.shape),
                    f"{k}.metric_names": np.array(list(met)), f"{k}.metric_values": np.array(list(met.values()), np.float64),
                    f"{k}.gc_cdf_sha256": np.array(hashlib.sha256(gc._quantized_cdf.numpy().astype(np.int32).tobytes()).hexdigest()),
                    f"{k}.gc_cdf_length": gc._cdf_length.numpy().astype(np.int32), f"{k}.gc_offset": gc._offset.numpy().astype(np.int32),
                    f"{k}.cfg": np.array([N, M, B, H, W])})
        print(f"  {k}: {len(data)} bytes, {ys[0][0].size} y symbols per image, mse {met.get('mse', float('nan')):.4f}, "
              f"y std {float(lat['y'].std()):.2f}, index range {int(out[f'{k}.indexes'].min())}..{int(out[f'{k}.indexes'].max())}, "
              f"prior_entropy {met.get('prior_entropy', float('nan')):.2f}")
        finish(k, codec, touched, seed, 820 + ci, calib)

    # ---- multi-edge aggregation in the inference pass (latent_graph.py:741-749): the hyper-latent z is the AVERAGE of two
    #      edges -- y_z(y) and w_z(w), with w a second analysis of x that is not coded itself -- folded by the reference's
    #      AverageNodeAggregatorModel; everything else as in the plain hyperprior graph
    from cbench.modules.entropy_coder.latent_graph import AverageNodeAggregatorModel
    agg_keys = []
    for ci, (B, H, W) in enumerate([(2, 64, 96)]):
        k = f"g{ci}"
        ec = LatentGraphicalANSEntropyCoder(
            latent_node_inference_topo_order=["x", "y", "w", "z"], latent_node_generative_topo_order=["z", "y", "x"],
            latent_node_entropy_coder_dict=dict(x=LossyDummyEntropyCoder(lambda_rd=145.2225),
                                                y=CompressAIGaussianConditionalCoder(),
                                                z=CompressAIEntropyBottleneckPriorCoder(entropy_bottleneck_channels=N, use_inner_aux_opt=True)),
            latent_inference_dict=dict(x_y=HyperpriorAnalysisModel(N=N, M=M), x_w=HyperpriorAnalysisModel(N=N, M=M),
                                       y_z=HyperpriorHyperAnalysisModel(N=N, M=M), w_z=HyperpriorHyperAnalysisModel(N=N, M=M)),
            latent_generative_dict=dict(z_y=HyperpriorHyperSynthesisModel(N=N, M=M), y_x=HyperpriorSynthesisModel(N=N, M=M)),
            latent_inference_node_aggregator_dict=dict(z=AverageNodeAggregatorModel()))
        codec = GeneralCodec(entropy_coder=ec).eval()
        seed = 740 + ci
        pre = "entropy_coder.latent_inference_modules."
        gen = "entropy_coder.latent_generative_modules."
        calib = [(pre + "x_y.model.6.weight", 8.0, 0.0), (pre + "x_w.model.6.weight", 8.0, 0.0), (pre + "y_z.model.4.weight", 6.0, 0.0),
                 (pre + "w_z.model.4.weight", 6.0, 0.0), (gen + "z_y.model.4.bias", 1.0, 1.5), (gen + "y_x.model.0.weight", 0.05, 0.0)]
        touched = named_seed_weights(codec, seed, calib)
        codec.update_state()
        x = recipe_input(840 + ci, (B, 3, H, W))
        log = []
        undo = spy_native(log)
        with torch.no_grad():
            data = codec.compress(x)
        undo()
        assert len(log) == 2 * B
        zs, ys = log[:B], log[B:]
        with torch.no_grad():
            xhat = codec.decompress(data)
            node = ec._node_generate_process(**ec._get_default_node_dict(force_add_default_dynamic_nodes=True))
            lat = ec._inference_process({"x": x, **node})
            codec.reset_all_cache()
            xfwd = codec(x)
            met = {n.split("metric_dict/entropy_coder/")[-1]: float(v) for n, v in codec.get_cache("metric_dict").items()}
        assert not isinstance(lat["z"], list)
        out.update({f"{k}.bytes": b2a(data), f"{k}.symbols": np.stack([s_ for s_, _ in ys]), f"{k}.indexes": np.stack([i for _, i in ys]),
                    f"{k}.z_symbols": np.stack([s_ for s_, _ in zs]), f"{k}.z_indexes": np.stack([i for _, i in zs]),
                    f"{k}.xhat": xhat.numpy(), f"{k}.y": lat["y"].numpy(), f"{k}.w": lat["w"].numpy(), f"{k}.z": lat["z"].numpy(),
                    f"{k}.xfwd_minus_xhat_max": np.float64((xfwd - xhat[..., :xfwd.shape[-2], :xfwd.shape[-1]]).abs().max()),
                    f"{k}.xfwd_shape": np.array(xfwd.shape),
                    f"{k}.metric_names": np.array(list(met)), f"{k}.metric_values": np.array(list(met.values()), np.float64),
                    f"{k}.cfg": np.array([N, M, B, H, W])})
        print(f"  {k}: {len(data)} bytes (z = mean of two edges), mse {met.get('mse', float('nan')):.4f}, z std {float(lat['z'].std()):.2f}, "
              f"prior_entropy {met.get('prior_entropy', float('nan')):.2f}")
        finish(k, codec, touched, seed, 840 + ci, calib)
        keys.remove(k)          # (its own list: the oracle-driven tests walk `keys`)
        agg_keys.append(k)
    out["agg_keys"] = np.array(agg_keys)

    # ---- BaSIC slimmable graph
    codec, ec, levels, ctl, touched, calib, Wd, M = _basic_graph()
    log = []
    spy_y(ec.latent_node_entropy_coders["y"], log)
    x = recipe_input(890, (1, 3, 64, 64))
    for li in range(len(levels)):
        codec.set_complex_level(li)
        record(f"b0.l{li}", codec, x, log)
        try:
            out[f"b0.l{li}.flops"] = np.float64(sum(float(m.get_current_flops()) for m in list(ec.latent_inference_modules.values()) + list(ec.latent_generative_modules.values())))
        except Exception as e:
            print("   flops unavailable:", repr(e))
    out["b0.levels"] = np.array([[lv[c] for c in ctl] for lv in levels])
    out["b0.controllers"] = np.array(ctl)
    out["b0.cfg"] = np.array([M, 1, 64, 64] + Wd)
    finish("b0", codec, touched, 790, 890, calib)
    out["keys"] = np.array(keys)
    logging.disable(logging.NOTSET)
    save("codec_graph.npz", **out)


# ---------------------------------------------------------------- 12. GroupedVariableRateCodec (codecs/base.py:138-243)
from recipe import GROUPED_OPS, GROUPED_VR_CONFIG, run_grouped_ops  # noqa: E402


def grouped():
    from cbench.codecs.base import (GroupedVariableRateCodec, NNTrainableCodec, VariableComplexityCodecInterface,
                                    VariableRateCodecInterface, VariableTaskCodecInterface)

    def make_member(i, log, rate=0, complex=0, tasks=0):
        bases = [NNTrainableCodec] + ([VariableRateCodecInterface] if rate else []) + \
                ([VariableComplexityCodecInterface] if complex else []) + ([VariableTaskCodecInterface] if tasks else [])
        ns = dict(
            compress=lambda self, data, *a, **k: (log.append((i, "compress", data)), f"bytes{i}:{data}")[1],
            decompress=lambda self, data, *a, **k: (log.append((i, "decompress", data)), f"out{i}:{data}")[1],
            forward=lambda self, *a, **k: (log.append((i, "forward", a)), f"fwd{i}")[1],
            forward_estimate_bitlen=lambda self, *a, **k: (log.append((i, "feb", a)), (f"fwd{i}", 10.0 + i))[1],
            update_state=lambda self, *a, **k: log.append((i, "update_state")),
            post_training_process=lambda self, *a, **k: log.append((i, "post_training_process")),
            set_rate_level=lambda self, level, *a, **k: log.append((i, "set_rate_level", level)),
            set_complex_level=lambda self, level, *a, **k: log.append((i, "set_complex_level", level)),
            get_current_complex_metrics=lambda self, *a, **k: (log.append((i, "metrics")), {"FLOPs": 100.0 * (i + 1)})[1],
            set_task=lambda self, task, *a, **k: (log.append((i, "set_task", task)), task < tasks)[1],
            num_rate_levels=property(lambda self: rate), num_complex_levels=property(lambda self: complex),
            num_tasks=property(lambda self: tasks))
        return type(f"Member{i}", tuple(bases), ns)()

    out = {}
    for tag, cfg in (("plain", None), ("cfg", GROUPED_VR_CONFIG)):
        log, rets = run_grouped_ops(make_member, GroupedVariableRateCodec, cfg)
        out[f"{tag}.log"], out[f"{tag}.rets"] = np.array(log), np.array(rets)
        print(tag, len(log), rets[:6])
    g = GroupedVariableRateCodec([make_member(j, []) for j in range(3)])
    out["module_names"] = np.array([n for n, _ in g.named_children()])
    save("grouped_codec.npz", **out)


def ar_coder_dynamic():
    """Dynamic-kernel PGMs (pgm_coder.py:996-1001,1314-1339,1941-1955): encode(..., pgm=(topo groups, context-conv weight
    [1, 2C, C, k, k], bias [1, 2C])), with and without pgm_dynamic_kernel_add_self; forward() rate estimate with the same
    pgm."""
    def seeded(coder, seed):
        names, shapes = [], []
        torch.manual_seed(seed)
        with torch.no_grad():
            for name, p in coder.named_parameters():
                p.copy_(torch.randn(p.shape) * (0.05 if p.dim() > 1 else 0.02))
                names.append(name)
                shapes.append(",".join(str(d) for d in p.shape))
        return names, shapes

    out, keys = {}, []
    g = torch.Generator().manual_seed(99)
    # (C, G, add_self, default method, (B, H, W), topo patch shape)
    cases = [(16, 2, False, "none", (1, 4, 6), (4, 6)), (16, 1, True, "scanline", (1, 5, 5), (5, 5)),
             (16, 2, True, "checkerboard", (2, 6, 6), (2, 2)), (16, 1, False, "none", (1, 4, 4), None)]
    for i, (C, G, add_self, method, (B, H, W), patch) in enumerate(cases):
        coder = GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder(
            in_channels=C, channel_groups=G, pgm_include_dynamic_kernel=True, pgm_dynamic_kernel_add_self=add_self,
            default_topo_group_method=method).eval()
        names, shapes = seeded(coder, 700 + i)
        coder.update_state()
        y = torch.randn(B, C, H, W, generator=g) * 3
        prior = torch.randn(B, 2 * C, H, W, generator=g)
        cp = coder.context_prediction
        if patch is None:
            pgm = None     # the coder then uses its own kernel and its default topo groups (:1302-1312)
        else:
            topo = torch.randint(0, 4, (1, G) + patch, generator=g)
            pgm = (topo, torch.randn((1,) + tuple(cp.weight.shape), generator=g) * 0.05, torch.randn((1,) + tuple(cp.bias.shape), generator=g) * 0.02)
        cap = {}
        orig = coder.ans_encoder

        class Spy:
            def encode_with_indexes(self, data, indexes, **k):
                cap["symbols"], cap["indexes"] = np.array(data), np.array(indexes)
                return orig.encode_with_indexes(data, indexes, **k)
        coder.ans_encoder = Spy()
        with torch.no_grad():
            data = coder.encode(y, prior=prior, pgm=pgm)
            yhat = coder.decode(data, prior=prior, pgm=pgm)
            coder(y, prior=prior, pgm=pgm)
            pe = float(coder.get_raw_cache("metric_dict")["prior_entropy"])
        assert float((yhat - y).abs().max()) <= 0.5 + 1e-5
        k = f"d{i}"
        out.update({f"{k}.pnames": np.array(names), f"{k}.pshapes": np.array(shapes),
                    f"{k}.wsum": np.array([float(sum(p.double().sum() for p in coder.parameters()))]),
                    f"{k}.cfg": np.array([C, G, int(add_self), B, H, W, int(pgm is None)]), f"{k}.method": np.array(method),
                    f"{k}.seed": np.array(700 + i), f"{k}.y": y.numpy(), f"{k}.prior": prior.numpy(), f"{k}.bytes": b2a(data),
                    f"{k}.symbols": cap["symbols"].astype(np.int32), f"{k}.indexes": cap["indexes"].astype(np.int32),
                    f"{k}.yhat": yhat.numpy(), f"{k}.prior_entropy": np.array(pe)})
        if pgm is not None:
            out.update({f"{k}.topo": pgm[0].numpy(), f"{k}.kernel_weight": pgm[1].numpy(), f"{k}.kernel_bias": pgm[2].numpy()})
        keys.append(k)
        print(f"  {k}: {len(data)} bytes, prior_entropy {pe:.2f}")
    out["keys"] = np.array(keys)
    save("ar_coder_dynamic.npz", **out)


def train_mode():
    """The reference's TRAIN-mode evaluation (additive-uniform-noise proxies): a random variable, recorded as draws.
      c0 / c1  PGM y-coder alone (scanline, context model), training_no_quantize_for_likelihood False / True: eval-mode
               prior_entropy (deterministic) and 24 draws of the train-mode loss_rate (pgm_coder.py:391-520)
      g.l*     the tiny BaSIC graph with the presets' training_no_quantize_for_likelihood=True y-coder: 16 draws of
               _test_dataset_complexity_performance(performance_method="loss", complexity_method="FLOPs") at three
               controller settings (latent_graph.py:1320-1395) and the eval-mode forward metrics at the same settings"""
    import copy
    import logging
    logging.disable(logging.CRITICAL)
    out = {}
    g = torch.Generator().manual_seed(31)
    for ci, flag in enumerate((False, True)):
        C, B, H, W = 16, 2, 8, 8
        coder = GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder(
            in_channels=C, default_topo_group_method="scanline", training_no_quantize_for_likelihood=flag,
            topo_group_context_model=TopoGroupDynamicMaskConv2dContextModel(in_channels=C, out_channels=2 * C))
        names, shapes = [], []
        torch.manual_seed(900 + ci)
        with torch.no_grad():
            for name, p in coder.named_parameters():
                p.copy_(torch.randn(p.shape) * (0.05 if p.dim() > 1 else 0.02))
                names.append(name)
                shapes.append(",".join(str(d) for d in p.shape))
        coder.update_state()
        y = torch.randn(B, C, H, W, generator=g) * 3
        prior = torch.cat([torch.randn(B, C, H, W, generator=g), torch.rand(B, C, H, W, generator=g) * 4 + 0.2], 1)
        prior = prior.reshape(B, 2, C, H, W).transpose(1, 2).reshape(B, 2 * C, H, W).contiguous()
        coder.eval()
        with torch.no_grad():
            coder(y, prior=prior)
            pe = float(coder.get_raw_cache("metric_dict")["prior_entropy"])
        coder.reset_all_cache()
        coder.train()
        draws = []
        for d in range(24):
            torch.manual_seed(3000 + d)
            with torch.no_grad():
                coder(y, prior=prior)
            draws.append(float(coder.get_raw_cache("loss_dict")["loss_rate"]))
            coder.reset_all_cache()
        k = f"c{ci}"
        out.update({f"{k}.pnames": np.array(names), f"{k}.pshapes": np.array(shapes), f"{k}.seed": np.array(900 + ci),
                    f"{k}.wsum": np.array([float(sum(p.double().sum() for p in coder.parameters()))]),
                    f"{k}.flag": np.array(int(flag)), f"{k}.y": y.numpy(), f"{k}.prior": prior.numpy(),
                    f"{k}.eval_prior_entropy": np.array(pe), f"{k}.train_loss_rate": np.array(draws)})
        print(f"  {k}: eval prior_entropy {pe:.2f} nats ({pe / math.log(2):.2f} bits), train loss_rate {np.mean(draws):.2f} +- {np.std(draws):.2f} bits")

    codec, ec, levels, ctl, touched, calib, Wd, M = _basic_graph(dict(training_no_quantize_for_likelihood=True))
    x = recipe_input(890, (1, 3, 64, 64))
    state = copy.deepcopy(codec.state_dict())
    for li in (0, 4, 6):
        lv = levels[li]
        codec.eval()
        codec.set_complex_level(li)
        codec.reset_all_cache()
        with torch.no_grad():
            codec(x)
        met = {n.split("metric_dict/entropy_coder/")[-1]: float(v) for n, v in codec.get_cache("metric_dict").items()}
        codec.reset_all_cache()
        vals = []
        for d in range(16):
            codec.load_state_dict(state)     # the z-coder's inner aux optimiser moves its quantiles in train mode (compressai_coder.py:186-190)
            torch.manual_seed(5000 + 16 * li + d)
            params = {name: ec.node_generators[name](lv[name]) for name in ctl}
            c, p_ = ec._test_dataset_complexity_performance([x], performance_method="loss", complexity_method="FLOPs", **params)
            vals.append((c, p_))
        codec.load_state_dict(state)
        vals = np.array(vals, np.float64)
        out[f"g.l{li}.train"] = vals
        out[f"g.l{li}.eval_metric_names"] = np.array(list(met))
        out[f"g.l{li}.eval_metric_values"] = np.array(list(met.values()), np.float64)
        print(f"  g.l{li}: complexity {vals[0, 0]:.4f} (spread {np.ptp(vals[:, 0]):.2e}), train loss {vals[:, 1].mean():.5f} +- {vals[:, 1].std():.5f}; "
              f"eval prior_entropy {met.get('prior_entropy', float('nan')):.2f}")
    out["g.levels"] = np.array([0, 4, 6])
    out["g.pnames"] = np.array([n for n, _ in touched])
    out["g.pshapes"] = np.array([s_ for _, s_ in touched])
    out["g.calib_names"] = np.array([c[0] for c in calib])
    out["g.calib_mul"] = np.array([c[1] for c in calib])
    out["g.calib_add_odd"] = np.array([c[2] for c in calib])
    logging.disable(logging.NOTSET)
    save("train_mode.npz", **out)


def ar_coder_quant():
    """Non-identity quantisers of the PGM y-coder (torch_ans.py:16-50,105-121,163-178): "uniform" [offset, -, step] at
    construction and overridden per call, "uniform_scale" [step]: bytes, integers, decode() output, forward() output and
    rate estimate."""
    out, keys = {}, []
    g = torch.Generator().manual_seed(55)
    # (quantizer_type, ctor params, per-call params or None, method, G)
    cases = [("uniform", [0.3, 128.0, 0.5], None, "checkerboard", 1), ("uniform", None, [-1.25, 128.0, 2.0], "scanline", 1),
             ("uniform_scale", [0.75], None, "none", 2)]
    for i, (qt, qp, call_qp, method, G) in enumerate(cases):
        C, B, H, W = 16, 1, 6, 6
        kw = dict(in_channels=C, channel_groups=G, default_topo_group_method=method, quantizer_type=qt)
        if qp is not None:
            kw["quantizer_params"] = qp
        coder = GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder(**kw).eval()
        names, shapes = [], []
        torch.manual_seed(600 + i)
        with torch.no_grad():
            for name, p in coder.named_parameters():
                p.copy_(torch.randn(p.shape) * (0.05 if p.dim() > 1 else 0.02))
                names.append(name)
                shapes.append(",".join(str(d) for d in p.shape))
        coder.update_state()
        y = torch.randn(B, C, H, W, generator=g) * 3
        prior = torch.cat([torch.randn(B, C, H, W, generator=g), torch.rand(B, C, H, W, generator=g) * 4 + 0.2], 1)
        prior = prior.reshape(B, 2, C, H, W).transpose(1, 2).reshape(B, 2 * C, H, W).contiguous()
        cap = {}
        orig = coder.ans_encoder

        class Spy:
            def encode_with_indexes(self, data, indexes, **k):
                cap["symbols"], cap["indexes"] = np.array(data), np.array(indexes)
                return orig.encode_with_indexes(data, indexes, **k)
        coder.ans_encoder = Spy()
        cq = None if call_qp is None else torch.tensor(call_qp)
        with torch.no_grad():
            data = coder.encode(y, prior=prior, quantizer_params=cq)
            yhat = coder.decode(data, prior=prior, quantizer_params=cq)
            yfwd = coder(y, prior=prior, quantizer_params=cq)
            pe = float(coder.get_raw_cache("metric_dict")["prior_entropy"])
        k = f"q{i}"
        out.update({f"{k}.pnames": np.array(names), f"{k}.pshapes": np.array(shapes), f"{k}.seed": np.array(600 + i),
                    f"{k}.wsum": np.array([float(sum(p.double().sum() for p in coder.parameters()))]),
                    f"{k}.qtype": np.array(qt), f"{k}.ctor_params": np.array(qp if qp is not None else [], np.float32),
                    f"{k}.call_params": np.array(call_qp if call_qp is not None else [], np.float32),
                    f"{k}.method": np.array(method), f"{k}.cfg": np.array([C, G, B, H, W]),
                    f"{k}.y": y.numpy(), f"{k}.prior": prior.numpy(), f"{k}.bytes": b2a(data),
                    f"{k}.symbols": cap["symbols"].astype(np.int32), f"{k}.indexes": cap["indexes"].astype(np.int32),
                    f"{k}.yhat": yhat.numpy(), f"{k}.yfwd": yfwd.numpy(), f"{k}.prior_entropy": np.array(pe)})
        keys.append(k)
        print(f"  {k}: {qt} {qp} call {call_qp}: {len(data)} bytes, max|yhat - y| {float((yhat - y).abs().max()):.3f}, prior_entropy {pe:.2f}")
    out["keys"] = np.array(keys)
    save("ar_coder_quant.npz", **out)


def ar_ops_kats():
    """Custom AR ops (csrc/ans/ar_funcs.hpp:29-87, ANSBase::init_custom_ar_ops ans_interface.hpp:40-84): bytes of the reference's
    compiled Rans64Encoder with ar_limited_scaled_add_linear_op index remaps of 1, 2 and 3 predecessors, and direct calls of the
    two bound op classes."""
    out, names = {}, []
    rng = np.random.default_rng(606)
    for order in (1, 2, 3):
        nd, ns, n, k = 16, 24, 1200, 3
        freqs = rng.integers(1, 600, (nd, ns)).astype(np.int32)
        nsym, off = np.full(nd, ns, np.int32), rng.integers(-3, 3, nd).astype(np.int32)
        ops = []
        for j in range(k):
            sc = float([1.0, 2.0, 4.0][j])
            ops.append(((rng.random(order) * 0.6 - 0.3).round(4).tolist(), float(np.round(rng.random() - 0.5, 4)), sc, 0.0, float(nd // int(sc) - 1)))
        sym = (rng.integers(-2, ns + 2, n)).astype(np.int32)
        idx = rng.integers(0, nd, n).astype(np.int32)
        ai = rng.integers(0, k, n).astype(np.int32)
        aoff = np.stack([np.minimum(np.arange(n), 1 + 2 * q) for q in range(order)]).astype(np.int32)
        enc, dec = ref_ans.Rans64Encoder(16, True, 4), ref_ans.Rans64Decoder(16, True, 4)
        for c in (enc, dec):
            c.init_params(freqs, nsym, off)
            c.init_custom_ar_ops([ref_ans.ar_limited_scaled_add_linear_op(*o) for o in ops])
        data = enc.encode_with_indexes(sym, idx, ai, aoff)
        assert np.array_equal(dec.decode_with_indexes(data, idx, ai, aoff), sym)
        name = f"o{order}"
        out.update({f"{name}.freqs": freqs, f"{name}.nsym": nsym, f"{name}.offsets": off, f"{name}.symbols": sym, f"{name}.indexes": idx,
                    f"{name}.ar_indexes": ai, f"{name}.ar_offsets": aoff, f"{name}.bytes": b2a(data),
                    f"{name}.ops": np.array([list(o[0]) + [0.0] * (3 - order) + list(o[1:]) for o in ops], np.float64)})
        names.append(name)
        print(f"  {name}: {len(data)} bytes")
    # direct calls
    lim = ref_ans.ar_limited_scaled_add_linear_op([0.37, -0.21], 0.4, 2.0, 0.0, 7.0)
    lin = ref_ans.ar_linear_op([1.0, 0.5, -0.25], 0.125, 2.0)
    vecs = rng.integers(-6, 16, (40, 3)).astype(np.int32)
    out["call.vectors"] = vecs
    out["call.limited"] = np.array([lim(v.tolist()) for v in vecs], np.int32)
    out["call.linear"] = np.array([lin(v.tolist()) for v in vecs], np.int32)
    out["names"] = np.array(names)
    save("ar_ops_kat.npz", **out)


def tans_kats():
    """Known answers of the reference's compiled TansEncoder / TansDecoder (cbench/csrc/ans/tans.cpp in oracle/_ref):
    bytes, decoded symbols, the error cases (too small an output budget -> ValueError; stream larger than its budget ->
    b"") and two AR-remap cases."""
    out, names = {}, []
    rng = np.random.default_rng(4242)
    cases = []
    # (name, table_log, rows, max symbols, n, bypass, skewed counts)
    for i, (L, nd, ns, n, byp, skew) in enumerate([(12, 8, 200, 3000, True, False), (11, 5, 40, 777, True, True), (10, 3, 9, 64, False, False),
                                                   (12, 64, 64, 4096, True, True), (9, 2, 30, 2000, False, True), (12, 1, 2, 6, False, False),
                                                   (12, 4, 17, 5, False, False), (5, 2, 8, 500, True, False), (11, 6, 120, 1500, False, True),
                                                   (12, 3, 500, 900, True, True)]):
        if skew:
            freqs = np.maximum((rng.random((nd, ns)) ** 7 * 60000).astype(np.int64), 1)
        else:
            freqs = rng.integers(1, 1024, (nd, ns))
        nsym = rng.integers(2, ns + 1, nd)
        off = rng.integers(-6, 6, nd)
        idx = rng.integers(0, nd, n)
        if byp:
            sym = off[idx] + rng.integers(-3, 1 << 30, n) % (nsym[idx] + 6)
            sym[::11] = rng.integers(-70000, 70000, sym[::11].size)
        else:
            sym = off[idx] + rng.integers(0, 1 << 30, n) % nsym[idx]
        cases.append((f"rand{i}", L, freqs, nsym, off, byp, sym, idx, None))
    # incompressible input: the stream outgrows the reference's n * L / 8 budget and comes back empty
    nd, ns, n, L = 2, 256, 400, 9
    freqs = rng.integers(1, 1024, (nd, ns))
    cases.append(("overflow", L, freqs, np.full(nd, ns), np.zeros(nd, np.int64), True, rng.integers(-5000, 5000, n), rng.integers(0, nd, n), None))
    # AR remap, order 1 and 2 (ans_interface.cpp:75-137)
    for order in (1, 2):
        nd, ns, n, L = 4, 12, 800, 11
        freqs = rng.integers(1, 500, (nd, ns))
        tab = rng.integers(0, nd, (2, nd) + (ns + 1,) * order)
        ai = rng.integers(0, 2, n)
        aoff = np.stack([np.minimum(np.arange(n), 1 + k) for k in range(order)])
        cases.append((f"ar{order}", L, freqs, np.full(nd, ns), np.zeros(nd, np.int64), False, rng.integers(0, ns, n), rng.integers(0, nd, n),
                      (tab, np.ones((2, order, 1), np.int64), ai, aoff)))
    for name, L, freqs, nsym, off, byp, sym, idx, ar in cases:
        f, n_, o = (np.asarray(a).astype(np.int32) for a in (freqs, nsym, off))
        s, ix = np.asarray(sym).astype(np.int32), np.asarray(idx).astype(np.int32)
        print(f"  {name} ...", flush=True)
        enc, dec = ref_ans.TansEncoder(L, 255, byp, 4), ref_ans.TansDecoder(L, 255, byp, 4)
        rec = {f"{name}.freqs": f, f"{name}.nsym": n_, f"{name}.offsets": o, f"{name}.cfg": np.array([L, int(byp), 4]),
               f"{name}.symbols": s, f"{name}.indexes": ix}
        kw = {}
        try:
            enc.init_params(f, n_, o)
            dec.init_params(f, n_, o)
            if ar is not None:
                tab, cfg, ai, aoff = (np.asarray(a).astype(np.int32) for a in ar)
                enc.init_ar_params(tab, cfg)
                dec.init_ar_params(tab, cfg)
                kw = dict(ar_indexes=ai, ar_offsets=aoff)
                rec.update({f"{name}.ar_table": tab, f"{name}.ar_cfg": cfg, f"{name}.ar_indexes": ai, f"{name}.ar_offsets": aoff})
            data = enc.encode_with_indexes(s, ix, **kw)
            rec[f"{name}.bytes"] = b2a(data)
            rec[f"{name}.error"] = np.array(0)
            if data:
                back = dec.decode_with_indexes(data, ix, **kw)
                rec[f"{name}.decoded"] = back.astype(np.int32)   # == symbols with bypass coding; clamped without it
                if byp or ar is not None or name != "overflow":
                    assert np.array_equal(back, s), name
        except ValueError as e:
            rec[f"{name}.bytes"] = np.zeros(0, np.uint8)
            rec[f"{name}.error"] = np.array(1)
            print(f"  {name}: reference raises ValueError({e})")
        out.update(rec)
        names.append(name)
        print(f"  {name}: L={L} n={s.size} -> {rec[f'{name}.bytes'].size} bytes")
    out["names"] = np.array(names)
    save("tans_kat.npz", **out)


def rans_cache_kats():
    """Rans64Encoder's symbol cache (encode_with_indexes(..., cache=True) / peek_cache / flush, rans64.cpp:237-386,
    rans64.hpp:78-86) and ANSBase::create_ar_ptrs (ans_interface.cpp:34-73) of the reference's compiled cbench.ans."""
    rng = np.random.default_rng(77)
    out, names = {}, []
    for ci, (nd, ns, bypass, calls) in enumerate([(3, 9, True, [40, 25]), (2, 6, False, [30]), (4, 20, True, [10, 1, 33])]):
        freqs = rng.integers(1, 400, (nd, ns)).astype(np.int32)
        nsym = rng.integers(2, ns + 1, nd).astype(np.int32)
        off = rng.integers(-4, 3, nd).astype(np.int32)
        enc = ref_ans.Rans64Encoder(16, bypass, 4)
        enc.init_params(freqs, nsym, off)
        k = f"c{ci}"
        out.update({f"{k}.freqs": freqs, f"{k}.nsym": nsym, f"{k}.offsets": off, f"{k}.bypass": np.array(int(bypass)), f"{k}.ncalls": np.array(len(calls))})
        for j, n in enumerate(calls):
            idx = rng.integers(0, nd, n).astype(np.int32)
            if bypass:
                sym = (off[idx] + rng.integers(-3, ns + 4, n)).astype(np.int32)
                sym[::7] += rng.integers(-3000, 3000, sym[::7].size).astype(np.int32)
            else:
                sym = (off[idx] + rng.integers(0, 1 << 30, n) % nsym[idx]).astype(np.int32)
            assert enc.encode_with_indexes(sym, idx, cache=True) == b""
            out[f"{k}.sym{j}"], out[f"{k}.idx{j}"] = sym, idx
            out[f"{k}.peek{j}"] = np.array(enc.peek_cache()).astype(np.int32)
        out[f"{k}.flush"] = b2a(enc.flush())
        out[f"{k}.peek_after"] = np.array(enc.peek_cache()).astype(np.int32).reshape(-1, 3)
        names.append(k)
        print(f"  {k}: {sum(calls)} symbols cached in {len(calls)} calls -> {out[f'{k}.peek{len(calls) - 1}'].shape[0]} rANS symbols, {out[f'{k}.flush'].size} bytes")
    enc = ref_ans.Rans64Encoder(16, True, 4)
    pcases = [((1, 3, 4), [[-1, 0]]), ((2, 3, 4), [[0, -1], [-1, -1]]), ((1, 2, 3, 4), [[-1, 0, 0], [0, -2, -1], [0, 0, 0]]), ((1, 5), [[-2]])]
    for pi, (shape, offs) in enumerate(pcases):
        r = enc.create_ar_ptrs(np.zeros(shape, np.int32), offs)
        out[f"p{pi}.shape"] = np.array(shape)
        out[f"p{pi}.offsets"] = np.array([o + [0] * (3 - len(o)) for o in offs])
        out[f"p{pi}.nd"] = np.array([len(o) for o in offs])
        out[f"p{pi}.ptrs"] = np.array(r, dtype=np.int64)
    try:
        enc.create_ar_ptrs(np.zeros((1, 3, 3), np.int32), [[1, 0]])
        out["p.positive_raises"] = np.array(0)
    except ValueError:
        out["p.positive_raises"] = np.array(1)
    # cached encoding on AUTOREGRESSIVE table sets (rans64.cpp:237-361: an AR call's rows are resolved from that call's own
    # symbols when it is cached; flush() codes the cached (start, range) pairs as they are): remap tables of order 1 and 2,
    # and the custom limited scaled-add op with two predecessors; two or three cached calls each
    arnames = []
    for ai_, (mode, order, calls) in enumerate([("table", 1, [60, 35]), ("table", 2, [50, 20, 41]), ("ops", 2, [70, 45])]):
        nd, ns = 8, 12
        freqs = rng.integers(1, 500, (nd, ns)).astype(np.int32)
        nsym, off = np.full(nd, ns, np.int32), np.zeros(nd, np.int32)
        enc = ref_ans.Rans64Encoder(16, True, 4)
        enc.init_params(freqs, nsym, off)
        k = f"a{ai_}"
        out.update({f"{k}.freqs": freqs, f"{k}.nsym": nsym, f"{k}.offsets": off, f"{k}.mode": np.array(mode), f"{k}.order": np.array(order),
                    f"{k}.ncalls": np.array(len(calls))})
        if mode == "table":
            ktab = 2
            tab = rng.integers(0, nd, (ktab, nd) + (ns + 1,) * order).astype(np.int32)
            enc.init_ar_params(tab, np.zeros((ktab, order, 1), np.int32))
            out[f"{k}.ar_table"] = tab
        else:
            ktab = 3
            ops = [((rng.random(order) * 0.6 - 0.3).round(4).tolist(), float(np.round(rng.random() - 0.5, 4)), float([1.0, 2.0, 4.0][j]), 0.0,
                    float(nd // int([1.0, 2.0, 4.0][j]) - 1)) for j in range(ktab)]
            enc.init_custom_ar_ops([ref_ans.ar_limited_scaled_add_linear_op(*o) for o in ops])
            out[f"{k}.ops"] = np.array([list(o[0]) + [0.0] * (3 - order) + list(o[1:]) for o in ops], np.float64)
        for j, n in enumerate(calls):
            idx = rng.integers(0, nd, n).astype(np.int32)
            sym = rng.integers(0, ns, n).astype(np.int32)     # inside the remap table's range (the reference does not check)
            ai = rng.integers(0, ktab, n).astype(np.int32)
            aoff = np.stack([np.minimum(np.arange(n), 1 + 2 * q) for q in range(order)]).astype(np.int32)
            assert enc.encode_with_indexes(sym, idx, ai, aoff, True) == b""
            out.update({f"{k}.sym{j}": sym, f"{k}.idx{j}": idx, f"{k}.ai{j}": ai, f"{k}.aoff{j}": aoff,
                        f"{k}.peek{j}": np.array(enc.peek_cache()).astype(np.int32)})
        out[f"{k}.flush"] = b2a(enc.flush())
        arnames.append(k)
        print(f"  {k}: cached AR ({mode}, order {order}): {sum(calls)} symbols in {len(calls)} calls -> {out[f'{k}.flush'].size} bytes")
    out["names"], out["npcases"], out["arnames"] = np.array(names), np.array(len(pcases)), np.array(arnames)
    save("rans_cache_kat.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["rans", "tables", "topo", "mconv", "ar", "framing", "harness", "search", "arpgm", "arjoint", "codec", "grouped", "tans", "ardyn", "train", "arquant", "arops", "ranscache"]
    fn = dict(rans=rans_kats, tables=gauss_tables, topo=topo_maps, mconv=masked_conv, ar=ar_coder, framing=framing, harness=harness, search=complexity_search, arpgm=ar_coder_pgm, arjoint=ar_coder_joint, codec=codec_graph, grouped=grouped, tans=tans_kats, ardyn=ar_coder_dynamic, train=train_mode, arquant=ar_coder_quant, arops=ar_ops_kats, ranscache=rans_cache_kats)
    for w in which:
        fn[w]()
