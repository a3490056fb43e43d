"""Codec builders for the BASELINE.json configurations (what the reference's config files
instantiate through ClassBuilder; here as plain functions).

  hyperprior_codec(N, M)  <- configs/lossy_graph_scalable_exp_hp.py:182-215
"""
import torch

from .codecs.general_codec import GeneralCodec
from .modules.entropy_coder.latent_graph import LatentGraphicalANSEntropyCoder, LossyDummyEntropyCoder
from .modules.prior_model.prior_coder.compressai_coder import (CompressAIEntropyBottleneckPriorCoder,
                                                               CompressAIGaussianConditionalCoder)
from .nn.models.google import (HyperpriorAnalysisModel, HyperpriorHyperAnalysisModel,
                               HyperpriorHyperSynthesisModel, HyperpriorSynthesisModel)


def hyperprior_codec(N=128, M=192):
    ec = LatentGraphicalANSEntropyCoder(
        latent_node_inference_topo_order=["x", "y", "z"],
        latent_node_generative_topo_order=["z", "y", "x"],
        latent_node_entropy_coder_dict=dict(
            x=LossyDummyEntropyCoder(lambda_rd=145.2225),
            y=CompressAIGaussianConditionalCoder(),
            z=CompressAIEntropyBottleneckPriorCoder(entropy_bottleneck_channels=N, use_inner_aux_opt=True),
        ),
        latent_inference_dict=dict(x_y=HyperpriorAnalysisModel(N=N, M=M), y_z=HyperpriorHyperAnalysisModel(N=N, M=M)),
        latent_generative_dict=dict(z_y=HyperpriorHyperSynthesisModel(N=N, M=M), y_x=HyperpriorSynthesisModel(N=N, M=M)),
    )
    return GeneralCodec(entropy_coder=ec)


def topogroup_ar_codec(method="checkerboard", N=128, M=192, channel_groups=1, expand_bottleneck=True, use_param_merger=True):
    """configs/lossy_latent_graph_topogroup.py:203-244 ("hyperprior-ar-base" and its topo-group variants
    :253-781): hyperprior transforms, h_s = HyperpriorHyperSynthesisModel(N, 2M) giving (mean, scale)
    interleaved, y coded by the topo-group AR Gaussian coder with the in-coder masked param merger."""
    from .modules.prior_model.prior_coder.pgm_coder import GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder
    hs = HyperpriorHyperSynthesisModel(N=N, M=2 * M)
    ec = LatentGraphicalANSEntropyCoder(
        latent_node_inference_topo_order=["x", "y", "z"],
        latent_node_generative_topo_order=["z", "y", "x"],
        latent_node_entropy_coder_dict=dict(
            x=LossyDummyEntropyCoder(lambda_rd=145.2225),
            y=GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder(in_channels=M, channel_groups=channel_groups,
                                                                   default_topo_group_method=method, use_param_merger=use_param_merger,
                                                                   param_merger_expand_bottleneck=expand_bottleneck),
            z=CompressAIEntropyBottleneckPriorCoder(entropy_bottleneck_channels=N, use_inner_aux_opt=True),
        ),
        latent_inference_dict=dict(x_y=HyperpriorAnalysisModel(N=N, M=M), y_z=HyperpriorHyperAnalysisModel(N=N, M=M)),
        latent_generative_dict=dict(z_y=hs, y_x=HyperpriorSynthesisModel(N=N, M=M)),
    )
    return GeneralCodec(entropy_coder=ec)


BASIC_WIDTHS = [48, 72, 96, 144, 192]


def basic_codec(widths=BASIC_WIDTHS, M=192, num_complex_levels=8, search_dataset=None, combined_entropy_coder=False):
    """BaSIC "hyperprior-ar-sc-slimmable-full-dynamic" (configs/presets/lossy_latent_graph_scalable_ar_models.py:
    73-197): slimmable g_a/g_s, MS-slimmable h_a/h_s, 192-ch EntropyBottleneck, scanline AR y-coder with the
    masked-conv context model, four slim controller nodes selected per complexity level.

    ``search_dataset`` (iterable of image batches) = the "...-greedy-search-8level" variant (:733-757): the levels are
    found by ``post_training_process`` (once weights are loaded and the codec sits on the GPU); without it a fixed
    monotone ladder of controller settings is installed.

    ``combined_entropy_coder`` = the "...-combined-dynamic-entropy-coder" variant (:198-372): the y-coder is a bank
    {scanline AR, 8-, 6-, 4-, 2-stage grouped coders with learned topo groups} selected by a fifth controller node
    ``pgmy`` (blend_weight); the learned groups are the coders' ``topo_group_predictor_cache`` buffers (random logits
    until a checkpoint is loaded)."""
    from .modules.prior_model.prior_coder.pgm_coder import (CombinedNNTrainablePGMPriorCoder,
                                                            GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder,
                                                            TopoGroupDynamicMaskConv2dContextModel)
    from .nn.layers.param_generator import IndexSelectParameterGeneratorWrapper, NNParameterGenerator
    from .nn.layers.pgm_layers import (HyperpriorAnalysisSlimmableConv2dPGMModel, HyperpriorSynthesisSlimmableConv2dPGMModel,
                                       MeanScaleHyperpriorHyperAnalysisSlimmableConv2dPGMModel,
                                       MeanScaleHyperpriorHyperSynthesisSlimmableConv2dPGMModel)
    n = len(widths)

    def ar_coder(**kw):
        # training_no_quantize_for_likelihood as in the reference preset (lossy_latent_graph_scalable_ar_models.py:121,261-330): the
        # forward() rate estimate is taken on round(y - mu) under the zero-mean density (no effect on the coded bytes)
        return GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder(
            in_channels=M, training_no_quantize_for_likelihood=True, topo_group_context_model=TopoGroupDynamicMaskConv2dContextModel(in_channels=M, out_channels=2 * M), **kw)

    nodes = dict(pgmxy=None, pgmyx=None, pgmyz=None, pgmzy=None)
    controllers = ["pgmxy", "pgmyz", "pgmzy", "pgmyx"]
    y_mapping = {"z": "prior"}
    if combined_entropy_coder:
        g = torch.Generator().manual_seed(1234)
        logits = lambda G, L: torch.randn(1, G * L, 2, 2, generator=g)   # predictor output: out_channels = G * L, 2 x 2 patch
        y_coder = CombinedNNTrainablePGMPriorCoder([
            ar_coder(default_topo_group_method="scanline"),
            ar_coder(channel_groups=4, topo_group_predictor=logits(4, 8)),    # 8-stage (:270-289)
            ar_coder(channel_groups=4, topo_group_predictor=logits(4, 6)),    # 6-stage
            ar_coder(topo_group_predictor=logits(1, 16)),                      # 4-stage (channel_groups 1, 16 logits)
            ar_coder(channel_groups=2, param_merger_expand_bottleneck=True, topo_group_predictor=logits(2, 2)),  # 2-stage
        ], training_use_max_capacity=True)
        nb = len(y_coder.coders)
        nodes["pgmy"] = IndexSelectParameterGeneratorWrapper(
            batched_generator=NNParameterGenerator(shape=(nb, nb), init_method="value", init_value=torch.eye(nb), fix_params=True),
            fix_for_inference=True)
        controllers.append("pgmy")
        y_mapping = {"pgmy": "blend_weight", "z": "prior"}
    else:
        y_coder = ar_coder(default_topo_group_method="scanline")

    def slim_node():
        return IndexSelectParameterGeneratorWrapper(
            batched_generator=NNParameterGenerator(shape=(n, 1, 1, n), init_method="value",
                                                   init_value=torch.eye(n).flip(-1).unsqueeze(1).unsqueeze(1), fix_params=True),
            fix_for_inference=True)

    ec = LatentGraphicalANSEntropyCoder(
        node_generator_dict={k: (v if v is not None else slim_node()) for k, v in nodes.items()},
        use_lossy_compression=True, lossy_compression_lambda_rd=145.2225,
        latent_node_inference_topo_order=["x", "y", "z"],
        latent_node_generative_topo_order=["z", "y", "x"],
        latent_node_entropy_coder_dict=dict(
            y=y_coder,
            z=CompressAIEntropyBottleneckPriorCoder(entropy_bottleneck_channels=M, use_inner_aux_opt=True),
        ),
        latent_inference_dict=dict(
            x_y=HyperpriorAnalysisSlimmableConv2dPGMModel(in_channels=3, out_channels=M, mid_channels_list=widths),
            y_z=MeanScaleHyperpriorHyperAnalysisSlimmableConv2dPGMModel(in_channels=M, out_channels=M, mid_channels_list=widths)),
        latent_generative_dict=dict(
            z_y=MeanScaleHyperpriorHyperSynthesisSlimmableConv2dPGMModel(in_channels=M, out_channels=2 * M, mid_channels_list=widths),
            y_x=HyperpriorSynthesisSlimmableConv2dPGMModel(in_channels=M, out_channels=3, mid_channels_list=widths)),
        latent_inference_input_mapping=dict(x_y={"pgmxy": "pgm"}, y_z={"pgmyz": "pgm"}),
        latent_generative_input_mapping=dict(y_x={"pgmyx": "pgm"}, z_y={"pgmzy": "pgm"}, y=y_mapping),
        complexity_level_greedy_search=True, complexity_level_greedy_search_num_levels=num_complex_levels,
        complexity_level_greedy_search_dataset=search_dataset, complexity_level_greedy_search_dataset_cached=True,
        complexity_level_controller_nodes=controllers,
    )
    if search_dataset is None:
        # Searched levels need trained weights + a dataset; without them install a fixed monotone ladder of
        # controller indices (index 0 = widest, n-1 = narrowest).
        ladder = basic_default_ladder(n, num_complex_levels)
        if combined_entropy_coder:  # entropy-coder stage count falls with the level too: scanline ... 2-stage
            for i, lvl in enumerate(ladder):
                lvl["pgmy"] = round(i * (nb - 1) / max(1, num_complex_levels - 1))
        ec.set_complexity_level_params([{k: ec.node_generators[k](index=i) for k, i in lvl.items()} for lvl in ladder])
    return GeneralCodec(entropy_coder=ec)


def basic_default_ladder(n_widths, num_levels):
    """num_levels index tuples; level 0 = all-wide (most complex, the reference's convention for searched levels,
    latent_graph.py:1527), last level = all-narrow, narrowing one controller at a time in between."""
    names = ["pgmyx", "pgmxy", "pgmzy", "pgmyz"]
    cur = {k: n_widths - 1 for k in names}
    steps = [dict(cur)]
    while any(v > 0 for v in cur.values()):
        for k in names:
            if cur[k] > 0:
                cur[k] -= 1
                steps.append(dict(cur))
    pick = [round(i * (len(steps) - 1) / max(1, num_levels - 1)) for i in range(num_levels)]
    return [dict(steps[i]) for i in reversed(pick)]


def seed_synthetic_weights(codec, seed=0, y_std=0.5):
    """Random-init weights for synthetic benchmarking (no pretrained models exist: reference
    README.md:107-108).  Default torch init under a fixed seed, then the last analysis layer is
    rescaled so the latents use a realistic range of the 64-entry scale table."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in codec.named_parameters():
            if name.endswith(".weight") and p.dim() == 4:
                fan_in = p.shape[1] * p.shape[2] * p.shape[3]
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * (3.0 / fan_in) ** 0.5)
            elif name.endswith(".bias") and (".model." in name or ".pgm_model." in name):
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.05)
            elif ".entropy_bottleneck._bias" in name:
                # the constructor draws these from the GLOBAL generator (uniform(-.5, .5), as upstream): redraw them from
                # the seeded one, or two codecs built by the same call would code z with different tables
                p.copy_(torch.rand(p.shape, generator=g) - 0.5)
        ec = codec.entropy_coder
        import torch.nn.functional as F
        from .nn.layers.gdn import GDN
        from .nn.layers.slimmable_layers import DynamicConv2d, DynamicGDN

        def seq_of(m):
            return m.model if hasattr(m, "model") else m.pgm_model

        def conv_of(m):
            return m.conv if isinstance(m, DynamicConv2d) else m

        def cpu_forward(seq, t):  # widest configuration, plain torch ops, calibration only
            for m in seq:
                if isinstance(m, (torch.nn.Conv2d, DynamicConv2d)) and not getattr(m, "transposed", False) \
                        and not isinstance(m, torch.nn.ConvTranspose2d):
                    c = conv_of(m)
                    pad = m.padding if isinstance(m, DynamicConv2d) else c.padding
                    t = F.conv2d(t, c.weight[:, : t.shape[1]], c.bias, stride=c.stride, padding=pad)
                elif isinstance(m, (GDN, DynamicGDN)):
                    gamma, beta = m.effective() if isinstance(m, GDN) else m.effective(len(m.channels_list) - 1)
                    C = t.shape[1]
                    t = t * torch.rsqrt(F.conv2d(t * t, gamma.reshape(C, C, 1, 1), beta))
            return t

        g_a = seq_of(ec.latent_inference_modules["x_y"])
        h_s = seq_of(ec.latent_generative_modules["z_y"])
        g_s = seq_of(ec.latent_generative_modules["y_x"])
        x = torch.rand(1, 3, 64, 64, generator=g)
        t = cpu_forward(g_a, x)
        last = conv_of(g_a[len(g_a) - 1])
        k = y_std / float(t.std())
        last.weight.mul_(k)
        last.bias.mul_(k)
        # predicted scales: per-channel log-uniform bias in [0.2, 1.2) on the last hyper-synthesis conv and a
        # damped data-dependent part, so the coded rate lands near a trained model's (~0.5-1.5 bpp)
        last_hs = conv_of([m for m in h_s if isinstance(m, (torch.nn.Conv2d, DynamicConv2d))][-1])
        last_hs.weight.mul_(0.05)
        last_hs.bias.copy_(torch.exp(torch.rand(last_hs.bias.shape, generator=g) * 1.8 - 1.6))
        # keep the reconstruction in a sane range: small output layer, mid-grey bias
        last_gs = conv_of(g_s[len(g_s) - 1])
        last_gs.weight.mul_(0.05)
        last_gs.bias.fill_(0.5)
    return codec
