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
            elif name.endswith(".bias") and ".model." in name:
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.05)
        ec = codec.entropy_coder
        g_a = ec.latent_inference_modules["x_y"].model
        h_s = ec.latent_generative_modules["z_y"].model
        # scale y: calibrate on one seeded image on the CPU reference ops
        import torch.nn.functional as F
        x = torch.rand(1, 3, 64, 64, generator=g)
        t = x
        for m in g_a:
            if isinstance(m, torch.nn.Conv2d):
                t = F.conv2d(t, m.weight, m.bias, stride=m.stride, padding=m.padding)
            else:
                gamma, beta = m.effective()
                C = t.shape[1]
                t = t * torch.rsqrt(F.conv2d(t * t, gamma.reshape(C, C, 1, 1), beta))
        last = g_a[len(g_a) - 1]
        k = y_std / float(t.std())
        last.weight.mul_(k)
        last.bias.mul_(k)
        # predicted scales: per-channel log-uniform bias in [0.2, 1.2) on the last hyper-synthesis conv and a
        # damped data-dependent part, so the coded rate lands near a trained model's (~0.5-1 bpp)
        last_hs = [m for m in h_s if isinstance(m, torch.nn.Conv2d)][-1]
        last_hs.weight.mul_(0.05)
        last_hs.bias.copy_(torch.exp(torch.rand(last_hs.bias.shape, generator=g) * 1.8 - 1.6))
        # keep the reconstruction in a sane range: small output layer, mid-grey bias
        g_s = ec.latent_generative_modules["y_x"].model
        last_gs = g_s[len(g_s) - 1]
        last_gs.weight.mul_(0.05)
        last_gs.bias.fill_(0.5)
    return codec
