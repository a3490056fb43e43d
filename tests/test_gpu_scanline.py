"""The persistent scan-line AR kernel (csrc/scanline.hip: one launch for all H*W coding steps) against the per-step
path of the same coder (one masked-conv launch sequence per step).  Both sum every dot product in the canonical block
order of csrc/mconv.hip, so they must agree EXACTLY -- integer symbols / indexes, the coded latent bit for bit, the bytes --
for the BaSIC context-model coder, the in-coder merger and the joint-AR raster variant at several batch sizes: a stream may
be written by either path at any batch size and read by the other."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _coder(kind, C):
    from cbench_basic_amd.modules.prior_model.prior_coder.pgm_coder import (GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder as Coder,
                                                                            TopoGroupDynamicMaskConv2dContextModel as Ctx)
    if kind.startswith("ctxmodel"):   # "ctxmodel", or "ctxmodel-k3" for a 3x3 context window (the masked-convolution plans stop at 5x5)
        ks = int(kind.split("-k")[1]) if "-k" in kind else 5
        c = Coder(in_channels=C, default_topo_group_method="scanline", topo_group_context_model=Ctx(in_channels=C, out_channels=2 * C, kernel_size=ks))
    elif kind == "merger":
        c = Coder(in_channels=C, default_topo_group_method="scanline")
    elif kind == "merger-expand":
        c = Coder(in_channels=C, default_topo_group_method="scanline", param_merger_expand_bottleneck=True)
    else:
        c = Coder(in_channels=C, use_joint_ar_model_impl=True)
    g = torch.Generator().manual_seed(17)
    with torch.no_grad():
        for p in c.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() > 1 else 0.02))
    c = c.eval().cuda()
    c.update_state()
    return c


@pytest.mark.parametrize("kind,C,B,H,W", [("ctxmodel", 32, 1, 6, 5), ("ctxmodel", 32, 3, 4, 7), ("ctxmodel", 192, 2, 5, 6), ("merger", 32, 2, 5, 5),
                                           ("merger-expand", 16, 1, 4, 4), ("joint", 32, 2, 3, 6), ("ctxmodel", 192, 11, 3, 4), ("ctxmodel", 48, 1, 1, 1),
                                           ("ctxmodel", 192, 1, 7, 9), ("merger", 192, 2, 4, 5), ("joint", 64, 1, 5, 4), ("ctxmodel", 30, 2, 3, 5),
                                           ("ctxmodel", 192, 1, 5, 1), ("ctxmodel", 192, 2, 3, 2), ("ctxmodel", 48, 2, 1, 6),
                                           ("ctxmodel-k3", 192, 1, 4, 3), ("ctxmodel-k3", 48, 2, 5, 6),
                                           # the batched kernel (the batch as the N dimension of MFMA tiles): one and two column tiles, full and ragged
                                           ("ctxmodel", 192, 8, 3, 5), ("ctxmodel", 192, 24, 4, 4), ("ctxmodel", 192, 64, 3, 4), ("ctxmodel", 192, 33, 2, 6),
                                           ("ctxmodel", 192, 3, 5, 7), ("ctxmodel", 192, 5, 1, 4), ("ctxmodel-k3", 192, 40, 3, 3),
                                           # ... at the sizes bench.py times (64 images of 256 x 256) and the reference tests on (Kodak-shaped latents)
                                           ("ctxmodel", 192, 64, 16, 16), ("ctxmodel", 192, 8, 32, 48)])
def test_persistent_encode_equals_per_step_path(kind, C, B, H, W):
    coder = _coder(kind, C)
    g = torch.Generator().manual_seed(B * 100 + H * 10 + W)
    y = (torch.randn(B, C, H, W, generator=g) * 3).cuda()
    if kind == "joint":
        prior = torch.cat([torch.rand(B, C, H, W, generator=g) * 4 + 0.2, torch.randn(B, C, H, W, generator=g)], 1).cuda()
    else:
        prior = torch.stack([torch.randn(B, C, H, W, generator=g), torch.rand(B, C, H, W, generator=g) * 3 + 0.1], 2).reshape(B, 2 * C, H, W).cuda()
    coder.use_persistent_scanline = False
    s0, i0, y0, plan = coder._run_encode(y, prior)
    coder.use_persistent_scanline = True
    coder.persistent_scanline_max_batch = 64
    assert coder._scanline_plan(plan, prior, B) is not None
    # the persistent kernels: the generic one (any batch), the pipelined one (batches whose working set fits the LDS) and the
    # batched one (layers of its shape, up to 64 images)
    ran = []
    for kernel in ("generic", "pipelined", "batched", None):
        if kernel is None:
            os.environ.pop("BASIC_SCAN_KERNEL", None)
        else:
            os.environ["BASIC_SCAN_KERNEL"] = kernel
        try:
            try:
                s1, i1, y1, _ = coder._run_encode(y, prior)
            except (RuntimeError, ValueError) as e:
                assert kernel in ("pipelined", "batched") and "does not fit" in str(e), e   # batch too large, a layer size that is not a multiple of 4 (batched: of 32 / 64), a narrow latent
                continue
            coder._layers["scanline"][0].check()
            ms, mi = int((s0 != s1).sum()), int((i0 != i1).sum())
            print(f"{kind} C={C} B={B} {H}x{W} [{kernel}]: workgroups {coder._layers['scanline'][0].workgroups}, symbol diffs {ms}, index diffs {mi} of {s0.numel()}")
            assert ms == 0 and mi == 0
            assert torch.equal(y0, y1), float((y0 - y1).abs().max())
            # decode: the persistent launch (compute workgroups + one decoder wavefront per image stream) reproduces the encoder's
            # buffer exactly; the per-step path codes the same bytes and decodes the persistent path's stream to the same latent
            data = coder.encode(y, prior=prior)
            yhat = coder.decode(data, prior=prior)
            coder._layers["scanline"][0].check()
            assert torch.equal(yhat, y1), float((yhat - y1).abs().max())
            ran.append(kernel)
        finally:
            os.environ.pop("BASIC_SCAN_KERNEL", None)
    ks = int(kind.split("-k")[1]) if "-k" in kind else 5
    assert "generic" in ran and ("pipelined" in ran or not (C in (48, 192) and B <= 2 and W >= ks // 2 + 2 and ks <= 5))
    assert "batched" in ran or not (kind.startswith("ctxmodel") and C == 192 and W >= ks // 2 + 2)   # (other widths: merger layers that are not whole 32-row tiles)
    coder.use_persistent_scanline = False
    assert coder.encode(y, prior=prior) == data
    assert torch.equal(coder.decode(data, prior=prior), y1)
