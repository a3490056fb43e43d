"""The reference's TRAIN-mode evaluation (additive-uniform-noise proxies in every coder) -- the loss its complexity-level
search sums (latent_graph.py:1320-1395 under self.train()) -- against draws recorded from the reference itself
(tests/golden/train_mode.npz, make_golden.py::train_mode).  A random variable: means over the draws are compared, with a
tolerance of four standard errors of the difference plus 1e-3 relative.  Also the eval-mode rate estimate of a coder with
training_no_quantize_for_likelihood (the BaSIC presets' setting): deterministic, within 2e-3."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _close_in_mean(ours, ref, rel=1e-3):
    ours, ref = np.asarray(ours, np.float64), np.asarray(ref, np.float64)
    se = math.sqrt(ours.var(ddof=1) / ours.size + ref.var(ddof=1) / ref.size)
    return abs(ours.mean() - ref.mean()) <= 4 * se + rel * abs(ref.mean()), (ours.mean(), ref.mean(), se)


def test_pgm_coder_train_mode_rate_and_residual_likelihood():
    from cbench_basic_amd.modules.prior_model.prior_coder.pgm_coder import (GaussianChannelGroupMaskConv2DTopoGroupPGMPriorCoder as Coder,
                                                                            TopoGroupDynamicMaskConv2dContextModel as Ctx)
    from test_oracle_golden import pgm_case
    z = np.load(os.path.join(G, "train_mode.npz"), allow_pickle=False)
    for k in ("c0", "c1"):
        sd = pgm_case(z, k, int(z[f"{k}.seed"]))
        flag = bool(int(z[f"{k}.flag"]))
        coder = Coder(in_channels=16, default_topo_group_method="scanline", training_no_quantize_for_likelihood=flag,
                      topo_group_context_model=Ctx(in_channels=16, out_channels=32)).eval()
        missing, unexpected = coder.load_state_dict(sd, strict=False)
        assert not unexpected and not [m for m in missing if not m.startswith("_") and m != "lower_bound_scale.bound"], (missing, unexpected)
        coder = coder.cuda()
        coder.update_state()
        coder.estimate_rate = True
        y, prior = torch.from_numpy(z[f"{k}.y"]).cuda(), torch.from_numpy(z[f"{k}.prior"]).cuda()
        q = coder(y, prior=prior)
        assert torch.equal(q, torch.round(y))
        got, ref = float(coder.get_raw_cache("metric_dict")["prior_entropy"]), float(z[f"{k}.eval_prior_entropy"])
        assert abs(got - ref) <= 2e-3 * abs(ref), (k, got, ref)
        coder.rate_proxy = "noise"
        draws = []
        for d in range(24):
            torch.manual_seed(7000 + d)
            out = coder(y, prior=prior)
            assert float((out - y).abs().max()) <= 0.5 and not torch.equal(out, torch.round(y))
            draws.append(float(coder.get_raw_cache("metric_dict")["prior_entropy"]) / math.log(2))   # loss_rate is in bits
        ok, info = _close_in_mean(draws, z[f"{k}.train_loss_rate"])
        print(k, "train-mode loss_rate (ours mean, reference mean, s.e.):", info, "eval-mode:", got / math.log(2))
        assert ok, (k, info)
        # the proxy is not the eval value in disguise: the two differ by far more than the tolerance
        assert abs(np.mean(draws) - got / math.log(2)) > 20 * info[2]


def test_complexity_search_loss_in_train_mode_vs_reference_draws():
    import codec_cases as cc
    zc = cc.load()
    z = np.load(os.path.join(G, "train_mode.npz"), allow_pickle=False)
    codec, _ = cc.build_codec(zc, "b0", y_extra=dict(training_no_quantize_for_likelihood=True))
    codec = codec.cuda()
    codec.update_state()
    ec = codec.entropy_coder
    x = cc.case_input(zc, "b0").cuda()
    levels = cc.basic_cfg(zc)["levels"]
    for li in (int(v) for v in z["g.levels"]):
        ref = z[f"g.l{li}.train"]
        params = {n: ec.node_generators[n](levels[li][n]) for n in ec.complexity_level_controller_nodes}
        # eval-mode forward metrics at this setting with the residual likelihood (deterministic)
        codec.set_complex_level(li)
        codec.reset_all_cache()
        codec(x)
        met = dict(prior_entropy=float(ec.get_raw_cache("metric_dict")["prior_entropy"]))
        want = dict(zip((str(n) for n in z[f"g.l{li}.eval_metric_names"]), (float(v) for v in z[f"g.l{li}.eval_metric_values"])))
        assert abs(met["prior_entropy"] - want["prior_entropy"]) <= 2e-3 * abs(want["prior_entropy"]), (li, met, want)
        draws = []
        for d in range(16):
            torch.manual_seed(9000 + 16 * li + d)
            c, p = ec._test_dataset_complexity_performance([x], performance_method="loss", complexity_method="FLOPs", loss_mode="train", **params)
            draws.append((c, p))
        draws = np.array(draws)
        assert np.allclose(draws[:, 0], ref[:, 0], rtol=1e-9), (li, draws[0, 0], ref[0, 0])     # operation counters per input element
        ok, info = _close_in_mean(draws[:, 1], ref[:, 1], rel=2e-4)
        c_eval, p_eval = ec._test_dataset_complexity_performance([x], performance_method="loss", complexity_method="FLOPs", **params)
        print(f"level {li}: train-mode loss (ours mean, reference mean, s.e.) {info}; eval-mode loss {p_eval:.5f}")
        assert ok, (li, info)
        assert draws[:, 1].std() > 0 and c_eval == draws[0, 0]
    # the constructor switch routes post_training_process to the same evaluation
    codec2, _ = cc.build_codec(zc, "b0", complexity_level_greedy_search_loss_mode="train")
    assert codec2.entropy_coder.complexity_level_greedy_search_loss_mode == "train"
    with pytest.raises(ValueError):
        cc.build_codec(zc, "b0", complexity_level_greedy_search_loss_mode="noise")
