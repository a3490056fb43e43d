"""Frequency-table construction of the continuous-distribution ANS prior coders --
ContinuousDistributionANSPriorCoder._get_ans_params (cbench/modules/prior_model/prior_coder/
torch_ans.py:284-310) and the quantiser of TorchANSPriorCoder._data_preprocess (:105-161).

The tables are a ONE-TIME product of update_state() (torch_ans.py:237-251).  They are evaluated on
the host in float32 with exactly the reference's torch operations and operation order (icdf, cdf
differences, log, softmax, *2^16, clamp_min(1), int32 truncation) so that they are bit-identical
to the reference's CPU tables (pinned by tests/golden/gauss_pgm_tables.npz), then quantised and
uploaded by libbasic_hip (basic_rans_tables_from_freqs).
"""
import numpy as np
import torch
import torch.distributions as D


def gaussian_ans_params(scale_table: torch.Tensor, freq_precision=16, lower_bound_scale=0.11):
    """Returns (freqs int32 [n, max_len], num_symbols int32 [n], offsets int32 [n]) for zero-mean
    Gaussians with the given scales (GaussianPGMPriorCoderImpl._init_dist_params, pgm_coder.py:780-788)."""
    freq_cnt = 1 << freq_precision
    tail_mass = torch.tensor([0.5 / freq_cnt])
    counts, num_symbols, offsets = [], [], []
    bound = torch.tensor([float(lower_bound_scale)])
    for scale in scale_table.float().cpu():
        # _params_to_dist (pgm_coder.py:757-778): Normal(mean=0, scale=max(scale, 0.11)) with a batch dim
        means = torch.zeros(1)
        scales = torch.max(scale.reshape(1), bound)
        dist = D.Normal(means, scales)
        dist_min = int(dist.icdf(tail_mass).floor().item())
        dist_max = int(dist.icdf(1 - tail_mass).ceil().item())
        offsets.append(dist_min)
        num_symbols.append(dist_max - dist_min + 1)
        pts = torch.arange(dist_min - 1, dist_max + 1).type_as(dist.mean) + 0.5
        logprob = (dist.cdf(pts[1:].unsqueeze(0)) - dist.cdf(pts[:-1].unsqueeze(0))).log()[0]
        pmf = torch.softmax(logprob, dim=-1)
        cnt = (pmf * freq_cnt).clamp_min(1)
        counts.append(cnt.detach().cpu().contiguous().numpy().astype(np.int32))
    freqs = np.zeros((len(counts), max(len(c) for c in counts)), dtype=np.int32)
    for i, c in enumerate(counts):
        freqs[i, : len(c)] = c
    return freqs, np.array(num_symbols, dtype=np.int32), np.array(offsets, dtype=np.int32)
