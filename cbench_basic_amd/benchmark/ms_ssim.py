"""Multi-scale structural similarity, restated from the published algorithm (Wang, Simoncelli, Bovik 2003) in the form the reference
consumes it: ``pytorch_msssim.ms_ssim(x_hat, x, data_range=..., size_average=...)`` (cbench/modules/entropy_coder/latent_graph.py:14,92-96;
cbench/benchmark/metrics/pytorch_distortion.py:8,17-18).  ``pytorch_msssim`` itself is a PyPI dependency of the reference that is NOT
under /root/reference, so this file is PARITY-UNPINNED: it follows that package's documented defaults -- 11-tap Gaussian window
(sigma 1.5) applied separably WITHOUT padding per channel, K = (0.01, 0.03), five scales with weights
(0.0448, 0.2856, 0.3001, 0.2363, 0.1333), 2 x 2 average pooling between scales (odd sizes padded), contrast-structure terms of the
first four scales and the full SSIM of the last, each clamped at zero, combined as a weighted product per channel, then the mean over
channels -- but no output of the package could be compared here.  Plain torch ops on whatever device the images live on: a
forward / benchmark metric, not part of the coding path."""
import torch
import torch.nn.functional as F

_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _gauss_window(size, sigma, device, dtype):
    coords = torch.arange(size, dtype=dtype, device=device) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _filter(x, win):
    """Separable Gaussian filter, 'valid' (no padding), the same window for every channel."""
    C = x.shape[1]
    k = win.numel()
    out = x
    if out.shape[2] >= k:
        out = F.conv2d(out, win.reshape(1, 1, k, 1).repeat(C, 1, 1, 1), groups=C)
    if out.shape[3] >= k:
        out = F.conv2d(out, win.reshape(1, 1, 1, k).repeat(C, 1, 1, 1), groups=C)
    return out


def _ssim_terms(x, y, win, data_range, K):
    c1, c2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = _filter(x, win), _filter(y, win)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = _filter(x * x, win) - mu1_sq
    s2 = _filter(y * y, win) - mu2_sq
    s12 = _filter(x * y, win) - mu12
    cs_map = (2 * s12 + c2) / (s1 + s2 + c2)
    ssim_map = ((2 * mu12 + c1) / (mu1_sq + mu2_sq + c1)) * cs_map
    return ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)   # [B, C] each


def ms_ssim(x, y, data_range=1.0, size_average=True, win_size=11, win_sigma=1.5, weights=_WEIGHTS, K=(0.01, 0.03)):
    """[B, C, H, W] images -> MS-SSIM per image ([B]) or its mean (size_average).  The smaller side must exceed
    (win_size - 1) * 2**4 = 160 pixels (four halvings must leave room for the window), as the package requires."""
    if x.shape != y.shape or x.dim() != 4:
        raise ValueError("ms_ssim takes two [B, C, H, W] tensors of one shape")
    if min(x.shape[2:]) <= (win_size - 1) * 2 ** (len(weights) - 1):
        raise ValueError(f"image side should be larger than {(win_size - 1) * 2 ** (len(weights) - 1)} for {len(weights)} scales of a {win_size}-tap window")
    x, y = x.float(), y.float()
    win = _gauss_window(win_size, win_sigma, x.device, x.dtype)
    w = torch.tensor(weights, device=x.device, dtype=x.dtype)
    terms = []
    for i in range(len(weights)):
        ssim_c, cs = _ssim_terms(x, y, win, data_range, K)
        if i < len(weights) - 1:
            terms.append(torch.relu(cs))
            pad = [s % 2 for s in x.shape[2:]]
            x = F.avg_pool2d(x, kernel_size=2, padding=pad)
            y = F.avg_pool2d(y, kernel_size=2, padding=pad)
    terms.append(torch.relu(ssim_c))
    val = torch.prod(torch.stack(terms, 0) ** w.reshape(-1, 1, 1), dim=0).mean(1)   # [B]
    return val.mean() if size_average else val
