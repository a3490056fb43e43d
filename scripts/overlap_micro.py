#!/usr/bin/env python3
"""GPU micro-probe: how long do the batched rANS kernels take ALONE and BESIDE an MFMA convolution train running on a
second HIP stream (and what does the convolution train lose)?"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbench_basic_amd.nn import kernels as K
from cbench_basic_amd.modules.prior_model.prior_coder.compressai_coder import gaussian_conditional_tables, get_scale_table

dev = torch.device("cuda", 0)
B = int(os.environ.get("B", "128"))
cdf, lengths, offsets = gaussian_conditional_tables(get_scale_table())
tab = K.RansTables(cdfs=cdf, cdf_sizes=lengths, offsets=offsets, precision=16, bypass=True, bypass_precision=4)
n = 192 * 16 * 16
g = torch.Generator().manual_seed(0)
idx = torch.randint(0, 40, (B, n), generator=g, dtype=torch.int32).to(dev)
sym = torch.round(torch.randn(B, n, generator=g) * 2).to(torch.int32).to(dev)
seg = (torch.arange(B + 1, dtype=torch.int64) * n).to(dev)
w = torch.randn(128, 128, 5, 5) * 0.02
gam = 0.1 * torch.eye(128) + 0.01
plan = K.ConvPlan(w, torch.zeros(128), 2, 2, 0, False, K.ACT_GDN, gam, torch.ones(128))
x = torch.randn(B, 128, 128, 128, device=dev)
y = plan(x)
s_r, s_c = torch.cuda.Stream(), torch.cuda.Stream()


def rans_enc():
    return tab.encode_batch(sym.reshape(-1), idx.reshape(-1), seg, n + 2)


words, nwords = rans_enc()
h = tab.encode_batch_begin(sym.reshape(-1), idx.reshape(-1), n)
host, off = tab.encode_batch_end(h)
d_words = torch.from_numpy(host.view(np.int32).copy()).to(dev)
d_woff = torch.from_numpy(off).to(dev)


def rans_dec():
    return tab.decode_batch(d_words, d_woff, idx.reshape(-1), seg)


out, _, _ = rans_dec()
assert torch.equal(out.reshape(B, n), sym)
torch.cuda.synchronize()


def timed(fn, stream, reps=1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
    return e0, e1


for name, fn in (("encode", rans_enc), ("decode", rans_dec)):
    e = timed(fn, s_r)
    torch.cuda.synchronize()
    alone = e[0].elapsed_time(e[1])
    c = timed(lambda: plan(x, out=y), s_c, 8)
    torch.cuda.synchronize()
    conv_alone = c[0].elapsed_time(c[1])
    c = timed(lambda: plan(x, out=y), s_c, 8)
    time.sleep(0.002)
    e = timed(fn, s_r)
    torch.cuda.synchronize()
    print(f"WPB={os.environ.get('BASIC_RANS_WPB', '1')} B={B} rANS {name}: alone {alone:.2f} ms, beside the conv train {e[0].elapsed_time(e[1]):.2f} ms; "
          f"conv train (8 launches) alone {conv_alone:.2f} ms, beside rANS {c[0].elapsed_time(c[1]):.2f} ms", flush=True)
