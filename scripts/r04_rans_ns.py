#!/usr/bin/env python3
"""Stand-alone rANS chains: ns per symbol of the fast encoder / decoder on ONE stream and on 64 streams (HIP-event time of the
launches), for a low-entropy and a high-entropy symbol mix on a 64-row table set (precision 16, bypass on)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbench_basic_amd.nn.kernels import RansTables

rng = np.random.default_rng(0)
nd, ns = 64, 62   # 63 cdf entries per row: narrow rows (one 64-lane probe); WIDE=1: 66 symbols, every row takes the two-level search
if os.environ.get("WIDE"):
    ns = 66
x = np.arange(ns) - ns // 2
rows = []
for r in range(nd):   # discretised Gaussians, scale 0.3 .. 8
    sc = 0.3 * (8 / 0.3) ** (r / (nd - 1))
    p = np.exp(-0.5 * (x / sc) ** 2) + 1e-6
    rows.append(np.maximum(1, np.round(p / p.sum() * 60000)).astype(np.int32))
freqs = np.stack(rows)
T = RansTables(freqs=freqs, nsym=np.full(nd, ns, np.int32), offsets=np.full(nd, -(ns // 2), np.int32))


def run(name, nstreams, n, lo_row, hi_row):
    idx = rng.integers(lo_row, hi_row, nstreams * n).astype(np.int32)
    sc = 0.3 * (8 / 0.3) ** (idx / (nd - 1))
    sym = np.clip(np.round(rng.normal(0, 1, idx.size) * sc), -(ns // 2) + 1, ns - ns // 2 - 3).astype(np.int32)
    seg = (np.arange(nstreams + 1) * n).astype(np.int64)
    d_sym, d_idx, d_seg = torch.from_numpy(sym).cuda(), torch.from_numpy(idx).cuda(), torch.from_numpy(seg).cuda()
    slot = n * 3 + 4
    for _ in range(2):
        words, nwords = T.encode_batch(d_sym, d_idx, d_seg, slot)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    reps = 5
    ev[0].record()
    for _ in range(reps):
        words, nwords = T.encode_batch(d_sym, d_idx, d_seg, slot)
    ev[1].record()
    torch.cuda.synchronize()
    w, nw = words.cpu().numpy().view(np.uint32), nwords.cpu().numpy()
    streams = [w[i, slot - nw[i]:] for i in range(nstreams)]
    woff = np.concatenate([[0], np.cumsum([s.size for s in streams])]).astype(np.int64)
    allw = torch.from_numpy(np.concatenate(streams).view(np.int32)).cuda()
    d_woff = torch.from_numpy(woff).cuda()
    for _ in range(2):
        out, _, _ = T.decode_batch(allw, d_woff, d_idx, d_seg)
    ev[2].record()
    for _ in range(reps):
        out, _, _ = T.decode_batch(allw, d_woff, d_idx, d_seg)
    ev[3].record()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), sym)
    enc, dec = ev[0].elapsed_time(ev[1]) / reps, ev[2].elapsed_time(ev[3]) / reps
    bits = 32.0 * sum(s.size for s in streams) / (nstreams * n)
    print(f"{name:28s} {nstreams:3d} stream(s) x {n:7d} symbols, {bits:5.2f} bits/symbol: encode {enc:7.3f} ms = {enc * 1e6 / n:6.1f} ns/symbol, "
          f"decode {dec:7.3f} ms = {dec * 1e6 / n:6.1f} ns/symbol", flush=True)


run("low entropy (rows 0-15)", 1, 294912, 0, 16)
run("mixed (rows 0-63)", 1, 294912, 0, 64)
run("high entropy (rows 40-63)", 1, 294912, 40, 64)
run("mixed (rows 0-63)", 64, 49152, 0, 64)
