"""Thin torch-tensor front-ends of the C-ABI compute entry points (include/basic_hip.h).

PyTorch is used here for device memory and streams only; every operation below is a
hand-written HIP kernel.  All functions require CUDA(HIP) tensors and raise otherwise --
there is no CPU implementation behind them.
"""
import ctypes

import numpy as np
import torch

from .. import _lib

ACT_NONE, ACT_RELU, ACT_LEAKY_RELU, ACT_GDN, ACT_IGDN = 0, 1, 2, 3, 4


def _dev(t, dtype=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.BasicHipError("cbench_basic_amd kernels need tensors on the MI355X (got a CPU tensor); "
                                 "there is no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    return t.contiguous()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32(a):
    return np.ascontiguousarray(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a, dtype=np.float32)


class ConvPlan:
    """Packed weights + fused epilogue of one conv / deconv layer (basic_conv_plan_*)."""

    def __init__(self, weight, bias, stride, padding, output_padding=0, transposed=False, act=ACT_NONE,
                 gamma=None, beta=None, cin_active=None, cout_active=None):
        w = _f32(weight)
        self.transposed = bool(transposed)
        cin, cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
        self.cin = cin if cin_active is None else int(cin_active)
        self.cout = cout if cout_active is None else int(cout_active)
        b = _f32(bias) if bias is not None else None
        g = _f32(gamma) if gamma is not None else None
        be = _f32(beta) if beta is not None else None
        if g is not None and g.shape[0] != cout:
            # effective gamma/beta given for the active slice only: embed in a [cout][cout] frame
            gf = np.zeros((cout, cout), np.float32); gf[: g.shape[0], : g.shape[1]] = g
            bf = np.ones((cout,), np.float32); bf[: be.shape[0]] = be
            g, be = gf, bf
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().basic_conv_plan_create(
            w.ctypes.data, b.ctypes.data if b is not None else None, cin, cout, w.shape[2], int(stride), int(padding),
            int(output_padding), int(self.transposed), int(act), g.ctypes.data if g is not None else None,
            be.ctypes.data if be is not None else None, self.cin, self.cout, ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().basic_conv_plan_destroy(h)
            except Exception:
                pass

    def out_hw(self, h, w):
        oh, ow = ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.lib().basic_conv_plan_out_hw(self._h, int(h), int(w), ctypes.byref(oh), ctypes.byref(ow)))
        return oh.value, ow.value

    def flops(self, batch, h, w):
        return int(_lib.lib().basic_conv_plan_flops(self._h, int(batch), int(h), int(w)))

    def launches(self, batch, h, w):
        """Kernel launches of one forward call on a [batch, cin, h, w] input."""
        return int(_lib.lib().basic_conv_plan_launches(self._h, int(batch), int(h), int(w)))

    def __call__(self, x, out=None):
        x = _dev(x, torch.float32)
        B, C, H, W = x.shape
        if C != self.cin:
            raise ValueError(f"conv plan expects {self.cin} input channels, got {C}")
        oh, ow = self.out_hw(H, W)
        if out is None:
            out = torch.empty((B, self.cout, oh, ow), device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().basic_conv_forward_dev(self._h, x.data_ptr(), B, H, W, out.data_ptr(), _stream()))
        return out


class MaskedConvPlan:
    """basic_mconv_plan_*: topo-group masked conv evaluated at a position list."""

    def __init__(self, weight, bias, in_groups, out_groups, allow_same, act=ACT_NONE):
        w = _f32(weight)
        b = _f32(bias) if bias is not None else None
        self.cout, self.cin, self.k = w.shape[0], w.shape[1], w.shape[2]
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().basic_mconv_plan_create(w.ctypes.data, b.ctypes.data if b is not None else None, self.cin,
                                                      self.cout, self.k, int(in_groups), int(out_groups), int(bool(allow_same)),
                                                      int(act), ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().basic_mconv_plan_destroy(h)
            except Exception:
                pass

    def __call__(self, x, topo_in, topo_out, pos, out, out_offset=0, step=None, first_step=None, in_perm=None, out_perm=None):
        """step / first_step: the coding-loop variant that only evaluates the (output group, position) pairs of the current
        step.  in_perm / out_perm (int32 [H*W]): position permutation inside the planes of x / out (private buffers of a
        chain of 1x1 layers; see basic_mconv_forward_ex_dev)."""
        x = _dev(x, torch.float32)
        B, C, H, W = x.shape
        topo_in, topo_out, pos = _dev(topo_in, torch.int32), _dev(topo_out, torch.int32), _dev(pos, torch.int32)
        out = _dev(out, torch.float32)
        if step is not None:
            first_step = _dev(first_step, torch.int32)
        in_perm = _dev(in_perm, torch.int32) if in_perm is not None else None
        out_perm = _dev(out_perm, torch.int32) if out_perm is not None else None
        _lib.check(_lib.lib().basic_mconv_forward_ex_dev(
            self._h, x.data_ptr(), topo_in.data_ptr(), topo_out.data_ptr(), B, H, W, pos.data_ptr(), pos.numel(), out.data_ptr(),
            out.shape[1], int(out_offset), int(step is not None), int(step) if step is not None else 0,
            first_step.data_ptr() if step is not None else None, in_perm.data_ptr() if in_perm is not None else None,
            out_perm.data_ptr() if out_perm is not None else None, _stream()))
        return out


class RansTables:
    """Device-resident CDF tables (basic_rans_tables_*) for the batched stream coder."""

    def __init__(self, freqs=None, nsym=None, offsets=None, cdfs=None, cdf_sizes=None, precision=16, bypass=True,
                 bypass_precision=4):
        h = ctypes.c_void_p()
        off = np.ascontiguousarray(offsets, dtype=np.int32)
        if cdfs is not None:
            c = np.ascontiguousarray(cdfs, dtype=np.int32)
            s = np.ascontiguousarray(cdf_sizes, dtype=np.int32)
            _lib.check(_lib.lib().basic_rans_tables_from_cdfs(c.ctypes.data, c.shape[0], c.shape[1], s.ctypes.data, off.ctypes.data,
                                                              precision, int(bypass), bypass_precision, ctypes.byref(h)))
        else:
            f = np.ascontiguousarray(freqs, dtype=np.int32)
            n = np.ascontiguousarray(nsym, dtype=np.int32)
            _lib.check(_lib.lib().basic_rans_tables_from_freqs(f.ctypes.data, f.shape[0], f.shape[1], n.ctypes.data, off.ctypes.data,
                                                               precision, int(bypass), bypass_precision, ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().basic_rans_tables_destroy(h)
            except Exception:
                pass

    def get_cdfs(self):
        rows, mx = ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.lib().basic_rans_tables_info(self._h, ctypes.byref(rows), ctypes.byref(mx)))
        out = np.zeros((rows.value, mx.value), dtype=np.int32)
        _lib.check(_lib.lib().basic_rans_tables_get_cdfs(self._h, out.ctypes.data, mx.value))
        return out

    def encode_batch(self, symbols, indexes, seg, slot_words=None):
        """symbols/indexes int32 [total]; seg int64 [nstreams+1] -> (words [nstreams, slot], nwords [nstreams])."""
        symbols, indexes, seg = _dev(symbols, torch.int32), _dev(indexes, torch.int32), _dev(seg, torch.int64)
        ns = seg.numel() - 1
        if slot_words is None:
            raise ValueError("slot_words is required")
        if symbols.numel() == 0:  # every stream empty: the C ABI still wants valid pointers
            symbols = indexes = torch.zeros((1,), device=seg.device, dtype=torch.int32)
        words = torch.empty((ns, slot_words), device=symbols.device, dtype=torch.int32)
        nwords = torch.empty((ns,), device=symbols.device, dtype=torch.int32)
        _lib.check(_lib.lib().basic_rans_encode_batch_dev(self._h, symbols.data_ptr(), indexes.data_ptr(), seg.data_ptr(), ns,
                                                          words.data_ptr(), slot_words, nwords.data_ptr(), _stream()))
        return words, nwords

    # ---- batched streams <-> host bytes.  begin() only enqueues GPU work (encode, device-side offsets, compaction);
    # end() is the single host synchronisation, so a caller can launch more kernels in between.
    def encode_batch_begin(self, symbols, indexes, n_per_stream, slot=None):
        symbols, indexes = _dev(symbols, torch.int32), _dev(indexes, torch.int32)
        ns = symbols.numel() // n_per_stream
        seg = torch.arange(ns + 1, device=symbols.device, dtype=torch.int64) * n_per_stream
        slot = slot or n_per_stream + 2  # the reference's own buffer size (rans64.cpp:240)
        words, nwords = self.encode_batch(symbols, indexes, seg, slot)
        d_off = torch.zeros((ns + 1,), device=symbols.device, dtype=torch.int64)
        torch.cumsum(nwords.clamp(min=0), 0, out=d_off[1:])
        packed = torch.empty((ns * slot,), device=symbols.device, dtype=torch.int32)
        _lib.check(_lib.lib().basic_rans_compact_streams_dev(words.data_ptr(), slot, nwords.data_ptr(), d_off.data_ptr(), ns,
                                                             packed.data_ptr(), _stream()))
        return dict(symbols=symbols, indexes=indexes, n=n_per_stream, ns=ns, slot=slot, nwords=nwords, packed=packed, keep=words)

    def _pinned(self, which, nbytes):
        """Page-locked staging buffer owned by this table set (D2H of encoded words / H2D of words to decode): DMA at
        full PCIe rate instead of the pageable path's bounce copies.  Grows geometrically, one per direction."""
        buf = getattr(self, "_pin_" + which, None)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty((max(nbytes, 1 << 16) * 3 // 2,), dtype=torch.uint8, pin_memory=True)
            setattr(self, "_pin_" + which, buf)
        return buf

    def encode_batch_end(self, h):
        """-> (uint32 words of all streams back to back (numpy), int64 word offsets [ns + 1]).  The words live in this
        table set's pinned staging buffer: valid until its next encode_batch_end()."""
        nw = h["nwords"].cpu().numpy()
        if (nw < 0).any():  # bypass-heavy data overflowed the reference's bound: redo with the guaranteed one
            h = self.encode_batch_begin(h["symbols"], h["indexes"], h["n"], slot=3 * h["n"] + 4)
            nw = h["nwords"].cpu().numpy()
        off = np.zeros(h["ns"] + 1, dtype=np.int64)
        np.cumsum(nw, out=off[1:])
        n = int(off[-1])
        stage = self._pinned("out", 4 * n)[: 4 * n].view(torch.int32)
        stage.copy_(h["packed"][:n], non_blocking=True)
        torch.cuda.current_stream(h["packed"].device).synchronize()
        return stage.numpy().view(np.uint32), off

    def encode_batch_to_bytes(self, symbols, indexes, n_per_stream):
        """Equal-length streams (one per image): returns a list of ``bytes`` (reference py::bytes)."""
        host, off = self.encode_batch_end(self.encode_batch_begin(symbols, indexes, n_per_stream))
        return [host[off[i]:off[i + 1]].tobytes() for i in range(len(off) - 1)]

    def _stage_in(self, nwords):
        """uint32 numpy view of nwords words of the pinned H2D staging buffer (waits for the previous DMA out of it)."""
        ev = getattr(self, "_pin_in_event", None)
        if ev is not None:
            ev.synchronize()
        return self._pinned("in", 4 * max(nwords, 1))[: 4 * nwords].view(torch.int32)

    def decode_batch_from_frame(self, data, indexes, n_per_stream):
        """Framed body (frame_streams / write_body) -> int32 symbols like ``indexes``: the payloads are unframed by the C
        helper straight into the pinned staging buffer and DMA'd from there."""
        h, w, n = frame_header(data)
        stage = self._stage_in((len(data) - 12 - 4 * n) // 4)
        words, word_off, _ = unframe_streams(data, out=stage.numpy().view(np.uint32))
        return self._decode_staged(stage[: words.size], word_off, indexes, n_per_stream)

    def decode_batch_from_words(self, host_words, word_off, indexes, n_per_stream):
        """Streams given as one uint32 array + word offsets (see unframe_streams): int32 symbols like ``indexes``."""
        stage = self._stage_in(host_words.size)
        stage.numpy()[:] = host_words.view(np.int32)
        return self._decode_staged(stage, word_off, indexes, n_per_stream)

    def _decode_staged(self, stage, word_off, indexes, n_per_stream):
        indexes = _dev(indexes, torch.int32)
        ns = len(word_off) - 1
        d_words = stage.to(indexes.device, non_blocking=True)
        self._pin_in_event = torch.cuda.Event()
        self._pin_in_event.record(torch.cuda.current_stream(indexes.device))
        d_woff = torch.from_numpy(np.ascontiguousarray(word_off, dtype=np.int64)).to(indexes.device)
        seg = torch.arange(ns + 1, device=indexes.device, dtype=torch.int64) * n_per_stream
        out, _, _ = self.decode_batch(d_words, d_woff, indexes, seg)
        return out

    def decode_batch_from_bytes(self, strings, indexes, n_per_stream):
        """Inverse of encode_batch_to_bytes: int32 symbols shaped like ``indexes`` (device)."""
        for s in strings:
            if len(s) < 8 or len(s) % 4:
                raise ValueError("rANS stream must hold >= 2 whole 32-bit words")
        lens = np.array([len(s) // 4 for s in strings], dtype=np.int64)
        woff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        host = np.frombuffer(b"".join(strings), dtype=np.uint32).copy()
        return self.decode_batch_from_words(host, woff, indexes, n_per_stream)

    def decode_batch(self, words, word_off, indexes, seg, out=None, state=None, pos=None):
        words, word_off = _dev(words, torch.int32), _dev(word_off, torch.int64)
        indexes, seg = _dev(indexes, torch.int32), _dev(seg, torch.int64)
        ns = seg.numel() - 1
        if out is None:
            out = torch.empty_like(indexes)
        if indexes.numel() == 0:  # every stream empty: nothing to decode
            return out, state, pos
        if state is None:
            state = torch.zeros((ns,), device=words.device, dtype=torch.int64)
            pos = torch.full((ns,), -1, device=words.device, dtype=torch.int64)
        _lib.check(_lib.lib().basic_rans_decode_batch_dev(self._h, words.data_ptr(), word_off.data_ptr(), indexes.data_ptr(),
                                                          seg.data_ptr(), ns, out.data_ptr(), state.data_ptr(), pos.data_ptr(), _stream()))
        return out, state, pos


def gc_quantize_index(y, scales, table, bound=0.11, want_yhat=True):
    y, scales, table = _dev(y, torch.float32), _dev(scales, torch.float32), _dev(table, torch.float32)
    sym = torch.empty(y.shape, device=y.device, dtype=torch.int32)
    idx = torch.empty(y.shape, device=y.device, dtype=torch.int32)
    yhat = torch.empty_like(y) if want_yhat else None
    _lib.check(_lib.lib().basic_gc_quantize_index_dev(y.data_ptr(), scales.data_ptr(), y.numel(), table.data_ptr(), table.numel(),
                                                      float(bound), sym.data_ptr(), idx.data_ptr(),
                                                      yhat.data_ptr() if yhat is not None else None, _stream()))
    return sym, idx, yhat


def eb_quantize_index(z, medians):
    z, medians = _dev(z, torch.float32), _dev(medians, torch.float32)
    B, C = z.shape[0], z.shape[1]
    hw = z.numel() // (B * C)
    sym = torch.empty(z.shape, device=z.device, dtype=torch.int32)
    idx = torch.empty(z.shape, device=z.device, dtype=torch.int32)
    zhat = torch.empty_like(z)
    _lib.check(_lib.lib().basic_eb_quantize_index_dev(z.data_ptr(), medians.data_ptr(), B, C, hw, sym.data_ptr(), idx.data_ptr(),
                                                      zhat.data_ptr(), _stream()))
    return sym, idx, zhat


def eb_dequantize(sym, medians):
    sym, medians = _dev(sym, torch.int32), _dev(medians, torch.float32)
    B, C = sym.shape[0], sym.shape[1]
    hw = sym.numel() // (B * C)
    zhat = torch.empty(sym.shape, device=sym.device, dtype=torch.float32)
    _lib.check(_lib.lib().basic_eb_dequantize_dev(sym.data_ptr(), medians.data_ptr(), B, C, hw, zhat.data_ptr(), _stream()))
    return zhat


def i32_to_f32(sym):
    sym = _dev(sym, torch.int32)
    out = torch.empty(sym.shape, device=sym.device, dtype=torch.float32)
    _lib.check(_lib.lib().basic_i32_to_f32_dev(sym.data_ptr(), sym.numel(), out.data_ptr(), _stream()))
    return out


def mse_per_image(a, b):
    a, b = _dev(a, torch.float32), _dev(b, torch.float32)
    out = torch.empty((a.shape[0],), device=a.device, dtype=torch.float32)
    _lib.check(_lib.lib().basic_mse_per_image_dev(a.data_ptr(), b.data_ptr(), a.shape[0], a.numel() // a.shape[0], out.data_ptr(), _stream()))
    return out


def gauss_nll_per_image(q, scales_or_params, interleaved, scale_bound=0.11, likelihood_bound=1e-9):
    """Rate estimate (nats per image) of a Gaussian-coded latent; see basic_gauss_nll_per_image_dev."""
    q, sp = _dev(q, torch.float32), _dev(scales_or_params, torch.float32)
    B, C = q.shape[0], q.shape[1]
    hw = q.numel() // (B * C)
    out = torch.empty((B,), device=q.device, dtype=torch.float32)
    mode = 2 if interleaved == "round_residual" else int(bool(interleaved))
    _lib.check(_lib.lib().basic_gauss_nll_per_image_dev(q.data_ptr(), sp.data_ptr(), B, C, hw, mode, float(scale_bound),
                                                        float(likelihood_bound), out.data_ptr(), _stream()))
    return out


def eb_nll_per_image(zq, coef, likelihood_bound=1e-9):
    zq, coef = _dev(zq, torch.float32), _dev(coef, torch.float32)
    B, C = zq.shape[0], zq.shape[1]
    hw = zq.numel() // (B * C)
    out = torch.empty((B,), device=zq.device, dtype=torch.float32)
    _lib.check(_lib.lib().basic_eb_nll_per_image_dev(zq.data_ptr(), coef.data_ptr(), B, C, hw, float(likelihood_bound), out.data_ptr(), _stream()))
    return out


def frame_streams(host_words, word_off, shape_hw) -> bytes:
    """write_body (compressai_coder.py:75-84) for one stream per image, done by the C helper."""
    host_words = np.ascontiguousarray(host_words, dtype=np.uint32)
    word_off = np.ascontiguousarray(word_off, dtype=np.int64)
    n = len(word_off) - 1
    cap = 12 + 4 * n + 4 * int(word_off[-1] - word_off[0])
    out = np.empty(cap, dtype=np.uint8)
    out_len = ctypes.c_int64()
    _lib.check(_lib.lib().basic_frame_streams(host_words.ctypes.data, word_off.ctypes.data, n, int(shape_hw[0]), int(shape_hw[1]),
                                              out.ctypes.data, cap, ctypes.byref(out_len)))
    return out[: out_len.value].tobytes()


def frame_streams_size(word_off) -> int:
    """Bytes frame_streams() writes for these streams."""
    n = len(word_off) - 1
    return 12 + 4 * n + 4 * int(word_off[-1] - word_off[0])


def frame_streams_into(host_words, word_off, shape_hw, address: int, capacity: int) -> int:
    """frame_streams() straight into caller memory (e.g. the final bytes object of a codec): returns the length."""
    host_words = np.ascontiguousarray(host_words, dtype=np.uint32)
    word_off = np.ascontiguousarray(word_off, dtype=np.int64)
    out_len = ctypes.c_int64()
    _lib.check(_lib.lib().basic_frame_streams(host_words.ctypes.data, word_off.ctypes.data, len(word_off) - 1, int(shape_hw[0]),
                                              int(shape_hw[1]), address, capacity, ctypes.byref(out_len)))
    return out_len.value


def frame_header(data: bytes):
    """(h, w, n streams) of a framed body."""
    buf = np.frombuffer(data, dtype=np.uint8)
    h, w, n = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_int()
    _lib.check(_lib.lib().basic_unframe_streams(buf.ctypes.data, len(data), ctypes.byref(h), ctypes.byref(w), ctypes.byref(n),
                                                None, 0, None))
    return h.value, w.value, n.value


def unframe_streams(data: bytes, out=None):
    """read_body (compressai_coder.py:63-72) for one stream per image: (uint32 words, word offsets, (h, w)).
    ``out``: optional uint32 array of at least (len(data) - 12 - 4 n) / 4 words to receive the payloads."""
    buf = np.frombuffer(data, dtype=np.uint8)
    h, w, n = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_int()
    _lib.check(_lib.lib().basic_unframe_streams(buf.ctypes.data, len(data), ctypes.byref(h), ctypes.byref(w), ctypes.byref(n),
                                                None, 0, None))
    word_off = np.empty(n.value + 1, dtype=np.int64)
    payload = max((len(data) - 12 - 4 * n.value) // 4, 0)
    words = out if out is not None and payload > 0 and out.size >= payload else np.empty(max(payload, 1), dtype=np.uint32)
    _lib.check(_lib.lib().basic_unframe_streams(buf.ctypes.data, len(data), ctypes.byref(h), ctypes.byref(w), ctypes.byref(n),
                                                word_off.ctypes.data, n.value, words.ctypes.data))
    return words[: int(word_off[-1])], word_off, (h.value, w.value)


class HyperpriorSession:
    """basic_hp_session_*: the fused compress / decompress entry points of the plain hyperprior latent graph (one C call
    per batch instead of a Python walk over the graph).  Borrows the layer plans and table sets it is given (kept alive
    here); one session serves one call at a time."""

    def __init__(self, g_a, h_a, h_s, g_s, eb_medians, z_tables, scale_table, scale_bound, y_tables):
        self._keep = (list(g_a), list(h_a), list(h_s), list(g_s), z_tables, y_tables)
        arrs = []
        for plans in self._keep[:4]:
            a = (ctypes.c_void_p * len(plans))(*[p._h for p in plans])
            arrs.append(a)
        med = np.ascontiguousarray(eb_medians.detach().cpu().numpy() if isinstance(eb_medians, torch.Tensor) else eb_medians, dtype=np.float32)
        tab = np.ascontiguousarray(scale_table.detach().cpu().numpy() if isinstance(scale_table, torch.Tensor) else scale_table, dtype=np.float32)
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().basic_hp_session_create(arrs[0], len(g_a), arrs[1], len(h_a), arrs[2], len(h_s), arrs[3], len(g_s),
                                                      med.ctypes.data, med.size, z_tables._h, tab.ctypes.data, tab.size,
                                                      float(scale_bound), y_tables._h, ctypes.byref(h)))
        self._h = h
        self.key = tuple(p._h.value for plans in self._keep[:4] for p in plans) + (z_tables._h.value, y_tables._h.value)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().basic_hp_session_destroy(h)
            except Exception:
                pass

    def set_rans_waves(self, waves_per_block):
        _lib.check(_lib.lib().basic_hp_session_set_rans_waves(self._h, int(waves_per_block)))

    def set_transform_token(self, enable):
        """False / 0: no ordering; True / 1: the transform phases of all sessions one at a time; 2: two at a time."""
        _lib.check(_lib.lib().basic_hp_session_set_transform_token(self._h, int(enable)))

    def encode(self, x) -> bytes:
        """x: float32 [B, C, H, W], on the GPU or on the host (uploaded inside the call; pinned memory goes at DMA rate)."""
        if x.dtype != torch.float32:
            raise TypeError(f"expected float32, got {x.dtype}")
        x = x.contiguous()
        B, C, H, W = x.shape
        n = ctypes.c_int64()
        _lib.check(_lib.lib().basic_hp_encode_images(self._h, x.data_ptr(), 0 if x.is_cuda else 1, B, H, W, None, 0, ctypes.byref(n), _stream()))
        # the final bytes object is allocated once at its exact size and the streams are framed straight into it
        api = ctypes.pythonapi
        api.PyBytes_FromStringAndSize.restype, api.PyBytes_FromStringAndSize.argtypes = ctypes.py_object, [ctypes.c_char_p, ctypes.c_ssize_t]
        api.PyBytes_AsString.restype, api.PyBytes_AsString.argtypes = ctypes.c_void_p, [ctypes.py_object]
        out = api.PyBytes_FromStringAndSize(None, n.value)
        _lib.check(_lib.lib().basic_hp_encode_result(self._h, api.PyBytes_AsString(out), n.value, None))
        return out

    def decode(self, data, device=None):
        buf = np.frombuffer(data, dtype=np.uint8)
        b, c, h, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.lib().basic_hp_decoded_shape(self._h, buf.ctypes.data, buf.size, ctypes.byref(b), ctypes.byref(c),
                                                     ctypes.byref(h), ctypes.byref(w)))
        out = torch.empty((b.value, c.value, h.value, w.value), device=device or torch.device("cuda", torch.cuda.current_device()),
                          dtype=torch.float32)
        _lib.check(_lib.lib().basic_hp_decode_images(self._h, buf.ctypes.data, buf.size, out.data_ptr(), out.numel(), _stream()))
        return out


class ScanlinePlan:
    """basic_scanline_*: the persistent scan-line AR coding loop (one launch for all H*W coding steps)."""

    def __init__(self, ctx_weight, ctx_bias, dense, prior_channels, ctx_act=False):
        """dense: list of (weight [out, in(, 1, 1)], bias or None, leaky_after: bool[, in_groups: int = 1]); in_groups = the
        input channel groups of the masked convolution the layer stands for (fixes the canonical summation blocks)."""
        dense = [tuple(d) + (1,) * (4 - len(d)) for d in dense]
        groups = np.array([int(d[3]) for d in dense], dtype=np.int32)
        dense = [d[:3] for d in dense]
        cw = _f32(ctx_weight)
        cb = _f32(ctx_bias) if ctx_bias is not None else None
        self.channels, self.ksize = cw.shape[1], cw.shape[2]
        ws = [_f32(w).reshape(w.shape[0], -1) for w, _, _ in dense]
        bs = [(_f32(b) if b is not None else None) for _, b, _ in dense]
        n = len(dense)
        wp = (ctypes.c_void_p * n)(*[w.ctypes.data for w in ws])
        bp = (ctypes.c_void_p * n)(*[(b.ctypes.data if b is not None else None) for b in bs])
        outs = np.array([w.shape[0] for w in ws], dtype=np.int32)
        acts = np.array([int(bool(ctx_act))] + [int(bool(a)) for _, _, a in dense], dtype=np.int32)
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().basic_scanline_plan_create(cw.ctypes.data, cb.ctypes.data if cb is not None else None, self.channels,
                                                         cw.shape[0], self.ksize, int(prior_channels), n, wp, bp, outs.ctypes.data,
                                                         acts.ctypes.data, groups.ctypes.data, ctypes.byref(h)))
        self._h = h
        wg, lb = ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.lib().basic_scanline_plan_info(h, ctypes.byref(wg), ctypes.byref(lb)))
        self.workgroups, self.lds_weight_bytes = wg.value, lb.value

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().basic_scanline_plan_destroy(h)
            except Exception:
                pass

    def encode(self, y, prior, table):
        y, table = _dev(y, torch.float32), _dev(table, torch.float32)
        prior = _dev(prior, torch.float32) if prior is not None else None
        B, C, H, W = y.shape
        sym = torch.empty((B, H * W * C), device=y.device, dtype=torch.int32)
        idx = torch.empty((B, H * W * C), device=y.device, dtype=torch.int32)
        ybuf = torch.empty_like(y)
        _lib.check(_lib.lib().basic_scanline_encode_dev(self._h, y.data_ptr(), prior.data_ptr() if prior is not None else None, B, H, W,
                                                        table.data_ptr(), table.numel(), sym.data_ptr(), idx.data_ptr(), ybuf.data_ptr(),
                                                        _stream()))
        return sym, idx, ybuf

    def can_encode(self, batch):
        """The encoder launch needs its workgroups resident: one per compute unit."""
        return self.workgroups <= torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count

    def batched_max(self, width, decode=False):
        """Largest batch the batched persistent kernel serves for a latent `width` columns wide on this device (0: never)."""
        key = (int(width), bool(decode), torch.cuda.current_device())
        cache = self.__dict__.setdefault("_batched_max", {})
        if key not in cache:
            m = ctypes.c_int()
            _lib.check(_lib.lib().basic_scanline_batched_max(self._h, int(width), int(bool(decode)), ctypes.byref(m)))
            cache[key] = m.value
        return cache[key]

    def can_decode(self, tables, batch):
        ok = ctypes.c_int()
        _lib.check(_lib.lib().basic_scanline_can_decode(self._h, tables._h, int(batch), ctypes.byref(ok)))
        return bool(ok.value)

    def decode(self, tables, d_words, d_word_off, prior, batch, h, w, table):
        """-> (symbols, indexes int32 [B, H*W*C] in coding order, y_hat [B, C, H, W])."""
        table = _dev(table, torch.float32)
        prior = _dev(prior, torch.float32) if prior is not None else None
        d_words, d_word_off = _dev(d_words, torch.int32), _dev(d_word_off, torch.int64)
        C = self.channels
        sym = torch.empty((batch, h * w * C), device=table.device, dtype=torch.int32)
        idx = torch.empty((batch, h * w * C), device=table.device, dtype=torch.int32)
        ybuf = torch.empty((batch, C, h, w), device=table.device, dtype=torch.float32)
        _lib.check(_lib.lib().basic_scanline_decode_dev(self._h, tables._h, d_words.data_ptr(), d_word_off.data_ptr(),
                                                        prior.data_ptr() if prior is not None else None, batch, h, w, table.data_ptr(),
                                                        table.numel(), sym.data_ptr(), idx.data_ptr(), ybuf.data_ptr(), _stream()))
        return sym, idx, ybuf

    def check(self):
        """Synchronises the current stream; raises if a barrier of the last launch gave up."""
        bad = ctypes.c_int()
        _lib.check(_lib.lib().basic_scanline_status(self._h, _stream(), ctypes.byref(bad)))
        if bad.value:
            raise _lib.BasicHipError("persistent scan-line kernel: an in-kernel barrier timed out (grid not resident?)")
