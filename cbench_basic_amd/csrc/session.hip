// Fused image entry points of the plain hyperprior latent graph (include/basic_hip.h section 8).
//
// One C call = what GeneralCodec.compress / decompress (cbench/codecs/general_codec.py:44-130) run for the graph of
// configs/lossy_graph_scalable_exp_hp.py:182-215 through LatentGraphicalANSEntropyCoder.encode / decode
// (cbench/modules/entropy_coder/latent_graph.py:1232-1295): the same kernels of this library, in the same order as the
// module-by-module Python path launches them, so both paths give identical bytes.  Everything here is host
// orchestration: buffers, launches, two small synchronisations per compress (stream lengths, then the words).
#include "common.h"

#include <cstring>
#include <map>
#include <mutex>
#include <vector>

using namespace basic;

constexpr int kMaxTokenLanes = 4;


namespace {

struct DBuf {   // grow-only device buffer
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return BASIC_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 256;
        BASIC_HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return BASIC_OK;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
    ~DBuf() { if (p) (void)hipFree(p); }
};

struct HBuf {   // grow-only page-locked host buffer
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return BASIC_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        BASIC_HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return BASIC_OK;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
    ~HBuf() { if (p) (void)hipHostFree(p); }
};

// out[b][c][y][x] = in[b][c][y][x] for y < oh, x < ow  (prior[..., :h, :w].contiguous(), compressai_coder.py call sites)
__global__ void crop_planes_kernel(const float *__restrict__ in, int ih, int iw, float *__restrict__ out, int oh, int ow,
                                   int64_t planes)
{
    const int64_t total = planes * oh * ow;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int x = static_cast<int>(i % ow);
        const int y = static_cast<int>((i / ow) % oh);
        const int64_t pl = i / (static_cast<int64_t>(ow) * oh);
        out[i] = in[(pl * ih + y) * iw + x];
    }
}

// idx[b][c][hw] = c   (EntropyBottleneck indexes, compressai_coder.py:238-245 -> upstream _build_indexes)
__global__ void channel_index_kernel(int32_t *__restrict__ idx, int channels, int hw, int64_t total)
{
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x)
        idx[i] = static_cast<int32_t>((i / hw) % channels);
}

// seg[i] = i * n (i <= streams); pos[i] = -1 (decoder: start from the stream head)
__global__ void seg_init_kernel(int64_t *__restrict__ seg, int64_t n, int streams, int64_t *__restrict__ pos)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= streams) seg[i] = static_cast<int64_t>(i) * n;
    if (pos && i < streams) pos[i] = -1;
}

// off[0] = 0, off[i + 1] = off[i] + max(nwords[i], 0)   (one wavefront; streams are few)
__global__ void offsets_kernel(const int32_t *__restrict__ nwords, int n, int64_t *__restrict__ off)
{
    const int lane = threadIdx.x;
    int64_t base = 0;
    if (lane == 0) off[0] = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int i = c0 + lane;
        int64_t v = (i < n && nwords[i] > 0) ? nwords[i] : 0;
        for (int d = 1; d < 64; d <<= 1) {   // inclusive scan over the wave
            const int64_t up = __shfl_up(v, d, 64);
            if (lane >= d) v += up;
        }
        if (i < n) off[i + 1] = base + v;
        base += __shfl(v, 63, 64);
    }
}

int grid_for(int64_t total) { int64_t g = (total + 255) / 256; return static_cast<int>(g < 1 ? 1 : g > 4096 ? 4096 : g); }

}  // namespace

struct basic_hp_session {
    std::vector<const basic_conv_plan *> g_a, h_a, h_s, g_s;
    const basic_rans_tables *z_tables = nullptr, *y_tables = nullptr;
    int z_channels = 0, y_channels = 0, x_channels = 0, out_channels = 0, n_scales = 0;
    float scale_bound = 0.11f;
    int rans_waves = 0;
    DBuf d_medians, d_table;
    DBuf act[2], d_x, d_y, d_z, d_zhat, d_prior, d_scales;
    DBuf z_sym, z_idx, y_sym, y_idx, seg_z, seg_y, slots_z, slots_y, nw, off, packed, words, woff, state, pos;
    HBuf h_nw, h_words, h_in;
    std::vector<int64_t> off_z, off_y;   // word offsets of the last encoded batch (its words sit in h_words)
    int res_batch = 0, res_zh = 0, res_zw = 0, res_yh = 0, res_yw = 0;
    hipEvent_t in_done = nullptr;   // the last decode's upload out of h_in
    // Entropy side stream (only with the transform token, i.e. concurrent sessions): everything that is NOT a big
    // transform -- the hyper path's small kernels, both rANS stages, offsets / compaction / copies -- runs on a stream of
    // the highest priority, so that its few short workgroups are dispatched ahead of the thousands a transform of another
    // session has queued (measured: the hyper path took 3.0-3.7 ms beside another session's transforms, 0.7 ms alone).
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool use_token = false;
    int token_lanes = 0;   // transform phases of all sessions admitted at a time (0 = no ordering)
    hipEvent_t enc_phase = nullptr, dec_phase = nullptr;   // close this session's transform phases (see TransformToken)
    // Upload of a host-resident batch (general_codec.py:46-47): on its own stream, ordered only after the PREVIOUS call's
    // analysis transform (the last reader of d_x), so it runs beside whatever the caller's stream still has queued (the
    // previous decode's chains and synthesis transform) instead of between that and this call's first layer.
    hipStream_t copy = nullptr;
    hipEvent_t x_free = nullptr, x_ready = nullptr;
    bool x_read_pending = false;
    ~basic_hp_session();
};

namespace {

// runs a chain of layer plans; the result lands in *out_buf (a session buffer) -- or in d_final when given
int run_chain(basic_hp_session *s, const std::vector<const basic_conv_plan *> &plans, const float *d_in, int batch, int h, int w,
              DBuf *out_buf, float *d_final, int *oh_out, int *ow_out, hipStream_t st)
{
    const float *cur = d_in;
    int flip = 0;
    for (size_t i = 0; i < plans.size(); ++i) {
        int oh = 0, ow = 0, co = 0;
        int rc = basic_conv_plan_out_hw(plans[i], h, w, &oh, &ow);
        if (rc) return rc;
        rc = basic_conv_plan_channels(plans[i], nullptr, &co);
        if (rc) return rc;
        const size_t bytes = sizeof(float) * static_cast<size_t>(batch) * co * oh * ow;
        float *dst;
        if (i + 1 == plans.size()) {
            if (d_final) dst = d_final;
            else { rc = out_buf->ensure(bytes); if (rc) return rc; dst = out_buf->as<float>(); }
        } else {
            rc = s->act[flip].ensure(bytes);
            if (rc) return rc;
            dst = s->act[flip].as<float>();
            flip ^= 1;
        }
        rc = basic_conv_forward_dev(plans[i], cur, batch, h, w, dst, st);
        if (rc) return rc;
        cur = dst;
        h = oh;
        w = ow;
    }
    if (oh_out) *oh_out = h;
    if (ow_out) *ow_out = w;
    return BASIC_OK;
}

int chain_out_hw(const std::vector<const basic_conv_plan *> &plans, int h, int w, int *oh, int *ow)
{
    for (const auto *p : plans) {
        int a = 0, b = 0;
        int rc = basic_conv_plan_out_hw(p, h, w, &a, &b);
        if (rc) return rc;
        h = a;
        w = b;
    }
    *oh = h;
    *ow = w;
    return BASIC_OK;
}

// The "transform token": sessions that opt in (basic_hp_session_set_transform_token) run their MFMA-heavy phases one
// after another in GPU time, in the order their host threads enqueue them -- a phase starts with a stream-wait on the
// event that closed the previous holder's phase and ends by recording its own.  Without it, equal workers drift into
// lock-step (all in their transforms together, sharing the chip; then all in their rANS chains together, leaving it
// idle); with it, one worker's chains always run beside another worker's transforms.  Nothing blocks on the host
// except the short critical section that keeps wait / launches / record of one phase together.
struct TransformToken {
    std::mutex mu;
    // per device: the events that close the most recently enqueued phases, one per LANE (owned by the sessions that recorded
    // them).  A new phase waits for the phase that used its lane last, so `lanes` phases run side by side: 1 = strictly one
    // after another (full batches fill the chip on their own), 2 = two at a time (small batches: one session's launches
    // leave compute units idle and tile counts quantise badly, two sessions' launches interleave)
    struct Dev {
        hipEvent_t lane[kMaxTokenLanes] = {};
        int cursor = 0;
    };
    std::map<int, Dev> dev;
};
TransformToken g_token;

struct TokenPhase {   // RAII: a phase of one session on one stream
    basic_hp_session *s;
    hipStream_t st;
    hipEvent_t *evt;
    bool on;
    TransformToken::Dev *dev = nullptr;
    int slot = 0;
    int rc = BASIC_OK;
    TokenPhase(basic_hp_session *s_, hipStream_t st_, hipEvent_t *evt_);
    int close();
    ~TokenPhase() { if (on) (void)close(); }
};

struct WavesGuard {   // the session's rANS geometry / tile hand-out apply to this thread's launches for the duration of a call
    int prev;
    bool prev_dyn;
    explicit WavesGuard(int w, bool concurrent = false) : prev(set_rans_waves(w)), prev_dyn(set_dynamic_tiles(concurrent)) {}
    ~WavesGuard() { set_rans_waves(prev); set_dynamic_tiles(prev_dyn); }
};

// the stream the non-transform work of a call goes to: the caller's, or (token mode) the session's priority stream,
// ordered after everything enqueued on the caller's stream so far
int entropy_stream(basic_hp_session *s, hipStream_t st, hipStream_t *out)
{
    *out = st;
    if (!s->use_token) return BASIC_OK;
    if (!s->side) {
        int lo = 0, hi = 0;   // numerically lowest = greatest priority
        BASIC_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        BASIC_HIP_TRY(hipStreamCreateWithPriority(&s->side, hipStreamNonBlocking, hi));
        BASIC_HIP_TRY(hipEventCreateWithFlags(&s->fork, hipEventDisableTiming));
        BASIC_HIP_TRY(hipEventCreateWithFlags(&s->join, hipEventDisableTiming));
    }
    BASIC_HIP_TRY(hipEventRecord(s->fork, st));
    BASIC_HIP_TRY(hipStreamWaitEvent(s->side, s->fork, 0));
    *out = s->side;
    return BASIC_OK;
}

// the caller's stream continues after the side stream's work
int rejoin(basic_hp_session *s, hipStream_t st, hipStream_t es)
{
    if (es == st) return BASIC_OK;
    BASIC_HIP_TRY(hipEventRecord(s->join, es));
    BASIC_HIP_TRY(hipStreamWaitEvent(st, s->join, 0));
    return BASIC_OK;
}

TokenPhase::TokenPhase(basic_hp_session *s_, hipStream_t st_, hipEvent_t *evt_) : s(s_), st(st_), evt(evt_), on(s_->token_lanes > 0)
{
    if (!on) return;
    g_token.mu.lock();
    int d = 0;
    hipError_t e = hipGetDevice(&d);
    if (e == hipSuccess && !*evt) e = hipEventCreateWithFlags(evt, hipEventDisableTiming);
    if (e == hipSuccess) {
        dev = &g_token.dev[d];
        slot = dev->cursor % s->token_lanes;
        hipEvent_t prev = dev->lane[slot];
        if (prev && prev != *evt) e = hipStreamWaitEvent(st, prev, 0);
    }
    if (e != hipSuccess) rc = hip_fail(e, "transform token: wait", __FILE__, __LINE__);
}

int TokenPhase::close()
{
    if (!on) return rc;
    on = false;
    hipError_t e = hipEventRecord(*evt, st);
    if (e == hipSuccess && dev) {
        dev->lane[slot] = *evt;
        dev->cursor = (slot + 1) % s->token_lanes;
    }
    g_token.mu.unlock();
    if (e != hipSuccess && rc == BASIC_OK) rc = hip_fail(e, "transform token: record", __FILE__, __LINE__);
    return rc;
}

}  // namespace

extern "C" int basic_hp_session_set_transform_token(basic_hp_session *s, int enable)
{
    BASIC_REQUIRE(s && enable >= 0 && enable <= kMaxTokenLanes, "hp_session_set_transform_token: 0 (off) or 1 .. 4 phases at a time");
    s->token_lanes = enable;
    s->use_token = enable != 0;
    return BASIC_OK;
}

extern "C" int basic_hp_session_create(const basic_conv_plan *const *g_a, int n_g_a, const basic_conv_plan *const *h_a, int n_h_a,
                                       const basic_conv_plan *const *h_s, int n_h_s, const basic_conv_plan *const *g_s, int n_g_s,
                                       const float *eb_medians, int z_channels, const basic_rans_tables *z_tables,
                                       const float *scale_table, int n_scales, float scale_bound,
                                       const basic_rans_tables *y_tables, basic_hp_session **out)
{
    int rc = require_device();
    if (rc) return rc;
    BASIC_REQUIRE(g_a && h_a && h_s && g_s && n_g_a >= 1 && n_h_a >= 1 && n_h_s >= 1 && n_g_s >= 1 && eb_medians && z_tables &&
                      scale_table && n_scales >= 2 && y_tables && out && z_channels >= 1,
                  "hp_session_create: bad argument");
    auto *s = new basic_hp_session();
    s->g_a.assign(g_a, g_a + n_g_a);
    s->h_a.assign(h_a, h_a + n_h_a);
    s->h_s.assign(h_s, h_s + n_h_s);
    s->g_s.assign(g_s, g_s + n_g_s);
    for (auto *list : {&s->g_a, &s->h_a, &s->h_s, &s->g_s})
        for (const auto *p : *list)
            if (!p) { delete s; set_error("hp_session_create: null layer plan"); return BASIC_ERR_INVALID; }
    int ci = 0, co = 0;
    basic_conv_plan_channels(s->g_a.front(), &s->x_channels, nullptr);
    basic_conv_plan_channels(s->g_a.back(), nullptr, &s->y_channels);
    basic_conv_plan_channels(s->h_a.front(), &ci, nullptr);
    basic_conv_plan_channels(s->h_a.back(), nullptr, &co);
    bool ok = ci == s->y_channels && co == z_channels;
    basic_conv_plan_channels(s->h_s.front(), &ci, nullptr);
    basic_conv_plan_channels(s->h_s.back(), nullptr, &co);
    ok = ok && ci == z_channels && co == s->y_channels;   // scales only: the GaussianConditional graph
    basic_conv_plan_channels(s->g_s.front(), &ci, nullptr);
    basic_conv_plan_channels(s->g_s.back(), nullptr, &s->out_channels);
    ok = ok && ci == s->y_channels;
    if (!ok) { delete s; set_error("hp_session_create: channel counts of the transforms do not chain (x->y->z->scales(y)->x)"); return BASIC_ERR_INVALID; }
    s->z_channels = z_channels;
    s->z_tables = z_tables;
    s->y_tables = y_tables;
    s->n_scales = n_scales;
    s->scale_bound = scale_bound;
    rc = s->d_medians.ensure(sizeof(float) * z_channels);
    if (!rc) rc = s->d_table.ensure(sizeof(float) * n_scales);
    if (rc) { delete s; return rc; }
    hipError_t e = hipMemcpy(s->d_medians.p, eb_medians, sizeof(float) * z_channels, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->d_table.p, scale_table, sizeof(float) * n_scales, hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete s; return hip_fail(e, "hp_session_create upload", __FILE__, __LINE__); }
    *out = s;
    return BASIC_OK;
}

basic_hp_session::~basic_hp_session()
{
    {
        std::lock_guard<std::mutex> lock(g_token.mu);
        for (auto &kv : g_token.dev)
            for (hipEvent_t &e : kv.second.lane)
                if (e && (e == enc_phase || e == dec_phase)) {
                    (void)hipEventSynchronize(e);   // whoever waits on it has been released
                    e = nullptr;
                }
    }
    for (hipEvent_t e : {in_done, enc_phase, dec_phase, fork, join, x_free, x_ready})
        if (e) (void)hipEventDestroy(e);
    if (side) (void)hipStreamDestroy(side);
    if (copy) (void)hipStreamDestroy(copy);
}

extern "C" void basic_hp_session_destroy(basic_hp_session *s) { delete s; }

extern "C" int basic_hp_session_set_rans_waves(basic_hp_session *s, int waves_per_block)
{
    BASIC_REQUIRE(s && (waves_per_block == 0 || waves_per_block == 1 || waves_per_block == 2 || waves_per_block == 4 ||
                        waves_per_block == 8 || waves_per_block == 16),
                  "hp_session_set_rans_waves: 0, 1, 2, 4, 8 or 16");
    s->rans_waves = waves_per_block;
    return BASIC_OK;
}

extern "C" int64_t basic_hp_encode_bound(const basic_hp_session *s, int batch, int h, int w)
{
    if (!s || batch < 1 || h < 1 || w < 1) return -1;
    int yh, yw, zh, zw;
    if (chain_out_hw(s->g_a, h, w, &yh, &yw) || chain_out_hw(s->h_a, yh, yw, &zh, &zw)) return -1;
    const int64_t ny = static_cast<int64_t>(s->y_channels) * yh * yw, nz = static_cast<int64_t>(s->z_channels) * zh * zw;
    // worst case of the coder: 3 n + 4 words per stream (the retry slot of the bypass-heavy case)
    return 4 + 2 * (12 + 4ll * batch) + 4ll * batch * ((3 * ny + 4) + (3 * nz + 4));
}

namespace {

// enqueues the encoder of `streams` equal-length streams (n symbols each) into right-aligned slots of `slot` words
int encode_latent(basic_hp_session *s, const basic_rans_tables *t, const int32_t *d_sym, const int32_t *d_idx, int streams,
                  int64_t n, DBuf *seg, DBuf *slots, int32_t *d_nw, int64_t slot, hipStream_t st)
{
    int rc = seg->ensure(sizeof(int64_t) * (streams + 1));
    if (!rc) rc = slots->ensure(sizeof(uint32_t) * static_cast<size_t>(streams) * slot);
    if (rc) return rc;
    hipLaunchKernelGGL(seg_init_kernel, dim3((streams + 256) / 256), dim3(256), 0, st, seg->as<int64_t>(), n, streams,
                       static_cast<int64_t *>(nullptr));
    BASIC_HIP_TRY(hipGetLastError());
    return basic_rans_encode_batch_dev(t, d_sym, d_idx, seg->as<int64_t>(), streams, slots->as<uint32_t>(), slot, d_nw, st);
}

}  // namespace

extern "C" int basic_hp_encode_images(basic_hp_session *s, const float *x, int x_on_host, int batch, int h, int w, uint8_t *out,
                                      int64_t out_capacity, int64_t *out_len, void *hip_stream)
{
    BASIC_REQUIRE(s && x && out_len && batch >= 1 && h >= 1 && w >= 1, "hp_encode_images: bad argument");
    s->res_batch = 0;
    hipStream_t st = as_stream(hip_stream);
    WavesGuard guard(s->rans_waves, s->use_token);
    int rc;
    const float *d_x = x;
    if (x_on_host) {   // general_codec.py:46-47: the upload belongs to compress()
        const size_t bytes = sizeof(float) * static_cast<size_t>(batch) * s->x_channels * h * w;
        rc = s->d_x.ensure(bytes);
        if (rc) return rc;
        if (!s->copy) {
            BASIC_HIP_TRY(hipStreamCreateWithFlags(&s->copy, hipStreamNonBlocking));
            BASIC_HIP_TRY(hipEventCreateWithFlags(&s->x_free, hipEventDisableTiming));
            BASIC_HIP_TRY(hipEventCreateWithFlags(&s->x_ready, hipEventDisableTiming));
        }
        if (s->x_read_pending) BASIC_HIP_TRY(hipStreamWaitEvent(s->copy, s->x_free, 0));
        BASIC_HIP_TRY(hipMemcpyAsync(s->d_x.p, x, bytes, hipMemcpyHostToDevice, s->copy));
        BASIC_HIP_TRY(hipEventRecord(s->x_ready, s->copy));
        BASIC_HIP_TRY(hipStreamWaitEvent(st, s->x_ready, 0));
        d_x = s->d_x.as<float>();
    }
    // ---- inference pass x -> y -> z (latent_graph.py:721-758)
    int yh, yw, zh, zw, ph, pw;
    TokenPhase phase(s, st, &s->enc_phase);   // the analysis transform: ~95 % of the encoder's MFMA work
    if (phase.rc) return phase.rc;
    rc = run_chain(s, s->g_a, d_x, batch, h, w, &s->d_y, nullptr, &yh, &yw, st);
    if (rc) return rc;
    if (x_on_host) {
        BASIC_HIP_TRY(hipEventRecord(s->x_free, st));
        s->x_read_pending = true;
    }
    rc = phase.close();   // the small hyper-path kernels and both rANS stages run beside the next holder's transforms
    if (rc) return rc;
    hipStream_t caller_st = st;
    rc = entropy_stream(s, caller_st, &st);   // from here on `st` is the entropy stream
    if (rc) return rc;
    rc = run_chain(s, s->h_a, s->d_y.as<float>(), batch, yh, yw, &s->d_z, nullptr, &zh, &zw, st);
    if (rc) return rc;
    const int64_t nz = static_cast<int64_t>(s->z_channels) * zh * zw, ny = static_cast<int64_t>(s->y_channels) * yh * yw;
    // ---- node z: EntropyBottleneck symbols + de-quantised latent (compressai_coder.py:203-236), rANS
    rc = s->z_sym.ensure(sizeof(int32_t) * batch * nz);
    if (!rc) rc = s->z_idx.ensure(sizeof(int32_t) * batch * nz);
    if (!rc) rc = s->d_zhat.ensure(sizeof(float) * batch * nz);
    if (!rc) rc = s->nw.ensure(sizeof(int32_t) * 2 * batch);
    if (!rc) rc = s->off.ensure(sizeof(int64_t) * 2 * (batch + 1));
    if (rc) return rc;
    rc = basic_eb_quantize_index_dev(s->d_z.as<float>(), s->d_medians.as<float>(), batch, s->z_channels, zh * zw,
                                     s->z_sym.as<int32_t>(), s->z_idx.as<int32_t>(), s->d_zhat.as<float>(), st);
    if (rc) return rc;
    int64_t slot_z = nz + 2, slot_y = ny + 2;   // the reference's own buffer size (rans64.cpp:240)
    int32_t *d_nw_z = s->nw.as<int32_t>(), *d_nw_y = s->nw.as<int32_t>() + batch;
    rc = encode_latent(s, s->z_tables, s->z_sym.as<int32_t>(), s->z_idx.as<int32_t>(), batch, nz, &s->seg_z, &s->slots_z, d_nw_z,
                       slot_z, st);
    if (rc) return rc;
    // ---- edge z -> y: h_s(z_hat); node y: GaussianConditional indexes + round (compressai_coder.py:377-385), rANS
    rc = run_chain(s, s->h_s, s->d_zhat.as<float>(), batch, zh, zw, &s->d_prior, nullptr, &ph, &pw, st);
    if (rc) return rc;
    BASIC_REQUIRE(ph >= yh && pw >= yw, "hp_encode_images: the hyper-synthesis output is smaller than the latent");
    const float *d_scales = s->d_prior.as<float>();
    if (ph != yh || pw != yw) {
        rc = s->d_scales.ensure(sizeof(float) * batch * ny);
        if (rc) return rc;
        hipLaunchKernelGGL(crop_planes_kernel, dim3(grid_for(batch * ny)), dim3(256), 0, st, s->d_prior.as<float>(), ph, pw,
                           s->d_scales.as<float>(), yh, yw, static_cast<int64_t>(batch) * s->y_channels);
        BASIC_HIP_TRY(hipGetLastError());
        d_scales = s->d_scales.as<float>();
    }
    rc = s->y_sym.ensure(sizeof(int32_t) * batch * ny);
    if (!rc) rc = s->y_idx.ensure(sizeof(int32_t) * batch * ny);
    if (rc) return rc;
    rc = basic_gc_quantize_index_dev(s->d_y.as<float>(), d_scales, batch * ny, s->d_table.as<float>(), s->n_scales, s->scale_bound,
                                     s->y_sym.as<int32_t>(), s->y_idx.as<int32_t>(), nullptr, st);
    if (rc) return rc;
    rc = encode_latent(s, s->y_tables, s->y_sym.as<int32_t>(), s->y_idx.as<int32_t>(), batch, ny, &s->seg_y, &s->slots_y, d_nw_y,
                       slot_y, st);
    if (rc) return rc;
    // ---- stream lengths to the host; offsets and compaction on the device meanwhile
    rc = s->h_nw.ensure(sizeof(int32_t) * 2 * batch + sizeof(int64_t) * 2 * (batch + 1));
    if (rc) return rc;
    int32_t *h_nw = s->h_nw.as<int32_t>();
    for (int attempt = 0;; ++attempt) {
        int64_t *d_off_z = s->off.as<int64_t>(), *d_off_y = s->off.as<int64_t>() + (batch + 1);
        hipLaunchKernelGGL(offsets_kernel, dim3(1), dim3(64), 0, st, d_nw_z, batch, d_off_z);
        hipLaunchKernelGGL(offsets_kernel, dim3(1), dim3(64), 0, st, d_nw_y, batch, d_off_y);
        BASIC_HIP_TRY(hipGetLastError());
        // both latents' streams share one packed buffer: z first, then y (each at its own base)
        rc = s->packed.ensure(sizeof(uint32_t) * static_cast<size_t>(batch) * (slot_z + slot_y));
        if (rc) return rc;
        uint32_t *d_pz = s->packed.as<uint32_t>(), *d_py = d_pz + static_cast<size_t>(batch) * slot_z;
        rc = basic_rans_compact_streams_dev(s->slots_z.as<uint32_t>(), slot_z, d_nw_z, d_off_z, batch, d_pz, st);
        if (!rc) rc = basic_rans_compact_streams_dev(s->slots_y.as<uint32_t>(), slot_y, d_nw_y, d_off_y, batch, d_py, st);
        if (rc) return rc;
        BASIC_HIP_TRY(hipMemcpyAsync(h_nw, s->nw.p, sizeof(int32_t) * 2 * batch, hipMemcpyDeviceToHost, st));
        BASIC_HIP_TRY(hipStreamSynchronize(st));
        bool over_z = false, over_y = false;
        for (int i = 0; i < batch; ++i) { over_z |= h_nw[i] < 0; over_y |= h_nw[batch + i] < 0; }
        if (!over_z && !over_y) break;
        BASIC_REQUIRE(attempt == 0, "hp_encode_images: rANS slot overflow with the guaranteed slot size");
        // bypass-heavy data overflowed the reference's bound (undefined behaviour there): redo with the guaranteed one
        if (over_z) {
            slot_z = 3 * nz + 4;
            rc = encode_latent(s, s->z_tables, s->z_sym.as<int32_t>(), s->z_idx.as<int32_t>(), batch, nz, &s->seg_z, &s->slots_z, d_nw_z,
                               slot_z, st);
            if (rc) return rc;
        }
        if (over_y) {
            slot_y = 3 * ny + 4;
            rc = encode_latent(s, s->y_tables, s->y_sym.as<int32_t>(), s->y_idx.as<int32_t>(), batch, ny, &s->seg_y, &s->slots_y, d_nw_y,
                               slot_y, st);
            if (rc) return rc;
        }
    }
    s->off_z.assign(batch + 1, 0);
    s->off_y.assign(batch + 1, 0);
    for (int i = 0; i < batch; ++i) { s->off_z[i + 1] = s->off_z[i] + h_nw[i]; s->off_y[i + 1] = s->off_y[i] + h_nw[batch + i]; }
    const int64_t wz = s->off_z[batch], wy = s->off_y[batch];
    rc = s->h_words.ensure(sizeof(uint32_t) * static_cast<size_t>(wz + wy));
    if (rc) return rc;
    uint32_t *h_wz = s->h_words.as<uint32_t>(), *h_wy = h_wz + wz;
    const uint32_t *d_pz = s->packed.as<uint32_t>(), *d_py = d_pz + static_cast<size_t>(batch) * slot_z;
    if (wz) BASIC_HIP_TRY(hipMemcpyAsync(h_wz, d_pz, sizeof(uint32_t) * wz, hipMemcpyDeviceToHost, st));
    if (wy) BASIC_HIP_TRY(hipMemcpyAsync(h_wy, d_py, sizeof(uint32_t) * wy, hipMemcpyDeviceToHost, st));
    BASIC_HIP_TRY(hipStreamSynchronize(st));
    rc = rejoin(s, caller_st, st);
    if (rc) return rc;
    s->res_batch = batch; s->res_zh = zh; s->res_zw = zw; s->res_yh = yh; s->res_yw = yw;
    *out_len = 4 + (12 + 4ll * batch + 4 * wz) + (12 + 4ll * batch + 4 * wy);
    if (!out) return BASIC_OK;   // the caller fetches the bytes with basic_hp_encode_result() once it knows their size
    return basic_hp_encode_result(s, out, out_capacity, out_len);
}

// framing: merge_bytes([z body, y body], num_segments=2) with write_body bodies (bytes_ops.py:19-33)
extern "C" int basic_hp_encode_result(basic_hp_session *s, uint8_t *out, int64_t out_capacity, int64_t *out_len)
{
    BASIC_REQUIRE(s && out && s->res_batch >= 1, "hp_encode_result: no encoded batch is held by this session");
    const int batch = s->res_batch;
    const int64_t wz = s->off_z[batch], wy = s->off_y[batch];
    const uint32_t *h_wz = s->h_words.as<uint32_t>(), *h_wy = h_wz + wz;
    const int64_t len_z = 12 + 4ll * batch + 4 * wz, len_y = 12 + 4ll * batch + 4 * wy;
    if (out_len) *out_len = 4 + len_z + len_y;
    if (4 + len_z + len_y > out_capacity) { set_error("hp_encode_result: output buffer too small"); return BASIC_ERR_OVERFLOW; }
    BASIC_REQUIRE(len_z <= 0xFFFFFFFFll, "hp_encode_result: z body exceeds the 32-bit length prefix");
    const uint32_t lz32 = static_cast<uint32_t>(len_z);
    std::memcpy(out, &lz32, 4);   // native-endian struct "I"
    int64_t written = 0;
    int rc = basic_frame_streams(h_wz, s->off_z.data(), batch, static_cast<uint32_t>(s->res_zh), static_cast<uint32_t>(s->res_zw), out + 4,
                                 len_z, &written);
    if (rc) return rc;
    return basic_frame_streams(h_wy, s->off_y.data(), batch, static_cast<uint32_t>(s->res_yh), static_cast<uint32_t>(s->res_yw),
                               out + 4 + len_z, len_y, &written);
}

namespace {

struct Body {
    const uint8_t *p = nullptr;
    int64_t len = 0;
    uint32_t h = 0, w = 0;
    int n = 0;
};

int split_bodies(const uint8_t *data, int64_t len, Body *z, Body *y)
{
    BASIC_REQUIRE(data && len >= 4 + 12 + 12, "hp_decode_images: truncated stream");
    uint32_t lz = 0;
    std::memcpy(&lz, data, 4);
    BASIC_REQUIRE(4ll + lz + 12 <= len && lz >= 12, "hp_decode_images: bad z-body length");
    z->p = data + 4; z->len = lz;
    y->p = data + 4 + lz; y->len = len - 4 - lz;
    int rc = basic_unframe_streams(z->p, z->len, &z->h, &z->w, &z->n, nullptr, 0, nullptr);
    if (!rc) rc = basic_unframe_streams(y->p, y->len, &y->h, &y->w, &y->n, nullptr, 0, nullptr);
    if (rc) return rc;
    BASIC_REQUIRE(z->n == y->n && z->n >= 1, "hp_decode_images: z and y bodies hold different image counts");
    return BASIC_OK;
}

}  // namespace

extern "C" int basic_hp_decoded_shape(const basic_hp_session *s, const uint8_t *data, int64_t len, int *batch, int *channels, int *h,
                                      int *w)
{
    BASIC_REQUIRE(s, "hp_decoded_shape: null session");
    Body z, y;
    int rc = split_bodies(data, len, &z, &y);
    if (rc) return rc;
    int oh, ow;
    rc = chain_out_hw(s->g_s, static_cast<int>(y.h), static_cast<int>(y.w), &oh, &ow);
    if (rc) return rc;
    if (batch) *batch = y.n;
    if (channels) *channels = s->out_channels;
    if (h) *h = oh;
    if (w) *w = ow;
    return BASIC_OK;
}

extern "C" int basic_hp_decode_images(basic_hp_session *s, const uint8_t *data, int64_t len, float *d_xhat,
                                      int64_t xhat_capacity_floats, void *hip_stream)
{
    BASIC_REQUIRE(s && d_xhat, "hp_decode_images: bad argument");
    hipStream_t st = as_stream(hip_stream);
    WavesGuard guard(s->rans_waves, s->use_token);
    Body z, y;
    int rc = split_bodies(data, len, &z, &y);
    if (rc) return rc;
    hipStream_t caller_st = st;
    rc = entropy_stream(s, caller_st, &st);   // everything before g_s goes to the entropy stream
    if (rc) return rc;
    const int batch = y.n, zh = z.h, zw = z.w, yh = y.h, yw = y.w;
    int oh, ow;
    rc = chain_out_hw(s->g_s, yh, yw, &oh, &ow);
    if (rc) return rc;
    BASIC_REQUIRE(static_cast<int64_t>(batch) * s->out_channels * oh * ow <= xhat_capacity_floats, "hp_decode_images: output buffer too small");
    const int64_t nz = static_cast<int64_t>(s->z_channels) * zh * zw, ny = static_cast<int64_t>(s->y_channels) * yh * yw;
    // ---- unframe both bodies into ONE pinned staging area [z words | y words | z offsets | y offsets] and upload it
    const int64_t wz = (z.len - 12 - 4ll * batch) / 4, wy = (y.len - 12 - 4ll * batch) / 4;
    BASIC_REQUIRE(wz >= 0 && wy >= 0, "hp_decode_images: truncated body");
    const size_t words_bytes = sizeof(uint32_t) * static_cast<size_t>(wz + wy + 2);
    const size_t off_pad = (words_bytes + 7) & ~static_cast<size_t>(7);
    const size_t total = off_pad + sizeof(int64_t) * 2 * (batch + 1);
    if (s->in_done) BASIC_HIP_TRY(hipEventSynchronize(s->in_done));   // the previous call's upload out of h_in has finished
    else BASIC_HIP_TRY(hipEventCreateWithFlags(&s->in_done, hipEventDisableTiming));
    rc = s->h_in.ensure(total);
    if (!rc) rc = s->words.ensure(total);
    if (rc) return rc;
    uint8_t *hb = s->h_in.as<uint8_t>();
    uint32_t *h_wz = reinterpret_cast<uint32_t *>(hb), *h_wy = h_wz + wz;
    int64_t *h_oz = reinterpret_cast<int64_t *>(hb + off_pad), *h_oy = h_oz + (batch + 1);
    uint32_t hh, ww;
    int nn;
    rc = basic_unframe_streams(z.p, z.len, &hh, &ww, &nn, h_oz, batch, h_wz);
    if (!rc) rc = basic_unframe_streams(y.p, y.len, &hh, &ww, &nn, h_oy, batch, h_wy);
    if (rc) return rc;
    BASIC_HIP_TRY(hipMemcpyAsync(s->words.p, hb, total, hipMemcpyHostToDevice, st));
    BASIC_HIP_TRY(hipEventRecord(s->in_done, st));
    const uint8_t *db = s->words.as<uint8_t>();
    const uint32_t *d_wz = reinterpret_cast<const uint32_t *>(db), *d_wy = d_wz + wz;
    const int64_t *d_oz = reinterpret_cast<const int64_t *>(db + off_pad), *d_oy = d_oz + (batch + 1);
    // ---- node z: indexes = channel, rANS decode, de-quantise (compressai_coder.py:238-245)
    rc = s->z_sym.ensure(sizeof(int32_t) * batch * nz);
    if (!rc) rc = s->z_idx.ensure(sizeof(int32_t) * batch * nz);
    if (!rc) rc = s->d_zhat.ensure(sizeof(float) * batch * nz);
    if (!rc) rc = s->seg_z.ensure(sizeof(int64_t) * (batch + 1));
    if (!rc) rc = s->seg_y.ensure(sizeof(int64_t) * (batch + 1));
    if (!rc) rc = s->state.ensure(sizeof(uint64_t) * 2 * batch);
    if (!rc) rc = s->pos.ensure(sizeof(int64_t) * 2 * batch);
    if (rc) return rc;
    hipLaunchKernelGGL(channel_index_kernel, dim3(grid_for(batch * nz)), dim3(256), 0, st, s->z_idx.as<int32_t>(), s->z_channels, zh * zw,
                       batch * nz);
    hipLaunchKernelGGL(seg_init_kernel, dim3((batch + 256) / 256), dim3(256), 0, st, s->seg_z.as<int64_t>(), nz, batch, s->pos.as<int64_t>());
    hipLaunchKernelGGL(seg_init_kernel, dim3((batch + 256) / 256), dim3(256), 0, st, s->seg_y.as<int64_t>(), ny, batch,
                       s->pos.as<int64_t>() + batch);
    BASIC_HIP_TRY(hipGetLastError());
    rc = basic_rans_decode_batch_dev(s->z_tables, d_wz, d_oz, s->z_idx.as<int32_t>(), s->seg_z.as<int64_t>(), batch, s->z_sym.as<int32_t>(),
                                     s->state.as<uint64_t>(), s->pos.as<int64_t>(), st);
    if (rc) return rc;
    rc = basic_eb_dequantize_dev(s->z_sym.as<int32_t>(), s->d_medians.as<float>(), batch, s->z_channels, zh * zw, s->d_zhat.as<float>(), st);
    if (rc) return rc;
    // ---- edge z -> y, node y (compressai_coder.py:387-393)
    int ph, pw;
    rc = run_chain(s, s->h_s, s->d_zhat.as<float>(), batch, zh, zw, &s->d_prior, nullptr, &ph, &pw, st);
    if (rc) return rc;
    BASIC_REQUIRE(ph >= yh && pw >= yw, "hp_decode_images: the hyper-synthesis output is smaller than the latent");
    const float *d_scales = s->d_prior.as<float>();
    if (ph != yh || pw != yw) {
        rc = s->d_scales.ensure(sizeof(float) * batch * ny);
        if (rc) return rc;
        hipLaunchKernelGGL(crop_planes_kernel, dim3(grid_for(batch * ny)), dim3(256), 0, st, s->d_prior.as<float>(), ph, pw,
                           s->d_scales.as<float>(), yh, yw, static_cast<int64_t>(batch) * s->y_channels);
        BASIC_HIP_TRY(hipGetLastError());
        d_scales = s->d_scales.as<float>();
    }
    rc = s->y_sym.ensure(sizeof(int32_t) * batch * ny);
    if (!rc) rc = s->y_idx.ensure(sizeof(int32_t) * batch * ny);
    if (!rc) rc = s->d_y.ensure(sizeof(float) * batch * ny);
    if (rc) return rc;
    // indexes only: the symbol output of this launch is scratch (the module path does the same, compressai_coder.py:387-393)
    rc = basic_gc_quantize_index_dev(d_scales, d_scales, batch * ny, s->d_table.as<float>(), s->n_scales, s->scale_bound,
                                     s->y_sym.as<int32_t>(), s->y_idx.as<int32_t>(), nullptr, st);
    if (rc) return rc;
    rc = basic_rans_decode_batch_dev(s->y_tables, d_wy, d_oy, s->y_idx.as<int32_t>(), s->seg_y.as<int64_t>(), batch, s->y_sym.as<int32_t>(),
                                     s->state.as<uint64_t>() + batch, s->pos.as<int64_t>() + batch, st);
    if (rc) return rc;
    rc = basic_i32_to_f32_dev(s->y_sym.as<int32_t>(), batch * ny, s->d_y.as<float>(), st);
    if (rc) return rc;
    // ---- edge y -> x: g_s straight into the caller's buffer
    rc = rejoin(s, caller_st, st);
    if (rc) return rc;
    st = caller_st;
    TokenPhase phase(s, st, &s->dec_phase);
    if (phase.rc) return phase.rc;
    rc = run_chain(s, s->g_s, s->d_y.as<float>(), batch, yh, yw, nullptr, d_xhat, nullptr, nullptr, st);
    const int rc2 = phase.close();
    return rc ? rc : rc2;
}
