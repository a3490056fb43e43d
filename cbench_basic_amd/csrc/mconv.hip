// Topo-group masked convolution evaluated ONLY at a list of positions (the positions of the
// topo group being coded), on the fp32 matrix core.
//
// Reference: TopoGroupDynamicMaskConv2d.forward, nn/layers/masked_conv.py:102-228
//     out[b, co, p] = bias[co] + sum_{ci, tap} W[co, ci, tap] * x[b, ci, p + tap]
//                                 * [ topo_in[g_in(ci), p + tap]  (< or <=)  topo_out[g_out(co), p] ]
//   with g_in(ci) = ci / (Cin/Gi), g_out(co) = co / (Cout/Go) (contiguous channel groups,
//   masked_conv.py:170-172,211-212), zero padding excluded from the sum (:125), "<=" when
//   allow_same_topogroup_conv (:168-169).  The 5x5 context conv and the 1x1 "param merger"
//   layers (pgm_coder.py:1215-1239,1621-1630; masked_conv.py:262-300) are all this one operator.
//
// The reference recomputes the full H x W map for every topo group (pgm_coder.py:922-924,
// 958-961).  Here a launch touches only the listed positions, and a whole (tap, input group)
// slab is skipped when its mask is empty for all 32 positions of the wave -- for causal
// patterns that removes most of the K loop.
//
// Mapping, large launches (masked_conv_pos_kernel<1, MT>): one wavefront = MT 32-row tiles of ONE output channel
// group x 32 listed positions.  The B fragment (gathered activations, mask applied) is loaded once and feeds MT MFMAs,
// and the activations are re-read Cout / (32 MT) times instead of Cout / 32 times -- with one tile per wave the widest
// merger layer (1536 -> 1536 on 32k positions) moved 9.6 GB per launch and ran at 0.37 of the MFMA peak.  The weights
// are packed so that a lane's MT A values are adjacent: [tap][ci][chunk][32 rows][MT] (one 8- / 16-byte load for MT 2 / 4).
// Tiny launches (a scanline group is one position per image) are latency bound: masked_conv_pos_kernel<4, 1> gives a
// tile to one workgroup of 4 waves which split the K loop and sum through LDS.
// Every output element accumulates the same products in the same order for every MT, and the variant is a function of
// the plan and the position count only, so encoder and decoder agree bit for bit (the split-K variant sums its four
// K-slices in a fixed order, which rounds differently from the single-wave sum).  B fragments are gathered from x with the mask applied as a select (so masked
// garbage, even NaN, never enters the sum -- the reference multiplies by 0, which only differs for non-finite data).
#include "common.h"

#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace basic;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MaskedLaunch {
    const float *x;        // [B][cin][H][W]
    float *y;              // [B][out_total][H][W], this layer writes channels out_off .. out_off+cout
    const float *w;        // [ntaps][cin][coutp]
    const float *bias;     // [coutp]
    const int32_t *topo_in;   // [gi][H][W]
    const int32_t *topo_out;  // [go][H][W]
    const int32_t *pos;    // [n_pos] flat b*H*W + p
    int64_t n_pos;
    int batch, cin, cout, coutp, h, w_, gi, go, gs_in, gs_out, tiles_per_group;
    int out_total, out_off, ntaps, pad, ksize, allow_same, act;
    int step;              // coding step (topo group being coded), or kNoStep: see the skip rule in the kernel
    const int32_t *first;  // [H][W] first step that visits a position (min over channel groups), or nullptr
    // cross-workgroup split-K (smallest launches): gridDim.z = tap_slices * pair_slices K-slices per tile, each
    // writes its partial tile to scratch[(tile * slices + slice)][16][64]; masked_conv_reduce_kernel sums them
    float *scratch;
    int tap_slices, pair_slices;
    int mt;                // pack factor of w: row r of a group sits at (r / (32 mt)) * 32 mt + (r % 32) * mt + (r / 32) % mt
};

constexpr int kNoStep = INT32_MIN;
constexpr int kUnroll = 4;  // channel pairs whose loads are issued together (one L2 round trip per 4 MFMAs)

// kWaves > 1: K is split over the waves of a workgroup and the partial tiles are summed through LDS -- for tiny
// launches (a scanline group is 1 position per image) where one wave per tile would leave the chip idle.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// a lane's MT adjacent A values (8- / 16-byte aligned for MT = 2 / 4 by construction of the packing)
template <int MT> __device__ inline void load_a(const float *p, bool ok, float (&a)[MT])
{
    if (MT == 4) {
        const f32x4 v = ok ? *reinterpret_cast<const f32x4 *>(p) : f32x4(0.f);
        a[0] = v[0]; a[1 % MT] = v[1]; a[2 % MT] = v[2]; a[3 % MT] = v[3];
    } else if (MT == 2) {
        const f32x2 v = ok ? *reinterpret_cast<const f32x2 *>(p) : f32x2(0.f);
        a[0] = v[0]; a[1 % MT] = v[1];
    } else {
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = ok ? p[m] : 0.f;
    }
}

// kWaves > 1 (MT == 1): K is split over the waves of a workgroup and the partial tiles are summed through LDS -- for tiny
// launches (a scanline group is 1 position per image) where one wave per tile would leave the chip idle.
// kWaves == 1: one wave computes MT row tiles (the weights' pack factor g.mt == MT).
template <int kWaves, int MT>
__global__ __launch_bounds__(64 * kWaves) void masked_conv_pos_kernel(const MaskedLaunch g)
{
    static_assert(kWaves == 1 || MT == 1, "split-K variant handles one tile per workgroup");
    __shared__ float partial[kWaves > 1 ? kWaves - 1 : 1][16][64];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int lane = tid & 63, col = lane & 31, khalf = lane >> 5;
    const int chunks_per_group = g.tiles_per_group / MT;       // host guarantees divisibility when MT > 1
    const int grp_o = blockIdx.y / chunks_per_group, ti0 = (blockIdx.y - grp_o * chunks_per_group) * MT;
    const bool row_ok = ti0 * 32 + col < g.gs_out || MT > 1;  // MT > 1 only with whole tiles (gs_out % (32 MT) == 0)
    // packed position of this lane's A value(s): see MaskedLaunch::mt
    const int a_off = grp_o * g.gs_out + (ti0 / g.mt) * 32 * g.mt + col * g.mt + (ti0 % g.mt);

    const int64_t pj = static_cast<int64_t>(blockIdx.x) * 32 + col;
    const bool pos_ok = pj < g.n_pos;
    const int hw = g.h * g.w_;
    int b = 0, py = 0, px = 0;
    if (pos_ok) {
        const int32_t f = g.pos[pj];
        b = f / hw;
        const int p = f - b * hw;
        py = p / g.w_;
        px = p - py * g.w_;
    }
    const int32_t centre = pos_ok ? g.topo_out[grp_o * hw + py * g.w_ + px] : 0;
    const float *xb = g.x + static_cast<int64_t>(b) * g.cin * hw;

    // Step rule (coding loop): the value of (output group, position) only depends on elements coded before ITS step
    // (mask < / <=), so what an earlier step wrote is still exact and what a later step needs is computed then.  A tile
    // is therefore evaluated only if some position has this output group in the current step -- or, for groups that
    // carry no topo id (-1: the prior half of the merger's hidden layers), at the first step that visits the position.
    // With G channel groups this removes (G-1)/G of the work of channel-wise schedules.  Wave-uniform, and the same in
    // all waves of a split-K workgroup (same positions, same group).
    if (g.step != kNoStep) {
        const bool need = pos_ok && (centre == g.step || (centre < 0 && g.first[py * g.w_ + px] == g.step));
        if (__ballot(need) == 0ull) return;
    }

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    // K-slice of this workgroup (blockIdx.z) and of this wave inside it: taps t = tap_slice (mod tap_slices), channel
    // pairs vwave, vwave + nvwaves, ... of every slab
    const int tap_slice = blockIdx.z % g.tap_slices, vwave = (blockIdx.z / g.tap_slices) * kWaves + wave;
    const int nvwaves = g.pair_slices * kWaves;

    for (int t = 0; t < g.ntaps; ++t) {
        if (t % g.tap_slices != tap_slice) continue;
        const int dy = t / g.ksize - g.pad, dx = t % g.ksize - g.pad;
        const int yy = py + dy, xx = px + dx;
        const bool inside = pos_ok && yy >= 0 && yy < g.h && xx >= 0 && xx < g.w_;
        const int noff = yy * g.w_ + xx;
        for (int gin = 0; gin < g.gi; ++gin) {
            bool open = false;
            if (inside) {
                const int32_t tn = g.topo_in[gin * hw + noff];
                open = g.allow_same ? (tn <= centre) : (tn < centre);
            }
            if (__ballot(open) == 0ull) continue;  // wave-uniform skip of an all-masked slab (same in every wave)
            const int c_beg = gin * g.gs_in, c_end = c_beg + g.gs_in;
            const float *wt = g.w + (static_cast<int64_t>(t) * g.cin) * g.coutp + a_off;
            // this wave's share of the slab: channel pairs vwave, vwave + nvwaves, ...; kUnroll pairs per round trip
            for (int c = c_beg + 2 * vwave; c < c_end; c += 2 * nvwaves * kUnroll) {
                float af[kUnroll][MT], bf[kUnroll];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    const int ci = c + u * 2 * nvwaves + khalf;
                    const bool ci_ok = ci < c_end;
                    load_a<MT>(wt + static_cast<int64_t>(ci) * g.coutp, row_ok && ci_ok, af[u]);
                    bf[u] = (open && ci_ok) ? xb[static_cast<int64_t>(ci) * hw + noff] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < kUnroll; ++u)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u][m], bf[u], acc[m], 0, 0, 0);
            }
        }
    }

    // sum the K-slices of the waves (fixed order: deterministic, same result in encoder and decoder)
    if (kWaves > 1) {
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) partial[wave - 1][r][lane] = acc[0][r];
        }
        __syncthreads();
    }
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < kWaves - 1; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] += partial[w][r][lane];
        if (g.scratch) {  // one K-slice of several: hand the partial tile to masked_conv_reduce_kernel
            const int64_t tile = static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x;
            float *dst = g.scratch + ((tile * gridDim.z + blockIdx.z) * 16) * 64 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r * 64] = acc[0][r];
            return;
        }
        if (pos_ok) {
            float *yb = g.y + (static_cast<int64_t>(b) * g.out_total + g.out_off) * hw + py * g.w_ + px;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rg = (ti0 + m) * 32 + 8 * (r >> 2) + 4 * khalf + (r & 3);
                    if (rg < g.gs_out) {
                        const int co = grp_o * g.gs_out + rg;
                        float v = acc[m][r] + g.bias[co];
                        if (g.act == BASIC_ACT_LEAKY_RELU) v = v > 0.f ? v : 0.01f * v;
                        else if (g.act == BASIC_ACT_RELU) v = v > 0.f ? v : 0.f;
                        yb[static_cast<int64_t>(co) * hw] = v;
                    }
                }
        }
    }
}

// Second half of the cross-workgroup split-K: one workgroup per tile, wave w owns accumulator rows 4w .. 4w+3 of every
// lane; all K-slices' values of a row are loaded together (at most 16 x 4 loads in flight) and summed in slice order
// (deterministic); then bias, activation and store -- the epilogue of masked_conv_pos_kernel<*, 1>.
__global__ __launch_bounds__(256) void masked_conv_reduce_kernel(const MaskedLaunch g, int slices)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, khalf = lane >> 5;
    const int grp_o = blockIdx.y / g.tiles_per_group, ti = blockIdx.y - grp_o * g.tiles_per_group;
    const int64_t pj = static_cast<int64_t>(blockIdx.x) * 32 + col;
    const bool pos_ok = pj < g.n_pos;
    const int hw = g.h * g.w_;
    int b = 0, py = 0, px = 0;
    if (pos_ok) {
        const int32_t f = g.pos[pj];
        b = f / hw;
        const int p = f - b * hw;
        py = p / g.w_;
        px = p - py * g.w_;
    }
    if (g.step != kNoStep) {  // same rule as the partial kernel: skipped tiles have no partials
        const int32_t centre = pos_ok ? g.topo_out[grp_o * hw + py * g.w_ + px] : 0;
        const bool need = pos_ok && (centre == g.step || (centre < 0 && g.first[py * g.w_ + px] == g.step));
        if (__ballot(need) == 0ull) return;
    }
    const int64_t tile = static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x;
    const float *src = g.scratch + (tile * slices * 16 + 4 * wave) * 64 + lane;
    constexpr int kMaxSlices = 16;
    float v[kMaxSlices][4];
#pragma unroll
    for (int s = 0; s < kMaxSlices; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[s][q] = s < slices ? src[(s * 16 + q) * 64] : 0.f;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < kMaxSlices; ++s)
        if (s < slices) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += v[s][q];
        }
    if (!pos_ok) return;
    float *yb = g.y + (static_cast<int64_t>(b) * g.out_total + g.out_off) * hw + py * g.w_ + px;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rg = ti * 32 + 8 * wave + 4 * khalf + q;
        if (rg < g.gs_out) {
            const int co = grp_o * g.gs_out + rg;
            float o = acc[q] + g.bias[co];
            if (g.act == BASIC_ACT_LEAKY_RELU) o = o > 0.f ? o : 0.01f * o;
            else if (g.act == BASIC_ACT_RELU) o = o > 0.f ? o : 0.f;
            yb[static_cast<int64_t>(co) * hw] = o;
        }
    }
}

constexpr int kScratchTiles = 1024;  // partial tiles (4 KB each) a plan can hold: tiles x K-slices of the smallest launches

}  // namespace

struct basic_mconv_plan {
    int cin = 0, cout = 0, coutp = 0, ksize = 1, gi = 1, go = 1, allow_same = 0, act = 0;
    int mt = 1;  // row tiles per wave of the large-launch kernel = pack factor of d_w
    float *d_scratch = nullptr;  // K-slice partial tiles of the cross-workgroup split-K (one stream per plan at a time)
    float *d_w1 = nullptr;  // plain [tap][ci][co] copy for the split-K kernel when mt > 1 (coalesced 128-byte A rows)
    float *d_w = nullptr, *d_bias = nullptr;
};

extern "C" void basic_mconv_plan_destroy(basic_mconv_plan *p)
{
    if (!p) return;
    if (p->d_w) (void)hipFree(p->d_w);
    if (p->d_w1) (void)hipFree(p->d_w1);
    if (p->d_scratch) (void)hipFree(p->d_scratch);
    if (p->d_bias) (void)hipFree(p->d_bias);
    delete p;
}

extern "C" int basic_mconv_plan_create(const float *weight, const float *bias, int cin, int cout, int ksize,
                                       int in_groups, int out_groups, int allow_same_topogroup, int activation,
                                       basic_mconv_plan **out)
{
    BASIC_REQUIRE(weight && out && cin >= 1 && cout >= 1 && (ksize == 1 || ksize == 3 || ksize == 5),
                  "mconv_plan_create: bad geometry (odd kernel <= 5)");
    BASIC_REQUIRE(in_groups >= 1 && out_groups >= 1 && cin % in_groups == 0 && cout % out_groups == 0,
                  "mconv_plan_create: channels must divide into the channel groups");
    BASIC_REQUIRE(activation == BASIC_ACT_NONE || activation == BASIC_ACT_RELU || activation == BASIC_ACT_LEAKY_RELU,
                  "mconv_plan_create: bad activation");
    int rc = require_device();
    if (rc) return rc;
    auto *p = new (std::nothrow) basic_mconv_plan();
    if (!p) { set_error("out of host memory"); return BASIC_ERR_INVALID; }
    p->cin = cin; p->cout = cout; p->coutp = (cout + 3) / 4 * 4; p->ksize = ksize;
    p->gi = in_groups; p->go = out_groups; p->allow_same = allow_same_topogroup ? 1 : 0; p->act = activation;
    const int ntaps = ksize * ksize;
    const int gs_out = cout / out_groups;
    p->mt = 1;                                                   // row tiles per wave: whole 32-row tiles only
    if (gs_out % 32 == 0) {
        const int tiles = gs_out / 32;
        p->mt = tiles <= 5 ? tiles : tiles % 4 == 0 ? 4 : tiles % 3 == 0 ? 3 : tiles % 5 == 0 ? 5 : tiles % 2 == 0 ? 2 : 1;
    }
    if (const char *e = std::getenv("BASIC_MCONV_MAX_MT")) {  // tests: one tile per wave as the comparison point
        if (p->mt > std::atoi(e)) p->mt = 1;
    }
    if (const char *e = std::getenv("BASIC_MCONV_FORCE_MT")) {  // tuning: another pack factor where it divides the group's tiles
        const int f = std::atoi(e);
        if (f >= 1 && f <= 5 && gs_out % (32 * f) == 0) p->mt = f;
    }
    const int span = 32 * p->mt;
    std::vector<float> wp(static_cast<size_t>(ntaps) * cin * p->coutp, 0.f), hb(p->coutp, 0.f);
    std::vector<float> w1(p->mt > 1 ? wp.size() : 0, 0.f);
    for (int o = 0; o < cout; ++o) {
        const int grp = o / gs_out, r = o - grp * gs_out;
        const int packed = grp * gs_out + (r / span) * span + (r % 32) * p->mt + (r / 32) % p->mt;
        for (int c = 0; c < cin; ++c)
            for (int t = 0; t < ntaps; ++t) {
                const float v = weight[(static_cast<size_t>(o) * cin + c) * ntaps + t];
                wp[(static_cast<size_t>(t) * cin + c) * p->coutp + packed] = v;
                if (p->mt > 1) w1[(static_cast<size_t>(t) * cin + c) * p->coutp + o] = v;
            }
    }
    if (bias) std::memcpy(hb.data(), bias, sizeof(float) * cout);
    hipError_t e = hipMalloc(&p->d_w, wp.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->d_w, wp.data(), wp.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && p->mt > 1) e = hipMalloc(&p->d_w1, w1.size() * sizeof(float));
    if (e == hipSuccess && p->mt > 1) e = hipMemcpy(p->d_w1, w1.data(), w1.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&p->d_scratch, static_cast<size_t>(kScratchTiles) * 16 * 64 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&p->d_bias, hb.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->d_bias, hb.data(), hb.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { basic_mconv_plan_destroy(p); return hip_fail(e, "mconv_plan_create", __FILE__, __LINE__); }
    *out = p;
    return BASIC_OK;
}

static int mconv_forward(const basic_mconv_plan *p, const float *d_x, const int32_t *d_topo_in, const int32_t *d_topo_out,
                         int batch, int h, int w, const int32_t *d_pos, int64_t n_pos, float *d_y, int out_channels_total,
                         int out_channel_offset, int step, const int32_t *d_first, void *hip_stream);

extern "C" int basic_mconv_forward_pos_dev(const basic_mconv_plan *p, const float *d_x, const int32_t *d_topo_in,
                                           const int32_t *d_topo_out, int batch, int h, int w, const int32_t *d_pos,
                                           int64_t n_pos, float *d_y, int out_channels_total, int out_channel_offset,
                                           void *hip_stream)
{
    return mconv_forward(p, d_x, d_topo_in, d_topo_out, batch, h, w, d_pos, n_pos, d_y, out_channels_total, out_channel_offset,
                         kNoStep, nullptr, hip_stream);
}

extern "C" int basic_mconv_forward_step_dev(const basic_mconv_plan *p, const float *d_x, const int32_t *d_topo_in,
                                            const int32_t *d_topo_out, int batch, int h, int w, const int32_t *d_pos,
                                            int64_t n_pos, float *d_y, int out_channels_total, int out_channel_offset,
                                            int step, const int32_t *d_first_step, void *hip_stream)
{
    BASIC_REQUIRE(d_first_step && step != kNoStep, "mconv_forward_step: first-step map and a step are required");
    return mconv_forward(p, d_x, d_topo_in, d_topo_out, batch, h, w, d_pos, n_pos, d_y, out_channels_total, out_channel_offset,
                         step, d_first_step, hip_stream);
}

static int mconv_forward(const basic_mconv_plan *p, const float *d_x, const int32_t *d_topo_in, const int32_t *d_topo_out,
                         int batch, int h, int w, const int32_t *d_pos, int64_t n_pos, float *d_y, int out_channels_total,
                         int out_channel_offset, int step, const int32_t *d_first, void *hip_stream)
{
    BASIC_REQUIRE(p && d_x && d_topo_in && d_topo_out && d_pos && d_y && batch >= 1 && h >= 1 && w >= 1 && n_pos >= 0,
                  "mconv_forward_pos: bad argument");
    BASIC_REQUIRE(out_channel_offset >= 0 && out_channel_offset + p->cout <= out_channels_total,
                  "mconv_forward_pos: output channel window out of range");
    BASIC_REQUIRE(static_cast<int64_t>(batch) * h * w < (1ll << 31), "mconv_forward_pos: position index overflow");
    if (n_pos == 0) return BASIC_OK;
    MaskedLaunch g{};
    g.x = d_x; g.y = d_y; g.w = p->d_w; g.bias = p->d_bias; g.topo_in = d_topo_in; g.topo_out = d_topo_out;
    g.pos = d_pos; g.n_pos = n_pos; g.batch = batch; g.cin = p->cin; g.cout = p->cout; g.coutp = p->coutp;
    g.h = h; g.w_ = w; g.gi = p->gi; g.go = p->go; g.gs_in = p->cin / p->gi; g.gs_out = p->cout / p->go;
    g.tiles_per_group = (g.gs_out + 31) / 32;
    g.out_total = out_channels_total; g.out_off = out_channel_offset;
    g.ntaps = p->ksize * p->ksize; g.pad = p->ksize / 2; g.ksize = p->ksize; g.allow_same = p->allow_same; g.act = p->act;
    g.mt = p->mt; g.step = step; g.first = d_first;
    g.scratch = nullptr; g.tap_slices = 1; g.pair_slices = 1;
    const unsigned ptiles = static_cast<unsigned>((n_pos + 31) / 32), rtiles = static_cast<unsigned>(g.tiles_per_group * g.go);
    hipStream_t st = as_stream(hip_stream);
    // few tiles -> split K over 4 waves per tile; many tiles -> one wave per MT row tiles already fills the chip
    // (BASIC_MCONV_SPLITK_BELOW moves the switch-over: tests drive both variants over the same inputs)
    int64_t split_below = 4096;
    if (const char *e = std::getenv("BASIC_MCONV_SPLITK_BELOW")) split_below = std::atoll(e);
    const int64_t tiles = static_cast<int64_t>(ptiles) * rtiles;
    int64_t cross_below = 256;   // fewer tiles than this: K is also split over workgroups (BASIC_MCONV_CROSS_BELOW)
    if (const char *e = std::getenv("BASIC_MCONV_CROSS_BELOW")) cross_below = std::atoll(e);
    if (tiles < split_below) {
        if (p->mt > 1) { g.w = p->d_w1; g.mt = 1; }
        int slices = 1;
        if (tiles < cross_below) {
            // ~512 workgroups in flight; the per-wave K loop of a scanline step drops from ~70 to ~5 load round trips
            while (slices < 16 && tiles * slices * 2 <= 512 && tiles * slices * 2 <= kScratchTiles) slices *= 2;
            // a slab must keep at least one channel pair per wave
            while (slices > 1 && (slices / (g.ntaps > 1 ? (slices < 4 ? slices : 4) : 1)) * 4 * 2 > g.gs_in) slices /= 2;
        }
        if (slices > 1) {
            g.tap_slices = g.ntaps > 1 ? (slices < 4 ? slices : 4) : 1;
            g.pair_slices = slices / g.tap_slices;
            g.scratch = p->d_scratch;
            hipLaunchKernelGGL((masked_conv_pos_kernel<4, 1>), dim3(ptiles, rtiles, slices), dim3(256), 0, st, g);
            hipLaunchKernelGGL(masked_conv_reduce_kernel, dim3(ptiles, rtiles), dim3(256), 0, st, g, slices);
        } else {
            hipLaunchKernelGGL((masked_conv_pos_kernel<4, 1>), dim3(ptiles, rtiles), dim3(256), 0, st, g);
        }
    }
    else if (p->mt == 5)
        hipLaunchKernelGGL((masked_conv_pos_kernel<1, 5>), dim3(ptiles, rtiles / 5), dim3(64), 0, st, g);
    else if (p->mt == 4)
        hipLaunchKernelGGL((masked_conv_pos_kernel<1, 4>), dim3(ptiles, rtiles / 4), dim3(64), 0, st, g);
    else if (p->mt == 3)
        hipLaunchKernelGGL((masked_conv_pos_kernel<1, 3>), dim3(ptiles, rtiles / 3), dim3(64), 0, st, g);
    else if (p->mt == 2)
        hipLaunchKernelGGL((masked_conv_pos_kernel<1, 2>), dim3(ptiles, rtiles / 2), dim3(64), 0, st, g);
    else
        hipLaunchKernelGGL((masked_conv_pos_kernel<1, 1>), dim3(ptiles, rtiles), dim3(64), 0, st, g);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}
