// Persistent scan-line autoregressive coding loop: ONE launch walks all H*W coding steps of a batch.
//
// Reference: TopoGroupPGMPriorCoder._encode_with_pgm / _pgm_generate (cbench/modules/prior_model/prior_coder/pgm_coder.py:
// 912-981) with the "scanline" topo groups (one group per spatial position, raster order, :1416-1491), the masked 5x5
// context convolution (cbench/nn/layers/masked_conv.py:102-228: for scan-line groups the mask keeps exactly the causal
// raster neighbours) and the 1x1 parameter-merger layers (masked_conv.py:262-300 / pgm_coder.py:1606-1638: dense for one
// channel group), followed by the Gaussian index / quantise step (pgm_coder.py:735-821, 927-941).
//
// Per step the reference (and this library's per-step path) launches ~6-8 tiny dependent kernels; at batch 1 a Kodak-shaped
// latent has 1,536 steps, so the loop is pure launch + dependent-load latency (65-100 us per step).  Here the weights of
// every layer (7.6 MB for C = 192) are spread ONCE over the LDS of `nwg` compute units -- workgroup w owns a fixed slice of
// the output rows of every layer -- and a step is: every workgroup computes its rows of a layer for all images (inputs
// staged in LDS, one wave per dot product), publishes them with write-through (sc1) stores, and all workgroups meet at a
// device-wide barrier before the next layer.  Four barriers per coding step instead of four-plus kernel boundaries with
// their dependent weight re-reads.
//
// Summation order.  The integers coded here must equal, bit for bit, what the per-step path (csrc/mconv.hip) derives from
// the same layers at any batch size -- a stream may be encoded by one path and decoded by the other.  Every dot product is
// therefore evaluated in mconv.hip's CANONICAL order: the K axis of a layer (context layer: [causal tap][channel]; dense
// layer: its input channels in `in_groups` equal groups) is cut, group by group, into blocks of BASIC_MCONV_BLOCK_CHANNELS
// channels; a block's partial is one fp32 FMA chain over its channels in ascending order starting from zero (= the
// v_mfma_f32_32x32x2_f32 chain of the masked convolution, scripts/micro/mfma_arith.hip); the partials are added in block
// order, then the bias.  Taps outside the image enter as zeros, as the masked-out slabs do there.
//
// Cross-workgroup exchange (round 3): DATA-TAGGED GRANULES.  Every value one workgroup hands to the others -- a layer output,
// a coded latent, a (mean, table row) pair for the decoder wavefronts -- travels as ONE naturally aligned 8-byte word
// { float bits | tag << 32 } written with an agent-scope (sc1, write-through) store and read with agent-scope loads; the tag
// is the coding step + 1 for the per-step buffers and the position + 1 for the position-major copy of the coded latent.  A
// consumer polls the word until its tag is the one it expects: the 8-byte store is single-copy atomic, so a matching tag means
// the value beside it is the matching value, and no barrier, no store drain and no separate flag is needed -- a layer
// exchange costs one store-to-load propagation (~2 us) instead of store drain + device barrier + loads (~5 us, measured in
// round 2).  The buffers are zeroed before the launch (tag 0 never matches).  Overwriting is safe without a second
// handshake: workgroup X can write step p+1 of layer l only after it has consumed step p+1 of layer l-1 from EVERY workgroup
// (for l = 0: position p of the coded latent from every producer), and a workgroup that has produced that has -- in program
// order -- already consumed step p of layer l.  Every poll is bounded (kSpinLimit, and the launch's error flag is checked while
// spinning): a launch whose grid is not fully resident gives up and reports it instead of hanging.  One workgroup per compute
// unit (the LDS footprint guarantees it), as MI355X_MICROARCH.md's inter-workgroup visibility section requires of spin-waits.
#include "common.h"
#include "wave_decoder.h"

#include <algorithm>
#include <mutex>

#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

using namespace basic;

namespace {

constexpr int kThreads = 256;
constexpr int kMaxLayers = 5;   // context convolution + up to four dense layers
constexpr int kMaxTaps = 24;    // causal taps of a k x k window: (k/2) * k + k/2  (k = 7 -> 24)
constexpr unsigned kSpinLimit = 4000000u;

struct ScanArgs {
    const float *y;       // [B][C][HW]   (encoder input)
    const float *priorT;  // [B][HW][P]   position-major copy of the prior (nullptr when P == 0)
    float *ybuf;          // [B][C][HW]   coded latent (output)
    uint64_t *yT;         // [B][HW][C]   position-major working copy of the coded latent as granules (tag = position + 1)
    int32_t *sym, *idx;   // [B][HW * C]  position-major: element p * C + c
    const float *table;
    int table_len;
    int B, C, H, W, P;
    int nlayers, ntaps, vec4;   // vec4: every K, row count and C is a multiple of 4 -> 16-byte exchanges and LDS reads
    int rows[kMaxLayers], kdim[kMaxLayers], rpw[kMaxLayers], woff[kMaxLayers], act_after[kMaxLayers];
    int kgroup[kMaxLayers], bpg[kMaxLayers], kpad[kMaxLayers];   // channels / canonical blocks per K group; padded LDS row (see padded_k)
    const float *w[kMaxLayers], *bias[kMaxLayers];
    uint64_t *act[kMaxLayers];   // exchange buffers [B][rows_l] of granules (tag = coding step + 1)
    int tap_off[kMaxTaps], tap_dy[kMaxTaps], tap_dx[kMaxTaps];
    int bc, xs_off, ps_off, tab_off, part_off, part_floats, flag_off, bias_off;   // LDS float offsets (the whole LDS is dynamic)
    int x0_off, early_off, desc_off;   // pipelined kernel only: context window [B][K0], early sums, unit descriptors
    unsigned *bar;
    int *err;
    int debug;   // BASIC_SCAN_DEBUG timing ablations (wrong results): 2 no input staging, 4 no dot products
    long long *prof;   // BASIC_SCAN_PROFILE=1: [layer][4] 100 MHz ticks of workgroup 0 spent staging (incl. waiting) / in the block dots / finishing / in the
                       // Gaussian step, summed over the steps; [4 * kMaxLayers .. +1] = shader clocks and ticks of the whole loop
    // decoder only
    int ncompute;          // workgroups [0, ncompute) compute, the rest decode (4 image streams each)
    uint64_t *mu;          // [B][C] means of the current step (compute -> decoder workgroups), granules
    uint64_t *idx_step;    // [B][C] table rows of the current step, granules (the row's bits in the value half)
    RansFastView tv;
    const uint32_t *words; // all streams back to back
    const int64_t *word_off;   // [B + 1]
    // batched kernel only (scanline_batched_kernel; nbt == 0 for the other two): the batch is the fastest dimension of every
    // exchanged array -- yT [HW][C][nbt], act[l] [rows_l][nbt], priorT [HW][P][nbt]; mu / idx_step stay [image][C]
    int nbt;               // columns of the exchange arrays: 32 * column tiles (a column tile = 32 images = the N of an MFMA tile)
    int nw, nd;            // compute workgroups per column tile; the first nd of them hold the dense layers' row tiles
    int bpt;               // canonical blocks per context tap = C / 64
    int nblk[kMaxLayers], rt[kMaxLayers];   // canonical blocks of a layer's K axis; its 32-row tiles
    int ctx_blocks;        // blocks of the first dense layer fed by the context layer (the rest: the prior)
    int tile_off[kMaxLayers];   // dense-role workgroups: LDS float offset of a layer's partial tiles
    float *wlate;          // [compute workgroups][256 threads][32]: a context workgroup's late blocks as A fragments (written by the launch itself)
};

typedef float f4 __attribute__((ext_vector_type(4)));
__host__ __device__ __forceinline__ int64_t bm_gran(int c, int col, int nbt);   // batched kernel's exchange layout (defined with it)

__device__ __forceinline__ uint64_t ld_gran(const uint64_t *p)
{
    return __hip_atomic_load(const_cast<uint64_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_gran(uint64_t *p, uint32_t bits, uint32_t tag)
{
    __hip_atomic_store(p, static_cast<uint64_t>(bits) | (static_cast<uint64_t>(tag) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_gran(uint64_t *p, float v, uint32_t tag) { st_gran(p, __float_as_uint(v), tag); }

// Workgroup barrier for LDS hand-overs only: waits for this wave's LDS traffic, NOT for its global stores -- __syncthreads()
// also drains vmcnt, i.e. it would wait ~1-1.5 us per layer for the write-through stores of the granules just published.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// polls a granule until it carries `tag`; false when the launch is poisoned (bounded spin / another workgroup gave up)
__device__ __forceinline__ bool wait_gran(const ScanArgs &a, const uint64_t *p, uint32_t tag, uint64_t &g)
{
    unsigned spins = 0;
    while (static_cast<uint32_t>(g >> 32) != tag) {
        if (++spins > kSpinLimit || (spins % 1024u == 0u && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        g = ld_gran(p);
    }
    return true;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// argmin_j |s - table[j]|, first minimum (pgm_coder.py:802-821; same as entropy.hip::nearest_scale).  On a strictly
// increasing table (the scale table is: checked once per workgroup) the minimum lies at the first entry >= s or the one
// before it, the earlier of the two on a tie: a 6-step search instead of a 64-step scan on the step's critical path.
__device__ __forceinline__ int nearest_scale(float s, const float *tab, int n, bool sorted)
{
    if (sorted && __builtin_isfinite(s)) {   // (NaN / infinite scales: the scan below defines the result)
        int lo = 0, hi = n;   // first index with tab[idx] >= s, in [0, n]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (tab[mid] < s) lo = mid + 1; else hi = mid;
        }
        if (lo == 0) return 0;
        if (lo == n) return n - 1;
        return fabsf(s - tab[lo]) < fabsf(s - tab[lo - 1]) ? lo : lo - 1;
    }
    int best = 0;
    float bd = fabsf(s - tab[0]);
    for (int j = 1; j < n; ++j) {
        const float d = fabsf(s - tab[j]);
        if (d < bd) { bd = d; best = j; }
    }
    return best;
}

// ---- inputs of `nb` images for layer l at position p -> xs[bi][k]; every exchanged value is polled until it carries the
//      expected tag (context layer: the neighbour's position + 1; dense layers: this step + 1).  Returns false when poisoned.
__device__ __forceinline__ bool stage_inputs(const ScanArgs &a, int l, int p, int py, int px, int b0, int nb, float *xs)
{
    const int tid = threadIdx.x, HW = a.H * a.W, K = a.kdim[l];
    const int total = nb * K, prev = l ? a.rows[l - 1] : 0;
    constexpr int U = 12;   // loads in flight per thread: the context layer at batch 1 is 9 granules per thread -- ONE round trip, not two
    bool ok = true;
    for (int e0 = tid; e0 < total; e0 += kThreads * U) {
        const uint64_t *src[U];
        uint64_t g[U];
        uint32_t want[U];
        float plain[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {   // all loads of the batch in flight before the first tag is looked at
            const int e = e0 + u * kThreads;
            src[u] = nullptr; g[u] = 0ull; want[u] = 0u; plain[u] = 0.f;
            if (e < total) {
                const int bi = e / K, kk = e - bi * K;
                const int64_t b = b0 + bi;
                if (l == 0) {   // causal neighbourhood of the coded latent: [tap][c]
                    const int t = kk / a.C, c = kk - t * a.C;
                    const int ny = py + a.tap_dy[t], nx = px + a.tap_dx[t];
                    if (ny >= 0 && nx >= 0 && nx < a.W) {
                        src[u] = a.yT + ((b * HW + p + a.tap_off[t]) * a.C + c);
                        want[u] = static_cast<uint32_t>(p + a.tap_off[t] + 1);
                    }
                } else if (kk < prev) {
                    src[u] = a.act[l - 1] + b * prev + kk;
                    want[u] = static_cast<uint32_t>(p + 1);
                } else if (a.priorT) {   // cat(ctx, prior); the prior is an input of the launch: ordinary loads
                    plain[u] = a.priorT[(b * HW + p) * a.P + (kk - prev)];
                }
                if (src[u]) g[u] = ld_gran(src[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * kThreads;
            if (e < total) {
                if (src[u]) {
                    ok = wait_gran(a, src[u], want[u], g[u]) && ok;
                    xs[e] = __uint_as_float(static_cast<uint32_t>(g[u]));
                } else {
                    xs[e] = plain[u];
                }
            }
        }
    }
    // A position at the start of a row has no left neighbour, so nothing above makes it wait for the previous position --
    // but the overwrite argument of the exchange buffers (header comment) needs every workgroup's step p+1 to start after
    // position p is coded: wait for its granules without using the values.
    if (l == 0 && p > 0 && px == 0) {
        for (int e = tid; e < nb * a.C; e += kThreads) {
            const int bi = e / a.C, c = e - bi * a.C;
            const uint64_t *src = a.yT + ((static_cast<int64_t>(b0 + bi) * HW + p - 1) * a.C + c);
            uint64_t g = ld_gran(src);
            ok = wait_gran(a, src, static_cast<uint32_t>(p), g) && ok;
        }
    }
    return ok;
}

constexpr int kKB = BASIC_MCONV_BLOCK_CHANNELS;
constexpr int kBlockPad = 4;   // LDS floats between the blocks of a weight row: lanes working on different blocks of a row start on different banks

// LDS position of element k of a weight row whose K axis has groups of kg channels (bpg blocks each)
__host__ __device__ __forceinline__ int padded_k(int k, int kg, int bpg)
{
    const int g = k / kg, kk = k - g * kg;
    return g * (kg + kBlockPad * bpg) + kk + kBlockPad * (kk / kKB);
}

// ---- one canonical block of one output: an FMA chain over `len` channels in ascending order, starting from zero.  A THREAD
//      owns a (block, image, row) unit; threads of a wave take consecutive rows of the same block (weight rows are padded
//      so that they start on different banks; lanes of one image read the same input address: LDS broadcast).
__device__ __forceinline__ float block_dot(const float *wr, const float *xr, int len, int vec4)
{
    float p = 0.f;
    if (vec4) {
        const f4 *x4 = reinterpret_cast<const f4 *>(xr), *w4 = reinterpret_cast<const f4 *>(wr);
        const int n4 = len >> 2;
        int k4 = 0;
        for (; k4 + 4 <= n4; k4 += 4) {   // 16 channels of loads in flight, one chain
            const f4 x0 = x4[k4], x1 = x4[k4 + 1], x2 = x4[k4 + 2], x3 = x4[k4 + 3];
            const f4 w0 = w4[k4], w1 = w4[k4 + 1], w2 = w4[k4 + 2], w3 = w4[k4 + 3];
#define BASIC_FMA4(WV, XV) p = fmaf(WV[0], XV[0], p); p = fmaf(WV[1], XV[1], p); p = fmaf(WV[2], XV[2], p); p = fmaf(WV[3], XV[3], p)
            BASIC_FMA4(w0, x0); BASIC_FMA4(w1, x1); BASIC_FMA4(w2, x2); BASIC_FMA4(w3, x3);
        }
        for (; k4 < n4; ++k4) { const f4 xv = x4[k4], wv = w4[k4]; BASIC_FMA4(wv, xv); }
#undef BASIC_FMA4
    } else {
        for (int k = 0; k < len; ++k) p = fmaf(wr[k], xr[k], p);
    }
    return p;
}

// ---- in-place rANS decoder of ONE stream held by one wavefront (decode_stream semantics, csrc/ans/rans64.cpp:501-598;
//      the search image, the arithmetic and the chunk schedule of rans_decode_fast_kernel in rans.hip: a lone wave is bound by
//      its instruction count, so a symbol's 16-byte image entry -- which depends on its table row only, not on the coder
//      state -- is read from LDS two symbols ahead, every lane advances the state for ITS candidate symbol, and one
//      compare + ballot picks the lane that is right; renormalisation, the bypass sentinel and wide rows (image frequency 0)
//      all hide behind the single test "new state < 2^31")
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

using wavedec::static_pairs;
using wavedec::WaveDecoder;   // the serial rANS chain of one stream (wave_decoder.h)

// ================= decoder workgroups: one wavefront per image stream, the search image in LDS =================
__device__ __forceinline__ void decoder_workgroup(const ScanArgs &a, float *lds)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    const int HW = a.H * a.W;
    uint32_t *img = reinterpret_cast<uint32_t *>(lds);
    for (int e = tid; e < a.tv.image_words; e += kThreads) img[e] = a.tv.image[e];
    // per table row {image offset, size, symbol offset}: three dependent global loads per chunk on the step's critical path otherwise
    u32x4 *rowtab = reinterpret_cast<u32x4 *>(img + ((a.tv.image_words + 3) & ~3));
    for (int r = tid; r < a.tv.rows; r += kThreads)
        rowtab[r] = u32x4{a.tv.meta[r], static_cast<uint32_t>(a.tv.sizes[r]), static_cast<uint32_t>(a.tv.offsets[r]), 0u};
    __syncthreads();
    const int b = (wg - a.ncompute) * (kThreads / 64) + wave;
    if (b >= a.B) return;
    __builtin_amdgcn_s_setprio(3);   // a serial chain: never lose the issue arbitration to the waves spinning beside it
    WaveDecoder d;
    {
        const int64_t w0 = a.word_off[b];
        d.init(img, a.words + w0, static_cast<int>(a.word_off[b + 1] - w0), a.tv.precision, a.tv.bypass_precision, a.tv.bypass != 0, -1, 0ull, lane);
    }
    const uint64_t *pi0 = a.idx_step + static_cast<int64_t>(b) * a.C, *pm0 = a.mu + static_cast<int64_t>(b) * a.C;
    // where a chunk's results go: a uniform pointer per (step, chunk) + a lane term that never changes (channel c0 + lane; the
    // batched exchange layout is linear in c0: bm_gran(c0 + l) = bm_gran(l) + c0 * nbt for chunk starts c0)
    const int64_t bC = static_cast<int64_t>(b) * a.C;
    const uint32_t lane_g = a.nbt ? static_cast<uint32_t>(bm_gran(lane, b, a.nbt)) : static_cast<uint32_t>(lane);
    const uint32_t lane_y = static_cast<uint32_t>(lane) * static_cast<uint32_t>(HW);
    for (int p = 0; p < HW; ++p) {
        const uint32_t tag = static_cast<uint32_t>(p + 1);
        // this step's (table row, mean) granules come from the compute workgroups that own the channels; the next chunk's
        // are requested before the current chunk is decoded
        uint64_t gi = 0ull, gm = 0ull;
        if (lane < a.C) { gi = ld_gran(pi0 + lane); gm = ld_gran(pm0 + lane); }
        for (int c0 = 0; c0 < a.C; c0 += 64) {
            const int c = c0 + lane;
            bool ok = true;
            d.line_up(lane);   // the chunk's 64 stream words, cut while the step's parameters are still on their way
            const long long tw0 = a.prof ? wall_clock64() : 0;
            if (c < a.C) ok = wait_gran(a, pi0 + c, tag, gi) && wait_gran(a, pm0 + c, tag, gm);
            if (__ballot(!ok) != 0ull) return;   // poisoned launch: wave-uniform exit
            const long long tw1 = a.prof ? wall_clock64() : 0;
            int32_t row = static_cast<int32_t>(static_cast<uint32_t>(gi));
            const float mu = __uint_as_float(static_cast<uint32_t>(gm));
            gi = 0ull; gm = 0ull;
            if (c + 64 < a.C) { gi = ld_gran(pi0 + c + 64); gm = ld_gran(pm0 + c + 64); }
            row = row < 0 ? 0 : (row >= a.tv.rows ? a.tv.rows - 1 : row);
            const u32x4 rt = rowtab[row];
            const int cnt = (a.C - c0) < 64 ? (a.C - c0) : 64;
            const long long tc2 = a.prof ? clock64() : 0;
            const int32_t mine = d.decode_chunk(rt[0], static_cast<int32_t>(rt[1]), cnt, lane) - 1;
            const long long tc3 = a.prof ? clock64() : 0;
            if (c < a.C) {
                const int32_t value = mine + static_cast<int32_t>(rt[2]);
                const float v = static_cast<float>(value) + mu;           // pgm_coder.py:973-978
                const int64_t at = bC * HW + static_cast<int64_t>(p) * a.C + c0;   // [b][p][c0] of sym / idx
                uint64_t *yt = a.yT + (a.nbt ? (static_cast<int64_t>(p) * a.C + c0) * a.nbt : (static_cast<int64_t>(b) * HW + p) * a.C + c0);
                st_gran(yt + lane_g, v, tag);   // first: the compute workgroups wait for it
                (a.sym + at)[lane] = value;
                (a.idx + at)[lane] = row;
                (a.ybuf + ((bC + c0) * HW + p))[lane_y] = v;
            }
            if (a.prof) {   // first stream: ticks waiting for the step's parameters / decoding and publishing
                const long long tc4 = clock64(), tw4 = wall_clock64();   // (read before the slots are touched: their updates wait for memory)
                if (b == 0 && lane == 0) {
                    a.prof[4 * kMaxLayers + 2] += tw1 - tw0;
                    a.prof[4 * kMaxLayers + 3] += tw4 - tw1;
                    a.prof[4 * kMaxLayers + 4] += tc3 - tc2;            // shader clocks inside decode_chunk
                    a.prof[4 * kMaxLayers + 5] += tc4 - tc3;            // ... publishing
                    a.prof[4 * kMaxLayers + 6] = d.position();
                }
            }
        }
    }
}

template <bool DECODE>
__global__ __launch_bounds__(kThreads) void scanline_persistent_kernel(const ScanArgs a)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int wg = blockIdx.x;
    const int HW = a.H * a.W;
    const int last = a.nlayers - 1;

    if (DECODE && wg >= a.ncompute) { decoder_workgroup(a, lds); return; }

    // ================= compute workgroups: a fixed slice of every layer's rows, weights resident in LDS =================
    int rows_w[kMaxLayers];
    for (int l = 0; l < a.nlayers; ++l) {
        int r = a.rows[l] - wg * a.rpw[l];
        r = r < 0 ? 0 : (r > a.rpw[l] ? a.rpw[l] : r);
        rows_w[l] = r;
        const float *src = a.w[l] + static_cast<int64_t>(wg) * a.rpw[l] * a.kdim[l];
        float *dst = lds + a.woff[l];
        const int Kl = a.kdim[l], Kp = a.kpad[l];
        for (int e = tid; e < r * Kl; e += kThreads) { const int rr = e / Kl; dst[rr * Kp + padded_k(e - rr * Kl, a.kgroup[l], a.bpg[l])] = src[e]; }
    }
    float *tab = lds + a.tab_off;
    for (int e = tid; e < a.table_len; e += kThreads) tab[e] = a.table[e];
    float *xs = lds + a.xs_off, *ps = lds + a.ps_off, *part = lds + a.part_off;
    int *s_flag = reinterpret_cast<int *>(lds + a.flag_off);
    // this workgroup's biases in LDS (a global load per output on the finishing threads cost ~1 us per layer and step)
    float *bias_l[kMaxLayers];
    {
        int off = a.bias_off;
        for (int l = 0; l < a.nlayers; ++l) {
            bias_l[l] = lds + off;
            for (int e = tid; e < rows_w[l]; e += kThreads) bias_l[l][e] = a.bias[l] ? a.bias[l][wg * a.rpw[l] + e] : 0.f;
            off += a.rpw[l];
        }
    }
    if (tid == 0) { s_flag[0] = 0; s_flag[1] = 1; }
    __syncthreads();
    for (int e = tid; e + 1 < a.table_len; e += kThreads)
        if (!(tab[e] < tab[e + 1])) s_flag[1] = 0;   // not strictly increasing: nearest_scale scans
    __syncthreads();
    const bool tab_sorted = s_flag[1] != 0;
    const long long loop_c0 = a.prof ? clock64() : 0, loop_t0 = a.prof ? wall_clock64() : 0;
    for (int p = 0; p < HW; ++p) {
        const int py = p / a.W, px = p - py * a.W;
        const uint32_t tag = static_cast<uint32_t>(p + 1);
        for (int l = 0; l <= last; ++l) {
            const int K = a.kdim[l], rw = rows_w[l];
            if (rw == 0) continue;   // a workgroup without rows of this layer has nothing to produce -- and nothing to wait for
            const float *wl = lds + a.woff[l];
            const int r_first = wg * a.rpw[l];
            for (int b0 = 0; b0 < a.B; b0 += a.bc) {
                const int nb = (a.B - b0) < a.bc ? (a.B - b0) : a.bc;
                bool ok = true;
                const long long t0 = a.prof ? wall_clock64() : 0;
                // encoder: this thread's latent of the Gaussian step, requested before the last layer's inputs are waited for
                float y_pre = 0.f;
                if (!DECODE && l == last && tid < nb * (rw >> 1)) {
                    const int bi = tid / (rw >> 1), j = tid - bi * (rw >> 1);
                    y_pre = a.y[((static_cast<int64_t>(b0 + bi)) * a.C + (r_first >> 1) + j) * HW + p];
                }
                if (!(a.debug & 2)) ok = stage_inputs(a, l, p, py, px, b0, nb, xs);
                if (!ok) *s_flag = 1;    // (a flag in the dynamic LDS: __syncthreads_or would add static LDS on top of the 160 KB)
                __syncthreads();
                if (*s_flag) return;     // poisoned launch: the whole workgroup leaves
                const long long t1 = a.prof ? wall_clock64() : 0;
                long long t2 = t1, t3 = t1;
                // units = (canonical block, image, row), one FMA chain each; `part` holds one round of partials as
                // [block][item]; the finishing threads add an item's partials in block order, then bias and activation
                const int items = (a.debug & 4) ? 0 : nb * rw, Kp = a.kpad[l];
                const int kg = a.kgroup[l], bpg = a.bpg[l], nblk = (K / kg) * bpg;
                const int per_round = items < a.part_floats / nblk ? items : a.part_floats / nblk;
                for (int i0 = 0; i0 < items; i0 += per_round) {
                    const int n_it = (items - i0) < per_round ? (items - i0) : per_round;
                    for (int u = tid; u < n_it * nblk; u += kThreads) {
                        const int blk = u / n_it, it = u - blk * n_it, item = i0 + it;
                        const int bi = item / rw, r = item - bi * rw;
                        const int g = blk / bpg, j = blk - g * bpg;
                        const int k0 = g * kg + j * kKB, len = (kg - j * kKB) < kKB ? (kg - j * kKB) : kKB;
                        part[u] = block_dot(wl + r * Kp + g * (kg + kBlockPad * bpg) + j * (kKB + kBlockPad), xs + bi * K + k0, len, a.vec4);
                    }
                    lds_barrier();
                    if (a.prof && wg == 0 && tid == 0) t2 = wall_clock64();
                    for (int it = tid; it < n_it; it += kThreads) {
                        const int item = i0 + it, bi = item / rw, r = item - bi * rw;
                        float v = 0.f;
                        for (int blk = 0; blk < nblk; ++blk) v += part[blk * n_it + it];
                        v += bias_l[l][r];
                        if (a.act_after[l]) v = v > 0.f ? v : 0.01f * v;   // LeakyReLU(0.01)
                        if (l < last) st_gran(a.act[l] + static_cast<int64_t>(b0 + bi) * a.rows[l] + r_first + r, v, tag);
                        else ps[bi * a.rpw[l] + r] = v;
                    }
                    lds_barrier();   // `part` is reused by the next round; ps is complete for the Gaussian step
                    if (a.prof && wg == 0 && tid == 0) t3 = wall_clock64();
                }
                if (l == last) {
                    // ---- Gaussian step on this workgroup's (mean, scale) pairs: rows 2c, 2c + 1 ("split_interleave")
                    const int pairs = rw >> 1, c_first = r_first >> 1;
                    for (int it = tid; it < nb * pairs; it += kThreads) {
                        const int bi = it / pairs, j = it - bi * pairs;
                        const int64_t b = b0 + bi;
                        const int c = c_first + j;
                        const float mu = ps[bi * a.rpw[l] + 2 * j], sg = ps[bi * a.rpw[l] + 2 * j + 1];
                        const int row = nearest_scale(sg, tab, a.table_len, tab_sorted);
                        if (DECODE) {
                            st_gran(a.idx_step + b * a.C + c, static_cast<uint32_t>(row), tag);
                            st_gran(a.mu + b * a.C + c, mu, tag);
                        } else {
                            const int64_t e = (b * a.C + c) * HW + p;
                            const int64_t o = b * a.C * HW + static_cast<int64_t>(p) * a.C + c;
                            const float q = rintf((it == tid ? y_pre : a.y[e]) - mu);          // torch.round: half to even
                            st_gran(a.yT + (b * HW + p) * a.C + c, q + mu, tag);   // first: every workgroup's next step waits for it
                            a.idx[o] = row;
                            a.sym[o] = static_cast<int32_t>(q);
                            a.ybuf[e] = q + mu;
                        }
                    }
                }
                lds_barrier();   // xs / ps are reused by the next chunk
                if (a.prof && wg == 0 && tid == 0) {
                    a.prof[4 * l] += t1 - t0;
                    a.prof[4 * l + 1] += t2 - t1;
                    a.prof[4 * l + 2] += t3 - t2;
                    a.prof[4 * l + 3] += wall_clock64() - t3;
                }
            }
        }
    }
    if (a.prof && wg == 0 && tid == 0) {
        a.prof[4 * kMaxLayers] = clock64() - loop_c0;
        a.prof[4 * kMaxLayers + 1] = wall_clock64() - loop_t0;
    }
}

// ================================================================================================================
// Pipelined variant for the batches whose whole working set fits the LDS beside the weights (batch 1-2 of the BaSIC
// coder: the reference harness's setting).  Same exchange protocol, same canonical sums, same results; what changes is
// the length of the dependent chain of a coding step:
//   * context layer split in two.  Of the ntaps causal taps only the LAST -- the left neighbour, position p - 1 -- is
//     coded during the previous step; the other taps' blocks (the first (ntaps - 1) * blocks-per-tap of the canonical
//     order) are summed one step AHEAD, right after this workgroup has published its context rows and while the next
//     layer's inputs are still in flight.  On the critical path stay C granules, blocks-per-tap dot products per row
//     and a short sum, instead of ntaps * C granules and the whole layer.  (Needs a latent at least ksize / 2 + 2 columns
//     wide: in a narrower one the tap up and to the right of position p + 1 IS position p or later; the host then takes the
//     generic kernel);
//   * a (block, image, row) unit's addresses are computed once per launch (descriptor table in LDS), not with four
//     integer divisions per unit and step -- a workgroup is four lone waves, each issuing one instruction per ~8-10
//     clocks, so instruction count IS the latency;
//   * a unit loads its whole block (16 + 16 16-byte LDS reads) before the FMA chain starts;
//   * the last layer's finishing thread owns a (mean, scale) pair and runs the Gaussian step itself: no hand-over
//     through LDS, two workgroup barriers per layer instead of three.
// ================================================================================================================

// all `len` channels of a block in flight, then the chain (len == kKB, vec4); the generic block_dot otherwise
__device__ __forceinline__ float block_dot_preloaded(const float *wr, const float *xr, int len, int vec4)
{
    if (len == kKB && vec4) {
        const f4 *x4 = reinterpret_cast<const f4 *>(xr), *w4 = reinterpret_cast<const f4 *>(wr);
        f4 xv[kKB / 4], wv[kKB / 4];
#pragma unroll
        for (int i = 0; i < kKB / 4; ++i) { wv[i] = w4[i]; xv[i] = x4[i]; }
        float p = 0.f;
#pragma unroll
        for (int i = 0; i < kKB / 4; ++i) {
            p = fmaf(wv[i][0], xv[i][0], p); p = fmaf(wv[i][1], xv[i][1], p); p = fmaf(wv[i][2], xv[i][2], p); p = fmaf(wv[i][3], xv[i][3], p);
        }
        return p;
    }
    return block_dot(wr, xr, len, vec4);
}

// v + part[0] + part[stride] + ... (n terms, in that order); eight LDS reads in flight per round -- a finishing thread is
// alone on its SIMD, a read-wait-add loop costs ~130 clocks per term
__device__ __forceinline__ float sum_partials(float v, const float *part, int stride, int n)
{
    int blk = 0;
    for (; blk + 8 <= n; blk += 8) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = part[(blk + i) * stride];
#pragma unroll
        for (int i = 0; i < 8; ++i) v += t[i];
    }
    for (; blk + 2 <= n; blk += 2) {
        const float t0 = part[blk * stride], t1 = part[(blk + 1) * stride];
        v += t0; v += t1;
    }
    if (blk < n) v += part[blk * stride];
    return v;
}

// Two neighbouring granules with one 16-byte load.  Each 8-byte half validates itself by its own tag, so nothing is asked
// of the load beyond what an aligned 8-byte access gives; volatile = cache-bypassing (sc0 sc1) on gfx950, as ld_gran's.
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u64x2 ld_gran2(const uint64_t *p)
{
    typedef const volatile u64x2 __attribute__((address_space(1))) *global_ptr;   // (a generic pointer would make it a flat load)
    return *(global_ptr)(reinterpret_cast<uintptr_t>(p));
}
__device__ __forceinline__ bool tags_are(const u64x2 &g, uint32_t want)
{
    return static_cast<uint32_t>(g[0] >> 32) == want && static_cast<uint32_t>(g[1] >> 32) == want;
}
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 values_of(const u64x2 &g)
{
    return f2{__uint_as_float(static_cast<uint32_t>(g[0])), __uint_as_float(static_cast<uint32_t>(g[1]))};
}

// n granule pairs src[2i], src[2i + 1] (all carrying `want` once written) -> dst[2i], dst[2i + 1]; thread i, i + 256, ...,
// two pairs of a thread in flight.  No divisions, no per-element address arithmetic beyond one add: the workgroup's
// waves are alone on their SIMDs and issue one instruction per ~10 clocks.
__device__ __forceinline__ bool stage_pairs(const ScanArgs &a, const uint64_t *src, int n, uint32_t want, float *dst)
{
    bool ok = true;
    for (int i0 = threadIdx.x; i0 < n; i0 += 2 * kThreads) {
        const int i1 = i0 + kThreads;
        const bool two = i1 < n;
        u64x2 g0 = ld_gran2(src + 2 * i0), g1 = two ? ld_gran2(src + 2 * i1) : u64x2{0ull, 0ull};
        unsigned spins = 0;
        for (;;) {
            const bool r0 = !tags_are(g0, want), r1 = two && !tags_are(g1, want);
            if (!r0 && !r1) break;
            if (++spins > kSpinLimit || (spins % 1024u == 0u && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
            if (r0) g0 = ld_gran2(src + 2 * i0);
            if (r1) g1 = ld_gran2(src + 2 * i1);
        }
        *reinterpret_cast<f2 *>(dst + 2 * i0) = values_of(g0);
        if (two) *reinterpret_cast<f2 *>(dst + 2 * i1) = values_of(g1);
    }
    return ok;
}

constexpr int kWinU = 6;   // early-window granule pairs per thread: (ntaps - 1) * C / 2 <= 6 * 256 per image

// per-thread state of one layer of the pipelined kernel; the layer index is a compile-time constant wherever these are
// used, so they live in registers
struct LayerRegs {
    int rw, n_it, K, prev, nblk, units, dbase, r_first;
    float bias0, bias1;   // of this thread's finishing item (last layer: of its (mean, scale) pair)
    int fin_bi, fin_r;    // this thread's finishing item: image, row (last layer: pair)
};

template <int kFrom, int kTo, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (kFrom < kTo) {
        f(std::integral_constant<int, kFrom>{});
        static_for<kFrom + 1, kTo>(f);
    }
}

template <bool DECODE>
__global__ __launch_bounds__(kThreads) void scanline_pipelined_kernel(const ScanArgs a)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wg = blockIdx.x;
    const int HW = a.H * a.W, B = a.B, C = a.C;
    const int last = a.nlayers - 1;
    if (DECODE && wg >= a.ncompute) { decoder_workgroup(a, lds); return; }

    // ---- this workgroup's weight slices, scale table, unit descriptors
    float *tab = lds + a.tab_off;
    for (int e = tid; e < a.table_len; e += kThreads) tab[e] = a.table[e];
    float *x0 = lds + a.x0_off, *xs = lds + a.xs_off, *part = lds + a.part_off, *early = lds + a.early_off;
    float *part_e = lds + a.early_off + ((B * a.rpw[0] + 3) & ~3);   // the early blocks' partials: not shared with the dense layers
    uint2 *desc = reinterpret_cast<uint2 *>(lds + a.desc_off);
    int *s_flag = reinterpret_cast<int *>(lds + a.flag_off);
    LayerRegs L[kMaxLayers];
    int pairs = 0, my_c = 0, my_b = 0;   // the last layer's (mean, scale) pairs of this workgroup; this thread's channel and image
    {
        int db = 0;
#pragma unroll
        for (int l = 0; l < kMaxLayers; ++l) {
            L[l] = LayerRegs{};
            if (l >= a.nlayers) continue;
            int r = a.rows[l] - wg * a.rpw[l];
            r = r < 0 ? 0 : (r > a.rpw[l] ? a.rpw[l] : r);
            const int Kl = a.kdim[l], Kp = a.kpad[l], kg = a.kgroup[l], bpg = a.bpg[l];
            {
                const float *src = a.w[l] + static_cast<int64_t>(wg) * a.rpw[l] * Kl;
                float *dst = lds + a.woff[l];
                for (int e = tid; e < r * Kl; e += kThreads) { const int rr = e / Kl; dst[rr * Kp + padded_k(e - rr * Kl, kg, bpg)] = src[e]; }
            }
            L[l].rw = r; L[l].n_it = B * r; L[l].K = Kl; L[l].prev = l ? a.rows[l - 1] : 0;
            L[l].nblk = (Kl / kg) * bpg; L[l].units = L[l].n_it * L[l].nblk; L[l].dbase = db; L[l].r_first = wg * a.rpw[l];
            // unit u = blk * n_it + it (it = image * rows + row): LDS float offsets of its weight block and input block, its length
            const int x_base = l == 0 ? a.x0_off : a.xs_off;
            for (int u = tid; u < L[l].units; u += kThreads) {
                const int blk = u / L[l].n_it, it = u - blk * L[l].n_it, bi = it / r, rr = it - bi * r;
                const int g = blk / bpg, j = blk - g * bpg;
                const int len = (kg - j * kKB) < kKB ? (kg - j * kKB) : kKB;
                const int w_off = a.woff[l] + rr * Kp + g * (kg + kBlockPad * bpg) + j * (kKB + kBlockPad);
                const int x_off = x_base + bi * Kl + g * kg + j * kKB;
                desc[db + u] = make_uint2(static_cast<uint32_t>(w_off) | (static_cast<uint32_t>(x_off) << 16), static_cast<uint32_t>(len));
            }
            db += L[l].units;
            // this thread's finishing item (host guarantees B * rows-per-workgroup <= kThreads: one item per thread)
            const int items = l == last ? B * (r >> 1) : B * r, per = l == last ? (r >> 1) : r;
            if (l == last) pairs = r >> 1;
            if (tid < items) {
                L[l].fin_bi = tid / per; L[l].fin_r = tid - L[l].fin_bi * per;
                const float *bias = a.bias[l];
                if (l == last) {
                    L[l].bias0 = bias ? bias[L[l].r_first + 2 * L[l].fin_r] : 0.f;
                    L[l].bias1 = bias ? bias[L[l].r_first + 2 * L[l].fin_r + 1] : 0.f;
                    my_c = (L[l].r_first >> 1) + L[l].fin_r; my_b = L[l].fin_bi;
                } else {
                    L[l].bias0 = bias ? bias[L[l].r_first + L[l].fin_r] : 0.f;
                }
            }
        }
    }
    if (tid == 0) { s_flag[0] = 0; s_flag[1] = 1; }
    __syncthreads();
    for (int e = tid; e + 1 < a.table_len; e += kThreads)
        if (!(tab[e] < tab[e + 1])) s_flag[1] = 0;   // not strictly increasing: nearest_scale scans
    __syncthreads();
    const bool tab_sorted = s_flag[1] != 0;

    const int rw0 = L[0].rw, n_it0 = L[0].n_it, K0 = L[0].K;
    const int early_blk = (a.ntaps - 1) * a.bpg[0];      // blocks of the taps coded at least two steps ago
    const int early_units = early_blk * n_it0;
    const int late_off = (a.ntaps - 1) * C;              // the left neighbour's channels in a context window

    // ---- early window: this thread's granule pairs (tap t, channels 2 c2, 2 c2 + 1) of one image, as offsets from the window's position
    int win_rel[kWinU], win_dst[kWinU], win_tap[kWinU], win_toff[kWinU];
    {
        const int half = C >> 1, E = (a.ntaps - 1) * half;
#pragma unroll
        for (int u = 0; u < kWinU; ++u) {
            const int e = tid + u * kThreads;
            win_rel[u] = 0; win_dst[u] = 0; win_tap[u] = 63; win_toff[u] = 0;
            if (e < E) {
                const int t = e / half, c2 = e - t * half;
                win_tap[u] = t; win_toff[u] = a.tap_off[t];
                win_rel[u] = a.tap_off[t] * C + 2 * c2;
                win_dst[u] = t * C + 2 * c2;
            }
        }
    }
    // lane t of every wave keeps tap t's (dy, dx): the taps inside the image at a position become one ballot
    const int my_dy = lane < a.ntaps ? a.tap_dy[lane] : -(1 << 20), my_dx = lane < a.ntaps ? a.tap_dx[lane] : 0;

    auto dots = [&](float *out, int dbase, int u0, int u1) {
        for (int u = u0 + tid; u < u1; u += kThreads) {
            const uint2 d = desc[dbase + u];
            out[u] = block_dot_preloaded(lds + (d.x & 0xFFFFu), lds + (d.x >> 16), static_cast<int>(d.y), a.vec4);
        }
    };
    // early half of the context layer for position q -- the sums over the first early_blk blocks, in block order from zero --
    // in three pieces that need a workgroup barrier between them: the main loop runs one piece in the shadow of each of the
    // following layers' exchanges (their barriers separate the pieces); with fewer layers than pieces the rest run back to back
    auto early_piece = [&](int piece, int q) -> bool {
        bool ok = true;
        if (piece == 0) {
            const int qy = q / a.W, qx = q - qy * a.W;
            const uint64_t inside = __ballot(qy + my_dy >= 0 && qx + my_dx >= 0 && qx + my_dx < a.W);
            for (int bi = 0; bi < B; ++bi) {
                const uint64_t *base = a.yT + (static_cast<int64_t>(bi) * HW + q) * C;
                float *dst = x0 + bi * K0;
                u64x2 g[kWinU];
                bool live[kWinU];
#pragma unroll
                for (int u = 0; u < kWinU; ++u) {
                    live[u] = ((inside >> win_tap[u]) & 1ull) != 0ull;   // (tap 63: no element)
                    g[u] = u64x2{0ull, 0ull};
                    if (live[u]) g[u] = ld_gran2(base + win_rel[u]);
                }
                unsigned spins = 0;
                for (;;) {   // these positions were coded at least two steps ago: the tags match at once
                    bool all = true;
#pragma unroll
                    for (int u = 0; u < kWinU; ++u)
                        if (live[u] && !tags_are(g[u], static_cast<uint32_t>(q + win_toff[u] + 1))) { g[u] = ld_gran2(base + win_rel[u]); all = false; }
                    if (all) break;
                    if (++spins > kSpinLimit) { __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
                }
#pragma unroll
                for (int u = 0; u < kWinU; ++u)
                    if (win_tap[u] != 63) *reinterpret_cast<f2 *>(dst + win_dst[u]) = values_of(g[u]);   // zeros outside the image
            }
        } else if (piece == 1) {
            dots(part_e, L[0].dbase, 0, early_units);
        } else {
            if (tid < n_it0) early[tid] = sum_partials(0.f, part_e + tid, n_it0, early_blk);
        }
        return ok;
    };
    auto early_pieces_from = [&](int first, int q) {   // pieces first..2, barriers in between
        for (int piece = first; piece < 3; ++piece) {
            if (piece > first) lds_barrier();
            if (!early_piece(piece, q)) *s_flag = 1;
        }
    };

    if (rw0 > 0) {
        early_pieces_from(0, 0);
        lds_barrier();
        if (*s_flag) return;
    }
    // encoder: pointers of this thread's channel of the Gaussian step, advanced by one position per step
    const bool codes = tid < B * pairs;
    const float *y_ptr = a.y ? a.y + (static_cast<int64_t>(my_b) * C + my_c) * HW : nullptr;
    float *ybuf_ptr = a.ybuf + (static_cast<int64_t>(my_b) * C + my_c) * HW;
    int32_t *sym_ptr = a.sym + static_cast<int64_t>(my_b) * C * HW + my_c, *idx_ptr = a.idx + static_cast<int64_t>(my_b) * C * HW + my_c;
    uint64_t *yT_ptr = a.yT + static_cast<int64_t>(my_b) * HW * C + my_c;

    const long long loop_c0 = a.prof ? clock64() : 0, loop_t0 = a.prof ? wall_clock64() : 0;
    int px = 0;
    bool dead = false;
    for (int p = 0; p < HW; ++p) {
        const uint32_t tag = static_cast<uint32_t>(p + 1);
        long long t0 = a.prof ? wall_clock64() : 0;
        float y_pre = 0.f;
        if (!DECODE && codes) y_pre = y_ptr[p];   // requested a whole step before it is needed
        if (rw0 > 0) {
            // ---------- context layer, late half: the left neighbour's blocks, the bias, the activation.  A row's first
            // position has no left neighbour (zeros), but the overwrite argument of the exchange buffers (header comment)
            // needs this workgroup's step p to start after position p - 1 is coded: its granules are waited for all the same.
            bool ok = true;
            for (int bi = 0; bi < B; ++bi) {
                float *dst = x0 + bi * K0 + late_off;
                if (p > 0) ok = stage_pairs(a, a.yT + (static_cast<int64_t>(bi) * HW + p - 1) * C, C >> 1, static_cast<uint32_t>(p), dst) && ok;
                if (px == 0)
                    for (int i = tid; i < (C >> 1); i += kThreads) *reinterpret_cast<f2 *>(dst + 2 * i) = f2{0.f, 0.f};
            }
            if (!ok) *s_flag = 1;
            lds_barrier();
            if (*s_flag) return;
            const long long t1 = a.prof ? wall_clock64() : 0;
            dots(part, L[0].dbase, early_units, L[0].units);
            lds_barrier();
            const long long t2 = a.prof ? wall_clock64() : 0;
            if (tid < n_it0) {
                float v = sum_partials(early[tid], part + early_units + tid, n_it0, L[0].nblk - early_blk);
                v += L[0].bias0;
                if (a.act_after[0]) v = v > 0.f ? v : 0.01f * v;   // LeakyReLU(0.01)
                st_gran(a.act[0] + static_cast<int64_t>(L[0].fin_bi) * a.rows[0] + L[0].r_first + L[0].fin_r, v, tag);
            }
            const long long t3 = a.prof ? wall_clock64() : 0;
            // ---------- early half for the next position: first piece here, in the shadow of this layer's exchange (a poisoned
            // wait shows in the flag at the next layer's barrier; with fewer than three dense layers all pieces run here)
            if (p + 1 < HW) {
                if (last >= 3) { if (!early_piece(0, p + 1)) *s_flag = 1; }
                else early_pieces_from(0, p + 1);
            }
            if (a.prof && wg == 0 && tid == 0) {
                a.prof[0] += t1 - t0; a.prof[1] += t2 - t1; a.prof[2] += t3 - t2; a.prof[3] += wall_clock64() - t3;
            }
        }
        static_for<1, kMaxLayers>([&](auto lc) {
            constexpr int l = decltype(lc)::value;
            if (l > last) return;
            const LayerRegs &R = L[l];
            const bool piece_due = rw0 > 0 && p + 1 < HW && last >= 3 && l <= 2;   // see early_piece
            if (dead) return;
            if (R.rw == 0) {   // nothing to produce here -- and nothing to wait for
                if (piece_due) { lds_barrier(); early_piece(l, p + 1); }
                return;
            }
            const long long t0 = a.prof ? wall_clock64() : 0;
            bool ok = true;
            for (int bi = 0; bi < B; ++bi) {
                float *dst = xs + bi * R.K;
                if (R.K > R.prev && a.priorT) {   // cat(ctx, prior): the prior is an input of the launch
                    const f4 *src = reinterpret_cast<const f4 *>(a.priorT + (static_cast<int64_t>(bi) * HW + p) * a.P);
                    for (int i = tid; i < ((R.K - R.prev) >> 2); i += kThreads) *reinterpret_cast<f4 *>(dst + R.prev + 4 * i) = src[i];
                }
                ok = stage_pairs(a, a.act[l - 1] + static_cast<int64_t>(bi) * R.prev, R.prev >> 1, tag, dst) && ok;
            }
            if (!ok) *s_flag = 1;
            lds_barrier();
            // a poisoned launch (some wait gave up; every later wait then fails fast) is left once per step, after a barrier:
            // at the context layer's, or here by the workgroups without context rows
            if (l == 1 && rw0 == 0 && *s_flag) { dead = true; return; }
            const long long t1 = a.prof ? wall_clock64() : 0;
            dots(part, R.dbase, 0, R.units);
            lds_barrier();
            const long long t2 = a.prof ? wall_clock64() : 0;
            if (l < last) {
                if (tid < R.n_it) {
                    float v = sum_partials(0.f, part + tid, R.n_it, R.nblk);
                    v += R.bias0;
                    if (a.act_after[l]) v = v > 0.f ? v : 0.01f * v;
                    st_gran(a.act[l] + static_cast<int64_t>(R.fin_bi) * a.rows[l] + R.r_first + R.fin_r, v, tag);
                }
                // the context layer's early half for the next position, second / third piece (see early_piece)
                if (piece_due) early_piece(l, p + 1);
            } else if (codes) {
                // ---- (mean, scale) = rows 2j, 2j + 1 ("split_interleave") summed by ONE thread, which then codes the channel
                const int i0 = R.fin_bi * R.rw + 2 * R.fin_r;
                float mu = sum_partials(0.f, part + i0, R.n_it, R.nblk), sg = sum_partials(0.f, part + i0 + 1, R.n_it, R.nblk);
                mu += R.bias0; sg += R.bias1;
                if (a.act_after[l]) { mu = mu > 0.f ? mu : 0.01f * mu; sg = sg > 0.f ? sg : 0.01f * sg; }
                const int row = nearest_scale(sg, tab, a.table_len, tab_sorted);
                if (DECODE) {
                    st_gran(a.idx_step + static_cast<int64_t>(my_b) * C + my_c, static_cast<uint32_t>(row), tag);
                    st_gran(a.mu + static_cast<int64_t>(my_b) * C + my_c, mu, tag);
                } else {
                    const float q = rintf(y_pre - mu);          // torch.round: half to even
                    st_gran(yT_ptr + static_cast<int64_t>(p) * C, q + mu, tag);   // first: every workgroup's next step waits for it
                    idx_ptr[static_cast<int64_t>(p) * C] = row;
                    sym_ptr[static_cast<int64_t>(p) * C] = static_cast<int32_t>(q);
                    ybuf_ptr[p] = q + mu;
                }
            }
            if (a.prof && wg == 0 && tid == 0) {
                a.prof[4 * l] += t1 - t0; a.prof[4 * l + 1] += t2 - t1; a.prof[4 * l + 2] += wall_clock64() - t2;
            }
        });
        if (dead) return;
        if (++px == a.W) px = 0;
    }
    if (a.prof && wg == 0 && tid == 0) {
        a.prof[4 * kMaxLayers] = clock64() - loop_c0;
        a.prof[4 * kMaxLayers + 1] = wall_clock64() - loop_t0;
    }
}

// ================================================================================================================
// Batched variant (round 4): batches of 3 .. 64 images on the matrix core.
//
// The two kernels above give a thread one (block, image, row) FMA chain: right for one or two images, 0.6 ms per coding step
// at 64.  Here the BATCH is the N dimension of v_mfma_f32_32x32x2_f32 tiles and the weights never move:
//   * a column tile = 32 images; the launch runs one independent set of `nw` compute workgroups per column tile (64 images =
//     two sets), one workgroup per compute unit;
//   * a workgroup owns ONE 32-row tile of a layer with its whole K axis, as MFMA A fragments IN REGISTERS: a wave keeps
//     eight 32-row x 64-channel blocks (256 of the 512 registers a lone wave per SIMD may use; a context tile's 36 blocks are
//     4 x 8 resident + four "late" blocks whose fragments are re-fetched each step while the wave waits anyway) -- 1.9 M
//     weights = 928 blocks fit the register files of 32 compute units, and the LDS stays free for partial tiles.  Roles: the first `nd`
//     workgroups hold row tile j of EVERY dense layer that has one (layers run one after the other, so their tiles share
//     the unit), the others one row tile of the context layer (ntaps * C / 64 blocks);
//   * a canonical block's partial tile (32 rows x 32 images) is one 32-step MFMA chain from zero -- exactly the masked
//     convolution's block (mconv.hip), so the integers agree with every other path bit for bit -- written to LDS; after a
//     workgroup barrier the 256 threads add the tile's blocks in block order (16-byte LDS reads, four outputs per thread),
//     bias, activation.  Splitting a row tile's K over workgroups would need a second exchange per layer; keeping it in one
//     compute unit costs up to three MFMA rounds per layer instead;
//   * B fragments are loaded straight from the exchange arrays into registers (lane = (k parity, image): two 256-byte runs per
//     load), validated by their tags, and fed to the chain -- no LDS staging.  All exchanged arrays are batch-minor;
//   * context layer: the taps coded at least two steps ago (all but the left neighbour) are multiplied one step AHEAD, in the
//     shadow of the dense layers; on the critical path stay C / 64 blocks and their sum.  Dense layer 1: the blocks fed by the
//     prior (an input of the launch) run before the context layer's output is waited for.
// Exchange protocol, tags, overwrite argument and bounded spins: as in the header comment (a row tile's workgroup consumes ALL
// inputs of its layer, so "produced" still implies "every input consumed").
// ================================================================================================================
constexpr int kBSlots = 8;        // weight blocks a wave keeps in registers for the whole launch (256 registers: the accumulator half of the file)
constexpr int kBLate = 4;         // context role: the LAST four blocks of the K axis (the left neighbour's among them) run in the late half, one per
                                  // wave, with weights re-fetched every step from the workgroup's fragment copy (a ninth resident block spills)
constexpr int kBDenseSlots = 3;   // blocks per dense layer and wave (a layer's K axis has at most 12; the last layer's at most 8: slots 6, 7)
constexpr int kBTile = 1024;      // floats of a 32 x 32 partial tile

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- batch-minor exchange arrays.  A [channels][nbt] slab of granules is stored in 16-byte pieces that hold the granules of
//      channels 4 m + h and 4 m + 2 + h of one column: the B operands of two consecutive MFMA steps of lane (h, column % 32), so
//      a lane fetches a block's 32 operands with sixteen 16-byte loads (each 8-byte half still validates itself by its tag).
//      Granule index of (channel c, column col):
__host__ __device__ __forceinline__ int64_t bm_gran(int c, int col, int nbt)
{
    return (static_cast<int64_t>((c >> 2) * 2 + (c & 1)) * nbt + col) * 2 + ((c >> 1) & 1);
}
//      The prior (plain floats, an input of the launch) likewise in 16-byte pieces of channels 8 m + 2 j + h, j = 0 .. 3: four steps
//      per load.  Float index of (channel c, column col) inside a position's [P][nbt] slab:
__host__ __device__ __forceinline__ int64_t bm_prior(int c, int col, int nbt)
{
    return (static_cast<int64_t>((c >> 3) * 2 + (c & 1)) * nbt + col) * 4 + ((c >> 1) & 3);
}

// A lone wave per SIMD issues one instruction per ~8-10 clocks, so a block's cost beside its 32 MFMAs (0.85 us) is its
// instruction count.  Operands are fetched with BUFFER loads: descriptor = the array, soffset (SGPR) = the block's and the piece's
// wave-uniform byte offset, voffset = the lane's constant 32-bit offset -- no vector ALU work per load, sixteen `buffer_load_dwordx4
// ... offen sc1` back to back (a `volatile` 16-byte global load would do too, but hipcc follows every volatile access with
// s_waitcnt vmcnt(0): sixteen serial round trips per block, measured 4.4 us) -- and validated with one min-tree over the 32 tags
// instead of 32 compare-and-branch pairs.  The lane offset goes through an empty asm: it keeps hipcc from treating the loads of
// different steps (same address, no store in sight) as one, and from hoisting per-load addresses out of the coding loop.
typedef __amdgpu_buffer_rsrc_t brsrc;
__device__ __forceinline__ brsrc b_rsrc(const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7FFFFFFF, 0x00020000); }
constexpr int kSc1 = 16;   // cache-policy bit of the raw buffer builtins: sc1 (agent scope: served at the L2 / memory side, never by this CU's L1)

__device__ __forceinline__ u64x2 b_ld2(brsrc r, uint32_t voff, int soff)
{
    return __builtin_bit_cast(u64x2, __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>(voff), soff, kSc1));
}

__device__ __forceinline__ void b_issue(brsrc r, int sbase, int stride, uint32_t voff, u64x2 (&g)[16])
{
    uint32_t vo = voff;
    int sb = sbase;   // (opaque too: the sixteen scalar offsets of every block would otherwise be hoisted out of the coding loop and spilled)
    asm volatile("" : "+v"(vo), "+s"(sb)::"memory");
#pragma unroll
    for (int m = 0; m < 16; ++m) g[m] = b_ld2(r, vo, sb + m * stride);
}

// the smallest of the 32 tags: a stale granule carries an older (smaller) tag or 0, never a newer one (overwrite argument)
__device__ __forceinline__ uint32_t b_min_tag(const u64x2 (&g)[16])
{
    uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
    for (int m = 0; m < 16; ++m) mn = min(min(mn, static_cast<uint32_t>(g[m][0] >> 32)), static_cast<uint32_t>(g[m][1] >> 32));
    return mn;
}

// waits until all 32 granules carry `want` (re-issuing the block's loads while one is missing); false = poisoned launch
__device__ __forceinline__ bool b_wait(const ScanArgs &a, brsrc r, int sbase, int stride, uint32_t voff, uint32_t want, u64x2 (&g)[16])
{
    unsigned spins = 0;
    while (__ballot(b_min_tag(g) != want) != 0ull) {
        if (++spins > kSpinLimit || (spins % 256u == 0u && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        b_issue(r, sbase, stride, voff, g);
    }
    return true;
}

// one piece polled alone before a block's loads are issued: a wave that waits LONG (the context role, idle through the dense
// layers) keeps one load in flight instead of sixteen
__device__ __forceinline__ bool b_sentinel(const ScanArgs &a, brsrc r, int soff, uint32_t voff, uint32_t want)
{
    unsigned spins = 0;
    for (;;) {
        uint32_t vo = voff;
        asm volatile("" : "+v"(vo)::"memory");
        const u64x2 g = b_ld2(r, vo, soff);
        if (__ballot(min(static_cast<uint32_t>(g[0] >> 32), static_cast<uint32_t>(g[1] >> 32)) != want) == 0ull) return true;
        if (++spins > kSpinLimit || (spins % 256u == 0u && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
}

// ---- Resident weights.  The eight blocks a wave keeps for the whole launch live in the ACCUMULATOR half of the register file,
//      a[32 k + i] = A fragment i of slot k (an MFMA takes its A operand from there directly).  As a C++ array hipcc spills part of
//      it whatever the slack (its allocator treats the unified file as two halves and parks / spills around the long live ranges),
//      and a spilled weight is re-read inside the chain behind an s_waitcnt vmcnt(0).  So the 256 registers are named literally
//      (cdna_hip_programming.md 5.7 item 4): written once (b_put_weights), read only by the MFMAs below, reserved by the clobber
//      list of b_reserve_acc; EVERY MFMA of the kernel is an asm statement with its accumulator in VGPRs, so the compiler itself has
//      no use for the accumulator file -- `make audit` checks the generated code: no spill, no compiler-issued v_accvgpr_*.
//      Nothing inside an asm string is padded by the compiler: every MFMA is preceded by the two wait states a just-written B
//      operand needs, and a chain ends with the 19 states a 16-pass MFMA's result needs before anything but the next MFMA of the
//      chain touches it.
#define BASIC_A10(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
__device__ __forceinline__ void b_reserve_acc()
{
    asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", BASIC_A10(1), BASIC_A10(2), BASIC_A10(3), BASIC_A10(4),
                 BASIC_A10(5), BASIC_A10(6), BASIC_A10(7), BASIC_A10(8), BASIC_A10(9), BASIC_A10(10), BASIC_A10(11), BASIC_A10(12), BASIC_A10(13),
                 BASIC_A10(14), BASIC_A10(15), BASIC_A10(16), BASIC_A10(17), BASIC_A10(18), BASIC_A10(19), BASIC_A10(20), BASIC_A10(21),
                 BASIC_A10(22), BASIC_A10(23), BASIC_A10(24), "a250", "a251", "a252", "a253", "a254", "a255");
}
#undef BASIC_A10

template <int K, int I> __device__ __forceinline__ void b_put_weights_step(const float *src, bool valid)
{
    if constexpr (I < 32) {
        const float x = valid ? src[2 * I] : 0.f;
        asm volatile("v_accvgpr_write_b32 a[%c1], %0" ::"v"(x), "n"(K * 32 + I));
        b_put_weights_step<K, I + 1>(src, valid);
    }
}
template <int K> __device__ __forceinline__ void b_put_weights(const float *src, bool valid) { b_put_weights_step<K, 0>(src, valid); }

__device__ __forceinline__ f32x16 b_zero16()
{
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// acc += A(slot K, fragment I) x b   /   acc += wv x b   (one v_mfma_f32_32x32x2_f32 each)
template <int K, int I> __device__ __forceinline__ void b_mfma_res(f32x16 &acc, float b)
{
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, a[%c2], %1, %0" : "+v"(acc) : "v"(b), "n"(K * 32 + I));
}
__device__ __forceinline__ void b_mfma_vgpr(f32x16 &acc, float wv, float b)
{
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %2, %1, %0" : "+v"(acc) : "v"(b), "v"(wv));
}
__device__ __forceinline__ void b_chain_end(f32x16 &acc) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc)); }

// the canonical block: one MFMA chain from zero over the block's 32 channel pairs, in ascending order, A fragments = resident slot
// K, B = the block's granules.  With kRefill every piece is re-loaded from `nbase` (the block after next of this wave) as soon as
// its two MFMAs have read it: two blocks' loads are in flight or landed at any time with 128 registers of buffers
template <int K, bool kRefill, int M> __device__ __forceinline__ void b_chain_gran_step(f32x16 &acc, u64x2 (&g)[16], brsrc r, int nb, int stride, uint32_t vo)
{
    if constexpr (M < 16) {
        b_mfma_res<K, 2 * M>(acc, __uint_as_float(static_cast<uint32_t>(g[M][0])));
        b_mfma_res<K, 2 * M + 1>(acc, __uint_as_float(static_cast<uint32_t>(g[M][1])));
        if constexpr (kRefill) {
            g[M] = b_ld2(r, vo, nb + M * stride);
            __builtin_amdgcn_sched_barrier(0);   // (keeps the loads between the MFMAs: issued under the matrix pipe's 64-cycle steps)
        }
        b_chain_gran_step<K, kRefill, M + 1>(acc, g, r, nb, stride, vo);
    }
}
template <int K, bool kRefill> __device__ __forceinline__ f32x16 b_chain_gran(u64x2 (&g)[16], brsrc r, int nbase, int stride, uint32_t voff)
{
    f32x16 acc = b_zero16();
    uint32_t vo = voff;
    int nb = nbase;
    asm volatile("" : "+v"(vo), "+s"(nb)::"memory");
    b_chain_gran_step<K, kRefill, 0>(acc, g, r, nb, stride, vo);
    b_chain_end(acc);
    return acc;
}

// the same chain with A fragments and B operands both in VGPRs (a streamed block; a block fed by the prior)
template <class AF, class BF> __device__ __forceinline__ f32x16 b_chain_vv(AF &&af, BF &&bf)
{
    f32x16 acc = b_zero16();
#pragma unroll
    for (int i = 0; i < 32; ++i) b_mfma_vgpr(acc, af(i), bf(i));
    b_chain_end(acc);
    return acc;
}
template <int K, int I, class BF> __device__ __forceinline__ void b_chain_res_step(f32x16 &acc, BF &bf)
{
    if constexpr (I < 32) {
        b_mfma_res<K, I>(acc, bf(I));
        b_chain_res_step<K, I + 1>(acc, bf);
    }
}
template <int K, class BF> __device__ __forceinline__ f32x16 b_chain_res(BF &&bf)   // resident slot K, B operands from a callable
{
    f32x16 acc = b_zero16();
    b_chain_res_step<K, 0>(acc, bf);
    b_chain_end(acc);
    return acc;
}

// accumulator layout of the 32 x 32 tile: lane holds image lane % 32, rows 8 q + 4 (lane / 32) + j for register 4 q + j; the
// tile goes to LDS as [q][lane] 16-byte pieces, and thread (q, lane) of the workgroup later owns exactly those four outputs
__device__ __forceinline__ void b_store_tile(float *tile, int lane, const f32x16 &acc)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f4 *>(tile + (q * 64 + lane) * 4) = f4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
}

// v + tile 0 + tile 1 + ... (n tiles, in that order), this thread's four outputs
__device__ __forceinline__ f4 b_sum_tiles(f4 v, const float *tiles, int n, int tid)
{
    const float *p = tiles + tid * 4;
    int b = 0;
    for (; b + 4 <= n; b += 4) {
        const f4 t0 = *reinterpret_cast<const f4 *>(p + b * kBTile), t1 = *reinterpret_cast<const f4 *>(p + (b + 1) * kBTile);
        const f4 t2 = *reinterpret_cast<const f4 *>(p + (b + 2) * kBTile), t3 = *reinterpret_cast<const f4 *>(p + (b + 3) * kBTile);
        v += t0; v += t1; v += t2; v += t3;
    }
    for (; b < n; ++b) v += *reinterpret_cast<const f4 *>(p + b * kBTile);
    return v;
}

__device__ __forceinline__ f4 b_leaky(f4 v)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : 0.01f * v[i];   // LeakyReLU(0.01)
    return v;
}

template <bool DECODE>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 1))) void scanline_batched_kernel(const ScanArgs a)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    if (DECODE && wg >= a.ncompute) { decoder_workgroup(a, lds); return; }
    const int HW = a.H * a.W, C = a.C, NBT = a.nbt, T = NBT >> 5;
    const int t = wg % T, j = wg / T;                  // column tile; role index inside the tile's set of workgroups
    const int h = lane >> 5, n = lane & 31, col = t * 32 + n;
    const bool img = col < a.B;                        // columns beyond the batch: nothing published (they LOAD the last image's
    const int colc = img ? col : a.B - 1;              // operands: no predicates, no zero fill in the load paths)
    const bool dense = j < a.nd;
    const int r0 = dense ? j : j - a.nd;               // this workgroup's row tile (dense role: of every layer that has one)
    const int nb0 = a.ntaps * a.bpt;                   // context blocks; the last kBLate run in the late half (the left neighbour's bpt among them),
    const int early = nb0 - kBLate;                    // the others -- taps coded at least two steps ago -- one step ahead
    const int frow = 8 * wave + 4 * h;                 // first of this thread's four finishing rows inside the tile
    const uint32_t voff = static_cast<uint32_t>(h * NBT + colc) * 16u;   // the lane's 16-byte piece inside a (channel quad, column tile) slab
    const int pstride = 2 * NBT * 16;                  // bytes between consecutive pieces of a lane
    // granule index of this thread's first finishing row (channel r0 * 32 + frow) relative to its tile's first piece
    const uint32_t soff = static_cast<uint32_t>(((frow >> 2) * 2 * NBT + col) * 2);

    // LDS: [scale table][flags][biases of this workgroup's row tiles: 3 x 32][partial tiles]
    float *tab = lds;
    int *s_flag = reinterpret_cast<int *>(lds + ((a.table_len + 3) & ~3));
    float *bias_s = lds + ((a.table_len + 3) & ~3) + 4;
    float *part = bias_s + 96;
    for (int e = tid; e < a.table_len; e += kThreads) tab[e] = a.table[e];
    if (tid < 96) {
        const int li = tid >> 5, l = dense ? 1 + li : 0;
        const float *bias = (l < a.nlayers && (dense || li == 0) && r0 < a.rt[l]) ? a.bias[l] : nullptr;
        bias_s[tid] = bias ? bias[r0 * 32 + (tid & 31)] : 0.f;
    }
    if (tid == 0) s_flag[1] = 1;
    __syncthreads();
    for (int e = tid; e + 1 < a.table_len; e += kThreads)
        if (!(tab[e] < tab[e + 1])) s_flag[1] = 0;   // not strictly increasing: nearest_scale scans
    __syncthreads();
    const bool tab_sorted = s_flag[1] != 0;

    // ---- this wave's resident weight blocks: A fragments (lane = row lane % 32 of the tile, channel parity lane / 32) in the
    // accumulator file, see b_put_weights
    b_reserve_acc();
    static_for<0, kBSlots>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int l = 1 + k / kBDenseSlots;                      // dense role: the slot's layer and block
        const int b = wave + 4 * (k % kBDenseSlots), e = wave + 4 * k;   // context role: its early block
        const bool v = dense ? (l < a.nlayers && r0 < a.rt[l] && b < a.nblk[l]) : e < early;   // wave-uniform
        const int li = dense ? (l < a.nlayers ? l : 0) : 0;
        b_put_weights<k>(a.w[li] + static_cast<int64_t>(r0 * 32 + n) * a.kdim[li] + 64 * (dense ? b : e) + h, v);
    });
    asm volatile("s_nop 4" ::: "memory");   // (accumulator writes -> the first MFMA that reads them)
    const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
    u64x2 gA[16], gB[16];

    // publishes this thread's four finishing rows of its tile into a [rows][nbt] exchange slab (tile = slab + the tile's first piece)
    auto publish4 = [&](uint64_t *tile, const f4 &v, uint32_t tag) {
        if (!img) return;
        uint32_t so = soff;
        asm volatile("" : "+v"(so));   // (or the four addresses are hoisted out of the coding loop and spilled)
#pragma unroll
        for (int i = 0; i < 4; ++i) st_gran(tile + (so + static_cast<uint32_t>((i & 1) * 2 * NBT + (i >> 1))), v[i], tag);
    };

    if (!dense) {
        // ================= context role: row tile r0 of the masked context convolution =================
        const char *yT8 = reinterpret_cast<const char *>(a.yT);
        const int blk_bytes = 16 * pstride;   // a 64-channel block of a slab
        uint64_t *out_tile = a.act[0] + static_cast<int64_t>(r0 * 8) * 2 * NBT * 2;
        const int64_t pos_bytes = static_cast<int64_t>(C) * NBT * 8;   // a position's slab of the coded latent
        f4 esum = zero4;   // position 0 has no causal neighbour: its early sum is zero
        int px = 0, py = 0;
        // this wave's late block as A fragments, lane-major (8 x 16 bytes per lane): saved once, re-fetched every step
        const char *wfrag = reinterpret_cast<const char *>(a.wlate) + static_cast<int64_t>(wg) * (kThreads * 128);
        {
            const int e = early + wave;
            const float *src = a.w[0] + static_cast<int64_t>(r0 * 32 + n) * a.kdim[0] + 64 * e + h;
            float *dst = reinterpret_cast<float *>(const_cast<char *>(wfrag)) + tid * 32;
#pragma unroll
            for (int i = 0; i < 32; ++i) dst[i] = src[2 * i];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const bool prof1 = a.prof && wg == a.nd * T && tid == 64, prof0 = a.prof && wg == a.nd * T && tid == 0;   // BASIC_SCAN_PROFILE: first context workgroup, waves 1 / 0
        const long long loop_t0 = prof0 ? wall_clock64() : 0;
        for (int p = 0; p < HW; ++p) {
            const uint32_t tag = static_cast<uint32_t>(p + 1);
            const long long tp0 = a.prof ? wall_clock64() : 0;
            long long tp1 = tp0;
            // ---------- late half: the last four blocks of the K axis, one per wave.  The left neighbour's (the last tap's) need
            // position p - 1, coded a moment ago: sentinel, then tags.  A row's first position has no left neighbour (zeros), but
            // the overwrite argument of the exchange arrays needs this step to start after position p - 1 is coded: its granules
            // are waited for all the same.  The block's weights come from this wave's fragment copy, requested before the wait
            {
                const int e = early + wave, tp = e / a.bpt, cb = e - tp * a.bpt;
                const bool left = tp == a.ntaps - 1;
                const int ny = py + a.tap_dy[tp], nx = px + a.tap_dx[tp];
                const bool inside = ny >= 0 && nx >= 0 && nx < a.W;
                f4 wl4[8];
                {
                    uint32_t fo = static_cast<uint32_t>(tid) * 128u;
                    asm volatile("" : "+v"(fo)::"memory");
#pragma unroll
                    for (int m = 0; m < 8; ++m) wl4[m] = *reinterpret_cast<const f4 *>(wfrag + (fo + static_cast<uint32_t>(m * 16)));
                }
                f32x16 acc = b_zero16();
                if (inside || (left && p > 0)) {
                    const int pos = left ? p - 1 : p + a.tap_off[tp];
                    const brsrc yr = b_rsrc(yT8 + pos * pos_bytes);
                    const int sb = cb * blk_bytes;
                    if (left && !b_sentinel(a, yr, sb + 15 * pstride, voff, static_cast<uint32_t>(pos + 1))) return;
                    if (a.prof) tp1 = wall_clock64();
                    b_issue(yr, sb, pstride, voff, gA);
                    if (!b_wait(a, yr, sb, pstride, voff, static_cast<uint32_t>(pos + 1), gA)) return;
                    if (inside)
                        acc = b_chain_vv([&](int i) { return wl4[i >> 2][i & 3]; }, [&](int i) { return __uint_as_float(static_cast<uint32_t>(gA[i >> 1][i & 1])); });
                }
                b_store_tile(part + e * kBTile, lane, acc);
            }
            const long long tp2 = a.prof ? wall_clock64() : 0;
            lds_barrier();
            {
                f4 v = b_sum_tiles(esum, part + early * kBTile, kBLate, tid);
                v += *reinterpret_cast<const f4 *>(bias_s + frow);
                if (a.act_after[0]) v = b_leaky(v);
                publish4(out_tile, v, tag);
            }
            const long long tp3 = a.prof ? wall_clock64() : 0;
            if (prof1) { a.prof[0] += tp1 - tp0; a.prof[1] += tp2 - tp1; a.prof[2] += tp3 - tp2; }
            if (++px == a.W) { px = 0; ++py; }
            if (p + 1 == HW) break;
            // ---------- early half for the next position q = p + 1, in the shadow of the dense layers: partial tiles of the blocks
            // of the taps coded at least two steps ago, then their sum in block order.  Every granule read here was validated by
            // this workgroup's late waves in an earlier step (the newest, position p - 1, in this step's late half, before the
            // barrier above), so the tags are not looked at again.  Two buffers: while block k's chain runs, its buffer is
            // re-filled piece by piece with block k + 2 (the loads sit between the MFMAs, where a lone wave's issue slots are
            // free), block k + 1 landed during the chain before
            const int q = p + 1, qy = py, qx = px;
            auto early_valid = [&](int kk) { return wave + 4 * kk < early; };
            // block e of this position's early window: false when its tap lies outside the image (zeros); r / sb = where its operands are
            auto early_src = [&](int e, brsrc &r, int &sb) -> bool {
                const int tp = e / a.bpt, cb = e - tp * a.bpt;
                const int ny = qy + a.tap_dy[tp], nx = qx + a.tap_dx[tp];
                const bool inside = ny >= 0 && nx >= 0 && nx < a.W;
                r = b_rsrc(yT8 + (inside ? q + a.tap_off[tp] : q - 1) * pos_bytes);   // (outside: any coded position -- loaded into a buffer nobody reads)
                sb = cb * blk_bytes;
                return inside;
            };
            bool have_a = false, have_b = false;   // whether buffer A / B holds (or is receiving) a block that is used
            {
                brsrc r; int sb;
                if (early_valid(0) && (have_a = early_src(wave, r, sb))) b_issue(r, sb, pstride, voff, gA);
                if (early_valid(1) && (have_b = early_src(wave + 4, r, sb))) b_issue(r, sb, pstride, voff, gB);
            }
            static_for<0, kBSlots>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if (!early_valid(k)) return;
                u64x2(&cur)[16] = *((k & 1) ? &gB : &gA);
                bool &have_cur = (k & 1) ? have_b : have_a;
                brsrc r2 = b_rsrc(yT8);
                int sb2 = 0;
                const bool next = k + 2 < kBSlots && early_valid(k + 2);
                const bool have_next = next && early_src(wave + 4 * (k + 2), r2, sb2);
                f32x16 acc = b_zero16();
                if (have_cur) {
                    if (!next) early_src(wave + 4 * k, r2, sb2);   // nothing follows: the refill re-reads this block (no second copy of the chain)
                    acc = b_chain_gran<k, true>(cur, r2, sb2, pstride, voff);
                } else if (have_next) {
                    b_issue(r2, sb2, pstride, voff, cur);   // (a tap outside the image enters as zeros: no chain to hide the loads behind)
                }
                have_cur = have_next;
                b_store_tile(part + (wave + 4 * k) * kBTile, lane, acc);
            });
            lds_barrier();
            esum = b_sum_tiles(zero4, part, early, tid);
            if (prof0) a.prof[3] += wall_clock64() - tp3;
        }
        if (prof0) a.prof[4 * kMaxLayers + 2] = wall_clock64() - loop_t0;
        return;
    }

    // ================= dense role: row tile r0 of every dense layer that has one =================
    const int last = a.nlayers - 1;
    const bool codes = r0 < a.rt[last] && img;     // this thread finishes two (mean, scale) pairs of image `col`
    const int c0 = (r0 * 32 + frow) >> 1;          // its channels c0, c0 + 1  (rows 2c = mean, 2c + 1 = scale: "split_interleave")
    // element offsets of (image col, channel c0) in y / ybuf [B][C][HW], sym / idx [B][HW * C], the coded latent's slab, mu / idx_step [nbt][C]
    const uint32_t yoff = static_cast<uint32_t>((col * C + c0) * HW), ooff = static_cast<uint32_t>(col * C * HW + c0);
    const uint32_t toff = static_cast<uint32_t>(bm_gran(c0, col, NBT)), moff = static_cast<uint32_t>(col * C + c0);
    const bool prof = a.prof && wg == 0 && tid == 0;   // BASIC_SCAN_PROFILE: first dense workgroup, wave 0
    const long long loop_c0 = prof ? clock64() : 0, loop_t0 = prof ? wall_clock64() : 0;
    for (int p = 0; p < HW; ++p) {
        const uint32_t tag = static_cast<uint32_t>(p + 1);
        // (per-thread element offsets stay 32-bit and opaque: as 64-bit addresses hoisted out of the loop they would be spilled)
        float y_pre0 = 0.f, y_pre1 = 0.f;
        if (!DECODE && codes) {   // requested a whole step before they are needed
            uint32_t yo = yoff;
            asm volatile("" : "+v"(yo));
            y_pre0 = (a.y + p)[yo];
            y_pre1 = (a.y + p)[yo + static_cast<uint32_t>(HW)];
        }
        bool alive = true;
        static_for<1, 4>([&](auto lc) {
            constexpr int l = decltype(lc)::value;
            if (l > last || r0 >= a.rt[l] || !alive) return;
            float *tiles = part + a.tile_off[l];
            const int fed = l == 1 ? a.ctx_blocks : a.nblk[l];   // blocks fed by the previous layer; the rest (layer 1): the prior
            const long long tq0 = a.prof ? wall_clock64() : 0;
            // (1) the prior's blocks: inputs of the launch, nothing to wait for (16-byte pieces of four steps' operands)
            if (l == 1) {
                static_for<0, kBDenseSlots>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const int b = wave + 4 * k;
                    if (b < fed || b >= a.nblk[1]) return;
                    const brsrc pr = b_rsrc(reinterpret_cast<const char *>(a.priorT) + static_cast<int64_t>(p) * a.P * NBT * 4);
                    int sb = 8 * (b - fed) * pstride;
                    f4 pv[8];
                    uint32_t vo = voff;
                    asm volatile("" : "+v"(vo), "+s"(sb)::"memory");
#pragma unroll
                    for (int m = 0; m < 8; ++m) pv[m] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(pr, static_cast<int>(vo), sb + m * pstride, 0));
                    const f32x16 acc = b_chain_res<k>([&](int i) { return pv[i >> 2][i & 3]; });
                    b_store_tile(tiles + b * kBTile, lane, acc);
                });
            }
            // (2) the blocks fed by layer l - 1: the first two blocks' loads go out together, the third's piece by piece behind the first
            // chain's MFMAs (no sentinel: the producers are rarely behind, and a second round trip per layer costs more than the polling)
            const long long tq1 = a.prof ? wall_clock64() : 0;
            const brsrc xr = b_rsrc(a.act[l - 1]);
            const int nb = wave < fed ? (fed - wave + 3) >> 2 : 0;   // this wave's blocks wave, wave + 4, wave + 8 below `fed`
            const int b0 = 16 * wave * pstride, b1 = b0 + 64 * pstride, b2 = b1 + 64 * pstride;   // their byte offsets in the slab
            if (nb >= 1) b_issue(xr, b0, pstride, voff, gA);
            if (nb >= 2) b_issue(xr, b1, pstride, voff, gB);
            if (nb >= 1) {
                alive = b_wait(a, xr, b0, pstride, voff, tag, gA) && alive;
                const f32x16 acc = (nb >= 3 && (l - 1) * kBDenseSlots + 2 < kBSlots) ? b_chain_gran<(l - 1) * kBDenseSlots, true>(gA, xr, b2, pstride, voff)
                                                                                    : b_chain_gran<(l - 1) * kBDenseSlots, false>(gA, xr, 0, 0, 0u);
                b_store_tile(tiles + wave * kBTile, lane, acc);
            }
            if (nb >= 2) {
                alive = b_wait(a, xr, b1, pstride, voff, tag, gB) && alive;
                const f32x16 acc = b_chain_gran<(l - 1) * kBDenseSlots + 1, false>(gB, xr, 0, 0, 0u);
                b_store_tile(tiles + (wave + 4) * kBTile, lane, acc);
            }
            if constexpr ((l - 1) * kBDenseSlots + 2 < kBSlots) {   // (the last layer has two register slots per wave)
                if (nb >= 3) {
                    alive = b_wait(a, xr, b2, pstride, voff, tag, gA) && alive;
                    const f32x16 acc = b_chain_gran<(l - 1) * kBDenseSlots + 2, false>(gA, xr, 0, 0, 0u);
                    b_store_tile(tiles + (wave + 8) * kBTile, lane, acc);
                }
            }
            if (!alive) return;
            const long long tq2 = a.prof ? wall_clock64() : 0;
            lds_barrier();
            f4 v = b_sum_tiles(zero4, tiles, a.nblk[l], tid);
            v += *reinterpret_cast<const f4 *>(bias_s + (l - 1) * 32 + frow);
            if (a.act_after[l]) v = b_leaky(v);
            if (l < last) {
                publish4(a.act[l] + static_cast<int64_t>(r0 * 8) * 2 * NBT * 2, v, tag);
            } else if (img) {
                // ---- Gaussian step: this thread's two (mean, scale) pairs.  What the next step waits for goes out first: the coded
                // latent (encoder) -- the table search and the plain outputs follow in its shadow
                uint32_t yo = yoff, oo = ooff, to = toff, mo = moff;
                asm volatile("" : "+v"(yo), "+v"(oo), "+v"(to), "+v"(mo));
                if (DECODE) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int row = nearest_scale(v[2 * i + 1], tab, a.table_len, tab_sorted);
                        st_gran(a.idx_step + (mo + i), static_cast<uint32_t>(row), tag);
                        st_gran(a.mu + (mo + i), v[2 * i], tag);
                    }
                } else {
                    const float q0 = rintf(y_pre0 - v[0]), q1 = rintf(y_pre1 - v[2]);          // torch.round: half to even
                    uint64_t *ypos = a.yT + static_cast<int64_t>(p) * C * NBT;
                    st_gran(ypos + to, q0 + v[0], tag);
                    st_gran(ypos + (to + static_cast<uint32_t>(2 * NBT)), q1 + v[2], tag);     // channel c0 + 1: the odd-parity piece of the same quad
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float mu = v[2 * i], qq = i ? q1 : q0;
                        const int row = nearest_scale(v[2 * i + 1], tab, a.table_len, tab_sorted);
                        (a.idx + static_cast<int64_t>(p) * C)[oo + i] = row;
                        (a.sym + static_cast<int64_t>(p) * C)[oo + i] = static_cast<int32_t>(qq);
                        (a.ybuf + p)[yo + static_cast<uint32_t>(i * HW)] = qq + mu;
                    }
                }
            }
            if (prof) { a.prof[4 * l] += tq1 - tq0; a.prof[4 * l + 1] += tq2 - tq1; a.prof[4 * l + 2] += wall_clock64() - tq2; }
        });
        if (!alive) return;
        lds_barrier();   // a layer's partial tiles are rewritten in the next step (a workgroup with one layer has no other barrier in between)
    }
    if (prof) {
        a.prof[4 * kMaxLayers] = clock64() - loop_c0;
        a.prof[4 * kMaxLayers + 1] = wall_clock64() - loop_t0;
    }
}

// prior [B][P][HW] -> priorT [HW] x slab [P][nbt] in 16-byte pieces of four steps' operands (bm_prior)
__global__ void transpose_prior_batched_kernel(const float *__restrict__ in, float *__restrict__ out, int B, int P, int HW, int nbt, int64_t total)
{
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int p = static_cast<int>(i % HW);   // position fastest: coalesced reads
        const int64_t r = i / HW;
        const int c = static_cast<int>(r % P);
        const int b = static_cast<int>(r / P);
        out[static_cast<int64_t>(p) * P * nbt + bm_prior(c, b, nbt)] = in[i];
    }
}

// prior [B][P][HW] -> priorT [B][HW][P]
__global__ void transpose_prior_kernel(const float *__restrict__ in, float *__restrict__ out, int P, int HW, int64_t total)
{
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int c = static_cast<int>(i % P);
        const int64_t r = i / P;
        const int p = static_cast<int>(r % HW);
        const int64_t b = r / HW;
        out[i] = in[(b * P + c) * HW + p];
    }
}

}  // namespace

struct basic_scanline_plan {
    int C = 0, P = 0, ksize = 0, nlayers = 0, ntaps = 0, nwg = 1, vec4 = 0;
    int rows[kMaxLayers] = {}, kdim[kMaxLayers] = {}, rpw[kMaxLayers] = {}, woff[kMaxLayers] = {}, act_after[kMaxLayers] = {};
    int kgroup[kMaxLayers] = {}, bpg[kMaxLayers] = {}, kpad[kMaxLayers] = {};   // canonical blocks (see the header comment)
    int tap_dy[kMaxTaps] = {}, tap_dx[kMaxTaps] = {};
    float *d_w[kMaxLayers] = {}, *d_b[kMaxLayers] = {};
    int weight_floats = 0;   // LDS floats of one workgroup's weight slices
    // per-call scratch (grow-only): exchange buffers, position-major copies of the coded latent and the prior, step buffers
    float *d_scratch = nullptr;
    size_t scratch_cap = 0;
    unsigned *d_bar = nullptr;   // [0] barrier counter, [1] error flag
    hipEvent_t done = nullptr;   // completion of this plan's last launch (see ScanChain)
    // batched kernel (scanline_batched_kernel): whether the layers have its shape, and the split of a column tile's workgroups
    bool batched = false;
    int b_nw = 0, b_nd = 0, b_bpt = 0, b_ctx_blocks = 0;
    int b_nblk[kMaxLayers] = {}, b_rt[kMaxLayers] = {}, b_tile_off[kMaxLayers] = {};
    int b_tiles = 0;             // partial tiles (4 KB each) the larger role keeps in LDS
};

namespace {

// Persistent launches spin on a device-wide barrier and must be fully resident (one workgroup per compute unit), so grids
// started from different HIP streams (concurrent stream workers) could each hold compute units the other is waiting for
// until both give up.  They are therefore admitted in GPU time, in host enqueue order: the chip is cut into three slots of
// CUs / 3 compute units; a launch of G workgroups takes ceil(G / (CUs / 3)) consecutive slots (round robin), waits -- a
// stream-wait, nothing blocks on the host -- for the launches that last held them and leaves its own completion event
// there.  Launches in flight hold disjoint slots, so together they never need more compute units than the chip has.
// Ordinary kernels of other streams are no hazard: they drain, and the barrier's spin bound (seconds) covers the wait.
struct ScanChain {
    std::mutex mu;
    struct Dev { hipEvent_t slot[3] = {nullptr, nullptr, nullptr}; int cursor = 0; };
    Dev dev[16];   // per device: launches on different GPUs of one process share nothing (they used to wait on each other's events)
};
ScanChain g_scan_chain;

template <typename Launch> int chained_launch(basic_scanline_plan *p, hipStream_t st, int grid, int cus, Launch &&launch)
{
    int device = 0;
    BASIC_HIP_TRY(hipGetDevice(&device));
    std::lock_guard<std::mutex> lock(g_scan_chain.mu);
    ScanChain::Dev &d = g_scan_chain.dev[device & 15];
    if (!p->done) BASIC_HIP_TRY(hipEventCreateWithFlags(&p->done, hipEventDisableTiming));
    const int per_slot = cus / 3 > 0 ? cus / 3 : 1;
    const int need = std::min(3, (grid + per_slot - 1) / per_slot);
    for (int j = 0; j < need; ++j) {
        hipEvent_t e = d.slot[(d.cursor + j) % 3];
        if (e && e != p->done) BASIC_HIP_TRY(hipStreamWaitEvent(st, e, 0));   // (its own previous launch: same stream, already ordered)
    }
    launch();
    BASIC_HIP_TRY(hipGetLastError());
    BASIC_HIP_TRY(hipEventRecord(p->done, st));
    for (int j = 0; j < need; ++j) d.slot[(d.cursor + j) % 3] = p->done;
    d.cursor = (d.cursor + need) % 3;
    return BASIC_OK;
}

}  // namespace

extern "C" void basic_scanline_plan_destroy(basic_scanline_plan *p)
{
    if (!p) return;
    for (int l = 0; l < kMaxLayers; ++l) {
        if (p->d_w[l]) (void)hipFree(p->d_w[l]);
        if (p->d_b[l]) (void)hipFree(p->d_b[l]);
    }
    if (p->d_scratch) (void)hipFree(p->d_scratch);
    if (p->d_bar) (void)hipFree(p->d_bar);
    if (p->done) {
        std::lock_guard<std::mutex> lock(g_scan_chain.mu);
        for (auto &d : g_scan_chain.dev)
            for (int i = 0; i < 3; ++i)
                if (d.slot[i] == p->done) {
                    (void)hipEventSynchronize(p->done);   // whoever waits on it has been released
                    d.slot[i] = nullptr;
                }
        (void)hipEventDestroy(p->done);
    }
    delete p;
}

extern "C" int basic_scanline_plan_create(const float *ctx_weight, const float *ctx_bias, int channels, int ctx_out, int ksize,
                                          int prior_channels, int n_dense, const float *const *dense_weight,
                                          const float *const *dense_bias, const int *dense_out, const int *act_after,
                                          const int *dense_in_groups, basic_scanline_plan **out)
{
    int rc = require_device();
    if (rc) return rc;
    BASIC_REQUIRE(ctx_weight && out && channels >= 1 && ctx_out >= 2 && (ksize == 3 || ksize == 5 || ksize == 7) && n_dense >= 1 &&
                      n_dense <= kMaxLayers - 1 && dense_weight && dense_out && act_after && prior_channels >= 0,
                  "scanline_plan_create: bad argument");
    BASIC_REQUIRE(dense_out[n_dense - 1] == 2 * channels, "scanline_plan_create: the last layer must give (mean, scale) pairs: 2 * channels rows");
    auto *p = new basic_scanline_plan();
    p->C = channels; p->P = prior_channels; p->ksize = ksize; p->nlayers = 1 + n_dense;
    const int half = ksize / 2;
    for (int ky = 0; ky <= half; ++ky)
        for (int kx = 0; kx < ksize; ++kx) {
            if (ky == half && kx >= half) break;
            p->tap_dy[p->ntaps] = ky - half; p->tap_dx[p->ntaps] = kx - half;
            ++p->ntaps;
        }
    p->rows[0] = ctx_out; p->kdim[0] = p->ntaps * channels; p->act_after[0] = act_after[0];
    for (int l = 1; l <= n_dense; ++l) {
        p->rows[l] = dense_out[l - 1];
        p->kdim[l] = p->rows[l - 1] + (l == 1 ? prior_channels : 0);
        p->act_after[l] = act_after[l];
    }
    // canonical blocks: the context layer's K groups are its taps; a dense layer's are its in_groups equal channel groups
    // (the channel groups of the masked convolution it stands for: cat(ctx, prior) of the first merger layer is two)
    p->kgroup[0] = channels;
    for (int l = 1; l <= n_dense; ++l) {
        const int gi = dense_in_groups ? dense_in_groups[l - 1] : 1;
        if (gi < 1 || p->kdim[l] % gi) {
            delete p;
            set_error("scanline_plan_create: a dense layer's inputs do not divide into its channel groups");
            return BASIC_ERR_INVALID;
        }
        p->kgroup[l] = p->kdim[l] / gi;
    }
    p->vec4 = channels % 4 == 0 && prior_channels % 4 == 0;
    for (int l = 0; l < p->nlayers; ++l) p->vec4 = p->vec4 && p->rows[l] % 4 == 0 && p->kdim[l] % 4 == 0 && p->kgroup[l] % 4 == 0;
    for (int l = 0; l < p->nlayers; ++l) {
        p->bpg[l] = (p->kgroup[l] + kKB - 1) / kKB;
        // row = its groups' channels + kBlockPad floats after every block, + one more pad so that consecutive rows shift banks
        p->kpad[l] = (p->kdim[l] / p->kgroup[l]) * (p->kgroup[l] + kBlockPad * p->bpg[l]) + (p->vec4 ? 4 : 1);
    }
    // workgroups: the fewest (<= 192: the decoder adds its own) whose weight slices fit ~126 KB of LDS (BaSIC, C = 192: 64 --
    // three launches of concurrent stream workers then hold 195 of the 256 compute units and leave the rest to the transforms;
    // with 112 KB it was 77 and the workers' transforms queued behind the persistent launches); every layer in whole rows per workgroup, the
    // last one in whole (mean, scale) pairs
    const char *wk = getenv("BASIC_SCAN_WEIGHT_KB");   // experiments: LDS budget of a workgroup's weight slices
    const size_t weight_kb = wk && atoi(wk) >= 16 && atoi(wk) <= 150 ? static_cast<size_t>(atoi(wk)) : 126;
    int nwg = 1;
    for (;; ++nwg) {
        int floats = 0;
        for (int l = 0; l < p->nlayers; ++l) {
            int rpw = (p->rows[l] + nwg - 1) / nwg;
            if (l == p->nlayers - 1) rpw = (rpw + 1) & ~1;
            floats += rpw * p->kpad[l];
        }
        if (floats * sizeof(float) <= weight_kb * 1024 || nwg >= 192) { p->weight_floats = floats; break; }
    }
    if (p->weight_floats * sizeof(float) > 150 * 1024) {
        delete p;
        set_error("scanline_plan_create: the layers do not fit the LDS of 192 compute units");
        return BASIC_ERR_INVALID;
    }
    p->nwg = nwg;
    int off = 0;
    for (int l = 0; l < p->nlayers; ++l) {
        int rpw = (p->rows[l] + nwg - 1) / nwg;
        if (l == p->nlayers - 1) rpw = (rpw + 1) & ~1;
        p->rpw[l] = rpw;
        p->woff[l] = off;
        off += (rpw * p->kpad[l] + 3) & ~3;
    }
    p->weight_floats = off;
    // upload: context weights as [row][tap][c] (only the causal taps), dense layers as they are ([rows][k]); padded by
    // one workgroup's worth of rows so that the slice copy of the last workgroup never reads past the buffer
    auto upload = [&](const std::vector<float> &h, float **d) -> int {
        BASIC_HIP_TRY(hipMalloc(d, h.size() * sizeof(float)));
        BASIC_HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        return BASIC_OK;
    };
    {
        std::vector<float> h(static_cast<size_t>(p->rows[0] + p->rpw[0]) * p->kdim[0], 0.f);
        for (int r = 0; r < ctx_out; ++r)
            for (int t = 0; t < p->ntaps; ++t)
                for (int c = 0; c < channels; ++c)
                    h[(static_cast<size_t>(r) * p->ntaps + t) * channels + c] =
                        ctx_weight[((static_cast<size_t>(r) * channels + c) * ksize + (p->tap_dy[t] + half)) * ksize + (p->tap_dx[t] + half)];
        rc = upload(h, &p->d_w[0]);
        if (!rc && ctx_bias) { std::vector<float> hb(ctx_bias, ctx_bias + ctx_out); hb.resize(ctx_out + p->rpw[0], 0.f); rc = upload(hb, &p->d_b[0]); }
    }
    for (int l = 1; l <= n_dense && !rc; ++l) {
        std::vector<float> h(static_cast<size_t>(p->rows[l] + p->rpw[l]) * p->kdim[l], 0.f);
        std::copy(dense_weight[l - 1], dense_weight[l - 1] + static_cast<size_t>(p->rows[l]) * p->kdim[l], h.begin());
        rc = upload(h, &p->d_w[l]);
        if (!rc && dense_bias && dense_bias[l - 1]) {
            std::vector<float> hb(dense_bias[l - 1], dense_bias[l - 1] + p->rows[l]);
            hb.resize(p->rows[l] + p->rpw[l], 0.f);
            rc = upload(hb, &p->d_b[l]);
        }
    }
    if (!rc) {
        hipError_t e = hipMalloc(&p->d_bar, 2 * sizeof(unsigned));
        if (e != hipSuccess) rc = hip_fail(e, "scanline_plan_create", __FILE__, __LINE__);
    }
    if (rc) { basic_scanline_plan_destroy(p); return rc; }
    // batched kernel: every layer in whole 32-row tiles and whole 64-channel canonical blocks (no short block, none across a
    // K group or the ctx | prior seam); a context row tile's blocks in the nine register slots of four waves (the late blocks
    // on waves 1 .. bpt, slot 8); at most three blocks per dense layer and wave; at most three dense layers
    {
        bool ok = p->nlayers >= 2 && p->nlayers <= 4 && channels % kKB == 0 && channels / kKB <= kBLate - 1 && p->rows[0] % kKB == 0;
        const int bpt = channels / kKB, nb0 = p->ntaps * bpt;
        ok = ok && nb0 >= kBLate && nb0 - kBLate <= 4 * kBSlots;   // four late blocks (all of the left neighbour's among them), 8 resident per wave
        for (int l = 0; l < p->nlayers && ok; ++l) {
            ok = p->rows[l] % 32 == 0 && p->kdim[l] % kKB == 0 && p->kgroup[l] % kKB == 0;
            p->b_nblk[l] = p->kdim[l] / kKB;
            p->b_rt[l] = p->rows[l] / 32;
            if (l > 0) ok = ok && p->b_nblk[l] <= 4 * (l < 3 ? kBDenseSlots : kBSlots - 2 * kBDenseSlots);   // register slots 0-2, 3-5, 6-7
        }
        if (ok) {
            p->b_bpt = bpt;
            p->b_ctx_blocks = p->rows[0] / kKB;
            int tiles = 0;
            for (int l = 1; l < p->nlayers; ++l) {
                p->b_nd = std::max(p->b_nd, p->b_rt[l]);
                p->b_tile_off[l] = tiles * kBTile;
                tiles += p->b_nblk[l];
            }
            p->b_nw = p->b_nd + p->b_rt[0];
            p->b_tiles = std::max(tiles, p->b_nblk[0]);
            ok = static_cast<size_t>(p->b_tiles) * kBTile * sizeof(float) + 8192 <= 160 * 1024;
        }
        p->batched = ok;
    }
    *out = p;
    return BASIC_OK;
}

extern "C" int basic_scanline_plan_info(const basic_scanline_plan *p, int *workgroups, int *lds_weight_bytes)
{
    BASIC_REQUIRE(p, "scanline_plan_info: null plan");
    if (workgroups) *workgroups = p->nwg;
    if (lds_weight_bytes) *lds_weight_bytes = p->weight_floats * static_cast<int>(sizeof(float));
    return BASIC_OK;
}

namespace {

size_t align4(size_t n) { return (n + 3) & ~static_cast<size_t>(3); }

// fills the launch arguments shared by both directions; *lds_bytes = LDS of a compute workgroup
int fill_args(basic_scanline_plan *p, ScanArgs &a, int batch, int h, int w, const float *d_prior, const float *d_table, int table_len,
              size_t *lds_bytes, bool *pipelined, hipStream_t st)
{
    const int64_t HW = static_cast<int64_t>(h) * w;
    a.B = batch; a.C = p->C; a.H = h; a.W = w; a.P = p->P;
    a.nlayers = p->nlayers; a.ntaps = p->ntaps; a.vec4 = p->vec4;
    a.table = d_table; a.table_len = table_len;
    // scratch: [granule regions: layer exchange buffers, position-major coded latent, step means / rows][position-major prior]
    size_t floats = 0;
    for (int l = 0; l < p->nlayers; ++l) floats += align4(2 * static_cast<size_t>(batch) * p->rows[l]);
    const size_t yT_off = floats;     floats += align4(2 * static_cast<size_t>(batch) * HW * p->C);
    const size_t mu_off = floats;     floats += align4(2 * static_cast<size_t>(batch) * p->C);
    const size_t is_off = floats;     floats += align4(2 * static_cast<size_t>(batch) * p->C);
    const size_t gran_floats = floats;
    const size_t pT_off = floats;     floats += align4(static_cast<size_t>(batch) * HW * p->P);
    if (floats > p->scratch_cap) {
        if (p->d_scratch) (void)hipFree(p->d_scratch);
        p->d_scratch = nullptr; p->scratch_cap = 0;
        BASIC_HIP_TRY(hipMalloc(&p->d_scratch, floats * sizeof(float)));
        p->scratch_cap = floats;
    }
    BASIC_HIP_TRY(hipMemsetAsync(p->d_scratch, 0, gran_floats * sizeof(float), st));   // tag 0 = "not written in this launch"
    size_t ao = 0;
    int kmax = 0;
    for (int l = 0; l < p->nlayers; ++l) {
        a.rows[l] = p->rows[l]; a.kdim[l] = p->kdim[l]; a.rpw[l] = p->rpw[l]; a.woff[l] = p->woff[l]; a.act_after[l] = p->act_after[l];
        a.kgroup[l] = p->kgroup[l]; a.bpg[l] = p->bpg[l]; a.kpad[l] = p->kpad[l];
        a.w[l] = p->d_w[l]; a.bias[l] = p->d_b[l];
        a.act[l] = reinterpret_cast<uint64_t *>(p->d_scratch + ao);
        ao += align4(2 * static_cast<size_t>(batch) * p->rows[l]);
        kmax = p->kdim[l] > kmax ? p->kdim[l] : kmax;
    }
    a.yT = reinterpret_cast<uint64_t *>(p->d_scratch + yT_off);
    a.mu = reinterpret_cast<uint64_t *>(p->d_scratch + mu_off);
    a.idx_step = reinterpret_cast<uint64_t *>(p->d_scratch + is_off);
    a.priorT = nullptr;
    if (p->P > 0) {
        float *pT = p->d_scratch + pT_off;
        const int64_t total = static_cast<int64_t>(batch) * HW * p->P;
        int64_t g = (total + 255) / 256;
        hipLaunchKernelGGL(transpose_prior_kernel, dim3(static_cast<unsigned>(g > 4096 ? 4096 : g)), dim3(256), 0, st, d_prior, pT, p->P,
                           static_cast<int>(HW), total);
        BASIC_HIP_TRY(hipGetLastError());
        a.priorT = pT;
    }
    for (int t = 0; t < p->ntaps; ++t) { a.tap_dy[t] = p->tap_dy[t]; a.tap_dx[t] = p->tap_dx[t]; a.tap_off[t] = p->tap_dy[t] * w + p->tap_dx[t]; }
    // LDS: [weights][table][params of a chunk][inputs of a chunk][flag]; as many images per chunk as fit (at most 8)
    const int total_floats = 160 * 1024 / 4 - 16;
    a.tab_off = static_cast<int>(align4(p->weight_floats));
    a.ps_off = a.tab_off + static_cast<int>(align4(table_len));
    const int rpw_last = p->rpw[p->nlayers - 1];
    int bc = 8 < batch ? 8 : batch;
    // one round of block partials: [blocks][items]; 1024 floats hold a batch-1 layer in one round (<= 36 blocks x ~10 rows)
    a.part_floats = 1024;
    for (int l = 0; l < p->nlayers; ++l)
        BASIC_REQUIRE((p->kdim[l] / p->kgroup[l]) * p->bpg[l] <= a.part_floats, "scanline: too many summation blocks in a layer");
    int bias_need = 0;
    for (int l = 0; l < p->nlayers; ++l) bias_need += p->rpw[l];
    auto need = [&](int n) { return a.ps_off + static_cast<int>(align4(n * rpw_last)) + static_cast<int>(align4(n * kmax)) + a.part_floats + 4 + static_cast<int>(align4(bias_need)); };
    while (bc > 1 && need(bc) > total_floats) --bc;
    BASIC_REQUIRE(need(bc) <= total_floats, "scanline: layer inputs do not fit the LDS");
    a.bc = bc;
    a.xs_off = a.ps_off + static_cast<int>(align4(bc * rpw_last));
    a.part_off = a.xs_off + static_cast<int>(align4(bc * kmax));
    a.flag_off = a.part_off + a.part_floats;
    a.bias_off = a.flag_off + 4;
    int bias_floats = 0;
    for (int l = 0; l < p->nlayers; ++l) bias_floats += p->rpw[l];
    *lds_bytes = static_cast<size_t>(a.bias_off + static_cast<int>(align4(bias_floats))) * sizeof(float);
    // pipelined kernel: [weights][table][biases][context window B x K0][dense inputs B x Kd][partials][early sums][unit descriptors][flag]
    // -- taken when all of it fits (BASIC_SCAN_KERNEL=generic|pipelined overrides; results are identical either way)
    {
        int kd = 0, units_total = 0, units_max = 0;
        for (int l = 0; l < p->nlayers; ++l) {
            if (l > 0) kd = p->kdim[l] > kd ? p->kdim[l] : kd;
            const int u = batch * p->rpw[l] * (p->kdim[l] / p->kgroup[l]) * p->bpg[l];
            units_total += u;
            units_max = u > units_max ? u : units_max;
        }
        const size_t bias_off = align4(static_cast<size_t>(a.tab_off) + table_len);
        const size_t x0_off = bias_off + align4(bias_need);
        const size_t xs_off = x0_off + align4(static_cast<size_t>(batch) * p->kdim[0]);
        const size_t part_off = xs_off + align4(static_cast<size_t>(batch) * kd);
        const size_t early_off = part_off + align4(units_max);
        const size_t desc_off = early_off + align4(static_cast<size_t>(batch) * p->rpw[0]) +
                                align4(static_cast<size_t>(batch) * p->rpw[0] * (p->kdim[0] / p->kgroup[0]) * p->bpg[0]);   // early sums + the early blocks' partials
        const size_t flag_off = desc_off + align4(2 * static_cast<size_t>(units_total));
        // (the early half of a position's context window must be coded two steps before it: the tap up and to the right by
        // ksize / 2 columns is w - ksize / 2 positions back, so the latent must be at least ksize / 2 + 2 columns wide)
        bool fits = flag_off + 4 <= static_cast<size_t>(total_floats) && p->vec4 && (p->ntaps - 1) * (p->C / 2) <= kWinU * kThreads &&
                    w >= p->ksize / 2 + 2;
        for (int l = 0; l < p->nlayers; ++l) fits = fits && batch * p->rpw[l] <= kThreads;   // one finishing item per thread
        const char *e = getenv("BASIC_SCAN_KERNEL");
        if (e && !strcmp(e, "generic")) fits = false;
        if (e && !strcmp(e, "pipelined")) BASIC_REQUIRE(fits, "scanline: BASIC_SCAN_KERNEL=pipelined, but this batch does not fit the LDS");
        *pipelined = fits;
        if (fits) {
            a.bias_off = static_cast<int>(bias_off); a.x0_off = static_cast<int>(x0_off); a.xs_off = static_cast<int>(xs_off);
            a.part_off = static_cast<int>(part_off); a.early_off = static_cast<int>(early_off); a.desc_off = static_cast<int>(desc_off);
            a.flag_off = static_cast<int>(flag_off);
            a.bc = batch;
            *lds_bytes = (flag_off + 4) * sizeof(float);
        }
    }
    a.bar = p->d_bar;
    a.err = reinterpret_cast<int *>(p->d_bar + 1);
#ifdef BASIC_DEBUG_ABLATIONS   // timing ablations (wrong results): only in a library built with `make ABLATIONS=1`
    { const char *e = getenv("BASIC_SCAN_DEBUG"); a.debug = e ? atoi(e) : 0; }
#else
    a.debug = 0;
#endif
    BASIC_HIP_TRY(hipMemsetAsync(p->d_bar, 0, 2 * sizeof(unsigned), st));
    return BASIC_OK;
}

// ---- batched kernel: when it serves a call, and its launch arguments
constexpr int kBatchedMaxTiles = 2;   // column tiles of 32 images per launch

// whether the batched kernel can serve `batch` images of a latent `w` columns wide (the context window's early half must be coded
// two steps before it is used: w >= ksize / 2 + 2); grid = column tiles x workgroups per tile (+ decoder workgroups)
bool batched_fits(const basic_scanline_plan *p, int batch, int w, int ndec, int cus)
{
    if (!p->batched || batch < 1 || batch > 32 * kBatchedMaxTiles || w < p->ksize / 2 + 2) return false;
    const int tiles = (batch + 31) / 32;
    return tiles * p->b_nw + ndec <= cus;
}

// 0 = generic / pipelined kernels, 1 = batched.  BASIC_SCAN_KERNEL=batched|generic|pipelined forces one (identical results)
int choose_batched(const basic_scanline_plan *p, int batch, int w, int ndec, int cus, bool *batched)
{
    const bool fits = batched_fits(p, batch, w, ndec, cus);
    const char *e = getenv("BASIC_SCAN_KERNEL");
    if (e && !strcmp(e, "batched")) BASIC_REQUIRE(fits, "scanline: BASIC_SCAN_KERNEL=batched, but this call does not fit the batched kernel");
    if (e && (!strcmp(e, "generic") || !strcmp(e, "pipelined"))) { *batched = false; return BASIC_OK; }
    const char *m = getenv("BASIC_SCAN_BATCHED_FROM");   // smallest batch the batched kernel takes by default
    const int from = m && atoi(m) >= 1 ? atoi(m) : 3;
    *batched = fits && (batch >= from || (e && !strcmp(e, "batched")));
    return BASIC_OK;
}

int fill_args_batched(basic_scanline_plan *p, ScanArgs &a, int batch, int h, int w, const float *d_prior, const float *d_table, int table_len,
                      size_t *lds_bytes, hipStream_t st)
{
    const int64_t HW = static_cast<int64_t>(h) * w;
    const int tiles = (batch + 31) / 32, nbt = 32 * tiles;
    a.B = batch; a.C = p->C; a.H = h; a.W = w; a.P = p->P;
    a.nlayers = p->nlayers; a.ntaps = p->ntaps; a.vec4 = p->vec4;
    a.table = d_table; a.table_len = table_len;
    a.nbt = nbt; a.nw = p->b_nw; a.nd = p->b_nd; a.bpt = p->b_bpt; a.ctx_blocks = p->b_ctx_blocks;
    // scratch: [granule regions: layer exchange arrays [rows][nbt], coded latent [HW][C][nbt], step means / rows [nbt][C]][prior [HW][P][nbt]]
    size_t floats = 0;
    for (int l = 0; l + 1 < p->nlayers; ++l) floats += align4(2 * static_cast<size_t>(nbt) * p->rows[l]);
    const size_t yT_off = floats;     floats += align4(2 * static_cast<size_t>(nbt) * HW * p->C);
    const size_t mu_off = floats;     floats += align4(2 * static_cast<size_t>(nbt) * p->C);
    const size_t is_off = floats;     floats += align4(2 * static_cast<size_t>(nbt) * p->C);
    const size_t gran_floats = floats;
    const size_t pT_off = floats;     floats += align4(static_cast<size_t>(nbt) * HW * p->P);
    const size_t wl_off = floats;     floats += static_cast<size_t>(tiles) * p->b_nw * kThreads * 32;
    if (floats > p->scratch_cap) {
        if (p->d_scratch) (void)hipFree(p->d_scratch);
        p->d_scratch = nullptr; p->scratch_cap = 0;
        BASIC_HIP_TRY(hipMalloc(&p->d_scratch, floats * sizeof(float)));
        p->scratch_cap = floats;
    }
    BASIC_HIP_TRY(hipMemsetAsync(p->d_scratch, 0, gran_floats * sizeof(float), st));   // tag 0 = "not written in this launch"
    a.wlate = p->d_scratch + wl_off;
    size_t ao = 0;
    for (int l = 0; l < p->nlayers; ++l) {
        a.rows[l] = p->rows[l]; a.kdim[l] = p->kdim[l]; a.act_after[l] = p->act_after[l];
        a.nblk[l] = p->b_nblk[l]; a.rt[l] = p->b_rt[l]; a.tile_off[l] = p->b_tile_off[l];
        a.w[l] = p->d_w[l]; a.bias[l] = p->d_b[l];
        a.act[l] = nullptr;
        if (l + 1 < p->nlayers) {
            a.act[l] = reinterpret_cast<uint64_t *>(p->d_scratch + ao);
            ao += align4(2 * static_cast<size_t>(nbt) * p->rows[l]);
        }
    }
    a.yT = reinterpret_cast<uint64_t *>(p->d_scratch + yT_off);
    a.mu = reinterpret_cast<uint64_t *>(p->d_scratch + mu_off);
    a.idx_step = reinterpret_cast<uint64_t *>(p->d_scratch + is_off);
    a.priorT = nullptr;
    if (p->P > 0) {
        float *pT = p->d_scratch + pT_off;
        const int64_t total = static_cast<int64_t>(batch) * HW * p->P;
        int64_t g = (total + 255) / 256;
        hipLaunchKernelGGL(transpose_prior_batched_kernel, dim3(static_cast<unsigned>(g > 8192 ? 8192 : g)), dim3(256), 0, st, d_prior, pT, batch,
                           p->P, static_cast<int>(HW), nbt, total);
        BASIC_HIP_TRY(hipGetLastError());
        a.priorT = pT;
    }
    for (int t = 0; t < p->ntaps; ++t) { a.tap_dy[t] = p->tap_dy[t]; a.tap_dx[t] = p->tap_dx[t]; a.tap_off[t] = p->tap_dy[t] * w + p->tap_dx[t]; }
    *lds_bytes = (align4(table_len) + 4 + 96 + static_cast<size_t>(p->b_tiles) * kBTile) * sizeof(float);   // table, flags, biases, partial tiles
    a.bar = p->d_bar;
    a.err = reinterpret_cast<int *>(p->d_bar + 1);
    a.debug = 0;
    BASIC_HIP_TRY(hipMemsetAsync(p->d_bar, 0, 2 * sizeof(unsigned), st));
    return BASIC_OK;
}

// BASIC_SCAN_PROFILE=1 (debugging aid): where does workgroup 0 spend a coding step?  Synchronises the stream.
struct ScanProfile {
    static constexpr int kSlots = 4 * kMaxLayers + 8;
    long long *d = nullptr;
    int begin(ScanArgs &a, hipStream_t st)
    {
        if (!getenv("BASIC_SCAN_PROFILE")) return BASIC_OK;
        BASIC_HIP_TRY(hipMalloc(&d, kSlots * sizeof(long long)));
        BASIC_HIP_TRY(hipMemsetAsync(d, 0, kSlots * sizeof(long long), st));
        a.prof = d;
        return BASIC_OK;
    }
    void report(const ScanArgs &a, hipStream_t st, const char *what)
    {
        if (!d) return;
        long long h[kSlots];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipFree(d);
        d = nullptr;
        const double steps = static_cast<double>(a.H) * a.W;
        if (a.nbt) {   // batched kernel: first context workgroup (late: wave 1, early: wave 0), first dense workgroup (wave 0)
            fprintf(stderr, "scan-line %s profile, 10 ns ticks per coding step | context tile: wait for y %.1f, late block %.1f, barrier + finish %.1f, early half %.1f | ", what,
                    h[0] / steps, h[1] / steps, h[2] / steps, h[3] / steps);
            for (int l = 1; l < a.nlayers; ++l)
                fprintf(stderr, "dense L%d: prior blocks + wait %.1f, loads + chains %.1f, barrier + finish %.1f | ", l, h[4 * l] / steps, h[4 * l + 1] / steps, h[4 * l + 2] / steps);
            fprintf(stderr, "loop %.1f ticks per step, %.2f shader clocks per tick", h[4 * kMaxLayers + 1] / steps,
                    h[4 * kMaxLayers + 1] ? static_cast<double>(h[4 * kMaxLayers]) / h[4 * kMaxLayers + 1] : 0.0);
            if (h[4 * kMaxLayers + 3])
                fprintf(stderr, " | decoder wave of stream 0: waiting %.1f, decoding %.1f (shader clocks per step: decode_chunk %.0f, publishing %.0f; stream words per step %.1f)", h[4 * kMaxLayers + 2] / steps,
                        h[4 * kMaxLayers + 3] / steps, h[4 * kMaxLayers + 4] / steps, h[4 * kMaxLayers + 5] / steps, h[4 * kMaxLayers + 6] / steps);
            fprintf(stderr, "\n");
            return;
        }
        fprintf(stderr, "scan-line %s profile (workgroup 0; 10 ns ticks per coding step: stage+wait / dots / finish / gauss): ", what);
        for (int l = 0; l < a.nlayers; ++l)
            fprintf(stderr, "L%d %.1f / %.1f / %.1f / %.1f | ", l, h[4 * l] / steps, h[4 * l + 1] / steps, h[4 * l + 2] / steps, h[4 * l + 3] / steps);
        fprintf(stderr, "loop %.1f ticks per step, %.2f shader clocks per tick", h[4 * kMaxLayers + 1] / steps,
                h[4 * kMaxLayers + 1] ? static_cast<double>(h[4 * kMaxLayers]) / h[4 * kMaxLayers + 1] : 0.0);
        if (h[4 * kMaxLayers + 3])
            fprintf(stderr, " | decoder wave of stream 0: waiting %.1f, decoding %.1f (shader clocks per step: decode_chunk %.0f, publishing %.0f; stream words per step %.1f)", h[4 * kMaxLayers + 2] / steps,
                    h[4 * kMaxLayers + 3] / steps, h[4 * kMaxLayers + 4] / steps, h[4 * kMaxLayers + 5] / steps, h[4 * kMaxLayers + 6] / steps);
        fprintf(stderr, "\n");
    }
};

int device_cus(int *cus)
{
    int dev = 0;
    BASIC_HIP_TRY(hipGetDevice(&dev));
    BASIC_HIP_TRY(hipDeviceGetAttribute(cus, hipDeviceAttributeMultiprocessorCount, dev));
    return BASIC_OK;
}

}  // namespace

extern "C" int basic_scanline_encode_dev(basic_scanline_plan *p, const float *d_y, const float *d_prior, int batch, int h, int w,
                                         const float *d_table, int table_len, int32_t *d_symbols, int32_t *d_indexes, float *d_ybuf,
                                         void *hip_stream)
{
    BASIC_REQUIRE(p && d_y && d_table && d_symbols && d_indexes && d_ybuf && batch >= 1 && h >= 1 && w >= 1 && table_len >= 1 &&
                      table_len <= 4096 && (d_prior || p->P == 0),
                  "scanline_encode: bad argument");
    hipStream_t st = as_stream(hip_stream);
    ScanArgs a{};
    size_t lds_bytes = 0;
    bool pipelined = false, batched = false;
    int cus = 0;
    int rc = device_cus(&cus);
    if (rc) return rc;
    rc = choose_batched(p, batch, w, 0, cus, &batched);
    if (rc) return rc;
    if (batched) {
        rc = fill_args_batched(p, a, batch, h, w, d_prior, d_table, table_len, &lds_bytes, st);
        if (rc) return rc;
        a.y = d_y; a.ybuf = d_ybuf; a.sym = d_symbols; a.idx = d_indexes;
        const int grid = (a.nbt / 32) * p->b_nw;
        a.ncompute = grid;
        if (lds_bytes < 96 * 1024) lds_bytes = 96 * 1024;   // one workgroup per compute unit
        BASIC_HIP_TRY(ensure_max_lds(reinterpret_cast<const void *>(scanline_batched_kernel<false>)));
        ScanProfile prof;
        rc = prof.begin(a, st);
        if (rc) return rc;
        rc = chained_launch(p, st, grid, cus, [&] { hipLaunchKernelGGL(scanline_batched_kernel<false>, dim3(grid), dim3(kThreads), lds_bytes, st, a); });
        prof.report(a, st, "encode (batched)");
        return rc;
    }
    rc = fill_args(p, a, batch, h, w, d_prior, d_table, table_len, &lds_bytes, &pipelined, st);
    if (rc) return rc;
    a.y = d_y; a.ybuf = d_ybuf; a.sym = d_symbols; a.idx = d_indexes;
    a.ncompute = p->nwg;
    BASIC_REQUIRE(p->nwg <= cus, "scanline_encode: more workgroups than compute units (the grid must be resident)");
    // more than half of a compute unit's LDS per workgroup: exactly one workgroup per unit, as the barrier protocol assumes
    if (lds_bytes < 96 * 1024) lds_bytes = 96 * 1024;
    BASIC_HIP_TRY(ensure_max_lds(reinterpret_cast<const void *>(scanline_persistent_kernel<false>)));
    BASIC_HIP_TRY(ensure_max_lds(reinterpret_cast<const void *>(scanline_pipelined_kernel<false>)));
    ScanProfile prof;
    rc = prof.begin(a, st);
    if (rc) return rc;
    rc = chained_launch(p, st, p->nwg, cus, [&] {
        if (pipelined) hipLaunchKernelGGL(scanline_pipelined_kernel<false>, dim3(p->nwg), dim3(kThreads), lds_bytes, st, a);
        else hipLaunchKernelGGL(scanline_persistent_kernel<false>, dim3(p->nwg), dim3(kThreads), lds_bytes, st, a);
    });
    prof.report(a, st, "encode");
    return rc;
}

extern "C" int basic_scanline_decode_dev(basic_scanline_plan *p, const basic_rans_tables *tables, const uint32_t *d_words,
                                         const int64_t *d_word_off, const float *d_prior, int batch, int h, int w, const float *d_table,
                                         int table_len, int32_t *d_symbols, int32_t *d_indexes, float *d_ybuf, void *hip_stream)
{
    BASIC_REQUIRE(p && tables && d_words && d_word_off && d_table && d_symbols && d_indexes && d_ybuf && batch >= 1 && h >= 1 && w >= 1 &&
                      table_len >= 1 && table_len <= 4096 && (d_prior || p->P == 0),
                  "scanline_decode: bad argument");
    hipStream_t st = as_stream(hip_stream);
    ScanArgs a{};
    size_t lds_bytes = 0;
    bool pipelined = false, batched = false;
    const int ndec = (batch + kThreads / 64 - 1) / (kThreads / 64);
    int cus = 0;
    int rc = device_cus(&cus);
    if (rc) return rc;
    rc = choose_batched(p, batch, w, ndec, cus, &batched);
    if (rc) return rc;
    if (batched) {
        rc = fill_args_batched(p, a, batch, h, w, d_prior, d_table, table_len, &lds_bytes, st);
        if (rc) return rc;
        rc = rans_fast_view(tables, &a.tv);
        if (rc) return rc;
        a.ybuf = d_ybuf; a.sym = d_symbols; a.idx = d_indexes; a.words = d_words; a.word_off = d_word_off;
        const int ncompute = (a.nbt / 32) * p->b_nw;
        a.ncompute = ncompute;
        const size_t dec_lds_b = (static_cast<size_t>((a.tv.image_words + 3) & ~3) + 4 * static_cast<size_t>(a.tv.rows) + 4) * sizeof(uint32_t);
        if (dec_lds_b > lds_bytes) lds_bytes = dec_lds_b;
        if (lds_bytes < 96 * 1024) lds_bytes = 96 * 1024;
        BASIC_REQUIRE(lds_bytes <= 160 * 1024, "scanline_decode: the search image does not fit the LDS");
        BASIC_HIP_TRY(ensure_max_lds(reinterpret_cast<const void *>(scanline_batched_kernel<true>)));
        ScanProfile prof;
        rc = prof.begin(a, st);
        if (rc) return rc;
        rc = chained_launch(p, st, ncompute + ndec, cus,
                            [&] { hipLaunchKernelGGL(scanline_batched_kernel<true>, dim3(ncompute + ndec), dim3(kThreads), lds_bytes, st, a); });
        prof.report(a, st, "decode (batched)");
        return rc;
    }
    rc = fill_args(p, a, batch, h, w, d_prior, d_table, table_len, &lds_bytes, &pipelined, st);
    if (rc) return rc;
    rc = rans_fast_view(tables, &a.tv);
    if (rc) return rc;
    a.ybuf = d_ybuf; a.sym = d_symbols; a.idx = d_indexes; a.words = d_words; a.word_off = d_word_off;
    a.ncompute = p->nwg;
    BASIC_REQUIRE(p->nwg + ndec <= cus, "scanline_decode: more workgroups than compute units (the grid must be resident)");
    const size_t dec_lds = (static_cast<size_t>((a.tv.image_words + 3) & ~3) + 4 * static_cast<size_t>(a.tv.rows) + 4) * sizeof(uint32_t);
    if (dec_lds > lds_bytes) lds_bytes = dec_lds;
    if (lds_bytes < 96 * 1024) lds_bytes = 96 * 1024;
    BASIC_REQUIRE(lds_bytes <= 160 * 1024, "scanline_decode: the search image does not fit the LDS");
    BASIC_HIP_TRY(ensure_max_lds(reinterpret_cast<const void *>(scanline_persistent_kernel<true>)));
    BASIC_HIP_TRY(ensure_max_lds(reinterpret_cast<const void *>(scanline_pipelined_kernel<true>)));
    ScanProfile prof;
    rc = prof.begin(a, st);
    if (rc) return rc;
    rc = chained_launch(p, st, p->nwg + ndec, cus, [&] {
        if (pipelined) hipLaunchKernelGGL(scanline_pipelined_kernel<true>, dim3(p->nwg + ndec), dim3(kThreads), lds_bytes, st, a);
        else hipLaunchKernelGGL(scanline_persistent_kernel<true>, dim3(p->nwg + ndec), dim3(kThreads), lds_bytes, st, a);
    });
    prof.report(a, st, "decode");
    return rc;
}

// Whether basic_scanline_decode_dev can serve `batch` streams of this table set on the current device (the set has a fast
// search image that fits the LDS, and compute + decoder workgroups fit the chip).  Both paths code the same integers, so a
// caller that gets 0 simply decodes with the per-step path.
extern "C" int basic_scanline_can_decode(const basic_scanline_plan *p, const basic_rans_tables *tables, int batch, int *ok)
{
    BASIC_REQUIRE(p && tables && ok && batch >= 1, "scanline_can_decode: bad argument");
    *ok = 0;
    RansFastView tv;
    if (rans_fast_view(tables, &tv) != BASIC_OK) return BASIC_OK;
    int cus = 0;
    int rc = device_cus(&cus);
    if (rc) return rc;
    const int ndec = (batch + kThreads / 64 - 1) / (kThreads / 64);
    const size_t dec_lds = (static_cast<size_t>((tv.image_words + 3) & ~3) + 4 * static_cast<size_t>(tv.rows) + 4) * sizeof(uint32_t);
    *ok = p->nwg + ndec <= cus && dec_lds <= 160 * 1024;
    return BASIC_OK;
}

// The largest batch the batched kernel (the batch as the N dimension of MFMA tiles, weights in registers) serves for a latent
// `w` columns wide on the current device: 0 = never (the layers do not have its shape, or the latent is too narrow);
// `decode` != 0 counts the decoder workgroups (one wavefront per image stream) too.
extern "C" int basic_scanline_batched_max(const basic_scanline_plan *p, int w, int decode, int *max_batch)
{
    BASIC_REQUIRE(p && max_batch && w >= 1, "scanline_batched_max: bad argument");
    *max_batch = 0;
    int cus = 0;
    int rc = device_cus(&cus);
    if (rc) return rc;
    for (int b = 32 * kBatchedMaxTiles; b >= 1; b -= 32) {
        const int ndec = decode ? (b + kThreads / 64 - 1) / (kThreads / 64) : 0;
        if (batched_fits(p, b, w, ndec, cus)) { *max_batch = b; break; }
    }
    return BASIC_OK;
}

// 0 = the last launch on this plan completed its barriers; 1 = a barrier timed out (results invalid).  Synchronises `hip_stream`.
extern "C" int basic_scanline_status(basic_scanline_plan *p, void *hip_stream, int *poisoned)
{
    BASIC_REQUIRE(p && poisoned, "scanline_status: bad argument");
    unsigned h[2] = {0, 0};
    BASIC_HIP_TRY(hipMemcpyAsync(h, p->d_bar, sizeof(h), hipMemcpyDeviceToHost, as_stream(hip_stream)));
    BASIC_HIP_TRY(hipStreamSynchronize(as_stream(hip_stream)));
    *poisoned = h[1] != 0;
    return BASIC_OK;
}
