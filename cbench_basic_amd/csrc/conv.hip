// Analysis / synthesis transforms: tap-list implicit-GEMM convolution on the fp32 matrix core
// (v_mfma_f32_32x32x2_f32) with LDS-staged activation patches and a fused
// bias + {ReLU, LeakyReLU, GDN, IGDN} epilogue.
//
// Replaces the ATen calls behind nn/models/google.py:25-101 (compressai conv/deconv/GDN) and
// nn/layers/slimmable_layers.py:157-183,258-282 (weight slicing = plan of the active slice).
//
// Formulation.  Every layer is a sum over TAPS of 1x1 GEMMs on a shifted input grid:
//     out[co][s_out*m + o0] = bias[co] + sum_t sum_ci W_t[co][ci] * in[ci][s_in*m + d_t]
//   * Conv2d(k, s, p):           one launch, taps = all k*k, s_in = s, s_out = 1, d = k_idx - p.
//   * ConvTranspose2d(k, s, p):  s*s launches (sub-pixel phases); phase (py,px) keeps the taps
//                                with ky = (py+p) mod s, s_in = 1, s_out = s, d = (py+p-ky)/s,
//                                so no multiply-by-zero work is ever issued.
// GEMM mapping (M = output channels, N = 32 output positions per wavefront, K = taps x Cin):
//   * workgroup = 4 (or 8) wavefronts; each wave owns ALL output channels x 32 positions, i.e. MT 32x32
//     accumulator tiles.  Owning every channel of its pixels is what lets the GDN normalisation
//     norm = beta + gamma . x^2 run as a SECOND MFMA GEMM straight from the accumulators:
//     the C/D layout (column = lane, rows in registers) of x^2 IS the B-operand layout of the
//     next 32x32x2 MFMA, with k-pairs (c, c+4) -- no LDS round trip, no lane shuffles.
//   * K is walked CK input channels at a time through TWO LDS stage buffers {weight slab
//     [taps][CK][32][MTP], input patch [TB][CK][PH][PW]} filled by LDS-DMA (global_load_lds) one stage
//     ahead of the MFMAs; B fragments are patch reads at (s_in*pos + tap shift), the A fragments of all
//     M-tiles of one (tap, ci) step are MTP consecutive floats (one ds_read_b128).
//   * Measured on gfx950 (scripts/micro/mfma_peak.hip, mfma_lds.hip): back-to-back fp32 MFMAs reach
//     99 % of the 157.3 TFLOP/s peak, and every OTHER instruction a SIMD issues costs matrix-pipe time.
//     So for the codec's tap grids the stage is fully unrolled with compile-time LDS offsets (two LDS
//     reads per MT MFMAs, nothing else); the remaining losses are stage barriers and DMA issue.
#include "common.h"

#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace basic;

namespace {
thread_local bool g_dynamic_tiles = false;   // see basic::set_dynamic_tiles
}
namespace basic {
bool set_dynamic_tiles(bool on)
{
    const bool prev = g_dynamic_tiles;
    g_dynamic_tiles = on;
    return prev;
}
}  // namespace basic

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Input channels per LDS stage: as many as two stage buffers allow (2 x ~31 KB per 4-wave workgroup at two
// workgroups per CU, 2 x ~72 KB for the 8-wave workgroup that has the CU to itself).
constexpr int kFewTaps = 9;      // 3x3 conv, the 3x3 sub-pixel phase of the 5x5 stride-2 transposed conv
constexpr int kVeryFewTaps = 6;  // its other three phases
constexpr int stage_channels(int ntaps, int waves)
{
    return (ntaps > kFewTaps ? 2 : (ntaps > kVeryFewTaps ? 4 : 8)) * (waves == 8 ? 2 : 1);
}
// 8-wave workgroups pay off for the 25-tap, 128-channel convolutions (measured: +3..5 %); the few-tap phases of
// the transposed convolutions run faster as two 4-wave workgroups per CU.
constexpr int plan_waves(int mt, int ntaps) { return (mt == 4 && ntaps > kFewTaps) ? 8 : 4; }
constexpr int kMaxTaps = 25;  // 5x5
constexpr int kTilePos = 128;  // output positions per workgroup
constexpr int kPSlots = 12;             // patch elements per thread and stage (patch <= 3072 floats) ...
constexpr int kPSlotsNarrow = 24;       // ... and for launches of <= 2 M-tiles, which have the registers for more descriptors
constexpr int patch_slots(int mt) { return mt <= 2 ? kPSlotsNarrow : kPSlots; }
constexpr int kPSlotsFused = 6;         // ... and for the fused column-phase launches, which carry two accumulator sets
constexpr int kFusedCK = 4;             // their channels per stage (15 / 10 taps: two 4-wave workgroups per CU still fit)
constexpr int kMaxCoutPerLaunch = 192;  // 6 accumulator tiles per wave
constexpr int kSplitBelowBlocks = 384;   // position grids smaller than this use the 32-channel-slice variant

struct TapLaunch {
    const float *in;
    float *out;
    const float *wpack;   // [gridDim.y][cin_pad/CK][ntaps][CK][32][MTP]  (MTP = mtile_pitch(coutp / 32))
    const float *wrow;    // first-layer row slab [ky * 16 + kx * 3 + c][32][4] (conv5x5_cin4_gdn_persistent_kernel<true>), or nullptr
    const float *bias;    // [gridDim.y][coutp] (zero padded) or nullptr
    int64_t split_wstride;  // floats between the weight packs of consecutive blockIdx.y output-channel slices
    const float *gammaT;  // [coutp(k)][32][MTP]: gamma[i = 32 m + col][k] at [k][col][m], zero padded
    const float *beta;    // [coutp]
    int batch, cin, cin_pad, cout, coutp;
    int out_ctotal, co_base;  // this launch writes channels co_base .. co_base+cout of out_ctotal (slice y: coutp*y onwards)
    int in_h, in_w, out_h, out_w;
    int mh, mw;  // m-grid of this launch
    int s_in, s_out, oy0, ox0;
    int ntaps, dymin, dxmin;
    int tb_log, th_log, tw_log;  // tile = 2^tb images x 2^th x 2^tw positions (product 128)
    int ph, pw, pwp;             // LDS patch rows / cols / padded cols
    int patch4;                  // 1: patch rows are whole 16-byte groups aligned to the image (x origin floor4(x0)): 16-byte DMA pieces
    int act;
    int debug;  // ablation switches for profiling only (BASIC_CONV_DEBUG): 1 = skip staging, 2 = skip MFMA loop
    int tiles_y, tiles_x;
    signed char dy[kMaxTaps], dx[kMaxTaps];  // relative to (dymin, dxmin)
};

// Workgroup -> tile map.  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one; speed only, never
// correctness), and each XCD has its own L2: with tile = blockIdx.x, horizontally adjacent tiles -- which share the 128-byte
// lines under their common halo columns -- always sit on different XCDs and every such line is pulled out of the
// fabric twice.  Handing each XCD a CONTIGUOUS range of tiles keeps neighbours behind one L2.
__device__ __forceinline__ int xcd_tile(int b, int n)
{
    const int per = n >> 3;
    return b < (per << 3) ? (b & 7) * per + (b >> 3) : b;
}

__device__ __forceinline__ float apply_act(float v, int act)
{
    if (act == BASIC_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == BASIC_ACT_LEAKY_RELU) return v > 0.f ? v : 0.01f * v;
    return v;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_cvoid;

__device__ const float basic_zero_page[64] = {0.f};  // source of every zero-filled patch element

// One lane's A fragments of a (tap, channel) step: MTP consecutive floats = one ds_read_b32/b64/b128 (two for MTP 8).
template <int MTP>
__device__ __forceinline__ void load_a(const float *p, float (&d)[MTP])
{
    if constexpr (MTP == 1) {
        d[0] = p[0];
    } else if constexpr (MTP == 2) {
        const f32x2 v = *reinterpret_cast<const f32x2 *>(p);
        d[0] = v[0]; d[1] = v[1];
    } else {
#pragma unroll
        for (int q = 0; q < MTP / 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(p + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[4 * q + e] = v[e];
        }
    }
}

constexpr int mtile_pitch(int mt) { return mt <= 1 ? 1 : (mt <= 2 ? 2 : (mt <= 4 ? 4 : 8)); }

// KH x KW > 0: the launch's taps form a dense KH x KW grid (tap t = row t / KW, column t % KW of the
// patch) and the whole stage is unrolled with compile-time LDS offsets; KH = 0: runtime tap table.
// WAVES = 4: 128 positions per workgroup, two workgroups per CU (MT <= 4); WAVES = 8: 256 positions, one
// workgroup per CU with twice the LDS per stage -- half as many barriers and weight DMAs per MFMA.
// KWB > 0: FUSED column phases of a stride-2 transposed convolution.  The launch computes the two output columns
// ox = 2 mx (taps KH x KW, accumulators acc) and ox = 2 mx + 1 (taps KH x KWB at patch columns shifted by KW - KWB,
// accumulators acc2) of every m-grid position from ONE input patch and stores them as 8-byte pairs: whole sectors
// instead of every other 4 bytes (a single phase writes half of each 32-byte sector; measured, the store tail of the
// four separate phases of g_s layer 3 cost 1.9 ms against 0.2 ms for the stride-2 convolution of the same size).
template <int MT, int kCK, int KH, int KW, int WAVES, int KWB = 0>
__global__ __launch_bounds__(64 * WAVES, (MT <= 4 && WAVES == 4 ? 2 : 1)) void conv_tap_mfma_kernel(const TapLaunch g)
{
    extern __shared__ float lds[];
    constexpr int kThreads = 64 * WAVES;   // shadows the 4-wave default of the host code
    constexpr int kWPiece = kThreads * 4;  // floats one 16-byte DMA instruction of the whole workgroup moves
    constexpr int kPSlots = KWB > 0 ? kPSlotsFused : patch_slots(MT);
    constexpr int kTapsA = KH * KW, kTapsAll = KH * (KW + KWB);
    static_assert(KWB == 0 || (KH > 0 && KWB <= KW && MT <= 4 && WAVES == 4), "fused phases: unrolled 4-wave launches of <= 4 tiles");
    constexpr int MTP = mtile_pitch(MT);  // A fragments of a lane sit MTP floats apart: [tap][ci][col][MTP]
    // output-channel slice of this block (small-grid launches spread Cout over blockIdx.y)
    const float *const wpack = g.wpack + blockIdx.y * g.split_wstride;
    const float *const bias = g.bias ? g.bias + blockIdx.y * g.coutp : nullptr;
    const int co_base = g.co_base + blockIdx.y * g.coutp;
    const int cout_here = (g.cout - static_cast<int>(blockIdx.y) * g.coutp < g.coutp) ? g.cout - static_cast<int>(blockIdx.y) * g.coutp : g.coutp;
    // LDS: two stage buffers { weight slab [tap][ci][col][MTP] | input patch [TB][CK][PH][PWP] }, filled by
    // LDS-DMA (global_load_lds: no VGPR round trip, no ds_write) one stage ahead of the MFMAs that read
    // them, then the tap-offset table of the runtime-tap variant.  The DMA image is lane-linear, which is
    // why both regions are padded to whole instructions of the workgroup (16 / 4 bytes per thread).
    const int wl_floats = g.ntaps * kCK * 32 * MTP;
    const int gam_floats = 32 * 32 * MTP;
    const int wl_pad = ((wl_floats > gam_floats ? wl_floats : gam_floats) + kWPiece - 1) / kWPiece * kWPiece;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: keeps LDS-DMA bases on the scalar unit
    const int khalf = lane >> 5, col = lane & 31;

    // tile origin
    int bid = xcd_tile(blockIdx.x, gridDim.x);
    const int tx_i = bid % g.tiles_x; bid /= g.tiles_x;
    const int ty_i = bid % g.tiles_y; bid /= g.tiles_y;
    const int TB = 1 << g.tb_log, TH = 1 << g.th_log, TW = 1 << g.tw_log;
    const int b0 = bid * TB, my0 = ty_i * TH, mx0 = tx_i * TW;

    // this lane's output position inside the tile
    const int q = wave * 32 + col;
    const int tx = q & (TW - 1);
    const int ty = (q >> g.tw_log) & (TH - 1);
    const int tb_raw = q >> (g.tw_log + g.th_log);
    const bool lane_live = tb_raw < TB;         // tiles of tiny maps may hold fewer than 128 positions
    const int tb = lane_live ? tb_raw : 0;

    const int chan_stride = g.ph * g.pwp;
    // with 16-byte patch pieces the patch starts at the 4-aligned column at or left of the tile's first column
    const int gx0_tile = mx0 * g.s_in + g.dxmin;
    const int xshift = g.patch4 ? (gx0_tile & 3) : 0;  // two's complement: also right for negative columns
    const int lane_b_base = ((tb * kCK + khalf) * g.ph + ty * g.s_in) * g.pwp + tx * g.s_in + xshift;
    const int lane_a_base = (khalf * 32 + col) * MTP;

    f32x16 acc[MT];
    f32x16 acc2[KWB > 0 ? MT : 1];  // second column phase (fused launches)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[m][r] = 0.f;
            if (KWB > 0) acc2[m][r] = 0.f;
        }

    const int patch_elems = TB * kCK * chan_stride;
    const int punit = g.patch4 ? 4 : 1;                       // floats per DMA lane
    const int patch_units = patch_elems / punit;              // (pwp is a multiple of 4 with patch4)
    const int patch_units_pad = (patch_units + 63) & ~63;     // DMA granularity: one wave-instruction (64 lanes)
    const int patch_pad = patch_units_pad * punit;
    const int stage_floats = wl_pad + patch_pad;
    const int n_wslots = wl_pad / kWPiece;
    const int nstages = g.cin_pad / kCK;
    const int gy0 = my0 * g.s_in + g.dymin, gx0 = mx0 * g.s_in + g.dxmin;
    const int64_t in_plane = static_cast<int64_t>(g.in_h) * g.in_w;
    const int64_t in_img = static_cast<int64_t>(g.cin) * in_plane;
    const float *in_b0 = g.in + static_cast<int64_t>(b0) * in_img;

    int *tapoff = reinterpret_cast<int *>(lds + 2 * stage_floats);  // [ntaps] LDS offsets of the taps
    if (KH == 0 && tid < g.ntaps) tapoff[tid] = g.dy[tid] * g.pwp + g.dx[tid];
    // Per-channel constants of the epilogue (bias, beta) go to LDS NOW, before any DMA is in flight: a global
    // load next to outstanding LDS-DMAs makes the compiler wait vmcnt(0) per load (16 serial L2 round trips).
    float *chan_const = lds + 2 * stage_floats + 32;  // [coutp] bias, [coutp] beta
    if (tid < g.coutp) {
        chan_const[tid] = bias ? bias[tid] : 0.f;
        chan_const[g.coutp + tid] = g.beta ? g.beta[tid] : 1.f;
    }
    if (nstages == 0) __syncthreads();  // otherwise the first stage barrier publishes them

    // Per-thread gather descriptors, computed once: patch element i = tid + 256*s of every stage is DMA'd
    // from pp[s], which then advances by pstride[s] bytes (CK channels); elements outside the image (and the
    // padding of the patch) read a zero word and never move.  Slots whose channel does not exist in the last
    // stage (cin not a multiple of CK) are flagged in lastmask.
    const float *pp[kPSlots];
    int pstride[kPSlots];
    unsigned lastmask = 0;
#pragma unroll
    for (int sl = 0; sl < kPSlots; ++sl) {
        int r = tid + sl * kThreads;
        const float *ptr = basic_zero_page;
        int stride = 0;
        if (r < patch_units) {
            const int upr = g.pwp / punit;  // DMA units per patch row
            const int px = (r % upr) * punit; r /= upr;
            const int py = r % g.ph; r /= g.ph;
            const int ci = r % kCK;
            const int pb = r / kCK;
            const int gy = gy0 + py, gx = gx0 - xshift + px;
            // a 16-byte group is entirely inside or entirely outside the image (in_w % 4 == 0, gx % 4 == 0)
            if ((g.patch4 || px < g.pw) && gy >= 0 && gy < g.in_h && gx >= 0 && gx < g.in_w && b0 + pb < g.batch && ci < g.cin) {
                ptr = in_b0 + pb * in_img + ci * in_plane + gy * g.in_w + gx;
                stride = static_cast<int>(kCK * in_plane * 4);
                if ((nstages - 1) * kCK + ci >= g.cin) lastmask |= 1u << sl;
            }
        }
        pp[sl] = ptr;
        pstride[sl] = stride;
    }

    constexpr int kStageTaps = (kCK == stage_channels(kMaxTaps, WAVES)) ? kMaxTaps : (kCK == stage_channels(kFewTaps, WAVES) ? kFewTaps : kVeryFewTaps);
    constexpr int kWSlotsMax = (((KH > 0 ? kTapsAll : kStageTaps) * kCK * 32 * MTP > 32 * 32 * MTP
                                     ? (KH > 0 ? kTapsAll : kStageTaps) * kCK * 32 * MTP : 32 * 32 * MTP) + kWPiece - 1) / kWPiece;

// DMA of stage S (weights + patch) into buffer BUF; advances the patch pointers to stage S+1.
#define BASIC_ISSUE_STAGE(S, BUF)                                                                              \
    do {                                                                                                       \
        float *dstw_ = lds + (BUF) * stage_floats;                                                             \
        const float *srcw_ = wpack + static_cast<int64_t>(S) * wl_floats + tid * 4;                            \
        _Pragma("unroll") for (int sl = 0; sl < kWSlotsMax; ++sl)                                              \
            if (sl < n_wslots)                                                                                 \
                __builtin_amdgcn_global_load_lds((glb_cvoid *)(srcw_ + sl * kWPiece),                          \
                                                 (lds_void *)(dstw_ + sl * kWPiece + wave * 256), 16, 0, 0);   \
        if ((S) == nstages - 1 && lastmask) {                                                                  \
            _Pragma("unroll") for (int sl = 0; sl < kPSlots; ++sl)                                             \
                if ((lastmask >> sl) & 1u) pp[sl] = basic_zero_page;                                           \
        }                                                                                                      \
        float *dstp_ = dstw_ + wl_pad;                                                                         \
        if (g.patch4) {                                                                                        \
            _Pragma("unroll") for (int sl = 0; sl < kPSlots; ++sl)                                             \
                if (sl * kThreads + wave * 64 < patch_units_pad) {                                             \
                    __builtin_amdgcn_global_load_lds((glb_cvoid *)pp[sl], (lds_void *)(dstp_ + (sl * kThreads + wave * 64) * 4), 16, 0, 0); \
                    pp[sl] = reinterpret_cast<const float *>(reinterpret_cast<const char *>(pp[sl]) + pstride[sl]); \
                }                                                                                              \
        } else {                                                                                               \
            _Pragma("unroll") for (int sl = 0; sl < kPSlots; ++sl)                                             \
                if (sl * kThreads + wave * 64 < patch_units_pad) {                                             \
                    __builtin_amdgcn_global_load_lds((glb_cvoid *)pp[sl], (lds_void *)(dstp_ + sl * kThreads + wave * 64), 4, 0, 0); \
                    pp[sl] = reinterpret_cast<const float *>(reinterpret_cast<const char *>(pp[sl]) + pstride[sl]); \
                }                                                                                              \
        }                                                                                                      \
    } while (0)

    if (nstages > 0 && !(g.debug & 1)) BASIC_ISSUE_STAGE(0, 0);
    for (int stg = 0; stg < nstages; ++stg) {
        // my DMAs of this stage have landed (vmcnt(0), part of the barrier's fence) + every wave is past the
        // previous stage, whose buffer the next DMA overwrites
        __syncthreads();
        if (stg + 1 < nstages && !(g.debug & 1)) BASIC_ISSUE_STAGE(stg + 1, (stg + 1) & 1);  // in flight during the MFMAs below
        const float *wl = lds + (stg & 1) * stage_floats;
        const float *patch = wl + wl_pad;
        // ---- MFMA over (tap, channel pair) steps.  The fragment reads of the next step are issued
        // before the MFMAs of the current one, on two alternating register sets (no copies, so the
        // compiler's counted lgkmcnt waits keep the next step's reads in flight while the matrix
        // core works); sched_barriers pin that order against the machine scheduler.
        if (!(g.debug & 2)) {
            constexpr int kPairs = kCK / 2;
            float fa[2][MTP], fb[2];
            if constexpr (KH > 0) {
                // Straight-line stage: every fragment address is lane base + compile-time offset (plus one
                // wave-uniform row/channel term), so a step is one ds_read_b32 (B), one ds_read_b128 (A) and
                // MT MFMAs.  The matrix pipe loses throughput to every other instruction the SIMD issues, so
                // nothing else may sit between the MFMAs.
                constexpr int kSteps = kTapsAll * kPairs;
                const float *a_lane = wl + lane_a_base;
                const float *b_lane = patch + lane_b_base;
                fb[0] = b_lane[0];
                load_a<MTP>(a_lane, fa[0]);
#pragma unroll
                for (int st = 0; st < kSteps; ++st) {
                    const int cur = st & 1, nxt = cur ^ 1;
                    const int sn = (st + 1 < kSteps) ? st + 1 : st;  // at the very end a harmless re-read
                    const int tn = sn / kPairs, cpn = sn % kPairs;
                    // patch row / column of tap tn: the first KH x KW taps are column phase 0, the KH x KWB after them
                    // column phase 1, whose tap grid starts KW - KWB columns to the right
                    const int trow = tn < kTapsA ? tn / KW : (tn - kTapsA) / (KWB > 0 ? KWB : 1);
                    const int tcol = tn < kTapsA ? tn % KW : (tn - kTapsA) % (KWB > 0 ? KWB : 1) + (KW - KWB);
                    fb[nxt] = b_lane[trow * g.pwp + cpn * 2 * chan_stride + tcol];
                    load_a<MTP>(a_lane + (tn * kCK + cpn * 2) * 32 * MTP, fa[nxt]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (KWB == 0 || st / kPairs < kTapsA) {
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m], fb[cur], acc[m], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            acc2[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m], fb[cur], acc2[m], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // Runtime tap table, one step = (tap, channel pair).
                const int nsteps = g.ntaps * kPairs;
                fb[0] = patch[lane_b_base + tapoff[0]];
                load_a<MTP>(wl + lane_a_base, fa[0]);
                for (int st = 0; st < nsteps; st += 2) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {  // two steps per iteration: the register sets alternate statically
                        const int cur = u, nxt = u ^ 1;
                        const int sn = (st + u + 1 < nsteps) ? st + u + 1 : st + u;
                        const int tn = sn / kPairs, cpn = sn % kPairs;
                        fb[nxt] = patch[lane_b_base + tapoff[tn] + cpn * 2 * chan_stride];
                        load_a<MTP>(wl + (tn * kCK + cpn * 2) * 32 * MTP + lane_a_base, fa[nxt]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (st + u < nsteps) {
#pragma unroll
                            for (int m = 0; m < MT; ++m)
                                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m], fb[cur], acc[m], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }

    // ---- epilogue: bias, GDN / IGDN -- on one accumulator set (twice for the fused column phases)
    auto finish = [&](f32x16 (&acc)[MT], bool again) __attribute__((always_inline)) {
    // accumulator register r of tile m, lane (khalf, col): channel 32m + 8(r>>2) + 4 khalf + (r&3)
    if (bias) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {  // registers 4rq..4rq+3 are 4 consecutive channels: one 16-byte LDS read
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(chan_const + 32 * m + 8 * rq + 4 * khalf);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[m][4 * rq + e] += b4[e];
            }
    }

    if (g.act == BASIC_ACT_GDN || g.act == BASIC_ACT_IGDN) {
        // norm[i][pos] = beta[i] + sum_k gammaT[k][i] * x[k][pos]^2 as a second MFMA GEMM.
        f32x16 nrm[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) nrm[m][r] = 0.f;
        // gamma^T is walked 32 k-rows at a time through the two (now free) stage buffers, again by LDS-DMA
        // one chunk ahead of the MFMAs that read it.
#define BASIC_ISSUE_GAMMA(MK)                                                                                   \
    do {                                                                                                       \
        float *dst_ = lds + ((nstages + (MK)) & 1) * stage_floats;                                             \
        const float *src_ = g.gammaT + static_cast<int64_t>(MK) * gam_floats + tid * 4;                        \
        _Pragma("unroll") for (int sl = 0; sl < (32 * 32 * MTP + kWPiece - 1) / kWPiece; ++sl)                 \
            __builtin_amdgcn_global_load_lds((glb_cvoid *)(src_ + sl * kWPiece), (lds_void *)(dst_ + sl * kWPiece + wave * 256), 16, 0, 0); \
    } while (0)
        if (again) __syncthreads();  // every wave is done with the buffers the first pass read last
        BASIC_ISSUE_GAMMA(0);
#pragma unroll
        for (int mk = 0; mk < MT; ++mk) {  // k rows 32mk .. 32mk+31
            __syncthreads();  // chunk mk landed; every wave is done with the buffer the next chunk overwrites
            if (mk + 1 < MT) BASIC_ISSUE_GAMMA(mk + 1);
            const float *wl = lds + ((nstages + mk) & 1) * stage_floats;
            float gk[2][MTP];  // A fragments one step ahead of their MFMAs, as in the main loop
            load_a<MTP>(wl + ((4 * khalf) * 32 + col) * MTP, gk[0]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cur = r & 1, nxt = cur ^ 1;
                const int rn = (r + 1 < 16) ? r + 1 : r;
                load_a<MTP>(wl + ((8 * (rn >> 2) + (rn & 3) + 4 * khalf) * 32 + col) * MTP, gk[nxt]);
                const float x = acc[mk][r];
                const float bfrag = x * x;  // k = 32mk + 8(r>>2) + (r&3) [+4 for lanes 32..63]
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    nrm[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(gk[cur][m], bfrag, nrm[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(chan_const + g.coutp + 32 * m + 8 * rq + 4 * khalf);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * rq + e;
                    const float nv = nrm[m][r] + b4[e];
                    // v_rsq_f32 / v_sqrt_f32 (1 ulp) instead of the ~10-instruction IEEE division / sqrt sequences:
                    // 64 of them per lane would cost as much VALU time as dozens of MFMAs
                    acc[m][r] *= (g.act == BASIC_ACT_GDN) ? __builtin_amdgcn_rsqf(nv) : __builtin_amdgcn_sqrtf(nv);
                }
            }
    }
    };
    finish(acc, false);
    if constexpr (KWB > 0) finish(acc2, true);

    // ---- store
    const int my = my0 + ty, mx = mx0 + tx, b = b0 + tb;
    if (lane_live && my < g.mh && mx < g.mw && b < g.batch) {
        const int oy = my * g.s_out + g.oy0, ox = mx * g.s_out + g.ox0;
        const int64_t plane = static_cast<int64_t>(g.out_h) * g.out_w;
        float *o = g.out + (static_cast<int64_t>(b) * g.out_ctotal + co_base) * plane + static_cast<int64_t>(oy) * g.out_w + ox;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * m + 8 * (r >> 2) + 4 * khalf + (r & 3);
                if (co < cout_here) {
                    if constexpr (KWB > 0) {  // columns ox, ox + 1 (ox even, rows 8-byte aligned: host-checked)
                        f32x2 v2;
                        v2[0] = apply_act(acc[m][r], g.act);
                        v2[1] = apply_act(acc2[m][r], g.act);
                        *reinterpret_cast<f32x2 *>(o + co * plane) = v2;
                    } else {
                        o[co * plane] = apply_act(acc[m][r], g.act);
                    }
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// First analysis layer (5x5 taps, <= 4 input channels, 128 output channels, GDN): its whole reduction
// is one stage, so the generic kernel spends most of a workgroup's life waiting for the DMA of the
// weight slab and of gamma (64 KB per 456 MFMAs of a wave).  Here both stay RESIDENT in LDS and a
// persistent 8-wave workgroup per CU walks the tiles: per tile one patch DMA (double buffered, issued a
// tile ahead), 50 conv steps, 64 GDN steps straight from the resident gamma, one barrier.
// Same packed weights / gamma / TapLaunch as conv_tap_mfma_kernel<4, 4, 5, 5, 8>.
// ---------------------------------------------------------------------------------------------
constexpr int kFirstSlots = 10;  // patch elements per thread (512 threads): <= 5120 floats, checked by the host

// Tiles: workgroup w starts with tile w; further tiles are handed out by `sched` (a device counter the host zeroes
// before the launch), fetched one tile ahead so that the patch DMA of the next tile still overlaps this tile's MFMAs.  A
// static stride (tile += gridDim.x, what runs when sched == nullptr) makes the launch as slow as its LAST workgroup to
// start: with rANS workgroups of another HIP stream holding the LDS of some compute units for milliseconds, the workgroups
// meant for those units start late with their full share of tiles still to do (measured 1.07 -> 2.05 ms at 128 images).
// ROWS (<= 3 input channels: the codec's RGB layer): the reduction runs kernel row by kernel row over (kx, channel) pairs --
// 15 real products + 1 zero-weight pad per row = 80 instead of 25 taps x 4 padded channels = 100, i.e. 40 MFMA steps per
// tile instead of 50 (of 114 with the GDN) -- and the patch is staged pixel-interleaved ([py][px][c], 3 floats per pixel, no
// padding channel: 25 % fewer DMA instructions), so that a row's 16 operands are 16 consecutive floats: every step's B
// read keeps an immediate offset.  Weights come as a second slab [ky * 16 + kx * 3 + c][32][4] (wrow).
template <bool ROWS>
__global__ __launch_bounds__(512, 1) void conv5x5_cin4_gdn_persistent_kernel(const TapLaunch g, int ntiles, int *sched)
{
    extern __shared__ float lds[];
    constexpr int MT = 4, kCK = ROWS ? 3 : 4, KW = 5, kSteps = ROWS ? 5 * 8 : 25 * 2;
    constexpr int kWFloats = ROWS ? 80 * 32 * 4 : 25 * 4 * 32 * 4;  // 10240 / 12800
    constexpr int kGFloats = 128 * 32 * 4;       // 16384
    float *wl = lds;
    float *gl = lds + kWFloats;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: keeps LDS-DMA bases on the scalar unit
    const int khalf = lane >> 5, col = lane & 31;
    const int TB = 1 << g.tb_log, TH = 1 << g.th_log, TW = 1 << g.tw_log;
    const int chan_stride = g.ph * g.pwp;
    const int patch_elems = TB * kCK * chan_stride;   // (ROWS: kCK = 3 floats per pixel)
    const int patch_pad = (patch_elems + 511) / 512 * 512;
    const int n_pslots = patch_pad / 512;
    float *pbuf = gl + kGFloats;  // two patch buffers of patch_pad floats
    float *chan_const = pbuf + 2 * patch_pad;  // [128] bias, [128] beta: read from LDS in every tile's epilogue
    int *next_slot = reinterpret_cast<int *>(chan_const + 256);  // [2]: the tile after next, published one barrier ahead
    if (tid < 128) {
        chan_const[tid] = g.bias ? g.bias[tid] : 0.f;
        chan_const[128 + tid] = g.beta ? g.beta[tid] : 1.f;
    }

    // resident operands: weight slab (6 whole 8 KB pieces + 2 KB) and gamma (8 pieces)
    {
        const float *srcw = (ROWS ? g.wrow : g.wpack) + tid * 4;
#pragma unroll
        for (int sl = 0; sl < kWFloats / 2048; ++sl)
            __builtin_amdgcn_global_load_lds((glb_cvoid *)(srcw + sl * 2048), (lds_void *)(wl + sl * 2048 + wave * 256), 16, 0, 0);
        if (kWFloats % 2048 && wave < (kWFloats % 2048) / 256)
            __builtin_amdgcn_global_load_lds((glb_cvoid *)(srcw + (kWFloats / 2048) * 2048), (lds_void *)(wl + (kWFloats / 2048) * 2048 + wave * 256), 16, 0, 0);
        const float *srcg = g.gammaT + tid * 4;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl)
            __builtin_amdgcn_global_load_lds((glb_cvoid *)(srcg + sl * 2048), (lds_void *)(gl + sl * 2048 + wave * 256), 16, 0, 0);
    }

    // this lane's output position inside a tile
    const int q = wave * 32 + col;
    const int tx = q & (TW - 1);
    const int ty = (q >> g.tw_log) & (TH - 1);
    const int tb_raw = q >> (g.tw_log + g.th_log);
    const bool lane_live = tb_raw < TB;
    const int tb = lane_live ? tb_raw : 0;
    const int row_pitch = ROWS ? 3 * g.pwp : g.pwp;   // floats per patch row
    const int lane_b_base = ROWS ? ((tb * g.ph + ty * g.s_in) * g.pwp + tx * g.s_in) * 3 + khalf
                                 : ((tb * kCK + khalf) * g.ph + ty * g.s_in) * g.pwp + tx * g.s_in;
    const int lane_a_base = (khalf * 32 + col) * 4;

    // tile-independent part of the gather descriptors: patch element tid + 512 s sits at (pb, ci, py, px)
    const int64_t in_plane = static_cast<int64_t>(g.in_h) * g.in_w;
    const int64_t in_img = static_cast<int64_t>(g.cin) * in_plane;
    int poff[kFirstSlots], pyx[kFirstSlots];  // offset from the tile's first element / (pb << 24 | py << 12 | px), -1 = padding
#pragma unroll
    for (int sl = 0; sl < kFirstSlots; ++sl) {
        int r = tid + sl * 512;
        int off = 0, code = -1;
        if (r < patch_elems) {
            int px, py, ci, pb;
            if (ROWS) {   // [pb][py][px][c]
                ci = r % 3; r /= 3;
                px = r % g.pwp; r /= g.pwp;
                py = r % g.ph;
                pb = r / g.ph;
            } else {      // [pb][c][py][px]
                px = r % g.pwp; r /= g.pwp;
                py = r % g.ph; r /= g.ph;
                ci = r % kCK;
                pb = r / kCK;
            }
            if (px < g.pw && ci < g.cin) {
                off = static_cast<int>(pb * in_img + ci * in_plane + py * g.in_w + px);
                code = (pb << 24) | (py << 12) | px;
            }
        }
        poff[sl] = off;
        pyx[sl] = code;
    }

#define BASIC_FIRST_ISSUE_PATCH(TILE, BUF)                                                                     \
    do {                                                                                                       \
        int bid_ = (TILE);                                                                                     \
        const int txi_ = bid_ % g.tiles_x; bid_ /= g.tiles_x;                                                  \
        const int tyi_ = bid_ % g.tiles_y; bid_ /= g.tiles_y;                                                  \
        const int b0_ = bid_ * TB;                                                                             \
        const int gy0_ = tyi_ * TH * g.s_in + g.dymin, gx0_ = txi_ * TW * g.s_in + g.dxmin;                    \
        const float *org_ = g.in + static_cast<int64_t>(b0_) * in_img + static_cast<int64_t>(gy0_) * g.in_w + gx0_; \
        float *dst_ = pbuf + (BUF) * patch_pad + wave * 64;                                                    \
        _Pragma("unroll") for (int sl = 0; sl < kFirstSlots; ++sl)                                             \
            if (sl < n_pslots) {                                                                               \
                const int c_ = pyx[sl];                                                                        \
                const int gy_ = gy0_ + ((c_ >> 12) & 0xFFF), gx_ = gx0_ + (c_ & 0xFFF);                        \
                const bool ok_ = c_ >= 0 && gy_ >= 0 && gy_ < g.in_h && gx_ >= 0 && gx_ < g.in_w && b0_ + (c_ >> 24) < g.batch; \
                const float *src_ = ok_ ? org_ + poff[sl] : basic_zero_page;                                   \
                __builtin_amdgcn_global_load_lds((glb_cvoid *)src_, (lds_void *)(dst_ + sl * 512), 4, 0, 0);   \
            }                                                                                                  \
    } while (0)

    int tile = blockIdx.x;
    if (tile < ntiles) BASIC_FIRST_ISSUE_PATCH(tile, 0);
    if (sched && tid == 0) next_slot[1] = static_cast<int>(gridDim.x) + atomicAdd(sched, 1);   // tile of iteration 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // weights, gamma and the first patch: this wave's DMAs have landed
    int tile_next = tile + static_cast<int>(gridDim.x);
    for (int it = 0; tile < ntiles; tile = tile_next, ++it) {
        // Every wave waited for its own DMAs of this tile's patch BEFORE it issued the previous tile's stores (below), so a
        // bare barrier publishes the patch -- and, unlike __syncthreads(), does not wait for those 64 stores per lane: the
        // HBM-bound store phase of tile i drains underneath the MFMAs of tile i + 1.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // the barrier also published next_slot[(it + 1) & 1], written by thread 0 during the previous iteration
        tile_next = sched ? __builtin_amdgcn_readfirstlane(next_slot[(it + 1) & 1]) : tile + static_cast<int>(gridDim.x);
        if (tile_next < ntiles && !(g.debug & 1)) BASIC_FIRST_ISSUE_PATCH(tile_next, (it + 1) & 1);
        int tile_after = 0;   // thread 0: the tile of iteration it + 2, requested now, published at the end of this iteration
        if (sched && tid == 0) tile_after = static_cast<int>(gridDim.x) + atomicAdd(sched, 1);
        const float *patch = pbuf + (it & 1) * patch_pad;

        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        if (!(g.debug & 2)) {
            float fa[2][4], fb[2];
            const float *a_lane = wl + lane_a_base;
            const float *b_lane = patch + lane_b_base;
            const float *b_row[KW];   // ROWS: this lane's 16 operands of kernel row ky start at b_row[ky] (its khalf folded in)
#pragma unroll
            for (int ky = 0; ky < KW; ++ky) b_row[ky] = b_lane + ky * row_pitch;
            fb[0] = b_lane[0];
            load_a<4>(a_lane, fa[0]);
#pragma unroll
            for (int st = 0; st < kSteps; ++st) {
                const int cur = st & 1, nxt = cur ^ 1;
                const int sn = (st + 1 < kSteps) ? st + 1 : st;
                if constexpr (ROWS) {   // step = (kernel row, pair of its 16 operands)
                    fb[nxt] = b_row[sn / 8][2 * (sn % 8)];
                    if (sn % 8 == 7 && khalf) fb[nxt] = 0.f;   // the row's 16th operand is the pad: whatever sits there (the next pixel) stays out, non-finite or not
                    load_a<4>(a_lane + (sn / 8 * 16 + 2 * (sn % 8)) * 32 * 4, fa[nxt]);
                } else {
                const int tn = sn / 2, cpn = sn % 2;
                fb[nxt] = b_lane[(tn / KW) * g.pwp + cpn * 2 * chan_stride + (tn % KW)];
                load_a<4>(a_lane + (tn * kCK + cpn * 2) * 32 * 4, fa[nxt]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m], fb[cur], acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (g.bias) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(chan_const + 32 * m + 8 * rq + 4 * khalf);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[m][4 * rq + e] += b4[e];
                }
        }
        if ((g.act == BASIC_ACT_GDN || g.act == BASIC_ACT_IGDN) && !(g.debug & 4)) {
            f32x16 nrm[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) nrm[m][r] = 0.f;
            float gk[2][4];
            load_a<4>(gl + ((4 * khalf) * 32 + col) * 4, gk[0]);
#pragma unroll
            for (int st = 0; st < 64; ++st) {  // k = 32 mk + 8 (r >> 2) + (r & 3) [+4 for lanes 32..63], st = 16 mk + r
                const int cur = st & 1, nxt = cur ^ 1;
                const int sn = (st + 1 < 64) ? st + 1 : st;
                const int kn = 32 * (sn >> 4) + 8 * ((sn & 15) >> 2) + (sn & 3);
                load_a<4>(gl + ((kn + 4 * khalf) * 32 + col) * 4, gk[nxt]);
                __builtin_amdgcn_sched_barrier(0);
                const float x = acc[st >> 4][st & 15];
                const float bfrag = x * x;
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    nrm[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(gk[cur][m], bfrag, nrm[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(chan_const + 128 + 32 * m + 8 * rq + 4 * khalf);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * rq + e;
                        const float nv = nrm[m][r] + b4[e];
                        acc[m][r] *= (g.act == BASIC_ACT_GDN) ? __builtin_amdgcn_rsqf(nv) : __builtin_amdgcn_sqrtf(nv);
                    }
                }
        }
        // ---- store (HBM-bound: 2 GB of fp32 activations leave this layer at B = 256)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next patch's DMAs (issued a whole tile ago) and older stores
        int bid = tile;
        const int tx_i = bid % g.tiles_x; bid /= g.tiles_x;
        const int ty_i = bid % g.tiles_y; bid /= g.tiles_y;
        const int my = ty_i * TH + ty, mx = tx_i * TW + tx, b = bid * TB + tb;
        if (lane_live && my < g.mh && mx < g.mw && b < g.batch && !(g.debug & 32)) {
            const int oy = my * g.s_out + g.oy0, ox = mx * g.s_out + g.ox0;
            const int64_t plane = static_cast<int64_t>(g.out_h) * g.out_w;
            // channels of (m, rq, e) are 32 m + 8 rq + 4 khalf + e: one running pointer instead of 64 address products
            float *oc = g.out + (static_cast<int64_t>(b) * g.out_ctotal + g.co_base + 4 * khalf) * plane + static_cast<int64_t>(oy) * g.out_w + ox;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        *oc = apply_act(acc[m][4 * rq + e], g.act);
                        oc += plane;
                    }
                    oc += 4 * plane;
                }
        }
        // slot it & 1 was last read right after THIS iteration's barrier: free to take the tile of iteration it + 2
        if (sched && tid == 0) next_slot[it & 1] = tile_after;
    }
#undef BASIC_FIRST_ISSUE_PATCH
}

// ---------------------------------------------------------------------------------------------
// Last synthesis layer (ConvTranspose2d 5x5, stride 2, pad 2, output_padding 1, Cout <= 4):
// three output channels would waste 29/32 of every MFMA tile, and on gfx950 the fp32 matrix rate
// equals the fp32 vector rate anyway, so this layer runs on the VALU.  One lane owns one INPUT
// position and produces its 2x2 output pixels x Cout: per input channel 9 LDS reads (the 3x3
// neighbourhood) feed all 25 taps x Cout FMAs, with the weights as wave-uniform scalar operands.
// ---------------------------------------------------------------------------------------------
constexpr int kSmTileH = 16;              // input positions per workgroup: 16 rows x 64 columns,
constexpr int kSmTileW = 64;              // one lane = a 1 x 4 strip -> 2 x 8 output pixels x Cout
constexpr int kSmCK = 8;                  // input channels per LDS stage
constexpr int kSmPH = kSmTileH + 2;       // +-1 halo
constexpr int kSmPW = kSmTileW + 2;
constexpr int kSmPitch = 68;              // row pitch in floats (16-byte multiple: aligned ds_read_b128)
constexpr int kSmWRow = 80;               // packed weights per input channel: 25 taps x 3, padded to 5 x 16

struct SmallLaunch {
    const float *in;      // [B][cin][H][W]
    float *out;           // [B][cout][2H][2W]
    const float *wsm;     // [cin][80]   W[ci][co][ky][kx] -> [ci][(ky*5+kx)*3 + co], zero padded
    const float *bias;    // [4]
    int batch, cin, cout, in_h, in_w, tiles_y, tiles_x, act;
};

typedef float f32x16s __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(4))) f32x16s cf32x16s;

// The 75 weights of input channel C as wave-uniform scalars (5 x s_load_dwordx16) times the pair table pr[3][5]
// of a lane's strip: 150 packed FMAs into acc[pair][py][px][co].
#define BASIC_SM_ACCUMULATE(C)                                                                                 \
    do {                                                                                                       \
        /* constant address space: keeps these wave-uniform loads on the scalar unit (s_load_dwordx16) even */ \
        /* after the LDS-DMA intrinsic, which the compiler treats as a store that could alias them          */ \
        const cf32x16s *w16 = (const cf32x16s *)(g.wsm + static_cast<int64_t>(C) * kSmWRow);                   \
        const f32x16s w0 = w16[0], w1 = w16[1], w2 = w16[2], w3 = w16[3], w4 = w16[4];                         \
        _Pragma("unroll") for (int ky = 0; ky < 5; ++ky)                                                       \
        _Pragma("unroll") for (int kx = 0; kx < 5; ++kx) {                                                     \
            /* output row 2*my + py receives input row my + dy through ky = py + 2 - 2*dy */                   \
            const int py = ky & 1, px = kx & 1;                                                                \
            const int dy = (py + 2 - ky) / 2, dx = (px + 2 - kx) / 2; /* in {-1, 0, 1} */                      \
            _Pragma("unroll") for (int co = 0; co < 3; ++co) {                                                 \
                const int wi = (ky * 5 + kx) * 3 + co;                                                         \
                const float w = wi < 16 ? w0[wi & 15] : wi < 32 ? w1[wi & 15] : wi < 48 ? w2[wi & 15] : wi < 64 ? w3[wi & 15] : w4[wi & 15]; \
                const f32x2 ww = {w, w};                                                                       \
                _Pragma("unroll") for (int qp = 0; qp < 2; ++qp)                                               \
                    acc[qp][py][px][co] = __builtin_elementwise_fma(pr[dy + 1][2 * qp + dx + 1], ww, acc[qp][py][px][co]); \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)

__global__ __launch_bounds__(256) void deconv5s2_cout3_kernel(const SmallLaunch g)
{
    __shared__ __attribute__((aligned(16))) float patch[kSmCK][kSmPH][kSmPitch];
    const int tid = threadIdx.x;
    int bid = xcd_tile(blockIdx.x, gridDim.x);
    const int tx_i = bid % g.tiles_x; bid /= g.tiles_x;
    const int ty_i = bid % g.tiles_y; bid /= g.tiles_y;
    const int b = bid;
    const int lxq = tid & 15, ly = tid >> 4;
    const int my = ty_i * kSmTileH + ly, mx0 = tx_i * kSmTileW + 4 * lxq;
    const int64_t plane = static_cast<int64_t>(g.in_h) * g.in_w;
    const float *inb = g.in + static_cast<int64_t>(b) * g.cin * plane;

    f32x2 acc[2][2][2][3];  // [pair of strip positions][py][px][co]
#pragma unroll
    for (int qp = 0; qp < 2; ++qp)
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int co = 0; co < 3; ++co) acc[qp][py][px][co] = f32x2{0.f, 0.f};

    for (int c0 = 0; c0 < g.cin; c0 += kSmCK) {
        __syncthreads();
        for (int i = tid; i < kSmCK * kSmPH * kSmPW; i += 256) {
            const int px = i % kSmPW, r = i / kSmPW;
            const int py = r % kSmPH, ci = r / kSmPH;
            const int gy = ty_i * kSmTileH - 1 + py, gx = tx_i * kSmTileW - 1 + px, c = c0 + ci;
            float v = 0.f;
            if (gy >= 0 && gy < g.in_h && gx >= 0 && gx < g.in_w && c < g.cin) v = inb[c * plane + gy * g.in_w + gx];
            patch[ci][py][px] = v;
        }
        __syncthreads();
        const int cmax = (g.cin - c0 < kSmCK) ? g.cin - c0 : kSmCK;
        for (int ci = 0; ci < cmax; ++ci) {
            // 3 x 6 neighbourhood of the strip as the five overlapping pairs (v0 v1) (v1 v2) ... (v4 v5) per row:
            // every FMA below is then a packed one on two neighbouring strip positions, with the weight as a
            // broadcast scalar operand and no register shuffling.
            f32x2 pr[3][5];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float *row = &patch[ci][ly + a][4 * lxq];
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(row);
                const f32x2 hi = *reinterpret_cast<const f32x2 *>(row + 4);
                pr[a][0] = f32x2{lo[0], lo[1]};
                pr[a][2] = f32x2{lo[2], lo[3]};
                pr[a][4] = hi;
                pr[a][1] = f32x2{row[1], row[2]};  // odd pairs: ds_read2_b32
                pr[a][3] = f32x2{row[3], row[4]};
            }
            BASIC_SM_ACCUMULATE(c0 + ci);
        }
    }
    if (my < g.in_h) {
        const int oh = 2 * g.in_h, ow = 2 * g.in_w;
#pragma unroll
        for (int co = 0; co < 3; ++co) {
            if (co >= g.cout) break;
            const float bv = g.bias[co];
            float *o = g.out + (static_cast<int64_t>(b) * g.cout + co) * oh * ow + static_cast<int64_t>(2 * my) * ow + 2 * mx0;
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (mx0 + q >= g.in_w) break;
                    float2 r;
                    r.x = apply_act(acc[q >> 1][py][0][co][q & 1] + bv, g.act);
                    r.y = apply_act(acc[q >> 1][py][1][co][q & 1] + bv, g.act);
                    *reinterpret_cast<float2 *>(o + py * ow + 2 * q) = r;
                }
        }
    }
}

// The same layer with LDS-DMA staging (needs in_w % 4 == 0): an LDS row is the 16-byte aligned global span
// [x0 - 4, x0 + 68) of an input row -- 18 chunks that are each entirely inside or entirely outside the image --
// so a stage is a lane-linear image of 16-byte pieces, double buffered one stage ahead of the FMAs.
constexpr int kSmDCK = 4;                                   // input channels per stage
constexpr int kSmDRow = 18;                                 // 16-byte chunks per LDS row (pitch 72 floats)
constexpr int kSmDChunks = kSmDCK * kSmPH * kSmDRow;        // chunks per stage
constexpr int kSmDSlots = (kSmDChunks + 255) / 256;         // DMA instructions per thread and stage
constexpr int kSmDStage = kSmDSlots * 256 * 4;              // floats per stage buffer (whole instructions)

__global__ __launch_bounds__(256) void deconv5s2_cout3_dma_kernel(const SmallLaunch g)
{
    __shared__ __attribute__((aligned(16))) float buf[2 * kSmDStage];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: LDS-DMA bases stay scalar
    int bid = xcd_tile(blockIdx.x, gridDim.x);
    const int tx_i = bid % g.tiles_x; bid /= g.tiles_x;
    const int ty_i = bid % g.tiles_y; bid /= g.tiles_y;
    const int b = bid;
    const int lxq = tid & 15, ly = tid >> 4;
    const int my = ty_i * kSmTileH + ly, mx0 = tx_i * kSmTileW + 4 * lxq;
    const int64_t plane = static_cast<int64_t>(g.in_h) * g.in_w;
    const float *inb = g.in + static_cast<int64_t>(b) * g.cin * plane;
    const int nstages = (g.cin + kSmDCK - 1) / kSmDCK;

    // gather descriptors (see conv_tap_mfma_kernel): chunk k = tid + 256 s of every stage comes from pp[s]
    const float *pp[kSmDSlots];
    int pstride[kSmDSlots];
    unsigned lastmask = 0;
#pragma unroll
    for (int sl = 0; sl < kSmDSlots; ++sl) {
        const int k = tid + sl * 256;
        const float *ptr = basic_zero_page;
        int stride = 0;
        if (k < kSmDChunks) {
            const int j = k % kSmDRow, r = k / kSmDRow;
            const int py = r % kSmPH, ci = r / kSmPH;
            const int gy = ty_i * kSmTileH - 1 + py, gx = tx_i * kSmTileW - 4 + 4 * j;
            if (gy >= 0 && gy < g.in_h && gx >= 0 && gx < g.in_w && ci < g.cin) {
                ptr = inb + ci * plane + static_cast<int64_t>(gy) * g.in_w + gx;
                stride = static_cast<int>(kSmDCK * plane * 4);
                if ((nstages - 1) * kSmDCK + ci >= g.cin) lastmask |= 1u << sl;
            }
        }
        pp[sl] = ptr;
        pstride[sl] = stride;
    }
#define BASIC_SM_ISSUE(S)                                                                                      \
    do {                                                                                                       \
        if ((S) == nstages - 1 && lastmask) {                                                                  \
            _Pragma("unroll") for (int sl = 0; sl < kSmDSlots; ++sl)                                           \
                if ((lastmask >> sl) & 1u) pp[sl] = basic_zero_page;                                           \
        }                                                                                                      \
        float *dst_ = buf + ((S) & 1) * kSmDStage + wave * 256;                                                \
        _Pragma("unroll") for (int sl = 0; sl < kSmDSlots; ++sl) {                                             \
            __builtin_amdgcn_global_load_lds((glb_cvoid *)pp[sl], (lds_void *)(dst_ + sl * 1024), 16, 0, 0);   \
            pp[sl] = reinterpret_cast<const float *>(reinterpret_cast<const char *>(pp[sl]) + pstride[sl]);    \
        }                                                                                                      \
    } while (0)

    f32x2 acc[2][2][2][3];  // [pair of strip positions][py][px][co]
#pragma unroll
    for (int qp = 0; qp < 2; ++qp)
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int co = 0; co < 3; ++co) acc[qp][py][px][co] = f32x2{0.f, 0.f};

    BASIC_SM_ISSUE(0);
    for (int stg = 0; stg < nstages; ++stg) {
        __syncthreads();  // this stage has landed; every wave is done with the buffer the next DMA overwrites
        if (stg + 1 < nstages) BASIC_SM_ISSUE(stg + 1);
        const float *st = buf + (stg & 1) * kSmDStage;
        const int c0 = stg * kSmDCK;
        const int cmax = (g.cin - c0 < kSmDCK) ? g.cin - c0 : kSmDCK;
        for (int ci = 0; ci < cmax; ++ci) {
            // the strip's 3 x 6 neighbourhood starts one float before chunk lxq + 1 of rows ly .. ly + 2
            f32x2 pr[3][5];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float *row = st + ((ci * kSmPH + ly + a) * kSmDRow + lxq) * 4 + 3;
                const float v0 = row[0], v5 = row[5];
                const f32x4 mid = *reinterpret_cast<const f32x4 *>(row + 1);
                pr[a][0] = f32x2{v0, mid[0]};
                pr[a][1] = f32x2{mid[0], mid[1]};
                pr[a][2] = f32x2{mid[1], mid[2]};
                pr[a][3] = f32x2{mid[2], mid[3]};
                pr[a][4] = f32x2{mid[3], v5};
            }
            BASIC_SM_ACCUMULATE(c0 + ci);
        }
    }
#undef BASIC_SM_ISSUE
    if (my < g.in_h) {
        const int oh = 2 * g.in_h, ow = 2 * g.in_w;
#pragma unroll
        for (int co = 0; co < 3; ++co) {
            if (co >= g.cout) break;
            const float bv = g.bias[co];
            float *o = g.out + (static_cast<int64_t>(b) * g.cout + co) * oh * ow + static_cast<int64_t>(2 * my) * ow + 2 * mx0;
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (mx0 + q >= g.in_w) break;
                    float2 r;
                    r.x = apply_act(acc[q >> 1][py][0][co][q & 1] + bv, g.act);
                    r.y = apply_act(acc[q >> 1][py][1][co][q & 1] + bv, g.act);
                    *reinterpret_cast<float2 *>(o + py * ow + 2 * q) = r;
                }
        }
    }
}

// The same layer with TWO vertically adjacent strips per lane (a 2 x 4 block of input positions -> 4 x 8 output pixels x Cout):
// a channel's 75 wave-uniform weights now feed 300 packed FMAs per lane instead of 150 -- the scalar loads per FMA halve (the
// 40 KB of weights do not fit the scalar cache: counters of round 2 showed half of those loads missing) -- and the two strips
// share two of their four patch rows (24 LDS values per 8 positions instead of 36).  Same FMA order per output as the kernel
// above: identical results.  Tile = 32 x 64 input positions per workgroup.
constexpr int kSm2TileH = 32;
constexpr int kSm2PH = kSm2TileH + 2;
constexpr int kSm2CK = 2;                                    // input channels per stage
constexpr int kSm2Chunks = kSm2CK * kSm2PH * kSmDRow;        // 16-byte chunks per stage
constexpr int kSm2Slots = (kSm2Chunks + 255) / 256;
constexpr int kSm2Stage = kSm2Slots * 256 * 4;

__global__ __launch_bounds__(256) void deconv5s2_cout3_dma2_kernel(const SmallLaunch g)
{
    __shared__ __attribute__((aligned(16))) float buf[2 * kSm2Stage];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = xcd_tile(blockIdx.x, gridDim.x);
    const int tx_i = bid % g.tiles_x; bid /= g.tiles_x;
    const int ty_i = bid % g.tiles_y; bid /= g.tiles_y;
    const int b = bid;
    const int lxq = tid & 15, ly = tid >> 4;
    const int my = ty_i * kSm2TileH + 2 * ly, mx0 = tx_i * kSmTileW + 4 * lxq;
    const int64_t plane = static_cast<int64_t>(g.in_h) * g.in_w;
    const float *inb = g.in + static_cast<int64_t>(b) * g.cin * plane;
    const int nstages = (g.cin + kSm2CK - 1) / kSm2CK;

    const float *pp[kSm2Slots];
    int pstride[kSm2Slots];
    unsigned lastmask = 0;
#pragma unroll
    for (int sl = 0; sl < kSm2Slots; ++sl) {
        const int k = tid + sl * 256;
        const float *ptr = basic_zero_page;
        int stride = 0;
        if (k < kSm2Chunks) {
            const int j = k % kSmDRow, r = k / kSmDRow;
            const int py = r % kSm2PH, ci = r / kSm2PH;
            const int gy = ty_i * kSm2TileH - 1 + py, gx = tx_i * kSmTileW - 4 + 4 * j;
            if (gy >= 0 && gy < g.in_h && gx >= 0 && gx < g.in_w && ci < g.cin) {
                ptr = inb + ci * plane + static_cast<int64_t>(gy) * g.in_w + gx;
                stride = static_cast<int>(kSm2CK * plane * 4);
                if ((nstages - 1) * kSm2CK + ci >= g.cin) lastmask |= 1u << sl;
            }
        }
        pp[sl] = ptr;
        pstride[sl] = stride;
    }
#define BASIC_SM2_ISSUE(S)                                                                                     \
    do {                                                                                                       \
        if ((S) == nstages - 1 && lastmask) {                                                                  \
            _Pragma("unroll") for (int sl = 0; sl < kSm2Slots; ++sl)                                           \
                if ((lastmask >> sl) & 1u) pp[sl] = basic_zero_page;                                           \
        }                                                                                                      \
        float *dst_ = buf + ((S) & 1) * kSm2Stage + wave * 256;                                                \
        _Pragma("unroll") for (int sl = 0; sl < kSm2Slots; ++sl) {                                             \
            __builtin_amdgcn_global_load_lds((glb_cvoid *)pp[sl], (lds_void *)(dst_ + sl * 1024), 16, 0, 0);   \
            pp[sl] = reinterpret_cast<const float *>(reinterpret_cast<const char *>(pp[sl]) + pstride[sl]);    \
        }                                                                                                      \
    } while (0)

    f32x2 acc[2][2][2][2][3];  // [strip row][pair of strip positions][py][px][co]
#pragma unroll
    for (int sr = 0; sr < 2; ++sr)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp)
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px)
#pragma unroll
                    for (int co = 0; co < 3; ++co) acc[sr][qp][py][px][co] = f32x2{0.f, 0.f};

    BASIC_SM2_ISSUE(0);
    for (int stg = 0; stg < nstages; ++stg) {
        __syncthreads();
        if (stg + 1 < nstages) BASIC_SM2_ISSUE(stg + 1);
        const float *st = buf + (stg & 1) * kSm2Stage;
        const int c0 = stg * kSm2CK;
        const int cmax = (g.cin - c0 < kSm2CK) ? g.cin - c0 : kSm2CK;
        for (int ci = 0; ci < cmax; ++ci) {
            // rows 2 ly .. 2 ly + 3 of the patch = rows my - 1 .. my + 2 of the image: strip row sr uses patch rows sr .. sr + 2
            f32x2 pr[4][5];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float *row = st + ((ci * kSm2PH + 2 * ly + a) * kSmDRow + lxq) * 4 + 3;
                const float v0 = row[0], v5 = row[5];
                const f32x4 mid = *reinterpret_cast<const f32x4 *>(row + 1);
                pr[a][0] = f32x2{v0, mid[0]};
                pr[a][1] = f32x2{mid[0], mid[1]};
                pr[a][2] = f32x2{mid[1], mid[2]};
                pr[a][3] = f32x2{mid[2], mid[3]};
                pr[a][4] = f32x2{mid[3], v5};
            }
            const cf32x16s *w16 = (const cf32x16s *)(g.wsm + static_cast<int64_t>(c0 + ci) * kSmWRow);
            const f32x16s w0 = w16[0], w1 = w16[1], w2 = w16[2], w3 = w16[3], w4 = w16[4];
#pragma unroll
            for (int ky = 0; ky < 5; ++ky)
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    const int py = ky & 1, px = kx & 1;
                    const int dy = (py + 2 - ky) / 2, dx = (px + 2 - kx) / 2;
#pragma unroll
                    for (int co = 0; co < 3; ++co) {
                        const int wi = (ky * 5 + kx) * 3 + co;
                        const float w = wi < 16 ? w0[wi & 15] : wi < 32 ? w1[wi & 15] : wi < 48 ? w2[wi & 15] : wi < 64 ? w3[wi & 15] : w4[wi & 15];
                        const f32x2 ww = {w, w};
#pragma unroll
                        for (int sr = 0; sr < 2; ++sr)
#pragma unroll
                            for (int qp = 0; qp < 2; ++qp)
                                acc[sr][qp][py][px][co] = __builtin_elementwise_fma(pr[sr + dy + 1][2 * qp + dx + 1], ww, acc[sr][qp][py][px][co]);
                    }
                }
        }
    }
#undef BASIC_SM2_ISSUE
    const int oh = 2 * g.in_h, ow = 2 * g.in_w;
#pragma unroll
    for (int sr = 0; sr < 2; ++sr) {
        if (my + sr >= g.in_h) break;
#pragma unroll
        for (int co = 0; co < 3; ++co) {
            if (co >= g.cout) break;
            const float bv = g.bias[co];
            float *o = g.out + (static_cast<int64_t>(b) * g.cout + co) * oh * ow + static_cast<int64_t>(2 * (my + sr)) * ow + 2 * mx0;
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (mx0 + q >= g.in_w) break;
                    float2 r;
                    r.x = apply_act(acc[sr][q >> 1][py][0][co][q & 1] + bv, g.act);
                    r.y = apply_act(acc[sr][q >> 1][py][1][co][q & 1] + bv, g.act);
                    *reinterpret_cast<float2 *>(o + py * ow + 2 * q) = r;
                }
        }
    }
}

struct Phase {
    int ntaps = 0, dymin = 0, dxmin = 0, span_y = 1, span_x = 1;
    int oy0 = 0, ox0 = 0;
    int ck = 2, cin_pad = 0;        // channels per LDS stage of this launch, cin rounded up to it
    int kh = 0, kw = 0;             // the taps form a dense kh x kw grid, tap t at (t / kw, t % kw)
    int kwb = 0;                    // > 0: fused column phases -- kh x kwb more taps (output column ox0 + 1) follow the kh x kw
    int waves = 4;                  // wavefronts per workgroup of this launch (32 positions each)
    signed char dy[kMaxTaps], dx[kMaxTaps];
    float *d_wpack = nullptr;
    float *d_wrow = nullptr;    // see TapLaunch::wrow
    int64_t split_wstride = 0;  // floats per output-channel slice of d_wpack
};

struct Chunk {  // <= 192 output channels handled by one launch family
    int co0 = 0, cout = 0, coutp = 0, mt = 0;
    int nsplit = 1;  // > 1: cout is cut into nsplit slices of coutp channels over gridDim.y
    std::vector<Phase> phases;
    std::vector<Phase> fused;  // stride-2 transposed convolutions: the column phases of each row phase as ONE launch (may be empty)
    float *d_bias = nullptr;
};

}  // namespace

struct basic_conv_plan {
    int cin = 0, cout = 0, ksize = 0, stride = 1, padding = 0, output_padding = 0, transposed = 0, act = 0;
    int s_in = 1, s_out = 1;
    std::vector<Chunk> chunks;
    std::vector<Chunk> split;  // the same layer as 32-channel slices over gridDim.y (small position grids), may be empty
    float *d_gammaT = nullptr, *d_beta = nullptr;
    float *d_wsm = nullptr, *d_bias4 = nullptr;  // VALU path of the Cout <= 4 synthesis output layer
    int *d_sched = nullptr;                      // tile counter of the persistent first-layer kernel (allocated on first use)
};

extern "C" void basic_conv_plan_destroy(basic_conv_plan *p)
{
    if (!p) return;
    for (auto *list : {&p->chunks, &p->split})
        for (auto &ch : *list) {
            for (auto &ph : ch.phases) {
                if (ph.d_wpack) (void)hipFree(ph.d_wpack);
                if (ph.d_wrow) (void)hipFree(ph.d_wrow);
            }
            for (auto &ph : ch.fused)
                if (ph.d_wpack) (void)hipFree(ph.d_wpack);
            if (ch.d_bias) (void)hipFree(ch.d_bias);
        }
    if (p->d_sched) (void)hipFree(p->d_sched);
    if (p->d_wsm) (void)hipFree(p->d_wsm);
    if (p->d_bias4) (void)hipFree(p->d_bias4);
    if (p->d_gammaT) (void)hipFree(p->d_gammaT);
    if (p->d_beta) (void)hipFree(p->d_beta);
    delete p;
}

namespace {

int upload(const std::vector<float> &h, float **d)
{
    BASIC_HIP_TRY(hipMalloc(d, h.size() * sizeof(float)));
    BASIC_HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return BASIC_OK;
}

// Packs the weights of every (output-channel chunk, sub-pixel phase) launch.  per_chunk == 0: balanced chunks of
// <= 192 channels, one launch family each.  per_chunk == 32: ONE Chunk whose phases hold the packs of all
// 32-channel slices back to back (slice stride in Phase::split_wstride), launched with gridDim.y = slices.
int build_chunks(basic_conv_plan *p, const float *weight, const float *bias, int cin, int cout, int slice,
                 std::vector<Chunk> *out)
{
    const int ci_n = p->cin, co_n = p->cout, ksize = p->ksize, stride = p->stride, padding = p->padding;
    const bool transposed = p->transposed != 0;
    const int nph = transposed ? stride : 1;
    const int n_chunks = slice ? 1 : (co_n + kMaxCoutPerLaunch - 1) / kMaxCoutPerLaunch;
    const int per_chunk = slice ? co_n : ((co_n + n_chunks - 1) / n_chunks + 31) / 32 * 32;  // balanced, whole M-tiles
    for (int co0 = 0; co0 < co_n; co0 += per_chunk) {
        Chunk ch;
        ch.co0 = co0;
        ch.cout = (co_n - co0 < per_chunk) ? co_n - co0 : per_chunk;
        ch.mt = slice ? slice / 32 : (ch.cout + 31) / 32;
        ch.coutp = ch.mt * 32;
        ch.nsplit = slice ? (ch.cout + slice - 1) / slice : 1;
        for (int py = 0; py < nph; ++py)
            for (int px = 0; px < nph; ++px) {
                Phase ph;
                ph.oy0 = py; ph.ox0 = px;
                std::vector<int> kys, kxs, dys, dxs;
                for (int k = 0; k < ksize; ++k) {
                    if (!transposed) { kys.push_back(k); dys.push_back(k - padding); }
                    else if ((py + padding - k) % stride == 0) { kys.push_back(k); dys.push_back((py + padding - k) / stride); }
                }
                for (int k = 0; k < ksize; ++k) {
                    if (!transposed) { kxs.push_back(k); dxs.push_back(k - padding); }
                    else if ((px + padding - k) % stride == 0) { kxs.push_back(k); dxs.push_back((px + padding - k) / stride); }
                }
                if (kys.empty() || kxs.empty()) {  // phase receives only the bias
                    ph.ntaps = 0;
                    ph.waves = plan_waves(ch.mt, 0);
                    ph.ck = stage_channels(0, ph.waves);  // must match the instantiation launch_mt picks (LDS layout)
                    ch.phases.push_back(ph);
                    continue;
                }
                int dymin = dys[0], dymax = dys[0], dxmin = dxs[0], dxmax = dxs[0];
                for (int v : dys) { dymin = v < dymin ? v : dymin; dymax = v > dymax ? v : dymax; }
                for (int v : dxs) { dxmin = v < dxmin ? v : dxmin; dxmax = v > dxmax ? v : dxmax; }
                ph.dymin = dymin; ph.dxmin = dxmin; ph.span_y = dymax - dymin + 1; ph.span_x = dxmax - dxmin + 1;
                // taps sorted by patch position: tap t reads patch offset (t / kw, t % kw)
                ph.kh = ph.span_y; ph.kw = ph.span_x;
                std::vector<std::pair<int, int>> taps(static_cast<size_t>(ph.kh) * ph.kw, {-1, -1});  // (ky, kx)
                for (size_t a = 0; a < kys.size(); ++a)
                    for (size_t b = 0; b < kxs.size(); ++b) {
                        const int t = (dys[a] - dymin) * ph.kw + (dxs[b] - dxmin);
                        taps[t] = {kys[a], kxs[b]};
                    }
                ph.ntaps = ph.kh * ph.kw;
                for (int t = 0; t < ph.ntaps; ++t) {
                    ph.dy[t] = static_cast<signed char>(t / ph.kw);
                    ph.dx[t] = static_cast<signed char>(t % ph.kw);
                }
                // ---- pack weights: [slice][cin_pad/CK][ntaps][CK][32][MTP]
                ph.waves = plan_waves(ch.mt, ph.ntaps);
                const int kCK = stage_channels(ph.ntaps, ph.waves);
                const int mtp = mtile_pitch(ch.mt);
                ph.ck = kCK;
                ph.cin_pad = (ci_n + kCK - 1) / kCK * kCK;
                ph.split_wstride = static_cast<int64_t>(ph.cin_pad) * ph.ntaps * 32 * mtp;
                std::vector<float> wp(static_cast<size_t>(ph.split_wstride) * ch.nsplit, 0.f);
                for (int c = 0; c < ci_n; ++c)
                    for (int t = 0; t < ph.ntaps; ++t) {
                        const int ky = taps[t].first, kx = taps[t].second;
                        if (ky < 0) continue;  // a hole in the grid keeps zero weights
                        for (int o = 0; o < ch.cout; ++o) {
                            const int og = co0 + o, ol = o % ch.coutp;
                            const float w = transposed
                                ? weight[((static_cast<size_t>(c) * cout + og) * ksize + ky) * ksize + kx]
                                : weight[((static_cast<size_t>(og) * cin + c) * ksize + ky) * ksize + kx];
                            wp[static_cast<size_t>(o / ch.coutp) * ph.split_wstride +
                               (((static_cast<size_t>(c / kCK) * ph.ntaps + t) * kCK + (c % kCK)) * 32 + ol % 32) * mtp + ol / 32] = w;
                        }
                    }
                wp.resize(wp.size() + 2048, 0.f);  // the DMA copies whole 4 / 8 KB pieces and may read past the last stage
                int rc = upload(wp, &ph.d_wpack);
                // the first analysis layer's row slab (conv5x5_cin4_gdn_persistent_kernel<true>): [ky * 16 + kx * 3 + c][32][4]
                if (!rc && !transposed && ph.kh == 5 && ph.kw == 5 && ci_n <= 3 && ch.cout == 128 && ch.mt == 4 && ch.nsplit == 1) {
                    std::vector<float> wr(static_cast<size_t>(80) * 32 * 4 + 2048, 0.f);
                    for (int c = 0; c < ci_n; ++c)
                        for (int t = 0; t < ph.ntaps; ++t) {
                            const int ky = taps[t].first, kx = taps[t].second;
                            if (ky < 0) continue;
                            const int k = (t / ph.kw) * 16 + (t % ph.kw) * 3 + c;   // patch row t / kw, patch column t % kw
                            for (int o = 0; o < ch.cout; ++o)
                                wr[(static_cast<size_t>(k) * 32 + o % 32) * 4 + o / 32] =
                                    weight[((static_cast<size_t>(co0 + o) * cin + c) * ksize + ky) * ksize + kx];
                        }
                    rc = upload(wr, &ph.d_wrow);
                }
                ch.phases.push_back(ph);
                if (rc) { out->push_back(ch); return rc; }
            }
        // Fused column phases (see conv_tap_mfma_kernel, KWB): for every row phase py, the launch (py, px = 0) takes the
        // taps of (py, px = 1) after its own.  Needs the second grid inside the first (same rows, columns shifted right
        // by kw - kwb) -- true for the codec's k5 s2 p2 transposed convolutions -- and at most 4 accumulator tiles.
        if (transposed && stride == 2 && ch.mt <= 4 && ch.phases.size() == 4) {
            bool ok = true;
            std::vector<Phase> fused;
            for (int py = 0; py < 2 && ok; ++py) {
                const Phase &a = ch.phases[py * 2], &b = ch.phases[py * 2 + 1];
                ok = a.ntaps > 0 && b.ntaps > 0 && a.kh == b.kh && a.dymin == b.dymin && b.kw <= a.kw &&
                     b.dxmin - a.dxmin == a.kw - b.kw && a.ox0 == 0 && b.ox0 == 1 && a.cin_pad % kFusedCK == 0 &&
                     ((a.kh == 3 && a.kw == 3 && b.kw == 2) || (a.kh == 2 && a.kw == 3 && b.kw == 2));
                if (!ok) break;
                Phase f = a;
                f.kwb = b.kw;
                f.ntaps = a.ntaps + b.ntaps;
                f.ck = kFusedCK;
                f.waves = 4;
                f.cin_pad = (ci_n + kFusedCK - 1) / kFusedCK * kFusedCK;
                f.d_wpack = nullptr;
                const int mtp = mtile_pitch(ch.mt);
                f.split_wstride = static_cast<int64_t>(f.cin_pad) * f.ntaps * 32 * mtp;
                std::vector<float> wp(static_cast<size_t>(f.split_wstride) * ch.nsplit, 0.f);  // [slice][stage][tap][CK][32][MTP]
                // taps of a phase sit at grid position t = (dy - dymin) * kw + (dx - dxmin); recover (ky, kx) from the geometry
                auto tap_weight = [&](int phase_px, int t, int kw_phase, int dymin_p, int dxmin_p, int c, int og) -> float {
                    const int dyv = dymin_p + t / kw_phase, dxv = dxmin_p + t % kw_phase;
                    const int ky = py + padding - dyv * stride, kx = phase_px + padding - dxv * stride;
                    if (ky < 0 || ky >= ksize || kx < 0 || kx >= ksize) return 0.f;
                    return weight[((static_cast<size_t>(c) * cout + og) * ksize + ky) * ksize + kx];
                };
                for (int c = 0; c < ci_n; ++c)
                    for (int t = 0; t < f.ntaps; ++t)
                        for (int o = 0; o < ch.cout; ++o) {
                            const float w = t < a.ntaps ? tap_weight(0, t, a.kw, a.dymin, a.dxmin, c, co0 + o)
                                                        : tap_weight(1, t - a.ntaps, b.kw, b.dymin, b.dxmin, c, co0 + o);
                            const int ol = o % ch.coutp;
                            wp[static_cast<size_t>(o / ch.coutp) * f.split_wstride +
                               (((static_cast<size_t>(c / kFusedCK) * f.ntaps + t) * kFusedCK + (c % kFusedCK)) * 32 + ol % 32) * mtp + ol / 32] = w;
                        }
                wp.resize(wp.size() + 2048, 0.f);
                const int rcf = upload(wp, &f.d_wpack);
                fused.push_back(f);
                if (rcf) { ok = false; }
            }
            if (ok) ch.fused = fused;
            else for (auto &f : fused) if (f.d_wpack) (void)hipFree(f.d_wpack);
        }
        std::vector<float> hb(static_cast<size_t>(ch.coutp) * ch.nsplit, 0.f);
        if (bias) std::memcpy(hb.data(), bias + co0, sizeof(float) * ch.cout);
        const int rc = upload(hb, &ch.d_bias);
        out->push_back(ch);
        if (rc) return rc;
    }
    return BASIC_OK;
}

}  // namespace

extern "C" int basic_conv_plan_create(const float *weight, const float *bias, int cin, int cout, int ksize, int stride,
                                      int padding, int output_padding, int transposed, int activation,
                                      const float *gamma, const float *beta, int cin_active, int cout_active,
                                      basic_conv_plan **out)
{
    BASIC_REQUIRE(weight && out && cin >= 1 && cout >= 1 && ksize >= 1 && ksize <= 5 && stride >= 1 && stride <= 2 &&
                      padding >= 0 && padding <= ksize,
                  "conv_plan_create: unsupported geometry (k<=5, stride<=2)");
    BASIC_REQUIRE(cin_active >= 1 && cin_active <= cin && cout_active >= 1 && cout_active <= cout,
                  "conv_plan_create: active channel slice out of range");
    BASIC_REQUIRE(activation >= BASIC_ACT_NONE && activation <= BASIC_ACT_IGDN, "conv_plan_create: bad activation");
    const bool gdn = activation == BASIC_ACT_GDN || activation == BASIC_ACT_IGDN;
    BASIC_REQUIRE(!gdn || (gamma && beta), "conv_plan_create: GDN needs gamma and beta");
    int rc = require_device();
    if (rc) return rc;

    auto *p = new (std::nothrow) basic_conv_plan();
    if (!p) { set_error("out of host memory"); return BASIC_ERR_INVALID; }
    const int ci_n = cin_active, co_n = cout_active;
    p->cin = ci_n; p->cout = co_n; p->ksize = ksize; p->stride = stride; p->padding = padding;
    p->output_padding = output_padding; p->transposed = transposed ? 1 : 0; p->act = activation;
    p->s_in = transposed ? 1 : stride;
    p->s_out = transposed ? stride : 1;

    if (transposed && ksize == 5 && stride == 2 && padding == 2 && output_padding == 1 && co_n <= 3 && !gdn) {
        std::vector<float> ws(static_cast<size_t>(ci_n) * kSmWRow, 0.f), b4(4, 0.f);
        for (int c = 0; c < ci_n; ++c)
            for (int o = 0; o < co_n; ++o)
                for (int t = 0; t < 25; ++t)
                    ws[static_cast<size_t>(c) * kSmWRow + t * 3 + o] = weight[(static_cast<size_t>(c) * cout + o) * 25 + t];
        if (bias) std::memcpy(b4.data(), bias, sizeof(float) * co_n);
        rc = upload(ws, &p->d_wsm);
        if (!rc) rc = upload(b4, &p->d_bias4);
        if (rc) { basic_conv_plan_destroy(p); return rc; }
        *out = p;
        return BASIC_OK;
    }

    rc = build_chunks(p, weight, bias, cin, cout, 0, &p->chunks);
    if (rc) { basic_conv_plan_destroy(p); return rc; }
    // Small-grid variant: 32-channel slices spread over blockIdx.y, used when the position grid alone
    // cannot fill the chip (the 4x4 .. 16x16 maps of the hyper transforms).  GDN needs all channels
    // of a position in one block, so only the plain-activation layers get it.
    if (!gdn && co_n >= 64) {
        rc = build_chunks(p, weight, bias, cin, cout, 32, &p->split);
        if (rc) { basic_conv_plan_destroy(p); return rc; }
    }
    if (gdn) {
        if (p->chunks.size() != 1) { basic_conv_plan_destroy(p); set_error("conv_plan_create: GDN needs cout <= 192"); return BASIC_ERR_INVALID; }
        const int coutp = p->chunks[0].coutp, mtp = mtile_pitch(p->chunks[0].mt);
        // effective gamma [cout][cout] -> A-fragment order [k][col][m] = gamma[i = 32 m + col][k], zero padded; beta padded with 1
        std::vector<float> gt(static_cast<size_t>(coutp) * 32 * mtp, 0.f), bt(coutp, 1.f);
        for (int i = 0; i < co_n; ++i) {
            bt[i] = beta[i];
            for (int k = 0; k < co_n; ++k)
                gt[(static_cast<size_t>(k) * 32 + i % 32) * mtp + i / 32] = gamma[static_cast<size_t>(i) * cout + k];
        }
        gt.resize(gt.size() + 2048, 0.f);
        rc = upload(gt, &p->d_gammaT);
        if (!rc) rc = upload(bt, &p->d_beta);
        if (rc) { basic_conv_plan_destroy(p); return rc; }
    }
    *out = p;
    return BASIC_OK;
}

extern "C" int basic_conv_plan_channels(const basic_conv_plan *p, int *cin_active, int *cout_active)
{
    BASIC_REQUIRE(p, "conv_plan_channels: null plan");
    if (cin_active) *cin_active = p->cin;
    if (cout_active) *cout_active = p->cout;
    return BASIC_OK;
}

extern "C" int basic_conv_plan_out_hw(const basic_conv_plan *p, int in_h, int in_w, int *out_h, int *out_w)
{
    BASIC_REQUIRE(p && in_h >= 1 && in_w >= 1, "conv_plan_out_hw: bad argument");
    int oh, ow;
    if (p->transposed) {
        oh = (in_h - 1) * p->stride - 2 * p->padding + p->ksize + p->output_padding;
        ow = (in_w - 1) * p->stride - 2 * p->padding + p->ksize + p->output_padding;
    } else {
        oh = (in_h + 2 * p->padding - p->ksize) / p->stride + 1;
        ow = (in_w + 2 * p->padding - p->ksize) / p->stride + 1;
    }
    BASIC_REQUIRE(oh >= 1 && ow >= 1, "conv_plan_out_hw: empty output");
    if (out_h) *out_h = oh;
    if (out_w) *out_w = ow;
    return BASIC_OK;
}

extern "C" int64_t basic_conv_plan_flops(const basic_conv_plan *p, int batch, int in_h, int in_w)
{
    int oh = 0, ow = 0;
    if (!p || basic_conv_plan_out_hw(p, in_h, in_w, &oh, &ow)) return -1;
    // conv MACs: every (input position, tap, ci, co) of a transposed conv / (output position, tap, ci, co) of a conv
    int64_t macs = p->transposed
        ? static_cast<int64_t>(in_h) * in_w * p->ksize * p->ksize * p->cin * p->cout
        : static_cast<int64_t>(oh) * ow * p->ksize * p->ksize * p->cin * p->cout;
    if (p->act == BASIC_ACT_GDN || p->act == BASIC_ACT_IGDN) macs += static_cast<int64_t>(oh) * ow * p->cout * p->cout;
    return 2 * macs * batch;
}

namespace {

template <int MT, int CK, int KH, int KW, int WAVES, int KWB = 0>
int launch_one(const TapLaunch &g, int blocks, int nsplit, size_t lds_bytes, hipStream_t st)
{
    BASIC_HIP_TRY(ensure_max_lds(reinterpret_cast<const void *>(conv_tap_mfma_kernel<MT, CK, KH, KW, WAVES, KWB>)));
    hipLaunchKernelGGL((conv_tap_mfma_kernel<MT, CK, KH, KW, WAVES, KWB>), dim3(blocks, nsplit), dim3(64 * WAVES), lds_bytes, st, g);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}

// Unrolled instantiations for the tap grids of the codec's layers (5x5 and 3x3 convolutions, the four
// sub-pixel phases of the 5x5 stride-2 transposed convolution); anything else takes the runtime tap table.
template <int MT>
int launch_fused(const TapLaunch &g, int kh, int kw, int kwb, int plan_ck, int plan_waves_, int blocks, int nsplit, size_t lds_bytes, hipStream_t st)
{
    BASIC_REQUIRE(plan_ck == kFusedCK && plan_waves_ == 4 && kw == 3 && kwb == 2 && (kh == 3 || kh == 2),
                  "conv_forward: fused plan / kernel instantiation mismatch");
    if constexpr (MT <= 4) {
        if (kh == 3) return launch_one<MT, kFusedCK, 3, 3, 4, 2>(g, blocks, nsplit, lds_bytes, st);
        return launch_one<MT, kFusedCK, 2, 3, 4, 2>(g, blocks, nsplit, lds_bytes, st);
    } else {
        set_error("conv_forward: fused phases need <= 4 accumulator tiles");
        return BASIC_ERR_INVALID;
    }
}

template <int MT>
int launch_mt(const TapLaunch &g, int kh, int kw, int plan_ck, int plan_waves_, int blocks, int nsplit, size_t lds_bytes,
              hipStream_t st)
{
    constexpr int WM = plan_waves(MT, kMaxTaps), WF = plan_waves(MT, kFewTaps), WV = plan_waves(MT, kVeryFewTaps);
    constexpr int kMany = stage_channels(kMaxTaps, WM), kFew = stage_channels(kFewTaps, WF), kVeryFew = stage_channels(kVeryFewTaps, WV);
    // the packing (channels per stage) and the LDS size were derived from the plan's values: they must be the
    // instantiation's, or the kernel would walk off its LDS allocation
    const int inst_ck = g.ntaps > kFewTaps ? kMany : (g.ntaps > kVeryFewTaps ? kFew : kVeryFew);
    const int inst_waves = g.ntaps > kFewTaps ? WM : (g.ntaps > kVeryFewTaps ? WF : WV);
    BASIC_REQUIRE(inst_ck == plan_ck && inst_waves == plan_waves_, "conv_forward: plan / kernel instantiation mismatch");
    if (g.ntaps > kFewTaps) {
        if (kh == 5 && kw == 5) return launch_one<MT, kMany, 5, 5, WM>(g, blocks, nsplit, lds_bytes, st);
        return launch_one<MT, kMany, 0, 0, WM>(g, blocks, nsplit, lds_bytes, st);
    }
    if (g.ntaps > kVeryFewTaps) {
        if (kh == 3 && kw == 3) return launch_one<MT, kFew, 3, 3, WF>(g, blocks, nsplit, lds_bytes, st);
        return launch_one<MT, kFew, 0, 0, WF>(g, blocks, nsplit, lds_bytes, st);
    }
    if (kh == 3 && kw == 2) return launch_one<MT, kVeryFew, 3, 2, WV>(g, blocks, nsplit, lds_bytes, st);
    if (kh == 2 && kw == 3) return launch_one<MT, kVeryFew, 2, 3, WV>(g, blocks, nsplit, lds_bytes, st);
    if (kh == 2 && kw == 2) return launch_one<MT, kVeryFew, 2, 2, WV>(g, blocks, nsplit, lds_bytes, st);
    return launch_one<MT, kVeryFew, 0, 0, WV>(g, blocks, nsplit, lds_bytes, st);
}

int ilog2(int v) { int l = 0; while ((1 << (l + 1)) <= v) ++l; return l; }
int pow2_ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

}  // namespace

namespace {
// which launch list forward() walks for this input: 32-channel slices for small position grids, fused column phases when
// the output rows allow 8-byte pair stores
struct LaunchChoice { bool use_split, fuse_ok; int dbg; };
LaunchChoice choose_launches(const basic_conv_plan *p, int batch, int oh, int ow, const void *d_out)
{
    const int64_t pos_blocks = (static_cast<int64_t>(batch) * ((oh + p->s_out - 1) / p->s_out) * ((ow + p->s_out - 1) / p->s_out) + kTilePos - 1) / kTilePos;
    const char *dbg_env = getenv("BASIC_CONV_DEBUG");
    // profiling ablations: 1 skip staging, 2 skip MFMA loop, 4 force slices, 8 forbid slices, 64 no persistent first layer,
    // 128 no 16-byte patch pieces, 512 no fused column phases
    int dbg = dbg_env ? atoi(dbg_env) : 0;
#ifndef BASIC_DEBUG_ABLATIONS
    dbg &= ~(1 | 2 | 32);   // the timing ablations give WRONG results: only a library built with `make ABLATIONS=1` honours them
#endif
    LaunchChoice c;
    c.dbg = dbg;
    c.use_split = !p->split.empty() && !(dbg & 8) && (pos_blocks < kSplitBelowBlocks || (dbg & 4));
    c.fuse_ok = ow % 2 == 0 && (reinterpret_cast<uintptr_t>(d_out) & 7) == 0 && !(dbg & 512);
    return c;
}
}  // namespace

extern "C" int basic_conv_plan_launches(const basic_conv_plan *p, int batch, int in_h, int in_w)
{
    int oh = 0, ow = 0;
    if (!p || batch < 1 || basic_conv_plan_out_hw(p, in_h, in_w, &oh, &ow)) return -1;
    if (p->d_wsm) return 1;
    const LaunchChoice c = choose_launches(p, batch, oh, ow, nullptr);
    int n = 0;
    for (const Chunk &ch : (c.use_split ? p->split : p->chunks))
        for (const Phase &ph : ((c.fuse_ok && !ch.fused.empty()) ? ch.fused : ch.phases)) {
            const int mh = (oh - ph.oy0 + p->s_out - 1) / p->s_out, mw = (ow - ph.ox0 + p->s_out - 1) / p->s_out;
            if (mh > 0 && mw > 0) ++n;
        }
    return n;
}

extern "C" int basic_conv_forward_dev(const basic_conv_plan *p, const float *d_in, int batch, int in_h, int in_w,
                                      float *d_out, void *hip_stream)
{
    BASIC_REQUIRE(p && d_in && d_out && batch >= 1, "conv_forward: bad argument");
    int oh = 0, ow = 0;
    int rc = basic_conv_plan_out_hw(p, in_h, in_w, &oh, &ow);
    if (rc) return rc;
    if (p->d_wsm) {
        SmallLaunch g{};
        g.in = d_in; g.out = d_out; g.wsm = p->d_wsm; g.bias = p->d_bias4;
        g.batch = batch; g.cin = p->cin; g.cout = p->cout; g.in_h = in_h; g.in_w = in_w; g.act = p->act;
        g.tiles_y = (in_h + kSmTileH - 1) / kSmTileH;
        g.tiles_x = (in_w + kSmTileW - 1) / kSmTileW;
        const int blocks = batch * g.tiles_y * g.tiles_x;
        const char *two = getenv("BASIC_CONV_LAST_2ROW");   // experiment switch: two strips per lane (identical results)
        const bool dma_ok = in_w % 4 == 0 && (reinterpret_cast<uintptr_t>(d_in) & 15) == 0;
        if (dma_ok && (two ? atoi(two) != 0 : in_h >= 64)) {
            g.tiles_y = (in_h + kSm2TileH - 1) / kSm2TileH;
            hipLaunchKernelGGL(deconv5s2_cout3_dma2_kernel, dim3(batch * g.tiles_y * g.tiles_x), dim3(256), 0, as_stream(hip_stream), g);
        } else if (dma_ok)
            hipLaunchKernelGGL(deconv5s2_cout3_dma_kernel, dim3(blocks), dim3(256), 0, as_stream(hip_stream), g);
        else
            hipLaunchKernelGGL(deconv5s2_cout3_kernel, dim3(blocks), dim3(256), 0, as_stream(hip_stream), g);
        BASIC_HIP_TRY(hipGetLastError());
        return BASIC_OK;
    }
    // small position grids: spread the output channels over gridDim.y instead of looping them inside a block;
    // fused column phases: even output width (both phases have the same m-grid) and 8-byte aligned rows for the pair stores
    const LaunchChoice choice = choose_launches(p, batch, oh, ow, d_out);
    const int dbg = choice.dbg;
    const bool use_split = choice.use_split, fuse_ok = choice.fuse_ok;
    for (const Chunk &ch : (use_split ? p->split : p->chunks))
    for (const Phase &ph : ((fuse_ok && !ch.fused.empty()) ? ch.fused : ch.phases)) {
        TapLaunch g{};
        g.in = d_in; g.out = d_out; g.wpack = ph.d_wpack; g.wrow = ph.d_wrow; g.split_wstride = ph.split_wstride; g.bias = ch.d_bias; g.gammaT = p->d_gammaT; g.beta = p->d_beta;
        g.batch = batch; g.cin = p->cin; g.cin_pad = ph.ntaps ? ph.cin_pad : 0; g.cout = ch.cout; g.coutp = ch.coutp; g.out_ctotal = p->cout; g.co_base = ch.co0;
        g.in_h = in_h; g.in_w = in_w; g.out_h = oh; g.out_w = ow;
        g.s_in = p->s_in; g.s_out = p->s_out; g.oy0 = ph.oy0; g.ox0 = ph.ox0;
        g.mh = (oh - ph.oy0 + p->s_out - 1) / p->s_out;
        g.mw = (ow - ph.ox0 + p->s_out - 1) / p->s_out;
        if (g.mh <= 0 || g.mw <= 0) continue;
        g.ntaps = ph.ntaps; g.dymin = ph.dymin; g.dxmin = ph.dxmin;
        std::memcpy(g.dy, ph.dy, sizeof(g.dy));
        std::memcpy(g.dx, ph.dx, sizeof(g.dx));
        g.act = p->act;
        const int kCK = ph.ck;
        g.debug = dbg;
        // tile shape: 128 positions = TB images x TH x TW, powers of two, preferring wide rows
        int tw = pow2_ceil(g.mw); if (tw > 16) tw = 16;
        const int threads = 64 * ph.waves, tile_pos = 32 * ph.waves;  // one position per half-wave lane
        int th = pow2_ceil(g.mh); if (th > tile_pos / tw) th = tile_pos / tw;
        int tb = tile_pos / (tw * th);
        const bool first_layer_path = ch.mt == 4 && ch.nsplit == 1 && ph.waves == 8 && ph.kh == 5 && ph.kw == 5 &&
                                      ph.cin_pad == kCK && kCK == 4 && ch.cout == 128 && p->d_gammaT && !(dbg & 64);
        g.ph = (th - 1) * g.s_in + ph.span_y;
        g.pw = (tw - 1) * g.s_in + ph.span_x;
        // Patch rows as whole 16-byte groups aligned to the image (a group is then entirely inside or outside it):
        // 4x fewer DMA instructions per stage.  Needs 16-byte aligned rows; otherwise 4-byte pieces, odd row pitch.
        // (measured: pays for the 8-wave 25-tap convolutions, +1..1.5 %; the wider rows cost the few-tap launches more than they save)
        g.patch4 = (ph.waves == 8 && !first_layer_path && in_w % 4 == 0 && (reinterpret_cast<uintptr_t>(d_in) & 15) == 0 && !(dbg & 128)) ? 1 : 0;
        g.pwp = g.patch4 ? (g.pw + 3 + 3) / 4 * 4 : (g.pw | 1);
        const int punit = g.patch4 ? 4 : 1;
        const int mtp = mtile_pitch(ch.mt);
        const int wl_floats = g.ntaps * kCK * 32 * mtp, gam_floats = 32 * 32 * mtp;
        const int wl_pad = ((wl_floats > gam_floats ? wl_floats : gam_floats) + 4 * threads - 1) / (4 * threads) * (4 * threads);
        auto patch_slots_needed = [&](int tbv) { return (tbv * kCK * g.ph * g.pwp / punit + threads - 1) / threads; };
        auto lds_need = [&](int tbv) {  // two stage buffers + tap table
            const int patch_pad = ((tbv * kCK * g.ph * g.pwp / punit + 63) & ~63) * punit;  // whole wave-instructions
            return sizeof(float) * (2 * static_cast<size_t>(wl_pad + patch_pad) + 32 + 2 * g.coutp);  // + tap table, bias, beta
        };
        // the patch must fit the gather descriptors and both stage buffers the LDS
        const int slots_avail = ph.kwb ? kPSlotsFused : patch_slots(ch.mt);
        while (tb > 1 && (patch_slots_needed(tb) > slots_avail || lds_need(tb) > 160 * 1024)) tb >>= 1;
        g.tw_log = ilog2(tw); g.th_log = ilog2(th); g.tb_log = ilog2(tb);
        g.tiles_y = (g.mh + th - 1) / th;
        g.tiles_x = (g.mw + tw - 1) / tw;
        const int blocks = ((batch + tb - 1) / tb) * g.tiles_y * g.tiles_x;
        const size_t lds_bytes = lds_need(tb);
        BASIC_REQUIRE(patch_slots_needed(tb) <= slots_avail, "conv_forward: input patch exceeds the gather descriptors");
        BASIC_REQUIRE(static_cast<int64_t>(tb) * p->cin * in_h * in_w < (1ll << 29), "conv_forward: input tile too large");
        BASIC_REQUIRE(lds_bytes <= 160 * 1024, "conv_forward: LDS budget exceeded");
        hipStream_t st = as_stream(hip_stream);
        if (first_layer_path) {
            // single-stage GDN layer (the first analysis layer): persistent workgroups with resident weights and gamma;
            // <= 3 input channels: the row-interleaved reduction (BASIC_CONV_DEBUG & 1024 keeps the padded-channel one)
            const bool rows = ph.d_wrow != nullptr && !(dbg & 1024);
            const int patch_pad1 = (tb * (rows ? 3 : kCK) * g.ph * g.pwp + 511) / 512 * 512;
            const size_t lds1 = sizeof(float) * ((rows ? 10240 : 12800) + 16384 + 2 * static_cast<size_t>(patch_pad1) + 256 + 4);
            if (lds1 <= 160 * 1024 && patch_pad1 <= kFirstSlots * 512) {
                BASIC_HIP_TRY(ensure_max_lds(reinterpret_cast<const void *>(conv5x5_cin4_gdn_persistent_kernel<false>)));
                BASIC_HIP_TRY(ensure_max_lds(reinterpret_cast<const void *>(conv5x5_cin4_gdn_persistent_kernel<true>)));
                int dev = 0, cus = 256;
                (void)hipGetDevice(&dev);
                (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                const int grid = blocks < cus ? blocks : cus;
                // dynamic tile hand-out only where other HIP streams compete for compute units (a codec session with the
                // transform token sets it for its calls): alone on the chip the static stride is 0.2 ms faster (2.14 vs 2.35 ms)
                int *sched = nullptr;
                if (g_dynamic_tiles) {
                    auto *pm = const_cast<basic_conv_plan *>(p);   // the counter is scratch of the launch, not plan state
                    if (!pm->d_sched) BASIC_HIP_TRY(hipMalloc(&pm->d_sched, sizeof(int)));
                    BASIC_HIP_TRY(hipMemsetAsync(pm->d_sched, 0, sizeof(int), st));
                    sched = pm->d_sched;
                }
                if (rows) hipLaunchKernelGGL(conv5x5_cin4_gdn_persistent_kernel<true>, dim3(grid), dim3(512), lds1, st, g, blocks, sched);
                else hipLaunchKernelGGL(conv5x5_cin4_gdn_persistent_kernel<false>, dim3(grid), dim3(512), lds1, st, g, blocks, sched);
                BASIC_HIP_TRY(hipGetLastError());
                continue;
            }
        }
        if (ph.kwb) {
            switch (ch.mt) {
                case 1: rc = launch_fused<1>(g, ph.kh, ph.kw, ph.kwb, ph.ck, ph.waves, blocks, ch.nsplit, lds_bytes, st); break;
                case 2: rc = launch_fused<2>(g, ph.kh, ph.kw, ph.kwb, ph.ck, ph.waves, blocks, ch.nsplit, lds_bytes, st); break;
                case 3: rc = launch_fused<3>(g, ph.kh, ph.kw, ph.kwb, ph.ck, ph.waves, blocks, ch.nsplit, lds_bytes, st); break;
                case 4: rc = launch_fused<4>(g, ph.kh, ph.kw, ph.kwb, ph.ck, ph.waves, blocks, ch.nsplit, lds_bytes, st); break;
                default: set_error("conv_forward: fused phases need <= 4 accumulator tiles"); rc = BASIC_ERR_INVALID;
            }
            if (rc) return rc;
            continue;
        }
        switch (ch.mt) {
            case 1: rc = launch_mt<1>(g, ph.kh, ph.kw, ph.ck, ph.waves, blocks, ch.nsplit, lds_bytes, st); break;
            case 2: rc = launch_mt<2>(g, ph.kh, ph.kw, ph.ck, ph.waves, blocks, ch.nsplit, lds_bytes, st); break;
            case 3: rc = launch_mt<3>(g, ph.kh, ph.kw, ph.ck, ph.waves, blocks, ch.nsplit, lds_bytes, st); break;
            case 4: rc = launch_mt<4>(g, ph.kh, ph.kw, ph.ck, ph.waves, blocks, ch.nsplit, lds_bytes, st); break;
            case 5: rc = launch_mt<5>(g, ph.kh, ph.kw, ph.ck, ph.waves, blocks, ch.nsplit, lds_bytes, st); break;
            case 6: rc = launch_mt<6>(g, ph.kh, ph.kw, ph.ck, ph.waves, blocks, ch.nsplit, lds_bytes, st); break;
            default: set_error("conv_forward: cout > 192 unsupported"); rc = BASIC_ERR_INVALID;
        }
        if (rc) return rc;
    }
    return BASIC_OK;
}
