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
// slab is skipped when its mask is empty for all positions of the tile -- for causal
// patterns that removes most of the K loop.
//
// ONE SUMMATION ORDER (round 3).  The integers the entropy coder sees are rounded from these sums, and an autoregressive
// stream only decodes if the decoder reproduces the encoder's sums bit for bit -- whatever the batch size, the number of
// listed positions or the launch shape either side happens to use (the reference runs one code path for any batch,
// pgm_coder.py:912-981).  Every kernel of this file (and the persistent scan-line kernel, scanline.hip) therefore
// evaluates an output element in the same CANONICAL order, a property of the layer alone:
//   * K is walked tap by tap (t = 0 .. k*k-1), inside a tap input group by input group, inside a group channel by channel;
//   * each (tap, input group) slab is cut into BLOCKS of kKB = 64 consecutive channels (the last block of a slab may be
//     shorter); a block's partial sum is ONE chain of v_mfma_f32_32x32x2_f32 over its channel pairs (c, c + 1) starting
//     from zero -- measured (scripts/micro/mfma_arith.hip): exactly fma(a1, b1, fma(a0, b0, acc)) per step;
//   * the block partials are added to the running total one by one, in block order (fp32 adds); bias last.
//   Masked / padded elements enter as zeros and all-masked slabs are skipped: both leave every partial and the total
//   unchanged (x + 0 = x), so skipping is a pure optimisation and may differ between tiles and kernels.
// The three kernels differ only in where the parallelism comes from:
//   masked_conv_dma_kernel      GEMM-shaped launches (>= thousands of positions, 128-row output chunks): 8-wave workgroup =
//                               128 rows x 256 positions, operands through LDS by LDS-DMA (weights: 16-byte pieces of a
//                               pre-packed slab; activations: buffer loads with the channel offset on the scalar unit and
//                               out-of-range = masked lanes writing zeros), two stage buffers, an unrolled MFMA stage of
//                               LDS reads with immediate offsets -- an activation element is fetched once per workgroup;
//   masked_conv_gather_kernel   any shape: one wave = MT row tiles x 32 positions, operands gathered into registers;
//   masked_conv_block_kernel    tiny launches (a scan-line step is one position per image): one wave per (tile, BLOCK),
//   + masked_conv_reduce_kernel partial tiles through scratch, summed in block order -- the chip sees hundreds of waves
//                               with ~32-step chains instead of a dozen waves with ~1000-step chains.
// B fragments are gathered from x with the mask applied as a select / an out-of-range address (so masked garbage, even NaN,
// never enters the sum -- the reference multiplies by 0, which only differs for non-finite data).
#include "common.h"

#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <type_traits>
#include <vector>

using namespace basic;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_cvoid;

constexpr int kKB = BASIC_MCONV_BLOCK_CHANNELS;   // canonical block (see the header comment); scanline.hip uses the same constant

struct MaskedLaunch {
    const float *x;        // [B][cin][H][W]
    float *y;              // [B][out_total][H][W], this layer writes channels out_off .. out_off+cout
    const float *w;        // gather kernel: [ntaps][cin][coutp] mt-packed; block kernel: plain; dma kernel: [rchunk][ntaps][cin][32][4]
    const float *bias;     // [coutp]
    const int32_t *topo_in;   // [gi][H][W]
    const int32_t *topo_out;  // [go][H][W]
    const int32_t *pos;    // [n_pos] flat b*H*W + p
    int64_t n_pos;
    int batch, cin, cout, coutp, h, w_, gi, go, gs_in, gs_out, tiles_per_group;
    int out_total, out_off, ntaps, pad, ksize, allow_same, act;
    // optional position permutations of the planes of x / y (nullptr = row-major p = py * W + px): element (b, c, p) sits at
    // (b * C + c) * HW + perm[p].  The coder keeps its PRIVATE hidden activations (between the 1x1 merger layers) with the
    // positions of a coding step contiguous, so that a step's gathers and stores cover whole cache lines (a checkerboard
    // step in row-major planes touches every other float: half of every line fetched and half of every sector written).
    const int32_t *in_perm, *out_perm;
    int step;              // coding step (topo group being coded), or kNoStep: see the skip rule in the kernel
    const int32_t *first;  // [H][W] first step that visits a position (min over channel groups), or nullptr
    int mt;                // pack factor of w (gather kernel): row r of a group sits at (r / (32 mt)) * 32 mt + (r % 32) * mt + (r / 32) % mt
    // block kernel: units = (slab, block of the slab) pairs; scratch[(tile * units + unit)][16][64], flags[tile * units + unit]
    float *scratch;
    int32_t *flags;
    int units, blocks_per_slab;
    // dma kernel
    int n_pchunks, n_rchunks, x_bytes;
    int debug;   // BASIC_MCONV_DEBUG dma kernel: timing ablations (wrong results) 1 no staging, 4 no stores
};

constexpr int kNoStep = INT32_MIN;
constexpr int kUnroll = 4;  // channel pairs whose loads are issued together in the gather kernel (one L2 round trip per 4 MFMAs)

struct Pos {
    int b, py, px;
    bool ok;
};

__device__ __forceinline__ Pos decode_pos(const MaskedLaunch &g, int64_t pj)
{
    Pos p{0, 0, 0, pj < g.n_pos};
    if (p.ok) {
        const int hw = g.h * g.w_;
        const int32_t f = g.pos[pj];
        p.b = f / hw;
        const int q = f - p.b * hw;
        p.py = q / g.w_;
        p.px = q - p.py * g.w_;
    }
    return p;
}

// Step rule (coding loop): the value of (output group, position) only depends on elements coded before ITS step
// (mask < / <=), so what an earlier step wrote is still exact and what a later step needs is computed then.  A tile
// is therefore evaluated only if some position has this output group in the current step -- or, for groups that
// carry no topo id (-1: the prior half of the merger's hidden layers), at the first step that visits the position.
// With G channel groups this removes (G-1)/G of the work of channel-wise schedules.
__device__ __forceinline__ bool step_needs(const MaskedLaunch &g, const Pos &p, int32_t centre)
{
    return p.ok && (centre == g.step || (centre < 0 && g.first[p.py * g.w_ + p.px] == g.step));
}

// mask of input group gin at the neighbour `noff` of a position whose own id is `centre`
__device__ __forceinline__ bool tap_open(const MaskedLaunch &g, int gin, bool inside, int noff, int32_t centre)
{
    if (!inside) return false;
    const int32_t tn = g.topo_in[gin * g.h * g.w_ + noff];
    return g.allow_same ? (tn <= centre) : (tn < centre);
}

__device__ __forceinline__ int in_slot(const MaskedLaunch &g, int noff) { return g.in_perm ? g.in_perm[noff] : noff; }
__device__ __forceinline__ int out_slot(const MaskedLaunch &g, const Pos &p)
{
    const int q = p.py * g.w_ + p.px;
    return g.out_perm ? g.out_perm[q] : q;
}

// bias, then the activation FUSED after the layer -- written with selects on launch-uniform values so that the 64 epilogue
// values of a lane stay straight-line code
__device__ __forceinline__ float activate(float v, int act)
{
    const float slope = act == BASIC_ACT_LEAKY_RELU ? 0.01f : 1.f;
    const float neg = act == BASIC_ACT_RELU ? 0.f : slope * v;
    return v > 0.f ? v : neg;
}

__device__ __forceinline__ float finish_value(const MaskedLaunch &g, float total, int co)
{
    return activate(total + g.bias[co], g.act);
}

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

// ---------------------------------------------------------------------------------------------------------------------
// Any shape: one wave = MT 32-row tiles of ONE output channel group x 32 listed positions, operands gathered into registers.
// The B fragment (gathered activations, mask applied) is loaded once and feeds MT MFMAs; the weights are packed so that a
// lane's MT A values are adjacent: [tap][ci][chunk][32 rows][MT] (one 8- / 16-byte load for MT 2 / 4).
// ---------------------------------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(64) void masked_conv_gather_kernel(const MaskedLaunch g)
{
    const int lane = threadIdx.x & 63, col = lane & 31, khalf = lane >> 5;
    const int chunks_per_group = g.tiles_per_group / MT;       // host guarantees divisibility when MT > 1
    const int grp_o = blockIdx.y / chunks_per_group, ti0 = (blockIdx.y - grp_o * chunks_per_group) * MT;
    const bool row_ok = ti0 * 32 + col < g.gs_out || MT > 1;  // MT > 1 only with whole tiles (gs_out % (32 MT) == 0)
    // packed position of this lane's A value(s): see MaskedLaunch::mt
    const int a_off = grp_o * g.gs_out + (ti0 / g.mt) * 32 * g.mt + col * g.mt + (ti0 % g.mt);

    const Pos p = decode_pos(g, static_cast<int64_t>(blockIdx.x) * 32 + col);
    const int hw = g.h * g.w_;
    const int32_t centre = p.ok ? g.topo_out[grp_o * hw + p.py * g.w_ + p.px] : 0;
    const float *xb = g.x + static_cast<int64_t>(p.b) * g.cin * hw;
    if (g.step != kNoStep && __ballot(step_needs(g, p, centre)) == 0ull) return;   // wave-uniform

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    for (int t = 0; t < g.ntaps; ++t) {
        const int dy = t / g.ksize - g.pad, dx = t % g.ksize - g.pad;
        const int yy = p.py + dy, xx = p.px + dx;
        const bool inside = p.ok && yy >= 0 && yy < g.h && xx >= 0 && xx < g.w_;
        const int noff = yy * g.w_ + xx;
        for (int gin = 0; gin < g.gi; ++gin) {
            const bool open = tap_open(g, gin, inside, noff, centre);
            if (__ballot(open) == 0ull) continue;  // wave-uniform skip of an all-masked slab
            const int xoff = open ? in_slot(g, noff) : 0;
            const int c_beg = gin * g.gs_in, c_end = c_beg + g.gs_in;
            const float *wt = g.w + (static_cast<int64_t>(t) * g.cin) * g.coutp + a_off;
            for (int c0 = c_beg; c0 < c_end; c0 += kKB) {       // canonical block: its own chain, then one add per element
                const int c1 = c0 + kKB < c_end ? c0 + kKB : c_end;
                f32x16 blk[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) blk[m][r] = 0.f;
                for (int c = c0; c < c1; c += 2 * kUnroll) {
                    float af[kUnroll][MT], bf[kUnroll];
#pragma unroll
                    for (int u = 0; u < kUnroll; ++u) {
                        const int ci = c + u * 2 + khalf;
                        const bool ci_ok = ci < c1;
                        load_a<MT>(wt + static_cast<int64_t>(ci) * g.coutp, row_ok && ci_ok, af[u]);
                        bf[u] = (open && ci_ok) ? xb[static_cast<int64_t>(ci) * hw + xoff] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < kUnroll; ++u)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            blk[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u][m], bf[u], blk[m], 0, 0, 0);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][r] += blk[m][r];
            }
        }
    }

    if (p.ok) {
        float *yb = g.y + (static_cast<int64_t>(p.b) * g.out_total + g.out_off) * hw + out_slot(g, p);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rg = (ti0 + m) * 32 + 8 * (r >> 2) + 4 * khalf + (r & 3);
                if (rg < g.gs_out) {
                    const int co = grp_o * g.gs_out + rg;
                    yb[static_cast<int64_t>(co) * hw] = finish_value(g, acc[m][r], co);
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Tiny launches: one WAVE per (32 x 32 tile, canonical block).  A block is at most 32 channel pairs: all its loads are in
// flight together (two rounds of 16 pairs), one MFMA chain, and the partial tile goes to scratch; a flag per unit says
// whether the unit's slab was open for the tile (closed units write nothing).  Plain [tap][ci][co] weights: a k row of
// the A operand is one coalesced 128-byte line.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kBlockUnroll = 16;

__global__ __launch_bounds__(256) void masked_conv_block_kernel(const MaskedLaunch g)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, khalf = lane >> 5;
    const int unit = blockIdx.z * 4 + wave;
    if (unit >= g.units) return;
    const int slab = unit / g.blocks_per_slab, blk_i = unit - slab * g.blocks_per_slab;
    const int t = slab / g.gi, gin = slab - t * g.gi;
    const int grp_o = blockIdx.y / g.tiles_per_group, ti = blockIdx.y - grp_o * g.tiles_per_group;
    const bool row_ok = ti * 32 + col < g.gs_out;
    const int a_off = grp_o * g.gs_out + ti * 32 + col;

    const Pos p = decode_pos(g, static_cast<int64_t>(blockIdx.x) * 32 + col);
    const int hw = g.h * g.w_;
    const int32_t centre = p.ok ? g.topo_out[grp_o * hw + p.py * g.w_ + p.px] : 0;
    const float *xb = g.x + static_cast<int64_t>(p.b) * g.cin * hw;
    const int64_t tile = static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x;
    int32_t *flag = g.flags + tile * g.units + unit;
    bool live = true;
    if (g.step != kNoStep && __ballot(step_needs(g, p, centre)) == 0ull) live = false;   // the reduce kernel skips the tile too
    const int dy = t / g.ksize - g.pad, dx = t % g.ksize - g.pad;
    const int yy = p.py + dy, xx = p.px + dx;
    const bool inside = p.ok && yy >= 0 && yy < g.h && xx >= 0 && xx < g.w_;
    const int noff = yy * g.w_ + xx;
    const bool open = live && tap_open(g, gin, inside, noff, centre);
    if (__ballot(open) == 0ull) {
        if (lane == 0) *flag = 0;
        return;
    }
    const int xoff = open ? in_slot(g, noff) : 0;
    const int c_beg = gin * g.gs_in, c_end = c_beg + g.gs_in;
    const int c0 = c_beg + blk_i * kKB, c1 = c0 + kKB < c_end ? c0 + kKB : c_end;
    const float *wt = g.w + (static_cast<int64_t>(t) * g.cin) * g.coutp + a_off;
    f32x16 blk;
#pragma unroll
    for (int r = 0; r < 16; ++r) blk[r] = 0.f;
    for (int c = c0; c < c1; c += 2 * kBlockUnroll) {
        float af[kBlockUnroll], bf[kBlockUnroll];
#pragma unroll
        for (int u = 0; u < kBlockUnroll; ++u) {
            const int ci = c + u * 2 + khalf;
            const bool ci_ok = ci < c1;
            af[u] = (row_ok && ci_ok) ? wt[static_cast<int64_t>(ci) * g.coutp] : 0.f;
            bf[u] = (open && ci_ok) ? xb[static_cast<int64_t>(ci) * hw + xoff] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kBlockUnroll; ++u) blk = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u], bf[u], blk, 0, 0, 0);
    }
    float *dst = g.scratch + ((tile * g.units + unit) * 16) * 64 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[r * 64] = blk[r];
    if (lane == 0) *flag = 1;
}

// Second half: one workgroup per tile, wave w owns accumulator rows 4w .. 4w+3 of every lane; the open units' partials are
// loaded a batch at a time and added in unit (= canonical block) order; then bias, activation and store.
__global__ __launch_bounds__(256) void masked_conv_reduce_kernel(const MaskedLaunch g)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, khalf = lane >> 5;
    const int grp_o = blockIdx.y / g.tiles_per_group, ti = blockIdx.y - grp_o * g.tiles_per_group;
    const Pos p = decode_pos(g, static_cast<int64_t>(blockIdx.x) * 32 + col);
    const int hw = g.h * g.w_;
    if (g.step != kNoStep) {  // same rule as the partial kernel: a skipped tile keeps what an earlier step wrote
        const int32_t centre = p.ok ? g.topo_out[grp_o * hw + p.py * g.w_ + p.px] : 0;
        if (__ballot(step_needs(g, p, centre)) == 0ull) return;
    }
    const int64_t tile = static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x;
    const int32_t *flags = g.flags + tile * g.units;
    const float *src = g.scratch + (tile * g.units * 16 + 4 * wave) * 64 + lane;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int kBatch = 16;
    // The open units as bit masks first (one coalesced load of the flags per 64 units), then their partials a batch at a
    // time in ascending unit order with every load of a batch in flight: a loop that looked at a flag and then loaded that
    // unit's partial paid two dependent round trips per batch -- ten per context-layer launch of a scan-line step.
    for (int ub = 0; ub < g.units; ub += 128) {
        uint64_t m0 = __ballot(ub + lane < g.units && flags[ub + lane] != 0);
        uint64_t m1 = __ballot(ub + 64 + lane < g.units && flags[ub + 64 + lane] != 0);
        while (m0 | m1) {
            int u[kBatch];
            float v[kBatch][4];
#pragma unroll
            for (int s = 0; s < kBatch; ++s) {
                if (m0) { u[s] = ub + __builtin_ctzll(m0); m0 &= m0 - 1; }
                else if (m1) { u[s] = ub + 64 + __builtin_ctzll(m1); m1 &= m1 - 1; }
                else u[s] = -1;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[s][q] = u[s] >= 0 ? src[(static_cast<int64_t>(u[s]) * 16 + q) * 64] : 0.f;
            }
#pragma unroll
            for (int s = 0; s < kBatch; ++s)
                if (u[s] >= 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] += v[s][q];
                }
        }
    }
    if (!p.ok) return;
    float *yb = g.y + (static_cast<int64_t>(p.b) * g.out_total + g.out_off) * hw + out_slot(g, p);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rg = ti * 32 + 8 * wave + 4 * khalf + q;
        if (rg < g.gs_out) {
            const int co = grp_o * g.gs_out + rg;
            yb[static_cast<int64_t>(co) * hw] = finish_value(g, acc[q], co);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// GEMM-shaped launches.  Workgroup = 8 waves = one 128-row chunk of an output channel group x 256 listed positions; wave w
// owns all 128 rows (4 accumulator tiles) of positions 32w .. 32w+31.  K is walked kDmaCK = 32 channels of one (tap, input
// group) slab per stage through three LDS stage buffers { A [32 k][32 cols][4 tiles] | B [32 k][256 positions] } filled by
// LDS-DMA a stage and a half ahead of the MFMAs:
//   A  16-byte pieces of the pre-packed slab [row chunk][tap][ci][32][4] (contiguous per stage: two instructions per lane);
//   B  buffer_load_dword ... lds, one position per lane: the lane's byte offset of (image, neighbour position) in a VGPR --
//      0x80000000 = out of range = the hardware writes ZERO for a masked / padded / unlisted element -- and the channel
//      offset in an SGPR, so a stage's 16 gathers cost no vector ALU work at all.
// A stage is 16 unrolled steps of { ds_read_b32 (B), ds_read_b128 (A), 4 MFMAs } with compile-time LDS offsets; two stages
// make one canonical block, whose chain starts from zero and is added to the running total at the first step of the next
// block (the adds sit between MFMAs of other tiles).  Slabs that are closed for all 256 positions are skipped.
// Workgroups are dealt to the XCDs so that all row chunks of one position chunk run on the SAME XCD back to back: the
// position chunk's activations (256 x Cin x 4 bytes) come out of the fabric once and are re-read from that XCD's L2.
// ---------------------------------------------------------------------------------------------------------------------
// WAVES = 8: the shape above (three stage buffers of 48 KB, one workgroup per compute unit).  WAVES = 4: 128 positions per
// workgroup, two stage buffers of 32 KB, TWO workgroups per compute unit -- for launches whose 8-wave grid would be only a
// round or two of the chip (the 384-row layers: 384 workgroups on 256 compute units take two rounds for one and a half
// rounds of work); twice as many, half as long workgroups cut that tail, at the price of staging the weights twice as often.
constexpr int kDmaCK = 32, kDmaRows = 128;
constexpr int kDmaAFloats = kDmaCK * 32 * 4;
constexpr int kDmaMaxSlabs = 64;
constexpr unsigned kOutOfRange = 0x80000000u;
constexpr int dma_bufs(int waves) { return waves == 8 ? 3 : 2; }
constexpr int dma_stage_floats(int waves) { return kDmaAFloats + kDmaCK * 32 * waves; }
constexpr size_t dma_lds_bytes(int waves) { return (dma_bufs(waves) * dma_stage_floats(waves) + 64 + kDmaRows) * sizeof(float); }

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, (WAVES == 8 ? 1 : 2)) void masked_conv_dma_kernel(const MaskedLaunch g)
{
    constexpr int kDmaThreads = 64 * WAVES, kDmaPos = 32 * WAVES, kDmaStage = dma_stage_floats(WAVES), kDmaBufs = dma_bufs(WAVES);
    constexpr int kChunks = WAVES / 2;                 // 64-position pieces of a B row (4-byte gathers: one piece per wave instruction)
    constexpr int kAPieces = kDmaAFloats / (kDmaThreads * 4);
    extern __shared__ float lds[];
    unsigned *s_open = reinterpret_cast<unsigned *>(lds + kDmaBufs * kDmaStage);   // [2] slab bitmask, [2] step-rule flag
    float *s_bias = lds + kDmaBufs * kDmaStage + 64;
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, khalf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (position chunk, row chunk): ids b, b + 8, b + 16 .. share an XCD; consecutive slots of an XCD walk the
    // row chunks of one position chunk
    // ... and output group by output group: the workgroups of one group do the same amount of work, and when the groups differ
    // (the merger's id-less groups see only the prior half of the inputs) the cheap ones come last, where they cost the least tail
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int chunks_per_group = g.gs_out / kDmaRows, pcs_per_xcd = (g.n_pchunks + 7) >> 3;
    const int grp_o = slot / (pcs_per_xcd * chunks_per_group), rem = slot - grp_o * pcs_per_xcd * chunks_per_group;
    const int pc = (rem / chunks_per_group) * 8 + xcd, rc = grp_o * chunks_per_group + rem % chunks_per_group;
    if (pc >= g.n_pchunks) return;
    const int row0 = grp_o * g.gs_out + (rc - grp_o * chunks_per_group) * kDmaRows;   // first output channel of the chunk
    const int hw = g.h * g.w_;

    if (tid < 4) s_open[tid] = 0u;
    if (tid < kDmaRows) s_bias[tid] = g.bias[row0 + tid];
    __syncthreads();

    // the position this lane STAGES (B rows are kDmaPos positions wide: wave w fills columns 64 (w % kChunks) .. + 63 of k rows
    // 2 i + w / kChunks) and the position this lane COMPUTES (column 32 w + col)
    const Pos dp = decode_pos(g, static_cast<int64_t>(pc) * kDmaPos + (wave % kChunks) * 64 + lane);
    const Pos mp = decode_pos(g, static_cast<int64_t>(pc) * kDmaPos + wave * 32 + col);
    const int32_t dcentre = dp.ok ? g.topo_out[grp_o * hw + dp.py * g.w_ + dp.px] : 0;
    if (g.step != kNoStep) {
        const int32_t mcentre = mp.ok ? g.topo_out[grp_o * hw + mp.py * g.w_ + mp.px] : 0;
        if (__ballot(step_needs(g, mp, mcentre)) != 0ull && lane == 0) atomicOr(&s_open[2], 1u);
    }
    // which slabs are open for this lane's staged position (bit s = tap * gi + gin) and for any position of the workgroup
    uint64_t my_open = 0ull;
    const int nslabs = g.ntaps * g.gi;
    for (int s = 0; s < nslabs; ++s) {
        const int t = s / g.gi, gin = s - t * g.gi;
        const int yy = dp.py + t / g.ksize - g.pad, xx = dp.px + t % g.ksize - g.pad;
        const bool inside = dp.ok && yy >= 0 && yy < g.h && xx >= 0 && xx < g.w_;
        if (tap_open(g, gin, inside, yy * g.w_ + xx, dcentre)) my_open |= 1ull << s;
    }
    {
        unsigned lo = static_cast<unsigned>(my_open), hi = static_cast<unsigned>(my_open >> 32);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { lo |= __shfl_xor(lo, d, 64); hi |= __shfl_xor(hi, d, 64); }
        if (lane == 0) { atomicOr(&s_open[0], lo); atomicOr(&s_open[1], hi); }
    }
    __syncthreads();
    if (g.step != kNoStep && s_open[2] == 0u) return;   // workgroup-uniform
    uint64_t todo = static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(s_open[0])) |
                    (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(s_open[1])) << 32);
    const int stages_per_slab = g.gs_in / kDmaCK;
    const int nstages = __builtin_popcountll(todo) * stages_per_slab;

    const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc[4], blk[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = kZero16;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g.x), 0, g.x_bytes, 0x00020000);
    const unsigned img_off = static_cast<unsigned>(dp.b) * static_cast<unsigned>(g.cin) * static_cast<unsigned>(hw) * 4u;
    const float *wchunk = g.w + static_cast<int64_t>(rc) * g.ntaps * g.cin * 128;

    // issue side: (slab = (tap, input group), stage of the slab) of the next stage to stage.  The tap / group split is kept in
    // wave-uniform state computed outside any divergent branch (a division result that flows through a divergent join costs
    // a waterfall loop per DMA instruction).
    uint64_t is_todo = todo;
    int is_slab = todo ? __builtin_ctzll(todo) : 0, is_j = 0;
    int is_t = is_slab / g.gi, is_gin = is_slab - is_t * g.gi;
    auto slab_voffset = [&](int s, int t) __attribute__((always_inline)) {
        const int yy = dp.py + t / g.ksize - g.pad, xx = dp.px + t % g.ksize - g.pad;
        const bool on = (my_open >> s) & 1ull;
        return on ? img_off + static_cast<unsigned>(in_slot(g, yy * g.w_ + xx)) * 4u : kOutOfRange;
    };
    unsigned voff = nstages > 0 ? slab_voffset(is_slab, is_t) : kOutOfRange;

    // The DMA of one stage = 2 pieces of A (16 bytes per lane) + 16 gathers of B, then the issue state moves on.  It is issued in
    // PIECES spread over the MFMA steps that follow the stage barrier: a wave that issues its 18 vector-memory instructions
    // back to back blocks in the issue stage until the texture-address unit has taken them (~33 cycles per 64-lane gather, all
    // eight waves at once: thousands of cycles without an MFMA, measured as +20 % kernel time).
    int ci0_i = 0;
    float *dst_i = nullptr;
    const float *srca_i = nullptr;
#define BASIC_MCONV_ISSUE_BEGIN(BUF)                                                                                            \
    do {                                                                                                                        \
        ci0_i = is_gin * g.gs_in + is_j * kDmaCK;                                                                               \
        dst_i = lds + (BUF) * kDmaStage;                                                                                        \
        srca_i = wchunk + (static_cast<int64_t>(is_t) * g.cin + ci0_i) * 128 + tid * 4;                                         \
    } while (0)
#define BASIC_MCONV_ISSUE_A(J)                                                                                                  \
    __builtin_amdgcn_global_load_lds((glb_cvoid *)(srca_i + (J) * kDmaThreads * 4), (lds_void *)(dst_i + (J) * kDmaThreads * 4 + wave * 256), 16, 0, 0)
#define BASIC_MCONV_ISSUE_B(I)                                                                                                  \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void *)(dst_i + kDmaAFloats + (wave % kChunks) * 64 + (2 * (I) + wave / kChunks) * kDmaPos), \
                                             4, voff, (ci0_i + 2 * (I) + wave / kChunks) * hw * 4, 0, 0)
#define BASIC_MCONV_ISSUE_END()                                                                                                 \
    do {                                                                                                                        \
        if (++is_j == stages_per_slab) {                                                                                        \
            is_j = 0;                                                                                                           \
            is_todo &= is_todo - 1;                                                                                             \
            if (is_todo) {                                                                                                      \
                is_slab = __builtin_ctzll(is_todo);                                                                             \
                is_t = is_slab / g.gi;                                                                                          \
                is_gin = is_slab - is_t * g.gi;                                                                                 \
                voff = slab_voffset(is_slab, is_t);                                                                             \
            }                                                                                                                   \
        }                                                                                                                       \
    } while (0)
#define BASIC_MCONV_ISSUE(BUF)                                                                                                  \
    do {                                                                                                                        \
        BASIC_MCONV_ISSUE_BEGIN(BUF);                                                                                           \
        _Pragma("unroll") for (int j_ = 0; j_ < kAPieces; ++j_) BASIC_MCONV_ISSUE_A(j_);                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < kDmaCK / 2; ++i_) BASIC_MCONV_ISSUE_B(i_);                                      \
        BASIC_MCONV_ISSUE_END();                                                                                                \
    } while (0)

    // Pipeline, WAVES = 8: THREE stage buffers and ONE workgroup barrier per stage, placed in the MIDDLE of the stage.  At the
    // barrier of stage n every wave has waited for its own DMAs of stage n+1 (issued a whole stage earlier) -- so after it
    // stage n+1 is complete in LDS -- and every wave has left stage n-1, whose buffer the DMA of stage n+2 (issued in pieces
    // over the steps after the barrier) overwrites.  The MFMA stream of a wave runs across stage boundaries without a pause:
    // the first fragments of stage n+1 are read during the last step of stage n.
    // WAVES = 4: TWO buffers, the barrier at the START of stage n (stage n has landed, everyone has left stage n-1), then the
    // DMA of stage n+1 in pieces; the co-resident workgroup covers the pause at the barrier.
    constexpr int kSteps = kDmaCK / 2, kBar = kDmaBufs == 3 ? kSteps / 2 : 0;
    if (nstages > 0 && !(g.debug & 1)) {
        BASIC_MCONV_ISSUE(0);
        if (kDmaBufs == 3 && nstages > 1) BASIC_MCONV_ISSUE(1);
    }
    const int a_lane = (khalf * 32 + col) * 4, b_lane = khalf * kDmaPos + wave * 32 + col;
    float fb[2];
    f32x4 fa[2];
    if (kDmaBufs == 3) {
        __syncthreads();   // stages 0 and 1 have landed
        fb[0] = lds[kDmaAFloats + b_lane];
        fa[0] = *reinterpret_cast<const f32x4 *>(lds + a_lane);
    }
    int buf = 0;   // stage buffer of the current stage (stg % kDmaBufs)
    // one stage: 16 steps of { next step's LDS reads, 4 MFMAs }.  FIRST = first stage of a canonical block: the chain starts
    // from an inline-constant 0 (no register initialisation).
    auto run_stage = [&](int stg, auto first_tag) __attribute__((always_inline)) {
        constexpr bool kFirst = decltype(first_tag)::value;
        const int nbuf = buf == kDmaBufs - 1 ? 0 : buf + 1;
        const float *al = lds + buf * kDmaStage + a_lane, *bl = al + (kDmaAFloats + b_lane - a_lane);
        const float *aln = lds + nbuf * kDmaStage + a_lane, *bln = aln + (kDmaAFloats + b_lane - a_lane);
        const bool issuing = stg + (kDmaBufs - 1) < nstages && !(g.debug & 1);
#pragma unroll
        for (int st = 0; st < kSteps; ++st) {
            const int cur = st & 1, nxt = cur ^ 1;
            if (st == kBar) {
                __syncthreads();   // vmcnt(0): my DMAs of the next stage to be read have landed; barrier: everyone's have, and the oldest buffer is free
                if (issuing) BASIC_MCONV_ISSUE_BEGIN(kDmaBufs == 3 ? (buf == 0 ? 2 : buf - 1) : nbuf);
                if (kDmaBufs == 2) {
                    fb[0] = bl[0];
                    fa[0] = *reinterpret_cast<const f32x4 *>(al);
                }
            }
            if (st >= kBar && st < kBar + kSteps / 2 && issuing) {   // the next DMA, two or three pieces per step
                const int q = st - kBar;
                if (q < kAPieces) BASIC_MCONV_ISSUE_A(q);
                BASIC_MCONV_ISSUE_B(2 * q);
                BASIC_MCONV_ISSUE_B(2 * q + 1);
                if (q == kSteps / 2 - 1) BASIC_MCONV_ISSUE_END();
            }
            if (st + 1 < kSteps) {
                fb[nxt] = bl[(st + 1) * 2 * kDmaPos];
                fa[nxt] = *reinterpret_cast<const f32x4 *>(al + (st + 1) * 2 * 128);
            } else if (kDmaBufs == 3) {   // first fragments of the next stage (complete since this stage's barrier; a harmless read after the last stage)
                fb[nxt] = bln[0];
                fa[nxt] = *reinterpret_cast<const f32x4 *>(aln);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kFirst && st == 0) {
#pragma unroll
                for (int m = 0; m < 4; ++m) blk[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m], fb[cur], kZero16, 0, 0, 0);
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) blk[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m], fb[cur], blk[m], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        buf = nbuf;
    };
    static_assert(kKB == 2 * kDmaCK && (kDmaCK / 2) % 2 == 0, "a canonical block is two stages of an even number of steps");
    for (int stg = 0; stg < nstages; stg += 2) {     // gs_in % kKB == 0 (host-checked): every block is exactly two stages
        run_stage(stg, std::true_type{});
        run_stage(stg + 1, std::false_type{});
        // the finished block joins the total: tile m's adds run while the matrix core still works on tiles m+1 .. 3
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] += blk[m][r];
    }
#undef BASIC_MCONV_ISSUE
#undef BASIC_MCONV_ISSUE_BEGIN
#undef BASIC_MCONV_ISSUE_A
#undef BASIC_MCONV_ISSUE_B
#undef BASIC_MCONV_ISSUE_END

    if (mp.ok && !(g.debug & 4)) {
        float *yb = g.y + (static_cast<int64_t>(mp.b) * g.out_total + g.out_off + row0) * hw + out_slot(g, mp);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rg = m * 32 + 8 * (r >> 2) + 4 * khalf + (r & 3);
                yb[static_cast<int64_t>(rg) * hw] = activate(acc[m][r] + s_bias[rg], g.act);
            }
    }
}

constexpr int64_t kMaxUnits = 4096;      // (tile, block) units of the block kernel: 4 KB of scratch each, allocated on first use

}  // namespace

struct basic_mconv_plan {
    int cin = 0, cout = 0, coutp = 0, ksize = 1, gi = 1, go = 1, allow_same = 0, act = 0;
    int mt = 1;  // row tiles per wave of the gather kernel = pack factor of d_w
    float *d_w = nullptr, *d_bias = nullptr;
    float *d_w1 = nullptr;    // plain [tap][ci][co] copy for the block kernel when mt > 1 (coalesced 128-byte A rows)
    float *d_wa = nullptr;    // [row chunk of 128][tap][ci][32][4] for the dma kernel (nullptr when the layer does not qualify)
    // scratch of the block kernel (one stream per plan at a time); allocated by the first tiny launch
    mutable std::mutex mu;
    mutable float *d_scratch = nullptr;
    mutable int32_t *d_flags = nullptr;
};

extern "C" void basic_mconv_plan_destroy(basic_mconv_plan *p)
{
    if (!p) return;
    if (p->d_w) (void)hipFree(p->d_w);
    if (p->d_w1) (void)hipFree(p->d_w1);
    if (p->d_wa) (void)hipFree(p->d_wa);
    if (p->d_scratch) (void)hipFree(p->d_scratch);
    if (p->d_flags) (void)hipFree(p->d_flags);
    if (p->d_bias) (void)hipFree(p->d_bias);
    delete p;
}

static bool dma_layer_ok(const basic_mconv_plan *p)
{
    const int gs_in = p->cin / p->gi, gs_out = p->cout / p->go;
    return gs_in % kKB == 0 && gs_out % kDmaRows == 0 && p->ksize * p->ksize * p->gi <= kDmaMaxSlabs;
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
        p->mt = tiles <= 4 ? tiles : tiles % 4 == 0 ? 4 : tiles % 3 == 0 ? 3 : tiles % 2 == 0 ? 2 : 1;
    }
    if (const char *e = std::getenv("BASIC_MCONV_MAX_MT")) {  // tests: one tile per wave as the comparison point
        if (p->mt > std::atoi(e)) p->mt = 1;
    }
    if (const char *e = std::getenv("BASIC_MCONV_FORCE_MT")) {  // tuning: another pack factor where it divides the group's tiles
        const int f = std::atoi(e);
        if (f >= 1 && f <= 4 && gs_out % (32 * f) == 0) p->mt = f;
    }
    const int span = 32 * p->mt;
    std::vector<float> wp(static_cast<size_t>(ntaps) * cin * p->coutp, 0.f), hb(p->coutp, 0.f);
    std::vector<float> w1(p->mt > 1 ? wp.size() : 0, 0.f);
    const bool dma = dma_layer_ok(p);
    std::vector<float> wa(dma ? static_cast<size_t>(ntaps) * cin * cout : 0, 0.f);
    for (int o = 0; o < cout; ++o) {
        const int grp = o / gs_out, r = o - grp * gs_out;
        const int packed = grp * gs_out + (r / span) * span + (r % 32) * p->mt + (r / 32) % p->mt;
        const int chunk = o / kDmaRows, ro = o % kDmaRows;     // gs_out % 128 == 0: chunks never straddle groups
        for (int c = 0; c < cin; ++c)
            for (int t = 0; t < ntaps; ++t) {
                const float v = weight[(static_cast<size_t>(o) * cin + c) * ntaps + t];
                wp[(static_cast<size_t>(t) * cin + c) * p->coutp + packed] = v;
                if (p->mt > 1) w1[(static_cast<size_t>(t) * cin + c) * p->coutp + o] = v;
                if (dma) wa[((static_cast<size_t>(chunk) * ntaps + t) * cin + c) * 128 + (ro % 32) * 4 + ro / 32] = v;
            }
    }
    if (bias) std::memcpy(hb.data(), bias, sizeof(float) * cout);
    auto upload = [](float **dst, const std::vector<float> &src) {
        hipError_t e = hipMalloc(dst, src.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(*dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = upload(&p->d_w, wp);
    if (e == hipSuccess && p->mt > 1) e = upload(&p->d_w1, w1);
    if (e == hipSuccess && dma) e = upload(&p->d_wa, wa);
    if (e == hipSuccess) e = upload(&p->d_bias, hb);
    if (e == hipSuccess && dma) e = ensure_max_lds(reinterpret_cast<const void *>(&masked_conv_dma_kernel<8>));
    if (e == hipSuccess && dma) e = ensure_max_lds(reinterpret_cast<const void *>(&masked_conv_dma_kernel<4>));
    if (e != hipSuccess) { basic_mconv_plan_destroy(p); return hip_fail(e, "mconv_plan_create", __FILE__, __LINE__); }
    *out = p;
    return BASIC_OK;
}

static int mconv_forward(const basic_mconv_plan *p, const float *d_x, const int32_t *d_topo_in, const int32_t *d_topo_out,
                         int batch, int h, int w, const int32_t *d_pos, int64_t n_pos, float *d_y, int out_channels_total,
                         int out_channel_offset, int step, const int32_t *d_first, const int32_t *d_in_perm,
                         const int32_t *d_out_perm, void *hip_stream);

extern "C" int basic_mconv_forward_pos_dev(const basic_mconv_plan *p, const float *d_x, const int32_t *d_topo_in,
                                           const int32_t *d_topo_out, int batch, int h, int w, const int32_t *d_pos,
                                           int64_t n_pos, float *d_y, int out_channels_total, int out_channel_offset,
                                           void *hip_stream)
{
    return mconv_forward(p, d_x, d_topo_in, d_topo_out, batch, h, w, d_pos, n_pos, d_y, out_channels_total, out_channel_offset,
                         kNoStep, nullptr, nullptr, nullptr, hip_stream);
}

extern "C" int basic_mconv_forward_step_dev(const basic_mconv_plan *p, const float *d_x, const int32_t *d_topo_in,
                                            const int32_t *d_topo_out, int batch, int h, int w, const int32_t *d_pos,
                                            int64_t n_pos, float *d_y, int out_channels_total, int out_channel_offset,
                                            int step, const int32_t *d_first_step, void *hip_stream)
{
    BASIC_REQUIRE(d_first_step && step != kNoStep, "mconv_forward_step: first-step map and a step are required");
    return mconv_forward(p, d_x, d_topo_in, d_topo_out, batch, h, w, d_pos, n_pos, d_y, out_channels_total, out_channel_offset,
                         step, d_first_step, nullptr, nullptr, hip_stream);
}

extern "C" int basic_mconv_forward_ex_dev(const basic_mconv_plan *p, const float *d_x, const int32_t *d_topo_in,
                                          const int32_t *d_topo_out, int batch, int h, int w, const int32_t *d_pos,
                                          int64_t n_pos, float *d_y, int out_channels_total, int out_channel_offset,
                                          int use_step, int step, const int32_t *d_first_step, const int32_t *d_in_perm,
                                          const int32_t *d_out_perm, void *hip_stream)
{
    BASIC_REQUIRE(!use_step || (d_first_step && step != kNoStep), "mconv_forward_ex: first-step map and a step are required");
    BASIC_REQUIRE(p && (!d_in_perm || p->ksize == 1), "mconv_forward_ex: an input permutation needs a 1x1 layer (no neighbours)");
    return mconv_forward(p, d_x, d_topo_in, d_topo_out, batch, h, w, d_pos, n_pos, d_y, out_channels_total, out_channel_offset,
                         use_step ? step : kNoStep, use_step ? d_first_step : nullptr, d_in_perm, d_out_perm, hip_stream);
}

static int mconv_forward(const basic_mconv_plan *p, const float *d_x, const int32_t *d_topo_in, const int32_t *d_topo_out,
                         int batch, int h, int w, const int32_t *d_pos, int64_t n_pos, float *d_y, int out_channels_total,
                         int out_channel_offset, int step, const int32_t *d_first, const int32_t *d_in_perm,
                         const int32_t *d_out_perm, void *hip_stream)
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
    g.mt = p->mt; g.step = step; g.first = d_first; g.in_perm = d_in_perm; g.out_perm = d_out_perm;
    const unsigned ptiles = static_cast<unsigned>((n_pos + 31) / 32), rtiles = static_cast<unsigned>(g.tiles_per_group * g.go);
    hipStream_t st = as_stream(hip_stream);
    // Which kernel: every one of them sums in the canonical order, so this is a matter of speed only and may depend on the
    // launch size.  BASIC_MCONV_KERNEL = dma | gather | block forces one where it applies (tests drive all three over the
    // same inputs and require identical bits); BASIC_MCONV_BLOCK_BELOW / _DMA_FROM move the switch-overs (tile counts).
    const int64_t tiles = static_cast<int64_t>(ptiles) * rtiles;
    int64_t block_below = 256, dma_from = 4096;
    if (const char *e = std::getenv("BASIC_MCONV_BLOCK_BELOW")) block_below = std::atoll(e);
    if (const char *e = std::getenv("BASIC_MCONV_DMA_FROM")) dma_from = std::atoll(e);
    const char *force = std::getenv("BASIC_MCONV_KERNEL");
    const int64_t x_bytes = static_cast<int64_t>(batch) * p->cin * h * w * 4;
    const int blocks_per_slab = (g.gs_in + kKB - 1) / kKB;
    const int64_t units = static_cast<int64_t>(g.ntaps) * g.gi * blocks_per_slab;
    bool use_dma = p->d_wa && x_bytes < (1ll << 31) && tiles >= dma_from;
    bool use_block = !use_dma && tiles < block_below && tiles * units <= kMaxUnits;
    if (force) {
        if (!std::strcmp(force, "dma")) { use_dma = p->d_wa && x_bytes < (1ll << 31); use_block = false; }
        else if (!std::strcmp(force, "block")) { use_dma = false; use_block = tiles * units <= kMaxUnits; }
        else if (!std::strcmp(force, "gather")) { use_dma = false; use_block = false; }
    }
    if (use_dma) {
        g.w = p->d_wa;
        g.n_rchunks = p->cout / kDmaRows;
        g.x_bytes = static_cast<int>(x_bytes);
#ifdef BASIC_DEBUG_ABLATIONS   // timing ablations (wrong results): only in a library built with `make ABLATIONS=1`
        { const char *e = std::getenv("BASIC_MCONV_DEBUG"); g.debug = e ? std::atoi(e) : 0; }
#else
        g.debug = 0;
#endif
        // 4 waves (128 positions per workgroup, two workgroups per compute unit) or 8 (256 positions, one workgroup);
        // BASIC_MCONV_DMA_WAVES picks one
        // measured (scripts/mconv_probe.py, warm clocks): the 4-wave shape is never slower -- 0.899 vs 0.905 ms on the 1536 -> 1536
        // layer, 0.31 vs 0.40 / 0.48 vs 0.60 ms on the 384-row layers whose 8-wave grid is one and a half rounds of the chip
        int waves = 4;
        if (const char *e = std::getenv("BASIC_MCONV_DMA_WAVES")) waves = std::atoi(e) == 8 ? 8 : 4;
        const int ppw = 32 * waves;
        g.n_pchunks = static_cast<int>((n_pos + ppw - 1) / ppw);
        const unsigned grid = static_cast<unsigned>((g.n_pchunks + 7) / 8 * 8) * g.n_rchunks;
        if (waves == 8) hipLaunchKernelGGL(masked_conv_dma_kernel<8>, dim3(grid), dim3(512), dma_lds_bytes(8), st, g);
        else hipLaunchKernelGGL(masked_conv_dma_kernel<4>, dim3(grid), dim3(256), dma_lds_bytes(4), st, g);
    } else if (use_block) {
        {
            std::lock_guard<std::mutex> lock(p->mu);
            if (!p->d_scratch) {
                BASIC_HIP_TRY(hipMalloc(&p->d_scratch, static_cast<size_t>(kMaxUnits) * 16 * 64 * sizeof(float)));
                BASIC_HIP_TRY(hipMalloc(&p->d_flags, static_cast<size_t>(kMaxUnits) * sizeof(int32_t)));
            }
        }
        if (p->mt > 1) { g.w = p->d_w1; g.mt = 1; }
        g.scratch = p->d_scratch; g.flags = p->d_flags;
        g.units = static_cast<int>(units); g.blocks_per_slab = blocks_per_slab;
        hipLaunchKernelGGL(masked_conv_block_kernel, dim3(ptiles, rtiles, static_cast<unsigned>((units + 3) / 4)), dim3(256), 0, st, g);
        hipLaunchKernelGGL(masked_conv_reduce_kernel, dim3(ptiles, rtiles), dim3(256), 0, st, g);
    }
    else if (p->mt == 4)
        hipLaunchKernelGGL((masked_conv_gather_kernel<4>), dim3(ptiles, rtiles / 4), dim3(64), 0, st, g);
    else if (p->mt == 3)
        hipLaunchKernelGGL((masked_conv_gather_kernel<3>), dim3(ptiles, rtiles / 3), dim3(64), 0, st, g);
    else if (p->mt == 2)
        hipLaunchKernelGGL((masked_conv_gather_kernel<2>), dim3(ptiles, rtiles / 2), dim3(64), 0, st, g);
    else
        hipLaunchKernelGGL((masked_conv_gather_kernel<1>), dim3(ptiles, rtiles), dim3(64), 0, st, g);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}
