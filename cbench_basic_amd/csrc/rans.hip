// rANS (64-bit state, 32-bit words) for gfx950: tables, batched stream kernels and the
// host-buffer drop-in entry points.
//
// Bitstream contract (bit-exact with the reference):
//   tables   csrc/ans/rans64.cpp:69-182      encoder  csrc/ans/rans64.cpp:203-361
//   decoder  csrc/ans/rans64.cpp:389-598     raw bits csrc/ans/rans64.cpp:29-65
//   core     csrc/ans/rans64.h:59-142 (ryg rans64)
//
// MI355X mapping: a stream is a strictly serial chain, so parallelism is ACROSS streams:
// one 64-lane wavefront per stream (one image), hundreds of streams per launch.  A lone wave issues
// one instruction per ~8-11 cycles whatever it is (scripts/micro/lone_wave_latency.hip), so the
// kernels are written for INSTRUCTION COUNT per symbol.  The fast paths make the state update itself
// lane-parallel: every lane evaluates the update for "its" symbol (decoder: candidate symbol l of
// the row; encoder: symbol j of the 64-symbol chunk) against the current wave-uniform state with a
// few VALU ops, and only the chosen lane's new state is broadcast back; renormalisation, bypass
// symbols and wide rows hide behind one rare-path test.  The generic kernels (AR remap, any table)
// keep the state update on the scalar unit.
#include "common.h"
#include "wave_decoder.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

using namespace basic;

// ---------------------------------------------------------------------------------------
// Host side: tables
// ---------------------------------------------------------------------------------------
struct basic_rans_tables {
    int rows = 0, stride = 0;
    int precision = 16, bypass = 1, bypass_precision = 4;
    std::vector<int32_t> cdfs, sizes, offsets;  // host copies
    int32_t *d_cdfs = nullptr, *d_sizes = nullptr, *d_offsets = nullptr;
    int ar_k = 0, ar_rows = 0, ar_order = 0, ar_s1 = 0;
    bool ar_custom = false;   // d_ar holds float [k][7] custom-op parameters (init_custom_ar_ops) instead of a remap table
    std::vector<int32_t> ar;
    int32_t *d_ar = nullptr;
    // Packed copy for the LDS-resident decoder: rows back to back as uint16 (the final entry 2^precision
    // does not fit 16 bits when precision == 16; it is implied by its position), row r at base[r].
    std::vector<uint16_t> cdf16;
    std::vector<int32_t> base;
    uint16_t *d_cdf16 = nullptr;
    int32_t *d_base = nullptr;
    // Search image of the fast decoder (uint32, copied to LDS by every workgroup), 16 bytes per lane.
    // Rows of <= 64 entries: lane l = { entry l, start of symbol l-1, frequency of symbol l-1, 0 } (INT_MAX keys
    // past the row; frequency 0 for the bypass sentinel, so that its candidate state always looks "rare").
    // Wider rows: one lane { INT_MAX, 0, 0, 0 }, the last entry of each of 64 blocks, the full row.
    // meta[r] = byte offset of the row in the image.
    std::vector<uint32_t> image, meta;
    uint32_t *d_image = nullptr, *d_meta = nullptr;
    bool fast_ok = false;
    // Encoder image of the fast encoder: one 16-byte entry per (row, value) --
    //   x = freq | (2^precision - freq) << 16,  y = post_shift | start' << 8,  z,w = 64-bit reciprocal
    // with start' = start (+ 2^precision - 1 for freq == 1, whose reciprocal 2^64-1 yields q = x - 1) --
    // and per row (offset, max_value).  Everything the serial chain needs for a symbol is one gather.
    std::vector<uint32_t> enc;      // [rows][stride][4]
    std::vector<int32_t> rowinfo;   // [rows][2]
    uint32_t *d_enc = nullptr;
    int32_t *d_rowinfo = nullptr;
    bool fast_enc_ok = false;
};

namespace {

// float32 pmf -> integer cdf summing to 2^precision (reference: rans64.cpp:69-126).
// All arithmetic is kept in the reference's types: float product, round-half-away, u32 total,
// u64 rescale, then the "steal one count from the narrowest bin wider than 1" repair.
int quantize_pmf(const float *pmf, int n, int precision, int32_t *cdf)
{
    const uint64_t one = 1ull << precision;
    cdf[0] = 0;
    uint32_t total = 0;
    for (int i = 0; i < n; ++i) {
        cdf[i + 1] = static_cast<int32_t>(std::round(pmf[i] * static_cast<float>(1 << precision)));
        total += static_cast<uint32_t>(cdf[i + 1]);
    }
    if (total == 0) return BASIC_ERR_INVALID;
    int32_t run = 0;
    for (int i = 1; i <= n; ++i) {
        run += static_cast<int32_t>((one * static_cast<uint64_t>(static_cast<int64_t>(cdf[i]))) / total);
        cdf[i] = run;
    }
    cdf[n] = static_cast<int32_t>(one);
    for (int i = 0; i < n; ++i) {
        if (cdf[i] != cdf[i + 1]) continue;
        uint32_t narrowest = ~0u;
        int donor = -1;
        for (int j = 0; j < n; ++j) {
            const uint32_t width = static_cast<uint32_t>(cdf[j + 1] - cdf[j]);
            if (width > 1 && width < narrowest) { narrowest = width; donor = j; }
        }
        if (donor < 0) return BASIC_ERR_INVALID;
        if (donor < i) for (int j = donor + 1; j <= i; ++j) cdf[j] -= 1;
        else           for (int j = i + 1; j <= donor; ++j) cdf[j] += 1;
    }
    return BASIC_OK;
}

int upload_tables(basic_rans_tables *t)
{
    int rc = require_device();
    if (rc) return rc;
    BASIC_HIP_TRY(hipMalloc(&t->d_cdfs, t->cdfs.size() * sizeof(int32_t)));
    BASIC_HIP_TRY(hipMalloc(&t->d_sizes, t->sizes.size() * sizeof(int32_t)));
    BASIC_HIP_TRY(hipMalloc(&t->d_offsets, t->offsets.size() * sizeof(int32_t)));
    BASIC_HIP_TRY(hipMemcpy(t->d_cdfs, t->cdfs.data(), t->cdfs.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    BASIC_HIP_TRY(hipMemcpy(t->d_sizes, t->sizes.data(), t->sizes.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    BASIC_HIP_TRY(hipMemcpy(t->d_offsets, t->offsets.data(), t->offsets.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    t->base.resize(t->rows);
    t->cdf16.clear();
    for (int r = 0; r < t->rows; ++r) {
        t->base[r] = static_cast<int32_t>(t->cdf16.size());
        for (int j = 0; j < t->sizes[r]; ++j) t->cdf16.push_back(static_cast<uint16_t>(t->cdfs[static_cast<size_t>(r) * t->stride + j]));
    }
    if (t->cdf16.size() & 1) t->cdf16.push_back(0);  // whole 32-bit words for the LDS copy
    BASIC_HIP_TRY(hipMalloc(&t->d_cdf16, t->cdf16.size() * sizeof(uint16_t)));
    BASIC_HIP_TRY(hipMalloc(&t->d_base, t->base.size() * sizeof(int32_t)));
    BASIC_HIP_TRY(hipMemcpy(t->d_cdf16, t->cdf16.data(), t->cdf16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    BASIC_HIP_TRY(hipMemcpy(t->d_base, t->base.data(), t->base.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    // fast-decoder image
    t->image.clear();
    t->meta.resize(t->rows);
    t->fast_ok = true;
    for (int r = 0; r < t->rows; ++r) {
        const int size = t->sizes[r];
        const int32_t *row = &t->cdfs[static_cast<size_t>(r) * t->stride];
        if (size > 4096) { t->fast_ok = false; break; }
        t->meta[r] = static_cast<uint32_t>(t->image.size() * 4);  // byte offset
        if (size <= 64) {
            // lane l: { entry l (search key), start and frequency of symbol l - 1, pad }.  Only the row's `size` lanes are
            // stored: the lanes past them read whatever follows (the next row, the pad behind the last one), and are never
            // selected -- the first lane whose key exceeds the coded value lies inside the row, whose last key is the total
            // 2^precision.  A set of many short rows (a factorised prior: one row per channel) then still fits the LDS.
            for (int l = 0; l < size; ++l) {
                uint32_t key = 0x7FFFFFFFu, st = 0, fq = 0;
                if (l < size) key = static_cast<uint32_t>(row[l]);
                if (l >= 1 && l < size) {
                    st = static_cast<uint32_t>(row[l - 1]);
                    fq = key - st;
                    if (t->bypass && l - 1 == size - 2) fq = 0;  // the sentinel symbol always takes the rare path
                }
                t->image.push_back(key); t->image.push_back(st); t->image.push_back(fq); t->image.push_back(0);
            }
        } else {
            // lane 0 = { INT_MAX, 0, 0 }: every lookup selects it and lands on the rare path; then the 64
            // block-end probes and the row (lanes 1..63 of the fast path read into them, harmlessly)
            t->image.push_back(0x7FFFFFFFu); t->image.push_back(0); t->image.push_back(0); t->image.push_back(0);
            const int step = (size + 63) >> 6;
            for (int l = 0; l < 64; ++l) {
                int e = (l + 1) * step - 1;
                if (e > size - 1) e = size - 1;
                t->image.push_back(static_cast<uint32_t>(row[e]));
            }
            for (int j = 0; j < size; ++j) t->image.push_back(static_cast<uint32_t>(row[j]));
            while (t->image.size() & 3) t->image.push_back(0);  // rows start 16-byte aligned
        }
    }
    for (int l = 0; l < 256; ++l) t->image.push_back(0x7FFFFFFFu);  // lanes 1..63 of a trailing wide row stay inside
    if (t->image.size() * 4 > 156 * 1024) t->fast_ok = false;
    if (t->fast_ok) {
        BASIC_HIP_TRY(hipMalloc(&t->d_image, t->image.size() * sizeof(uint32_t)));
        BASIC_HIP_TRY(hipMalloc(&t->d_meta, t->meta.size() * sizeof(uint32_t)));
        BASIC_HIP_TRY(hipMemcpy(t->d_image, t->image.data(), t->image.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        BASIC_HIP_TRY(hipMemcpy(t->d_meta, t->meta.data(), t->meta.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    // fast-encoder image
    t->fast_enc_ok = t->rows <= 2048 && static_cast<size_t>(t->rows) * t->stride <= (4u << 20);
    if (t->fast_enc_ok) {
        const uint32_t one = 1u << t->precision;
        t->enc.assign(static_cast<size_t>(t->rows) * t->stride * 4, 0u);
        t->rowinfo.resize(static_cast<size_t>(t->rows) * 2);
        for (int r = 0; r < t->rows && t->fast_enc_ok; ++r) {
            const int32_t *row = &t->cdfs[static_cast<size_t>(r) * t->stride];
            t->rowinfo[2 * r] = t->offsets[r];
            t->rowinfo[2 * r + 1] = t->sizes[r] - 2;
            for (int v = 0; v + 1 < t->sizes[r]; ++v) {
                const uint32_t start = static_cast<uint32_t>(row[v]) & 0xFFFFu;           // uint16_t casts, rans64.cpp:289-291
                const uint32_t freq = static_cast<uint32_t>(row[v + 1] - row[v]) & 0xFFFFu;
                if (freq == 0 || freq >= one + (t->precision == 16 ? 0u : 1u) || freq > 0xFFFFu) { t->fast_enc_ok = false; break; }
                uint64_t rcp = ~0ull;
                uint32_t post_shift = 0, start_adj = start + (one - 1u);
                if (freq >= 2u) {
                    uint32_t shift = 0;
                    while ((1u << shift) < freq) ++shift;  // ceil(log2 freq)
                    const unsigned __int128 num = (static_cast<unsigned __int128>(1) << (63 + shift)) + (freq - 1u);
                    rcp = static_cast<uint64_t>(num / freq);
                    post_shift = shift - 1u;
                    start_adj = start;
                }
                uint32_t *e = &t->enc[(static_cast<size_t>(r) * t->stride + v) * 4];
                e[0] = freq | ((one - freq) << 16);
                e[1] = post_shift | (start_adj << 8);
                e[2] = static_cast<uint32_t>(rcp);
                e[3] = static_cast<uint32_t>(rcp >> 32);
            }
        }
    }
    if (t->fast_enc_ok) {
        BASIC_HIP_TRY(hipMalloc(&t->d_enc, t->enc.size() * sizeof(uint32_t)));
        BASIC_HIP_TRY(hipMalloc(&t->d_rowinfo, t->rowinfo.size() * sizeof(int32_t)));
        BASIC_HIP_TRY(hipMemcpy(t->d_enc, t->enc.data(), t->enc.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        BASIC_HIP_TRY(hipMemcpy(t->d_rowinfo, t->rowinfo.data(), t->rowinfo.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    t->enc.clear(); t->enc.shrink_to_fit();
    return BASIC_OK;
}

bool params_ok(int precision, int bypass_precision)
{
    return precision >= 1 && precision <= 16 && bypass_precision >= 1 && bypass_precision <= 16;
}

}  // namespace

extern "C" int basic_pmf_to_quantized_cdf(const float *pmf, int n, int precision, int32_t *cdf_out)
{
    BASIC_REQUIRE(pmf && cdf_out && n >= 1 && precision >= 1 && precision <= 30, "pmf_to_quantized_cdf: bad argument");
    int rc = quantize_pmf(pmf, n, precision, cdf_out);
    if (rc) set_error("pmf_to_quantized_cdf: degenerate pmf");
    return rc;
}

extern "C" int basic_rans_tables_from_freqs(const int32_t *freqs, int rows, int freq_stride, const int32_t *nsym,
                                            const int32_t *offsets, int freq_precision, int bypass_coding,
                                            int bypass_precision, basic_rans_tables **out)
{
    BASIC_REQUIRE(freqs && nsym && offsets && out && rows >= 1, "init_params: null/empty argument");
    BASIC_REQUIRE(params_ok(freq_precision, bypass_precision), "init_params: precision out of range");
    int max_n = 0;
    for (int r = 0; r < rows; ++r) {
        BASIC_REQUIRE(nsym[r] >= 1 && nsym[r] <= freq_stride,
                      "freqs should be 2-dimensional with shape (num_symbols.size(), >num_symbols.max())");
        max_n = std::max(max_n, nsym[r]);
    }
    auto *t = new (std::nothrow) basic_rans_tables();
    if (!t) { set_error("out of host memory"); return BASIC_ERR_INVALID; }
    t->rows = rows;
    t->stride = max_n + 2;
    t->precision = freq_precision;
    t->bypass = bypass_coding ? 1 : 0;
    t->bypass_precision = bypass_precision;
    t->cdfs.assign(static_cast<size_t>(rows) * t->stride, 0);
    t->sizes.resize(rows);
    t->offsets.assign(offsets, offsets + rows);
    std::vector<float> pmf(max_n + 1);
    for (int r = 0; r < rows; ++r) {
        // rans64.cpp:141-151: float accumulation of the counts, tail mass 1 appended.
        const int n = nsym[r];
        const int32_t *f = freqs + static_cast<size_t>(r) * freq_stride;
        float tot = 0.0f;
        for (int i = 0; i < n; ++i) tot += static_cast<float>(f[i]);
        tot += 1.0f;
        for (int i = 0; i < n; ++i) pmf[i] = static_cast<float>(f[i]) / tot;
        pmf[n] = 1.0f / tot;
        if (quantize_pmf(pmf.data(), n + 1, freq_precision, &t->cdfs[static_cast<size_t>(r) * t->stride])) {
            delete t;
            set_error("init_params: degenerate frequency row");
            return BASIC_ERR_INVALID;
        }
        t->sizes[r] = n + 2;
    }
    int rc = upload_tables(t);
    if (rc) { basic_rans_tables_destroy(t); return rc; }
    *out = t;
    return BASIC_OK;
}

extern "C" int basic_rans_tables_from_cdfs(const int32_t *cdfs, int rows, int cdf_stride, const int32_t *cdf_sizes,
                                           const int32_t *offsets, int freq_precision, int bypass_coding,
                                           int bypass_precision, basic_rans_tables **out)
{
    BASIC_REQUIRE(cdfs && cdf_sizes && offsets && out && rows >= 1, "init_cdf_params: null/empty argument");
    BASIC_REQUIRE(params_ok(freq_precision, bypass_precision), "init_cdf_params: precision out of range");
    int max_len = 0;
    for (int r = 0; r < rows; ++r) {
        BASIC_REQUIRE(cdf_sizes[r] >= 2 && cdf_sizes[r] <= cdf_stride,
                      "cdfs should be 2-dimensional with shape (cdfs_sizes.size(), >cdfs_sizes.max())");
        max_len = std::max(max_len, cdf_sizes[r]);
    }
    auto *t = new (std::nothrow) basic_rans_tables();
    if (!t) { set_error("out of host memory"); return BASIC_ERR_INVALID; }
    t->rows = rows;
    t->stride = max_len;
    t->precision = freq_precision;
    t->bypass = bypass_coding ? 1 : 0;
    t->bypass_precision = bypass_precision;
    t->cdfs.assign(static_cast<size_t>(rows) * t->stride, 0);
    for (int r = 0; r < rows; ++r)
        std::memcpy(&t->cdfs[static_cast<size_t>(r) * t->stride], cdfs + static_cast<size_t>(r) * cdf_stride,
                    sizeof(int32_t) * cdf_sizes[r]);
    t->sizes.assign(cdf_sizes, cdf_sizes + rows);
    t->offsets.assign(offsets, offsets + rows);
    int rc = upload_tables(t);
    if (rc) { basic_rans_tables_destroy(t); return rc; }
    *out = t;
    return BASIC_OK;
}

extern "C" int basic_rans_tables_set_ar(basic_rans_tables *t, const int32_t *ar_tab, int k, int rows, int order, int s1)
{
    BASIC_REQUIRE(t && ar_tab && k >= 1 && rows >= 1 && s1 >= 1, "init_ar_params: null/empty argument");
    BASIC_REQUIRE(order == 1 || order == 2, "Too many dimensions!");
    size_t n = static_cast<size_t>(k) * rows * s1 * (order == 2 ? s1 : 1);
    t->ar.assign(ar_tab, ar_tab + n);
    t->ar_k = k; t->ar_rows = rows; t->ar_order = order; t->ar_s1 = s1;
    t->ar_custom = false;
    if (t->d_ar) { (void)hipFree(t->d_ar); t->d_ar = nullptr; }
    BASIC_HIP_TRY(hipMalloc(&t->d_ar, n * sizeof(int32_t)));
    BASIC_HIP_TRY(hipMemcpy(t->d_ar, t->ar.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    return BASIC_OK;
}

// ANSBase::init_custom_ar_ops (ans_interface.hpp:40-48): k ops of type ar_limited_scaled_add_linear_op, ops float [k][7] =
// (w0, w1, w2, bias, scale, min, max).  The number of predecessors (1..3) is the number of ar_offsets rows of a call.
extern "C" int basic_rans_tables_set_ar_ops(basic_rans_tables *t, const float *ops, int k)
{
    BASIC_REQUIRE(t && ops && k >= 1, "init_custom_ar_ops: null/empty argument");
    for (int i = 0; i < k; ++i) BASIC_REQUIRE(ops[i * 7 + 4] != 0.f, "init_custom_ar_ops: scale must not be zero");
    if (t->d_ar) { (void)hipFree(t->d_ar); t->d_ar = nullptr; }
    BASIC_HIP_TRY(hipMalloc(&t->d_ar, static_cast<size_t>(k) * 7 * sizeof(float)));
    BASIC_HIP_TRY(hipMemcpy(t->d_ar, ops, static_cast<size_t>(k) * 7 * sizeof(float), hipMemcpyHostToDevice));
    t->ar_k = k; t->ar_rows = t->rows; t->ar_order = 0; t->ar_s1 = 0;
    t->ar_custom = true;
    return BASIC_OK;
}

extern "C" int basic_rans_tables_info(const basic_rans_tables *t, int *rows, int *max_cdf_len)
{
    BASIC_REQUIRE(t, "ANS not initialized!");
    if (rows) *rows = t->rows;
    if (max_cdf_len) *max_cdf_len = *std::max_element(t->sizes.begin(), t->sizes.end());
    return BASIC_OK;
}

extern "C" int basic_rans_tables_get_cdfs(const basic_rans_tables *t, int32_t *out, int out_stride)
{
    BASIC_REQUIRE(t && out, "ANS not initialized!");
    for (int r = 0; r < t->rows; ++r) {
        BASIC_REQUIRE(t->sizes[r] <= out_stride, "get_cdfs: output stride too small");
        int32_t *dst = out + static_cast<size_t>(r) * out_stride;
        std::memset(dst, 0, sizeof(int32_t) * out_stride);
        std::memcpy(dst, &t->cdfs[static_cast<size_t>(r) * t->stride], sizeof(int32_t) * t->sizes[r]);
    }
    return BASIC_OK;
}

extern "C" void basic_rans_tables_destroy(basic_rans_tables *t)
{
    if (!t) return;
    if (t->d_cdfs) (void)hipFree(t->d_cdfs);
    if (t->d_sizes) (void)hipFree(t->d_sizes);
    if (t->d_offsets) (void)hipFree(t->d_offsets);
    if (t->d_ar) (void)hipFree(t->d_ar);
    if (t->d_cdf16) (void)hipFree(t->d_cdf16);
    if (t->d_base) (void)hipFree(t->d_base);
    if (t->d_image) (void)hipFree(t->d_image);
    if (t->d_meta) (void)hipFree(t->d_meta);
    if (t->d_enc) (void)hipFree(t->d_enc);
    if (t->d_rowinfo) (void)hipFree(t->d_rowinfo);
    delete t;
}

// ---------------------------------------------------------------------------------------
// Device side
// ---------------------------------------------------------------------------------------
namespace {

constexpr uint64_t kRansL = 1ull << 31;
typedef uint32_t f32x4u __attribute__((ext_vector_type(4)));  // one 16-byte entry of the decoder image

struct TablesDev {
    const int32_t *cdfs, *sizes, *offsets;
    int rows, stride, precision, bypass, bypass_precision;
    const uint16_t *cdf16;  // packed rows (see basic_rans_tables)
    const int32_t *base;
    int total16;            // entries in cdf16 (even)
    const uint32_t *image, *meta;  // fast-decoder search image (see basic_rans_tables)
    int image_words;
    const uint4 *enc;        // fast-encoder image [rows][stride] (see basic_rans_tables)
    const int2 *rowinfo;     // [rows] (offset, max_value)
};

// Alternative to a seg[] array: stream b codes the elements [first + b*stride, first + b*stride + count).
struct StridedSeg { int64_t first, stride, count; };

struct ArDev {
    const int32_t *tab;  // nullptr = no AR remap
    int k, order, rows, s1;
    const int32_t *ar_indexes, *off0, *off1;  // per stream-element arrays (global element ids)
    const int32_t *off2;                       // third back distance (custom ops only)
    int custom;                                // tab = float [k][7]: w0 w1 w2 bias scale min max (init_custom_ar_ops)
};


__device__ __forceinline__ int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

// AR remap of the table row (ans_interface.hpp:89-104); every index is clamped so that a
// malformed input can never read outside the tables.
__device__ __forceinline__ int32_t ar_row(const ArDev &ar, int32_t a, int32_t row, int32_t v0, int32_t v1)
{
    a = clampi(a, 0, ar.k - 1);
    row = clampi(row, 0, ar.rows - 1);
    v0 = clampi(v0, 0, ar.s1 - 1);
    if (ar.order == 1) return ar.tab[(static_cast<int64_t>(a) * ar.rows + row) * ar.s1 + v0];
    v1 = clampi(v1, 0, ar.s1 - 1);
    return ar.tab[((static_cast<int64_t>(a) * ar.rows + row) * ar.s1 + v0) * ar.s1 + v1];
}

// ar_limited_scaled_add_linear_op::op on (index, the RAW previous symbols), csrc/ans/ar_funcs.hpp:58-87 called from
// ar_update_index (ans_interface.hpp:60-84).  float arithmetic with every product and sum rounded on its own, as the
// reference's x86 build does (no fused multiply-add: the intrinsics keep hipcc from contracting).
__device__ __forceinline__ int32_t ar_custom_row(const ArDev &ar, int32_t a, int32_t row, int32_t v0, int32_t v1, int32_t v2)
{
    const float *op = reinterpret_cast<const float *>(ar.tab) + static_cast<int64_t>(clampi(a, 0, ar.k - 1)) * 7;
    const float base = static_cast<float>(row), scale = op[4];
    const float base_unscaled = floorf(__fdiv_rn(base, scale));
    float adder = __fadd_rn(0.f, __fmul_rn(static_cast<float>(v0), op[0]));
    if (ar.order > 1) adder = __fadd_rn(adder, __fmul_rn(static_cast<float>(v1), op[1]));
    if (ar.order > 2) adder = __fadd_rn(adder, __fmul_rn(static_cast<float>(v2), op[2]));
    adder = __fadd_rn(adder, op[3]);
    float lim = __fadd_rn(base_unscaled, adder);
    lim = lim < op[6] ? lim : op[6];
    lim = op[5] > lim ? op[5] : lim;
    const float step = __fmul_rn(__fsub_rn(roundf(lim), base_unscaled), scale);
    return static_cast<int32_t>(__fadd_rn(base, step));
}

__device__ __forceinline__ uint32_t bcast_u32(uint32_t v, int lane)
{
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), lane));
}
__device__ __forceinline__ uint64_t bcast_u64(uint64_t v, int lane)
{
    return static_cast<uint64_t>(bcast_u32(static_cast<uint32_t>(v), lane)) |
           (static_cast<uint64_t>(bcast_u32(static_cast<uint32_t>(v >> 32), lane)) << 32);
}
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v)
{
    uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
    uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
    return static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32);
}

// ceil(2^(63+shift) / f) for f >= 2 with shift = ceil(log2 f): the 64-bit reciprocal that makes
// q = mulhi(x, rcp) >> (shift-1) == floor(x / f) for every x < 2^63 (Alverson 1991).  Built from
// two 64-by-32 divisions, per lane, in the parallel phase.
__device__ __forceinline__ void exact_reciprocal(uint32_t f, uint64_t &rcp, uint32_t &post_shift)
{
    const uint32_t shift = 32u - static_cast<uint32_t>(__clz(static_cast<int>(f - 1u)));
    const uint64_t hi_num = 1ull << (shift + 31u);  // top 64 bits of 2^(63+shift) split as hi:lo words
    const uint64_t q_hi = hi_num / f;
    const uint64_t lo_num = ((hi_num % f) << 32) + (f - 1u);
    const uint64_t q_lo = lo_num / f;
    rcp = (q_hi << 32) + q_lo;
    post_shift = shift - 1u;
}

struct Emitter {
    uint32_t *p;   // next free word is p[-1]; wave-uniform
    uint32_t *lo;  // slot start
    bool overflow;
    __device__ __forceinline__ void push(uint32_t w, int lane)
    {
        if (p == lo) { overflow = true; return; }
        --p;
        if (lane == 0) *p = w;
    }
};

// The fast encoder's emitter: the word a symbol's renormalisation emits is parked in THAT SYMBOL'S lane of a register (an
// immediate lane: no m0, no counter) and a bit of a 64-bit mask is set; the words leave with one coalesced store per chunk of
// 64 symbols (and in front of a bypass value's escape code), so a renormalisation on the serial chain issues no memory
// instruction.  Symbols are folded from lane 63 down and the stream is written towards lower addresses: the word parked in
// lane l lands at p[-1 - (number of parked lanes above l)].
struct WaveEmitter {
    uint32_t *p;   // next free word is p[-1]; wave-uniform
    uint32_t *lo;  // slot start
    bool overflow;
    uint32_t parked;   // lane j: the word symbol j emitted (where its mask bit is set)
    uint64_t mask;     // uniform
    __device__ __forceinline__ static uint32_t *store_parked(uint32_t *p, uint32_t *lo, uint32_t parked, uint64_t mask, int lane, bool &overflow)
    {
        const int n = __builtin_popcountll(mask);
        if (p - lo < n) { overflow = true; return p; }
        const uint64_t above = (mask >> lane) >> 1;   // parked lanes above this one: emitted earlier
        if ((mask >> lane) & 1ull) p[-1 - __builtin_popcountll(above)] = parked;
        return p - n;
    }
    __device__ __forceinline__ void flush(int lane)
    {
        if (mask == 0ull) return;
        p = store_parked(p, lo, parked, mask, lane, overflow);
        mask = 0ull;
    }
    __device__ __forceinline__ void push_direct(uint32_t w, int lane)   // straight to memory; nothing may be parked (the final state)
    {
        if (p == lo) { overflow = true; return; }
        --p;
        if (lane == 0) *p = w;
    }
};

// x = C(s, x) with renormalisation (rans64.h:65-84).  All operands are wave-uniform.
__device__ __forceinline__ void put_symbol(uint64_t &x, Emitter &em, int lane, uint32_t start, uint32_t freq,
                                           uint64_t rcp, uint32_t post_shift, uint32_t precision)
{
    const uint64_t x_max = static_cast<uint64_t>(freq) << (63u - precision);  // ((L>>prec)<<32)*freq
    if (x >= x_max) { em.push(static_cast<uint32_t>(x), lane); x >>= 32; }
    uint64_t q;
    if (freq == 1u) q = x;
    else q = __umul64hi(x, rcp) >> post_shift;
    x = x + start + q * ((1u << precision) - freq);  // (q << prec) + (x - q*freq) + start
}

template <class Em> __device__ __forceinline__ void put_raw(uint64_t &x, Em &em, int lane, uint32_t val, uint32_t nbits)
{ /* rans64.cpp:29-47 */
    const uint64_t x_max = 1ull << (63u - nbits);  // ((L>>16)<<32) << (16-nbits)
    if (x >= x_max) { em.push(static_cast<uint32_t>(x), lane); x >>= 32; }
    x = (x << nbits) | val;
}

// A bypass value's escape code in front of its sentinel symbol (written reversed: the decoder reads the sentinel, the count
// nibbles, then the payload low-first; rans64.cpp:312-339), straight to memory behind the words parked so far;
// then the sentinel's own renormalisation.  Not inlined, everything by value: the fast encoder keeps its state in named
// scalar registers, and these loops would pull it into vector registers (their integer division runs on the vector unit).
struct EscapeResult { uint64_t x; uint32_t *p; int overflow; };
__device__ __noinline__ EscapeResult encode_escape(uint64_t x, uint32_t *p, uint32_t *lo, int lane, uint32_t parked, uint64_t parked_mask, uint32_t r, uint32_t bprec,
                                                   uint32_t maxbv, uint32_t x_max_high)
{
    Emitter direct{p, lo, false};
    if (parked_mask != 0ull) {   // the words parked so far leave first (WaveEmitter::flush)
        bool overflow = false;
        direct.p = WaveEmitter::store_parked(p, lo, parked, parked_mask, lane, overflow);
        if (overflow) return EscapeResult{x, p, 1};
    }
    int nb = 0;
    while (nb * bprec < 32u && (r >> (nb * bprec)) != 0u) ++nb;
    for (int k = nb - 1; k >= 0; --k) put_raw(x, direct, lane, (r >> (k * bprec)) & maxbv, bprec);
    put_raw(x, direct, lane, static_cast<uint32_t>(nb) % maxbv, bprec);
    for (uint32_t k = 0; k < static_cast<uint32_t>(nb) / maxbv; ++k) put_raw(x, direct, lane, maxbv, bprec);
    if (static_cast<uint32_t>(x >> 32) >= x_max_high) {
        direct.push(static_cast<uint32_t>(x), lane);
        x >>= 32;
    }
    return EscapeResult{x, direct.p, direct.overflow ? 1 : 0};
}

// One wavefront per stream.  Symbols are consumed last-to-first in chunks of 64: the lanes
// prepare (start, freq, reciprocal, raw bypass payload) for their own symbol, then the chunk is
// folded into the state serially on broadcast (scalar) values.
__global__ __launch_bounds__(64) void rans_encode_kernel(TablesDev T, ArDev ar, const int32_t *__restrict__ symbols,
                                                         const int32_t *__restrict__ indexes,
                                                         const int64_t *__restrict__ seg, uint32_t *out_words,
                                                         int64_t slot_words, int32_t *out_nwords)
{
    const int stream = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t beg = seg[stream];
    const int64_t n = seg[stream + 1] - beg;
    const int32_t *sym = symbols + beg;
    const int32_t *idx = indexes + beg;
    uint32_t *slot = out_words + static_cast<int64_t>(stream) * slot_words;
    Emitter em{slot + slot_words, slot, false};
    uint64_t x = kRansL;
    const uint32_t prec = static_cast<uint32_t>(T.precision);
    const uint32_t bprec = static_cast<uint32_t>(T.bypass_precision);
    const uint32_t maxbv = (1u << bprec) - 1u;

    for (int64_t hi = n; hi > 0; hi -= 64) {
        const int64_t i = hi - 64 + lane;  // lane 63 holds the chunk's last symbol
        uint32_t start = 0, freq = 1, raw = 0, post_shift = 0;
        uint64_t rcp = 0;
        bool is_bypass = false;
        if (i >= 0) {
            int32_t row = idx[i];
            if (ar.tab) {
                const int64_t g = beg + i;
                const int32_t a = ar.ar_indexes ? ar.ar_indexes[g] : 0;
                const int32_t one = ar.custom ? 0 : 1;   // the table flavour indexes with symbol + 1 (0 = no predecessor)
                const int32_t d0 = ar.off0[g];
                const int32_t v0 = (d0 > 0 && d0 <= i) ? sym[i - d0] + one : 0;
                int32_t v1 = 0, v2 = 0;
                if (ar.order >= 2) {
                    const int32_t d1 = ar.off1[g];
                    v1 = (d1 > 0 && d1 <= i) ? sym[i - d1] + one : 0;
                }
                if (ar.order >= 3) {
                    const int32_t d2 = ar.off2[g];
                    v2 = (d2 > 0 && d2 <= i) ? sym[i - d2] + one : 0;
                }
                row = ar.custom ? ar_custom_row(ar, a, row, v0, v1, v2) : ar_row(ar, a, row, v0, v1);
            }
            row = clampi(row, 0, T.rows - 1);
            const int32_t *cdf = T.cdfs + static_cast<int64_t>(row) * T.stride;
            const int32_t max_value = T.sizes[row] - 2;
            int32_t value = sym[i] - T.offsets[row];
            if (T.bypass) {
                if (value < 0) { raw = static_cast<uint32_t>(-2 * value - 1); value = max_value; }
                else if (value >= max_value) { raw = static_cast<uint32_t>(2 * (value - max_value)); value = max_value; }
                is_bypass = (value == max_value);
            }
            value = clampi(value, 0, max_value);  // without bypass the reference is UB out of range
            const int32_t c0 = cdf[value], c1 = cdf[value + 1];
            start = static_cast<uint32_t>(c0) & 0xFFFFu;          // uint16_t casts, rans64.cpp:289-291
            freq = static_cast<uint32_t>(c1 - c0) & 0xFFFFu;
            if (freq >= 2u) exact_reciprocal(freq, rcp, post_shift);
        }
        const uint64_t bypass_mask = __ballot(is_bypass);
        const int j_lo = hi >= 64 ? 0 : static_cast<int>(64 - hi);
        for (int j = 63; j >= j_lo; --j) {
            if ((bypass_mask >> j) & 1ull) {
                // decode order: sentinel, count nibbles, payload low-first  =>  written reversed
                const uint32_t r = bcast_u32(raw, j);
                int nb = 0;
                while (nb * bprec < 32u && (r >> (nb * bprec)) != 0u) ++nb;
                for (int k = nb - 1; k >= 0; --k) put_raw(x, em, lane, (r >> (k * bprec)) & maxbv, bprec);
                put_raw(x, em, lane, static_cast<uint32_t>(nb) % maxbv, bprec);
                for (uint32_t k = 0; k < static_cast<uint32_t>(nb) / maxbv; ++k) put_raw(x, em, lane, maxbv, bprec);
            }
            put_symbol(x, em, lane, bcast_u32(start, j), bcast_u32(freq, j), bcast_u64(rcp, j),
                       bcast_u32(post_shift, j), prec);
        }
    }
    em.push(static_cast<uint32_t>(x >> 32), lane);  // flush, rans64.h:87-94
    em.push(static_cast<uint32_t>(x), lane);
    if (lane == 0) out_nwords[stream] = em.overflow ? -1 : static_cast<int32_t>((slot + slot_words) - em.p);
}

// Lean encoder for the common case (no AR remap, every frequency in [1, 2^16)).  Same arithmetic as
// rans_encode_kernel, reorganised around the single-wave issue rate that bounds it:
//   * the per-symbol division constants come from the encoder image (one 16-byte gather per lane)
//     instead of two 64-bit divisions per lane and chunk;
//   * the gathers of chunk k+1 and the symbol/index loads of chunk k+2 are in flight while chunk k is
//     folded into the state, so the serial chain never waits for memory;
//   * freq == 1 needs no branch (reciprocal 2^64-1 gives q = x-1, the missing 2^p-1 is folded into start).
struct EncLane {
    uint32_t a, b, rl, rh, raw;
};

__device__ __forceinline__ EncLane enc_prepare(const TablesDev &T, const int2 *rowinfo_lds, bool live, int32_t row, int32_t s)
{
    EncLane e{1u | (1u << 16), 0u, ~0u, ~0u, 0u};
    if (live) {
        row = clampi(row, 0, T.rows - 1);
        const int2 ri = rowinfo_lds[row];
        const int32_t max_value = ri.y;
        int32_t value = s - ri.x;
        uint32_t byp = 0;
        if (T.bypass) {
            if (value < 0) { e.raw = static_cast<uint32_t>(-2 * value - 1); value = max_value; }
            else if (value >= max_value) { e.raw = static_cast<uint32_t>(2 * (value - max_value)); value = max_value; }
            byp = (value == max_value) ? 0x80u : 0u;
        }
        value = clampi(value, 0, max_value);  // without bypass the reference is UB out of range
        const uint4 t = T.enc[static_cast<int64_t>(row) * T.stride + value];
        e.a = t.x; e.b = t.y | byp; e.rl = t.z; e.rh = t.w;
    }
    return e;
}

template <int WPB>
__global__ __launch_bounds__(64 * WPB) void rans_encode_fast_kernel(TablesDev T, const int32_t *__restrict__ symbols,
                                                                    const int32_t *__restrict__ indexes,
                                                                    const int64_t *__restrict__ seg, uint32_t *out_words,
                                                                    int64_t slot_words, int32_t *out_nwords, int nstreams)
{
    // A stream is a serial chain that issues one instruction every ~8 cycles: when its wave shares a SIMD with the waves of
    // an MFMA transform running on another HIP stream, losing the issue arbitration stretches the chain several times over
    // (measured: 2.4 -> 11.9 ms beside conv_tap_mfma_kernel), while winning it costs the other waves next to nothing.
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ uint32_t lds_words[];
    int2 *rowinfo_lds = reinterpret_cast<int2 *>(lds_words);
    const int stream = blockIdx.x * WPB + (WPB > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0);
    const int lane = threadIdx.x & 63;
    for (int r = threadIdx.x; r < T.rows; r += 64 * WPB) rowinfo_lds[r] = T.rowinfo[r];
    __syncthreads();
    if (stream >= nstreams) return;
    const int64_t beg = seg[stream];
    const int64_t n = seg[stream + 1] - beg;
    const int32_t *sym = symbols + beg;
    const int32_t *idx = indexes + beg;
    uint32_t *slot = out_words + static_cast<int64_t>(stream) * slot_words;
    WaveEmitter em{slot + slot_words, slot, false, 0u, 0ull};
    uint32_t xlo_ = static_cast<uint32_t>(kRansL), xhi_ = 0u;   // the state, in s[52:53] across the spelled-out steps
    const uint32_t xs = 31u - static_cast<uint32_t>(T.precision);  // x >= freq << (63-p)  <=>  (x >> 32) >= freq << (31-p)
    const uint32_t bprec = static_cast<uint32_t>(T.bypass_precision);
    const uint32_t maxbv = (1u << bprec) - 1u;

    // software pipeline over chunks of 64 symbols (last chunk first; lane 63 = last symbol of a chunk)
    int64_t i1 = n - 64 + lane;               // element of this lane in the chunk being prepared
    int32_t row1 = 0, s1 = 0;
    if (i1 >= 0) { row1 = idx[i1]; s1 = sym[i1]; }
    EncLane cur_ = enc_prepare(T, rowinfo_lds, i1 >= 0, row1, s1);
    int64_t i2 = i1 - 64;
    int32_t row2 = 0, s2 = 0;
    if (i2 >= 0) { row2 = idx[i2]; s2 = sym[i2]; }

    for (int64_t hi = n; hi > 0; hi -= 64) {
        // next chunk's gather and the loads of the chunk after it, in flight during the serial fold below
        const EncLane nxt = enc_prepare(T, rowinfo_lds, i2 >= 0, row2, s2);
        const int64_t i3 = i2 - 64;
        int32_t row3 = 0, s3 = 0;
        if (i3 >= 0) { row3 = idx[i3]; s3 = sym[i3]; }

        // Lane j owns symbol j of the chunk and already holds its frequency, start and exact-division constants, so C(s, x)
        // is evaluated by ALL lanes for their own symbol against the current uniform state and only the new state of lane j
        // is broadcast back: no per-symbol operand broadcasts, no scalar 64x64 multiply.  A lone wave issues one instruction
        // per ~8.4 clocks whatever it is (scripts/r04_chain_probe.hip), so what counts is the instruction count per symbol: 13
        // spelled out below + the compiler's scalar compare and branch for "renormalise?" (x_max forced to 0 for a bypass
        // symbol: the same test covers it).  A plain renormalisation parks the low word in the symbol's own lane of a register (one
        // coalesced store per chunk) and shifts the state: four instructions, no memory instruction, no bounds test behind the branch.
        const uint32_t fq = cur_.a & 0xFFFFu, cm_ = cur_.a >> 16, sh_ = cur_.b & 63u;
        const uint64_t st_ = cur_.b >> 8;
        const uint32_t xm_ = (cur_.b & 0x80u) ? 0u : (fq << xs);  // renormalise when (x >> 32) >= xm
        const int j_lo = hi >= 64 ? 0 : static_cast<int>(64 - hi);
        uint32_t xmj_ = bcast_u32(xm_, 63);
        uint64_t low_pair_ = 0ull;   // v[30:31]: {mulhi(x_lo, rcp_lo), 0}
        auto renormalise = [&](auto jc) {
            constexpr int J = decltype(jc)::value;
            if (__builtin_expect(xmj_ != 0u, 1)) {   // park the low word in this symbol's lane, x >>= 32
                uint32_t &xlo = xlo_, &xhi = xhi_;   // (generic lambda: an asm operand alone is no odr-use, the capture needs one)
                uint32_t &parked = em.parked;
                uint64_t &parked_mask = em.mask;
                asm volatile("v_writelane_b32 %2, s52, %4\n\ts_bitset1_b64 %3, %4\n\ts_mov_b32 s52, s53\n\ts_mov_b32 s53, 0"
                             : "={s52}"(xlo), "={s53}"(xhi), "+v"(parked), "+s"(parked_mask) : "n"(J), "0"(xlo), "1"(xhi));
            } else {   // a bypass symbol: its escape code goes first (the decoder reads sentinel, count nibbles, payload low-first)
                const EscapeResult er = encode_escape(static_cast<uint64_t>(xlo_) | (static_cast<uint64_t>(xhi_) << 32), em.p, em.lo, lane, em.parked, em.mask,
                                                      bcast_u32(cur_.raw, J), bprec, maxbv, bcast_u32(fq, J) << xs);
                em.p = er.p;
                em.mask = 0ull;
                if (er.overflow) em.overflow = true;
                const uint64_t x = er.x;
                {   // back into s[52:53] through vector registers: the loops above may leave the (uniform) state in them, and a
                    // copy from there to a NAMED scalar register is something the compiler cannot legalise itself
                    uint32_t &xlo = xlo_, &xhi = xhi_;
                    asm volatile("v_readfirstlane_b32 s52, %2\n\tv_readfirstlane_b32 s53, %3" : "={s52}"(xlo), "={s53}"(xhi)
                                 : "v"(static_cast<uint32_t>(x)), "v"(static_cast<uint32_t>(x >> 32)));
                }
            }
            {
                uint32_t &xlo = xlo_, &xhi = xhi_;
                asm volatile("" : "={s52}"(xlo), "={s53}"(xhi) : "0"(xlo), "1"(xhi));   // (the state stays in its registers on every path)
            }
        };
        auto step = [&](auto jc) {
            constexpr int J = decltype(jc)::value;
            uint32_t &xlo = xlo_, &xhi = xhi_, &xmj = xmj_;
            uint64_t &low_pair = low_pair_;
            const EncLane &cur = cur_;
            const uint32_t &sh = sh_, &cm = cm_, &xm = xm_;
            const uint64_t &st = st_;
            if (__builtin_expect(xhi >= xmj, 0)) renormalise(jc);
            // q = mulhi64(x, rcp) >> shift (== x / freq, Alverson; q = x - 1 for freq 1, see the encoder image):
            //   t = xh*rl + hi32(xl*rl); mid = xl*rh + t with its carry; h = xh*rh + (mid >> 32 | carry << 32); q = h >> shift
            // x' = (q << p) + (x - q*freq) + start = x + start' + q * (2^p - freq); then the next symbol's x_max and the new state
            asm volatile("v_mul_hi_u32 v30, s52, %[rl]\n\t"
                         "v_mad_u64_u32 v[32:33], s[54:55], s53, %[rl], v[30:31]\n\t"
                         "v_mad_u64_u32 v[32:33], s[54:55], s52, %[rh], v[32:33]\n\t"
                         "v_lshl_add_u64 v[34:35], s[52:53], 0, %[st]\n\t"
                         "v_addc_co_u32 v37, vcc, 0, 0, s[54:55]\n\t"
                         "v_mov_b32 v36, v33\n\t"
                         "v_mad_u64_u32 v[32:33], vcc, s53, %[rh], v[36:37]\n\t"
                         "v_lshrrev_b64 v[32:33], %[sh], v[32:33]\n\t"
                         "v_mad_u32_u24 v35, v33, %[cm], v35\n\t"
                         "v_mad_u64_u32 v[32:33], vcc, v32, %[cm], v[34:35]\n\t"
                         "v_readlane_b32 s56, %[xm], %[jn]\n\t"
                         "v_readlane_b32 s52, v32, %[j]\n\t"
                         "v_readlane_b32 s53, v33, %[j]"
                         : "={s52}"(xlo), "={s53}"(xhi), "={s56}"(xmj), "={v[30:31]}"(low_pair)
                         : "0"(xlo), "1"(xhi), "3"(low_pair), [rl] "v"(cur.rl), [rh] "v"(cur.rh), [st] "v"(st), [sh] "v"(sh), [cm] "v"(cm), [xm] "v"(xm),
                           [j] "n"(J), [jn] "n"(J > 0 ? J - 1 : 0)
                         : "s54", "s55", "vcc", "v32", "v33", "v34", "v35", "v36", "v37");
        };
        if (j_lo == 0) {   // a full chunk
            wavedec::static_down<63>(step);
        } else {           // the stream's first symbols: lanes j_lo .. 63
            wavedec::static_down<63>([&](auto jc) { if (decltype(jc)::value >= j_lo) step(jc); });
        }
        em.flush(lane);
        cur_ = nxt;
        i2 = i3; row2 = row3; s2 = s3;
    }
    em.push_direct(xhi_, lane);  // flush, rans64.h:87-94 (every chunk has flushed its parked words)
    em.push_direct(xlo_, lane);
    if (lane == 0) out_nwords[stream] = em.overflow ? -1 : static_cast<int32_t>((slot + slot_words) - em.p);
}

// Word reader: 64 words are kept in a VGPR (lane k = word base+k) and handed out by broadcast.
struct WordReader {
    const uint32_t *words;
    int64_t limit;  // words in this stream; reads past it return 0 (truncated stream != fault)
    int64_t pos;    // next word (uniform)
    int64_t base;   // first word cached
    uint32_t cache;
    __device__ __forceinline__ void fill(int lane)
    {
        base = pos & ~63ll;
        cache = (base + lane < limit) ? words[base + lane] : 0u;
    }
    __device__ __forceinline__ uint32_t next(int lane)
    {
        if (pos - base >= 64) fill(lane);
        const uint32_t w = bcast_u32(cache, static_cast<int>(pos - base));
        ++pos;
        return w;
    }
};

__device__ __forceinline__ uint32_t get_raw(uint64_t &x, WordReader &rd, int lane, uint32_t nbits)
{ /* rans64.cpp:49-65 */
    const uint32_t v = static_cast<uint32_t>(x) & ((1u << nbits) - 1u);
    x >>= nbits;
    if (x < kRansL) x = (x << 32) | rd.next(lane);
    return v;
}

// One wavefront per stream; the CDF search of each symbol is a 64-ary search over the row
// (one probe per lane, one ballot per level), everything else is scalar.
template <bool AR, bool LDS>
__global__ __launch_bounds__(64) void rans_decode_kernel(TablesDev T, ArDev ar, const uint32_t *__restrict__ words_all,
                                                         const int64_t *__restrict__ word_off,
                                                         const int32_t *__restrict__ indexes,
                                                         const int64_t *__restrict__ seg, int32_t *out_symbols,
                                                         uint64_t *state, int64_t *pos_io, StridedSeg ss)
{
    extern __shared__ uint32_t lds_words[];
    const uint16_t *lds16 = reinterpret_cast<const uint16_t *>(lds_words);
    const int stream = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t beg = seg ? seg[stream] : ss.first + stream * ss.stride;
    const int64_t n = seg ? seg[stream + 1] - beg : ss.count;
    const int32_t *idx = indexes + beg;
    int32_t *out = out_symbols + beg;
    if (LDS) {  // whole table set resident in LDS: the serial search never leaves the CU
        const uint32_t *src = reinterpret_cast<const uint32_t *>(T.cdf16);
        for (int i = lane; i < T.total16 / 2; i += 64) lds_words[i] = src[i];
        __syncthreads();
    }

    WordReader rd;
    rd.words = words_all + word_off[stream];
    rd.limit = word_off[stream + 1] - word_off[stream];
    uint64_t x;
    int64_t p0 = pos_io[stream];
    if (p0 < 0) {
        rd.pos = 0;
        rd.fill(lane);
        const uint32_t w0 = rd.next(lane), w1 = rd.next(lane);
        x = static_cast<uint64_t>(w0) | (static_cast<uint64_t>(w1) << 32);
    } else {
        rd.pos = p0;
        rd.fill(lane);
        x = state[stream];
    }
    x = uniform_u64(x);
    const uint32_t prec = static_cast<uint32_t>(T.precision);
    const uint32_t mask = (1u << prec) - 1u;
    const uint32_t bprec = static_cast<uint32_t>(T.bypass_precision);
    const uint32_t maxbv = (1u << bprec) - 1u;

    for (int64_t c0 = 0; c0 < n; c0 += 64) {
        const int64_t i = c0 + lane;
        int32_t row_l = 0, size_l = 2, off_l = 0, base_l = 0;
        if (i < n) {
            row_l = AR ? idx[i] : clampi(idx[i], 0, T.rows - 1);
            if (!AR) { size_l = T.sizes[row_l]; off_l = T.offsets[row_l]; if (LDS) base_l = T.base[row_l]; }
        }
        int32_t result = 0;
        const int cnt = (n - c0) < 64 ? static_cast<int>(n - c0) : 64;
        for (int j = 0; j < cnt; ++j) {
            int32_t row = static_cast<int32_t>(bcast_u32(static_cast<uint32_t>(row_l), j));
            int32_t size, offset;
            if (AR) {
                // remap from already-decoded symbols (ans_interface.hpp:89-104); lane-uniform loads
                const int64_t e = c0 + j, g = beg + e;
                const int32_t a = ar.ar_indexes ? ar.ar_indexes[g] : 0;
                const int32_t one = ar.custom ? 0 : 1;
                const int32_t d0 = ar.off0[g];
                const int32_t v0 = (d0 > 0 && d0 <= e) ? __hip_atomic_load(out + (e - d0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + one : 0;
                int32_t v1 = 0, v2 = 0;
                if (ar.order >= 2) {
                    const int32_t d1 = ar.off1[g];
                    v1 = (d1 > 0 && d1 <= e) ? __hip_atomic_load(out + (e - d1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + one : 0;
                }
                if (ar.order >= 3) {
                    const int32_t d2 = ar.off2[g];
                    v2 = (d2 > 0 && d2 <= e) ? __hip_atomic_load(out + (e - d2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + one : 0;
                }
                row = ar.custom ? ar_custom_row(ar, a, row, v0, v1, v2) : ar_row(ar, a, row, v0, v1);
                row = __builtin_amdgcn_readfirstlane(clampi(row, 0, T.rows - 1));
                size = T.sizes[row];
                offset = T.offsets[row];
                size = __builtin_amdgcn_readfirstlane(size);
                offset = __builtin_amdgcn_readfirstlane(offset);
            } else {
                size = static_cast<int32_t>(bcast_u32(static_cast<uint32_t>(size_l), j));
                offset = static_cast<int32_t>(bcast_u32(static_cast<uint32_t>(off_l), j));
            }
            const int32_t *cdf = T.cdfs + static_cast<int64_t>(row) * T.stride;
            int32_t rbase = 0;
            if (LDS) rbase = AR ? __builtin_amdgcn_readfirstlane(T.base[row]) : static_cast<int32_t>(bcast_u32(static_cast<uint32_t>(base_l), j));
            // entry e of the current row (the last entry, 2^precision, is implied in the packed copy)
            auto entry = [&](int32_t e) -> int32_t {
                if (!LDS) return cdf[e];
                return (e == size - 1) ? (1 << prec) : static_cast<int32_t>(lds16[rbase + e]);
            };
            const uint32_t cf = static_cast<uint32_t>(x) & mask;

            // find t = first entry with cdf[t] > cf (t >= 1 because cdf[0] = 0); s = t - 1.
            int32_t lo = 0;           // window start (uniform)
            int32_t span = size;      // entries still in play, starting at lo
            while (span > 64) {       // coarse level(s): probe the last entry of each of 64 blocks
                const int32_t step = (span + 63) >> 6;
                int32_t pi = lo + (lane + 1) * step - 1;
                const bool in = pi < lo + span;
                if (!in) pi = lo + span - 1;
                const int32_t v = entry(pi);
                const uint64_t m = __ballot(static_cast<uint32_t>(v) > cf);
                const int blk = __builtin_ctzll(m);  // m != 0: the final entry 2^prec > cf
                const int32_t nlo = lo + blk * step;
                const int32_t nspan = (nlo + step <= lo + span) ? step : (lo + span - nlo);
                lo = nlo;
                span = nspan;
            }
            // fine level: lanes cover entries lo-1 .. lo+span-1 (span+1 <= 65 -> lane 0 is entry lo-1
            // only when lo > 0; entry lo+span-1 by lane span).  Use two registers to stay within 64.
            const int32_t e0 = lo + lane;  // entry index for register A
            int32_t va = (lane < span) ? entry(e0) : 0x7FFFFFFF;
            const uint64_t m = __ballot(static_cast<uint32_t>(va) > cf && lane < span);
            const int tl = __builtin_ctzll(m);  // lane of t
            const uint32_t c_t = bcast_u32(static_cast<uint32_t>(va), tl);
            uint32_t c_s;
            if (tl > 0) c_s = bcast_u32(static_cast<uint32_t>(va), tl - 1);
            else c_s = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(entry(lo - 1)));
            const int32_t s = lo + tl - 1;
            const uint32_t freq = c_t - c_s;

            x = static_cast<uint64_t>(freq) * (x >> prec) + (static_cast<uint32_t>(x) & mask) - c_s;  // rans64.h:128-142
            if (x < kRansL) x = (x << 32) | rd.next(lane);

            int32_t value = s;
            if (T.bypass && value == size - 2) {
                uint32_t v = get_raw(x, rd, lane, bprec);
                uint32_t nb = v;
                while (v == maxbv) { v = get_raw(x, rd, lane, bprec); nb += v; }
                uint32_t raw = 0;
                for (uint32_t k = 0; k < nb; ++k) {
                    const uint32_t nib = get_raw(x, rd, lane, bprec);
                    if (k * bprec < 32u) raw |= nib << (k * bprec);
                }
                value = static_cast<int32_t>(raw >> 1);
                if (raw & 1u) value = -value - 1; else value += size - 2;
            }
            value += offset;
            if (AR) {
                if (lane == 0) __hip_atomic_store(out + (c0 + j), value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                result = (lane == j) ? value : result;
            }
        }
        if (!AR && i < n) out[i] = result;
    }
    if (lane == 0) { state[stream] = x; pos_io[stream] = rd.pos; }
}

// Fast decoder for the common case (no AR remap, search image resident in LDS, rows <= 4096 entries): one wavefront per
// stream runs the serial chain of wave_decoder.h (built for instruction count: a lone wave issues one instruction per ~8.5
// clocks); table rows of a chunk's 64 symbols are gathered lane-parallel before the chain starts, offsets added after it.
// WPB wavefronts (= streams) per workgroup share ONE LDS copy of the search image: a batch of streams then occupies
// nstreams / WPB compute units instead of nstreams, and the units it leaves alone keep running the MFMA transforms of
// another sub-batch on a second HIP stream (the image, not the wave slot, is what excludes a convolution workgroup).
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void rans_decode_fast_kernel(TablesDev T, const uint32_t *__restrict__ words_all,
                                                                    const int64_t *__restrict__ word_off,
                                                                    const int32_t *__restrict__ indexes,
                                                                    const int64_t *__restrict__ seg, int32_t *out_symbols,
                                                                    uint64_t *state, int64_t *pos_io, StridedSeg ss, int nstreams)
{
    __builtin_amdgcn_s_setprio(3);   // see rans_encode_fast_kernel
    extern __shared__ __attribute__((aligned(16))) uint32_t img[];
    const int lane = threadIdx.x & 63;
    const int stream_raw = blockIdx.x * WPB + (WPB > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0);
    const bool live = stream_raw < nstreams;           // a surplus wave of the last workgroup only helps to copy the image
    const int stream = live ? stream_raw : nstreams - 1;
    const int64_t beg = seg ? seg[stream] : ss.first + stream * ss.stride;
    const int n = static_cast<int>(seg ? seg[stream + 1] - beg : ss.count);
    const int32_t *idx = indexes + beg;
    int32_t *out = out_symbols + beg;
    {   // table image -> LDS in 16-byte pieces, eight loads in flight per lane (the image is a multiple of 16 bytes, rows
        // are 16-byte aligned): a 150 KB image costs ~5 us instead of ~25, which matters when the AR loop calls this
        // kernel once per topo group for a few hundred symbols per stream
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 *src = reinterpret_cast<const u32x4 *>(T.image);
        u32x4 *dst = reinterpret_cast<u32x4 *>(img);
        const int pieces = T.image_words >> 2;
        constexpr int kT = 64 * WPB;
        int i = threadIdx.x;
        for (; i + 7 * kT < pieces; i += 8 * kT) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[i + u * kT];
#pragma unroll
            for (int u = 0; u < 8; ++u) dst[i + u * kT] = v[u];
        }
        for (; i < pieces; i += kT) dst[i] = src[i];
        for (int j = (pieces << 2) + static_cast<int>(threadIdx.x); j < T.image_words; j += kT) img[j] = T.image[j];
    }
    __syncthreads();
    if (!live) return;

    wavedec::WaveDecoder d;
    {
        const int64_t p0 = pos_io[stream];   // < 0: a fresh stream; otherwise resume behind an earlier call (state[], pos_io[])
        d.init(img, words_all + word_off[stream], static_cast<int>(word_off[stream + 1] - word_off[stream]), T.precision, T.bypass_precision, T.bypass != 0,
               p0 < 0 ? -1 : static_cast<int>(p0), p0 < 0 ? 0ull : uniform_u64(state[stream]), lane);
    }
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int i = c0 + lane;
        uint32_t meta_l = 0;
        int32_t off_l = 0, size_l = 2;
        if (i < n) {
            const int32_t row = clampi(idx[i], 0, T.rows - 1);
            meta_l = T.meta[row];
            off_l = T.offsets[row];
            size_l = T.sizes[row];
        }
        const int cnt = (n - c0) < 64 ? (n - c0) : 64;
        d.line_up(lane);
        const int32_t result = d.decode_chunk(meta_l, size_l, cnt, lane);   // symbol + 1 on the symbol's lane
        if (i < n) out[i] = result - 1 + off_l;
    }
    if (lane == 0) {
        state[stream] = d.x;
        pos_io[stream] = d.position();
    }
}

// Gather the right-aligned streams of an encode batch into one contiguous buffer.
__global__ void compact_streams_kernel(const uint32_t *__restrict__ slots, int64_t slot_words,
                                       const int32_t *__restrict__ nwords, const int64_t *__restrict__ out_off,
                                       uint32_t *__restrict__ out)
{
    const int s = blockIdx.x;
    const int n = nwords[s];
    if (n <= 0) return;
    const uint32_t *src = slots + static_cast<int64_t>(s + 1) * slot_words - n;
    uint32_t *dst = out + out_off[s];
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) dst[i] = src[i];
}

TablesDev dev_view(const basic_rans_tables *t)
{
    return TablesDev{t->d_cdfs, t->d_sizes, t->d_offsets, t->rows, t->stride, t->precision, t->bypass, t->bypass_precision,
                     t->d_cdf16, t->d_base, static_cast<int>(t->cdf16.size()),
                     t->d_image, t->d_meta, static_cast<int>(t->image.size()),
                     reinterpret_cast<const uint4 *>(t->d_enc), reinterpret_cast<const int2 *>(t->d_rowinfo)};
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 4); }
    template <typename T> T *as() { return static_cast<T *>(p); }
};

}  // namespace

namespace {

constexpr size_t kLdsTableBudget = 144 * 1024;  // of the 160 KiB per CU

// Streams (wavefronts) per workgroup of the batched fast coders.  Default: one per workgroup, spread over the chip (a
// stream is a serial chain bound by the issue latency of a lone wave; measured on MI355X, 256 streams x 51,200 symbols:
// 4 per workgroup costs +2 %, 8 +10 %, 16 doubles the chain time).  More per workgroup (BASIC_RANS_WPB = 2/4/8/16, or the
// codec session's setting) packs a batch onto nstreams / WPB compute units, which leaves the others -- and their LDS,
// which the decoder's 124 KB search image would otherwise claim -- to transforms running on other HIP streams.
thread_local int g_wpb_override = 0;   // set by a codec session around its launches (basic::set_rans_waves)

int rans_waves_per_block(int nstreams)
{
    static const int forced = [] { const char *e = getenv("BASIC_RANS_WPB"); return e ? atoi(e) : 0; }();
    int w = g_wpb_override > 0 ? g_wpb_override : forced > 0 ? forced : 1;
    (void)nstreams;
    if (w >= 16) return 16;
    if (w >= 8) return 8;
    if (w >= 4) return 4;
    if (w >= 2) return 2;
    return 1;
}

template <bool AR, bool LDS>
int launch_decode_v(const basic_rans_tables *t, const ArDev &ar, int nstreams, hipStream_t st, const uint32_t *d_words,
                    const int64_t *d_word_off, const int32_t *d_indexes, const int64_t *d_seg, int32_t *d_out,
                    uint64_t *d_state, int64_t *d_pos, StridedSeg ss)
{
    const size_t lds = LDS ? t->cdf16.size() * sizeof(uint16_t) : 0;
    if (LDS) BASIC_HIP_TRY(ensure_max_lds(reinterpret_cast<const void *>(rans_decode_kernel<AR, LDS>)));
    hipLaunchKernelGGL((rans_decode_kernel<AR, LDS>), dim3(nstreams), dim3(64), lds, st, dev_view(t), ar, d_words, d_word_off,
                       d_indexes, d_seg, d_out, d_state, d_pos, ss);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}

int launch_decode(const basic_rans_tables *t, const ArDev &ar, int nstreams, hipStream_t st, const uint32_t *d_words,
                  const int64_t *d_word_off, const int32_t *d_indexes, const int64_t *d_seg, int32_t *d_out,
                  uint64_t *d_state, int64_t *d_pos, StridedSeg ss = StridedSeg{0, 0, 0})
{
    // LDS-resident tables pay a per-launch copy of the table set; worth it unless the launch is tiny.
    const bool lds = t->cdf16.size() * sizeof(uint16_t) <= kLdsTableBudget;
    int max_size = 0;
    for (int v : t->sizes) max_size = v > max_size ? v : max_size;
    (void)max_size;
    if (!ar.tab && t->fast_ok) {
        const size_t lds_img = t->image.size() * sizeof(uint32_t);
        const int wpb = rans_waves_per_block(nstreams);
#define BASIC_DEC_LAUNCH(W)                                                                                              \
        do {                                                                                                             \
            BASIC_HIP_TRY(ensure_max_lds(reinterpret_cast<const void *>(rans_decode_fast_kernel<W>)));                   \
            hipLaunchKernelGGL(rans_decode_fast_kernel<W>, dim3((nstreams + W - 1) / W), dim3(64 * W), lds_img, st, dev_view(t), \
                               d_words, d_word_off, d_indexes, d_seg, d_out, d_state, d_pos, ss, nstreams);              \
        } while (0)
        switch (wpb) {
            case 16: BASIC_DEC_LAUNCH(16); break;
            case 8: BASIC_DEC_LAUNCH(8); break;
            case 4: BASIC_DEC_LAUNCH(4); break;
            case 2: BASIC_DEC_LAUNCH(2); break;
            default: BASIC_DEC_LAUNCH(1); break;
        }
#undef BASIC_DEC_LAUNCH
        BASIC_HIP_TRY(hipGetLastError());
        return BASIC_OK;
    }
    if (ar.tab)
        return lds ? launch_decode_v<true, true>(t, ar, nstreams, st, d_words, d_word_off, d_indexes, d_seg, d_out, d_state, d_pos, ss)
                   : launch_decode_v<true, false>(t, ar, nstreams, st, d_words, d_word_off, d_indexes, d_seg, d_out, d_state, d_pos, ss);
    return lds ? launch_decode_v<false, true>(t, ar, nstreams, st, d_words, d_word_off, d_indexes, d_seg, d_out, d_state, d_pos, ss)
               : launch_decode_v<false, false>(t, ar, nstreams, st, d_words, d_word_off, d_indexes, d_seg, d_out, d_state, d_pos, ss);
}

}  // namespace

namespace basic {
int rans_fast_view(const basic_rans_tables *t, RansFastView *out)
{
    BASIC_REQUIRE(t && out, "rans_fast_view: null argument");
    BASIC_REQUIRE(t->fast_ok && !t->d_ar, "rans_fast_view: this table set has no fast-decoder image (rows > 4096 entries, image > 156 KB, or AR remap)");
    out->image = t->d_image; out->meta = t->d_meta; out->sizes = t->d_sizes; out->offsets = t->d_offsets;
    out->image_words = static_cast<int>(t->image.size()); out->rows = t->rows; out->precision = t->precision;
    out->bypass = t->bypass ? 1 : 0; out->bypass_precision = t->bypass_precision;
    return BASIC_OK;
}

int set_rans_waves(int waves_per_block)
{
    const int prev = g_wpb_override;
    g_wpb_override = waves_per_block;
    return prev;
}
}  // namespace basic

namespace {
inline void put_be32(uint8_t *p, uint32_t v) { p[0] = v >> 24; p[1] = (v >> 16) & 0xFF; p[2] = (v >> 8) & 0xFF; p[3] = v & 0xFF; }
inline uint32_t get_be32(const uint8_t *p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
}  // namespace

extern "C" int basic_frame_streams(const uint32_t *words, const int64_t *word_off, int n, uint32_t h, uint32_t w,
                                   uint8_t *out, int64_t out_capacity, int64_t *out_len)
{
    BASIC_REQUIRE(word_off && out_len && n >= 0 && (words || n == 0 || word_off[n] == word_off[0]), "frame_streams: bad argument");
    const int64_t total = 12 + 4ll * n + 4 * (word_off[n] - word_off[0]);
    *out_len = total;
    if (!out || out_capacity < total) { set_error("frame_streams: output buffer too small"); return BASIC_ERR_OVERFLOW; }
    put_be32(out, h); put_be32(out + 4, w); put_be32(out + 8, static_cast<uint32_t>(n));
    uint8_t *p = out + 12;
    for (int i = 0; i < n; ++i) {
        const int64_t bytes = 4 * (word_off[i + 1] - word_off[i]);
        BASIC_REQUIRE(bytes >= 0 && bytes <= 0xFFFFFFFFll, "frame_streams: offsets must ascend");
        put_be32(p, static_cast<uint32_t>(bytes));
        std::memcpy(p + 4, words + word_off[i], static_cast<size_t>(bytes));
        p += 4 + bytes;
    }
    return BASIC_OK;
}

extern "C" int basic_unframe_streams(const uint8_t *data, int64_t len, uint32_t *h, uint32_t *w, int *n,
                                     int64_t *word_off, int n_capacity, uint32_t *words_out)
{
    BASIC_REQUIRE(data && len >= 12 && n, "unframe_streams: truncated header");
    if (h) *h = get_be32(data);
    if (w) *w = get_be32(data + 4);
    const uint32_t cnt = get_be32(data + 8);
    BASIC_REQUIRE(cnt <= 0x7FFFFFFFu && 12 + 4ll * cnt <= len, "unframe_streams: truncated body");
    *n = static_cast<int>(cnt);
    if (!words_out) return BASIC_OK;
    BASIC_REQUIRE(word_off && n_capacity >= static_cast<int>(cnt), "unframe_streams: offset array too small");
    int64_t cur = 12, wpos = 0;
    word_off[0] = 0;
    for (uint32_t i = 0; i < cnt; ++i) {
        BASIC_REQUIRE(cur + 4 <= len, "unframe_streams: truncated body");
        const int64_t bytes = get_be32(data + cur);
        cur += 4;
        BASIC_REQUIRE(cur + bytes <= len, "unframe_streams: truncated body");
        BASIC_REQUIRE(bytes >= 8 && bytes % 4 == 0, "rANS stream must hold >= 2 whole 32-bit words");
        std::memcpy(words_out + wpos, data + cur, static_cast<size_t>(bytes));
        cur += bytes;
        wpos += bytes / 4;
        word_off[i + 1] = wpos;
    }
    return BASIC_OK;
}

extern "C" int64_t basic_rans_encode_bound(int64_t n)
{
    // worst case per symbol: one 16-bit symbol + <= 13 raw groups of bypass_precision bits, well under
    // 3 words; plus 2 flush words.
    return (3 * n + 4) * 4;
}

extern "C" int basic_rans_encode_batch_dev(const basic_rans_tables *t, const int32_t *d_symbols,
                                           const int32_t *d_indexes, const int64_t *d_seg, int nstreams,
                                           uint32_t *d_out_words, int64_t slot_words, int32_t *d_out_nwords,
                                           void *hip_stream)
{
    if (!t) { set_error("ANS not initialized!"); return BASIC_ERR_NOT_INIT; }
    BASIC_REQUIRE(d_symbols && d_indexes && d_seg && d_out_words && d_out_nwords && nstreams >= 1 && slot_words >= 2,
                  "rans_encode_batch: bad argument");
    BASIC_REQUIRE(!t->d_ar, "rans_encode_batch: AR tables are only supported by the host-buffer entry points");
    static const bool no_fast = getenv("BASIC_RANS_NO_FAST_ENCODE") != nullptr;  // profiling ablation
    if (t->fast_enc_ok && !no_fast) {
        // Packed launches (several streams per workgroup) exist to share the chip with transforms on other HIP streams.
        // The encoder needs almost no LDS, so a convolution workgroup would settle on the same compute unit -- and its
        // LDS-DMA traffic then owns that unit's vector-memory queue: the encoder's table gathers wait behind it and the
        // chain runs 6x slower (measured: 4.0 -> 25 ms beside conv_tap_mfma_kernel, while the decoder, whose 124 KB
        // search image keeps its unit to itself, goes 8.1 -> 8.6 ms).  So a packed encoder workgroup claims the whole
        // unit's LDS as well: nstreams / W units are then the coder's alone.
        const int wpb_ = rans_waves_per_block(nstreams);
        size_t lds_rows = static_cast<size_t>(t->rows) * sizeof(int2);
        if (wpb_ > 1 && lds_rows < 159 * 1024) lds_rows = 159 * 1024;
#define BASIC_ENC_LAUNCH(W)                                                                                          \
        do {                                                                                                         \
        if (lds_rows > 64 * 1024) BASIC_HIP_TRY(ensure_max_lds(reinterpret_cast<const void *>(rans_encode_fast_kernel<W>))); \
        hipLaunchKernelGGL(rans_encode_fast_kernel<W>, dim3((nstreams + W - 1) / W), dim3(64 * W), lds_rows,         \
                           as_stream(hip_stream), dev_view(t), d_symbols, d_indexes, d_seg, d_out_words, slot_words, \
                           d_out_nwords, nstreams);                                                                  \
        } while (0)
        switch (wpb_) {
            case 16: BASIC_ENC_LAUNCH(16); break;
            case 8: BASIC_ENC_LAUNCH(8); break;
            case 4: BASIC_ENC_LAUNCH(4); break;
            case 2: BASIC_ENC_LAUNCH(2); break;
            default: BASIC_ENC_LAUNCH(1); break;
        }
#undef BASIC_ENC_LAUNCH
        BASIC_HIP_TRY(hipGetLastError());
        return BASIC_OK;
    }
    ArDev ar{};
    hipLaunchKernelGGL(rans_encode_kernel, dim3(nstreams), dim3(64), 0, as_stream(hip_stream), dev_view(t), ar,
                       d_symbols, d_indexes, d_seg, d_out_words, slot_words, d_out_nwords);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}

extern "C" int basic_rans_decode_batch_dev(const basic_rans_tables *t, const uint32_t *d_words,
                                           const int64_t *d_word_off, const int32_t *d_indexes, const int64_t *d_seg,
                                           int nstreams, int32_t *d_out_symbols, uint64_t *d_state, int64_t *d_pos,
                                           void *hip_stream)
{
    if (!t) { set_error("ANS not initialized!"); return BASIC_ERR_NOT_INIT; }
    BASIC_REQUIRE(d_words && d_word_off && d_indexes && d_seg && d_out_symbols && d_state && d_pos && nstreams >= 1,
                  "rans_decode_batch: bad argument");
    BASIC_REQUIRE(!t->d_ar, "rans_decode_batch: AR tables are only supported by the host-buffer entry points");
    ArDev ar{};
    int rc = launch_decode(t, ar, nstreams, as_stream(hip_stream), d_words, d_word_off, d_indexes, d_seg, d_out_symbols,
                           d_state, d_pos);
    if (rc) return rc;
    return BASIC_OK;
}

extern "C" int basic_rans_compact_streams_dev(const uint32_t *d_slots, int64_t slot_words, const int32_t *d_nwords,
                                              const int64_t *d_out_off, int nstreams, uint32_t *d_out, void *hip_stream)
{
    BASIC_REQUIRE(d_slots && d_nwords && d_out_off && d_out && nstreams >= 1 && slot_words >= 2,
                  "rans_compact_streams: bad argument");
    hipLaunchKernelGGL(compact_streams_kernel, dim3(nstreams, 8), dim3(256), 0, as_stream(hip_stream), d_slots, slot_words,
                       d_nwords, d_out_off, d_out);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}

extern "C" int basic_rans_decode_batch_strided_dev(const basic_rans_tables *t, const uint32_t *d_words,
                                                   const int64_t *d_word_off, const int32_t *d_indexes, int64_t first,
                                                   int64_t stride, int64_t count, int nstreams, int32_t *d_out_symbols,
                                                   uint64_t *d_state, int64_t *d_pos, void *hip_stream)
{
    if (!t) { set_error("ANS not initialized!"); return BASIC_ERR_NOT_INIT; }
    BASIC_REQUIRE(d_words && d_word_off && d_indexes && d_out_symbols && d_state && d_pos && nstreams >= 1 && first >= 0 &&
                      stride >= 0 && count >= 0,
                  "rans_decode_batch_strided: bad argument");
    BASIC_REQUIRE(!t->d_ar, "rans_decode_batch_strided: AR tables are only supported by the host-buffer entry points");
    if (count == 0) return BASIC_OK;
    ArDev ar{};
    return launch_decode(t, ar, nstreams, as_stream(hip_stream), d_words, d_word_off, d_indexes, nullptr, d_out_symbols, d_state,
                         d_pos, StridedSeg{first, stride, count});
}

// ---------------------------------------------------------------------------------------
// Host-buffer drop-ins (single stream): stage -> kernel -> copy back.
// ---------------------------------------------------------------------------------------
namespace {

int stage_ar(const basic_rans_tables *t, int64_t n, const int32_t *ar_indexes, const int32_t *ar_off0,
             const int32_t *ar_off1, const int32_t *ar_off2, DevBuf &b_ai, DevBuf &b_o0, DevBuf &b_o1, DevBuf &b_o2, ArDev &ar)
{
    ar = ArDev{};
    if (!t->d_ar) return BASIC_OK;
    if (!ar_off0 || (!t->ar_custom && t->ar_order == 2 && !ar_off1)) {
        set_error("ar_offsets is required for ar coding!");
        return BASIC_ERR_INVALID;
    }
    const size_t bytes = static_cast<size_t>(n) * sizeof(int32_t);
    ar.tab = t->d_ar; ar.k = t->ar_k; ar.order = t->ar_order; ar.rows = t->ar_rows; ar.s1 = t->ar_s1;
    if (t->ar_custom) {   // the op's arity is the number of offset rows of this call (ans_interface.hpp:70-84)
        ar.custom = 1;
        ar.order = ar_off2 ? 3 : ar_off1 ? 2 : 1;
        if (ar_off2 && !ar_off1) { set_error("ar_offsets rows must be given in order"); return BASIC_ERR_INVALID; }
    }
    const int order = ar.order;
    if (ar_indexes) {
        BASIC_HIP_TRY(b_ai.alloc(bytes));
        BASIC_HIP_TRY(hipMemcpy(b_ai.p, ar_indexes, bytes, hipMemcpyHostToDevice));
        ar.ar_indexes = b_ai.as<int32_t>();
    }
    BASIC_HIP_TRY(b_o0.alloc(bytes));
    BASIC_HIP_TRY(hipMemcpy(b_o0.p, ar_off0, bytes, hipMemcpyHostToDevice));
    ar.off0 = b_o0.as<int32_t>();
    if (order >= 2) {
        BASIC_HIP_TRY(b_o1.alloc(bytes));
        BASIC_HIP_TRY(hipMemcpy(b_o1.p, ar_off1, bytes, hipMemcpyHostToDevice));
        ar.off1 = b_o1.as<int32_t>();
    }
    if (order >= 3) {
        BASIC_HIP_TRY(b_o2.alloc(bytes));
        BASIC_HIP_TRY(hipMemcpy(b_o2.p, ar_off2, bytes, hipMemcpyHostToDevice));
        ar.off2 = b_o2.as<int32_t>();
    }
    return BASIC_OK;
}

}  // namespace

extern "C" int basic_rans_encode_host(const basic_rans_tables *t, const int32_t *symbols, const int32_t *indexes,
                                      int64_t n, const int32_t *ar_indexes, const int32_t *ar_off0,
                                      const int32_t *ar_off1, uint8_t *out, int64_t out_capacity, int64_t *out_len)
{
    return basic_rans_encode_host_ex(t, symbols, indexes, n, ar_indexes, ar_off0, ar_off1, nullptr, out, out_capacity, out_len);
}

namespace {
int encode_host_impl(const basic_rans_tables *t, const int32_t *symbols, const int32_t *indexes, int64_t n, bool use_ar,
                     const int32_t *ar_indexes, const int32_t *ar_off0, const int32_t *ar_off1, const int32_t *ar_off2, uint8_t *out,
                     int64_t out_capacity, int64_t *out_len);
}

extern "C" int basic_rans_encode_host_ex(const basic_rans_tables *t, const int32_t *symbols, const int32_t *indexes,
                                         int64_t n, const int32_t *ar_indexes, const int32_t *ar_off0, const int32_t *ar_off1,
                                         const int32_t *ar_off2, uint8_t *out, int64_t out_capacity, int64_t *out_len)
{
    return encode_host_impl(t, symbols, indexes, n, true, ar_indexes, ar_off0, ar_off1, ar_off2, out, out_capacity, out_len);
}

// The same with the table rows taken as given even when the set carries an AR remap: what Rans64Encoder::flush() does with
// symbols cached by AR calls -- their rows were remapped when they were cached (rans64.cpp:258-263,343,363-386).
extern "C" int basic_rans_encode_host_rows(const basic_rans_tables *t, const int32_t *symbols, const int32_t *indexes, int64_t n,
                                           uint8_t *out, int64_t out_capacity, int64_t *out_len)
{
    return encode_host_impl(t, symbols, indexes, n, false, nullptr, nullptr, nullptr, nullptr, out, out_capacity, out_len);
}

namespace {
int encode_host_impl(const basic_rans_tables *t, const int32_t *symbols, const int32_t *indexes, int64_t n, bool use_ar,
                     const int32_t *ar_indexes, const int32_t *ar_off0, const int32_t *ar_off1, const int32_t *ar_off2, uint8_t *out,
                     int64_t out_capacity, int64_t *out_len)
{
    if (!t) { set_error("ANS not initialized!"); return BASIC_ERR_NOT_INIT; }
    BASIC_REQUIRE(n >= 0 && out && out_len && (n == 0 || (symbols && indexes)), "encode_with_indexes: bad argument");
    const int64_t slot_words = basic_rans_encode_bound(n) / 4;
    DevBuf b_sym, b_idx, b_seg, b_out, b_nw, b_ai, b_o0, b_o1, b_o2;
    const size_t bytes = static_cast<size_t>(n) * sizeof(int32_t);
    BASIC_HIP_TRY(b_sym.alloc(bytes));
    BASIC_HIP_TRY(b_idx.alloc(bytes));
    BASIC_HIP_TRY(b_seg.alloc(2 * sizeof(int64_t)));
    BASIC_HIP_TRY(b_out.alloc(static_cast<size_t>(slot_words) * 4));
    BASIC_HIP_TRY(b_nw.alloc(sizeof(int32_t)));
    if (n) {
        BASIC_HIP_TRY(hipMemcpy(b_sym.p, symbols, bytes, hipMemcpyHostToDevice));
        BASIC_HIP_TRY(hipMemcpy(b_idx.p, indexes, bytes, hipMemcpyHostToDevice));
    }
    const int64_t seg[2] = {0, n};
    BASIC_HIP_TRY(hipMemcpy(b_seg.p, seg, sizeof(seg), hipMemcpyHostToDevice));
    ArDev ar{};
    int rc = use_ar ? stage_ar(t, n, ar_indexes, ar_off0, ar_off1, ar_off2, b_ai, b_o0, b_o1, b_o2, ar) : BASIC_OK;
    if (rc) return rc;
    if (!ar.tab && t->fast_enc_ok)  // same kernel choice as the batched device entry point
        hipLaunchKernelGGL(rans_encode_fast_kernel<1>, dim3(1), dim3(64), static_cast<size_t>(t->rows) * sizeof(int2), nullptr,
                           dev_view(t), b_sym.as<int32_t>(), b_idx.as<int32_t>(), b_seg.as<int64_t>(), b_out.as<uint32_t>(),
                           slot_words, b_nw.as<int32_t>(), 1);
    else
        hipLaunchKernelGGL(rans_encode_kernel, dim3(1), dim3(64), 0, nullptr, dev_view(t), ar, b_sym.as<int32_t>(),
                           b_idx.as<int32_t>(), b_seg.as<int64_t>(), b_out.as<uint32_t>(), slot_words, b_nw.as<int32_t>());
    BASIC_HIP_TRY(hipGetLastError());
    int32_t nwords = 0;
    BASIC_HIP_TRY(hipMemcpy(&nwords, b_nw.p, sizeof(nwords), hipMemcpyDeviceToHost));
    if (nwords < 0) { set_error("rans encoder: slot overflow"); return BASIC_ERR_OVERFLOW; }
    const int64_t nbytes = static_cast<int64_t>(nwords) * 4;
    *out_len = nbytes;
    if (nbytes > out_capacity) { set_error("encode_with_indexes: output buffer too small"); return BASIC_ERR_OVERFLOW; }
    BASIC_HIP_TRY(hipMemcpy(out, b_out.as<uint32_t>() + (slot_words - nwords), nbytes, hipMemcpyDeviceToHost));
    return BASIC_OK;
}
}  // namespace

struct basic_rans_stream {
    const basic_rans_tables *t = nullptr;
    DevBuf words, state, pos, woff;
    int64_t nwords = 0;
};

extern "C" int basic_rans_stream_open(const basic_rans_tables *t, const uint8_t *stream, int64_t stream_len,
                                      basic_rans_stream **out)
{
    if (!t) { set_error("ANS not initialized!"); return BASIC_ERR_NOT_INIT; }
    BASIC_REQUIRE(stream && out && stream_len >= 8 && (stream_len % 4) == 0, "set_stream: stream must hold >= 2 whole 32-bit words");
    auto *s = new (std::nothrow) basic_rans_stream();
    if (!s) { set_error("out of host memory"); return BASIC_ERR_INVALID; }
    s->t = t;
    s->nwords = stream_len / 4;
    hipError_t e = s->words.alloc(static_cast<size_t>(stream_len));
    if (e == hipSuccess) e = hipMemcpy(s->words.p, stream, stream_len, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = s->state.alloc(8);
    if (e == hipSuccess) e = s->pos.alloc(8);
    if (e == hipSuccess) e = s->woff.alloc(16);
    const int64_t minus1 = -1;
    const int64_t woff[2] = {0, s->nwords};
    if (e == hipSuccess) e = hipMemcpy(s->pos.p, &minus1, 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->woff.p, woff, 16, hipMemcpyHostToDevice);
    if (e != hipSuccess) { delete s; return hip_fail(e, "rans_stream_open", __FILE__, __LINE__); }
    *out = s;
    return BASIC_OK;
}

namespace {
int stream_decode(basic_rans_stream *s, const int32_t *indexes, int64_t n, const int32_t *ar_indexes,
                  const int32_t *ar_off0, const int32_t *ar_off1, const int32_t *ar_off2, int32_t *out_symbols)
{
    BASIC_REQUIRE(s && n >= 0 && (n == 0 || (indexes && out_symbols)), "decode: bad argument");
    if (n == 0) return BASIC_OK;
    const basic_rans_tables *t = s->t;
    DevBuf b_idx, b_seg, b_out, b_ai, b_o0, b_o1, b_o2;
    const size_t bytes = static_cast<size_t>(n) * sizeof(int32_t);
    BASIC_HIP_TRY(b_idx.alloc(bytes));
    BASIC_HIP_TRY(b_out.alloc(bytes));
    BASIC_HIP_TRY(b_seg.alloc(2 * sizeof(int64_t)));
    BASIC_HIP_TRY(hipMemcpy(b_idx.p, indexes, bytes, hipMemcpyHostToDevice));
    const int64_t seg[2] = {0, n};
    BASIC_HIP_TRY(hipMemcpy(b_seg.p, seg, sizeof(seg), hipMemcpyHostToDevice));
    ArDev ar;
    int rc = stage_ar(t, n, ar_indexes, ar_off0, ar_off1, ar_off2, b_ai, b_o0, b_o1, b_o2, ar);
    if (rc) return rc;
    rc = launch_decode(t, ar, 1, nullptr, s->words.as<uint32_t>(), s->woff.as<int64_t>(), b_idx.as<int32_t>(),
                       b_seg.as<int64_t>(), b_out.as<int32_t>(), s->state.as<uint64_t>(), s->pos.as<int64_t>());
    if (rc) return rc;
    BASIC_HIP_TRY(hipMemcpy(out_symbols, b_out.p, bytes, hipMemcpyDeviceToHost));
    return BASIC_OK;
}
}  // namespace

extern "C" int basic_rans_stream_decode(basic_rans_stream *s, const int32_t *indexes, int64_t n, int32_t *out_symbols)
{
    // decode_stream ignores AR parameters in the reference (rans64.cpp:529,537)
    if (!s) { set_error("set_stream was not called"); return BASIC_ERR_NOT_INIT; }
    BASIC_REQUIRE(!s->t->d_ar, "decode_stream does not support AR tables (reference: rans64.cpp:529)");
    return stream_decode(s, indexes, n, nullptr, nullptr, nullptr, nullptr, out_symbols);
}

extern "C" void basic_rans_stream_close(basic_rans_stream *s) { delete s; }

extern "C" int basic_rans_decode_host(const basic_rans_tables *t, const uint8_t *stream, int64_t stream_len,
                                      const int32_t *indexes, int64_t n, const int32_t *ar_indexes,
                                      const int32_t *ar_off0, const int32_t *ar_off1, int32_t *out_symbols)
{
    return basic_rans_decode_host_ex(t, stream, stream_len, indexes, n, ar_indexes, ar_off0, ar_off1, nullptr, out_symbols);
}

extern "C" int basic_rans_decode_host_ex(const basic_rans_tables *t, const uint8_t *stream, int64_t stream_len,
                                         const int32_t *indexes, int64_t n, const int32_t *ar_indexes, const int32_t *ar_off0,
                                         const int32_t *ar_off1, const int32_t *ar_off2, int32_t *out_symbols)
{
    basic_rans_stream *s = nullptr;
    int rc = basic_rans_stream_open(t, stream, stream_len, &s);
    if (rc) return rc;
    rc = stream_decode(s, indexes, n, ar_indexes, ar_off0, ar_off1, ar_off2, out_symbols);
    basic_rans_stream_close(s);
    return rc;
}
