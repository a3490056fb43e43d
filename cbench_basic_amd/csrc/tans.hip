// Table ANS (tANS / FSE-style, switchable tables, bypass coding) for gfx950: tables, batched stream kernels and the
// host-buffer drop-ins behind cbench_basic_amd.ans.TansEncoder / TansDecoder.
//
// Bitstream contract (bit-exact with the reference):
//   count normalisation   csrc/ans/tans.cpp:27-147      encoder tables  csrc/ans/tans.cpp:149-226
//   decoder tables        csrc/ans/tans.cpp:261-318     symbol loops    csrc/ans/tans.cpp:527-680, :722-815
//   bit container         csrc/FSE/bitstream.h:185-247 (writer), :260-360 (reader)
//
// A tANS stream is one little-endian integer: the encoder appends bit fields at the top while it walks the symbols
// last to first, the decoder takes them off the top (below the end mark) first to last.  Every step is a table lookup
// that depends on the previous state, so -- as with rANS -- a stream is a serial chain and parallelism is ACROSS
// streams: one 64-lane wavefront per stream.  The lanes prepare a chunk of 64 symbols at a time (row remap, offset,
// clamp / bypass split, the symbol's two transform words: one coalesced gather), the chain itself runs on wave-uniform
// values; the state-dependent lookups hit the table set in L2 (a set of R rows is R x 2^L x 2 bytes for the encoder,
// R x 2^L x 4 bytes for the decoder).
#include <type_traits>
#include "common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace basic;

// ---------------------------------------------------------------------------------------
// Host side: tables
// ---------------------------------------------------------------------------------------
struct basic_tans_tables {
    int rows = 0, log = 11, bypass = 0, bypass_precision = 4, max_nsym = 0;
    std::vector<int32_t> nsym, offsets;
    // device images; row `rows` is the uniform bypass alphabet (present when bypass coding is on)
    uint16_t *d_next = nullptr;    // [(rows+1)][2^log]      encoder: next state, grouped by symbol
    uint2 *d_sym = nullptr;        // [(rows+1)][max_nsym]   encoder: { (bits << 16) - first wide state, group start - count }
    uint32_t *d_dec = nullptr;     // [(rows+1)][2^log]      decoder: base | bits << 12 | symbol << 16
    int2 *d_rowinfo = nullptr;     // [(rows+1)]             { offset, max_value }
    int ar_k = 0, ar_order = 0, ar_s1 = 0;
    int32_t *d_ar = nullptr;
};

namespace {

inline unsigned top_bit(uint32_t v) { return 31u - static_cast<unsigned>(__builtin_clz(v)); }

// Normalised counts of one distribution: -1 marks a symbol that gets exactly one cell at the top of the table.
// Returns false when the counts cannot be laid out (too small a table, or one symbol holding the whole mass -- the
// reference walks on with an unwritten table there, tans.cpp:118).
bool normalize_counts(const int32_t *freq, int nsym, unsigned L, std::vector<int16_t> &norm, std::string &why)
{
    static const uint32_t kRestToBeat[8] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
    std::vector<uint32_t> count(nsym);
    uint64_t total = 0;
    for (int s = 0; s < nsym; ++s) { count[s] = static_cast<uint32_t>(freq[s]); total += count[s]; }
    norm.assign(nsym, 0);
    if (total < 2) { why = "Error (generic)"; return false; }
    const unsigned need = std::min(top_bit(static_cast<uint32_t>(total - 1)) + 1, top_bit(static_cast<uint32_t>(nsym - 1)) + 2);
    if (L < need) { why = "Error (generic)"; return false; }   // FSE_minTableLog, tans.cpp:17-23,106
    const uint64_t scale = 62 - L, step = (1ull << 62) / total, vstep = 1ull << (scale - 20);
    const uint32_t rare_below = static_cast<uint32_t>(total >> L);
    int left = 1 << L, top = 0;
    int16_t top_p = 0;
    for (int s = 0; s < nsym; ++s) {
        if (count[s] == total) { why = "a distribution with a single possible symbol has no tANS table"; return false; }
        if (count[s] == 0) continue;
        if (count[s] <= rare_below) { norm[s] = -1; --left; continue; }
        const uint64_t scaled = static_cast<uint64_t>(count[s]) * step;
        int16_t p = static_cast<int16_t>(scaled >> scale);
        if (p < 8 && scaled - (static_cast<uint64_t>(p) << scale) > vstep * kRestToBeat[p]) ++p;
        if (p > top_p) { top_p = p; top = s; }
        norm[s] = p;
        left -= p;
    }
    if (-left < (norm[top] >> 1)) { norm[top] = static_cast<int16_t>(norm[top] + left); return true; }

    // the largest symbol cannot absorb the excess: distribute again, rare symbols first (tans.cpp:27-95)
    uint32_t placed = 0;
    uint32_t one_below = static_cast<uint32_t>((total * 3) >> (L + 1));
    for (int s = 0; s < nsym; ++s) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= rare_below) { norm[s] = -1; ++placed; total -= count[s]; continue; }
        if (count[s] <= one_below) { norm[s] = 1; ++placed; total -= count[s]; continue; }
        norm[s] = -2;
    }
    uint32_t cells = (1u << L) - placed;
    if (cells == 0) { why = "Error (generic)"; return false; }   // every cell taken by rare symbols: the reference divides by zero here
    if (total / cells > one_below) {
        one_below = static_cast<uint32_t>((total * 3) / (cells * 2));
        for (int s = 0; s < nsym; ++s)
            if (norm[s] == -2 && count[s] <= one_below) { norm[s] = 1; ++placed; total -= count[s]; }
        cells = (1u << L) - placed;
    }
    if (placed == static_cast<uint32_t>(nsym)) {
        int best = 0;
        for (int s = 1; s < nsym; ++s) if (count[s] > count[best]) best = s;
        norm[best] = static_cast<int16_t>(norm[best] + static_cast<int16_t>(cells));
        return true;
    }
    const uint64_t vlog = 62 - L, mid = (1ull << (vlog - 1)) - 1;
    const uint64_t rstep = (((1ull << vlog) * cells) + mid) / total;
    uint64_t acc = mid;
    for (int s = 0; s < nsym; ++s) {
        if (norm[s] != -2) continue;
        const uint64_t end = acc + static_cast<uint64_t>(count[s]) * rstep;
        const uint32_t weight = static_cast<uint32_t>(end >> vlog) - static_cast<uint32_t>(acc >> vlog);
        if (weight < 1) { why = "Error (generic)"; return false; }
        norm[s] = static_cast<int16_t>(weight);
        acc = end;
    }
    return true;
}

// Fills one row of the three device images (host staging copies) from its normalised counts.
bool build_row(const std::vector<int16_t> &norm, unsigned L, uint16_t *next, uint2 *sym, uint32_t *dec, std::string &why)
{
    const int nsym = static_cast<int>(norm.size());
    const uint32_t size = 1u << L, mask = size - 1, stride = (size >> 1) + (size >> 3) + 3;
    std::vector<uint16_t> cell(size);
    uint32_t high = size - 1, pos = 0;
    for (int s = 0; s < nsym; ++s) if (norm[s] == -1) cell[high--] = static_cast<uint16_t>(s);
    for (int s = 0; s < nsym; ++s)
        for (int k = 0; k < norm[s]; ++k) {
            cell[pos] = static_cast<uint16_t>(s);
            do pos = (pos + stride) & mask; while (pos > high);
        }
    if (pos != 0) { why = "Error (generic)"; return false; }
    std::vector<uint32_t> group(nsym + 1, 0), seen(nsym, 0);
    for (int s = 0; s < nsym; ++s) {
        const uint32_t c = norm[s] == -1 ? 1u : norm[s] > 0 ? static_cast<uint32_t>(norm[s]) : 0u;
        group[s + 1] = group[s] + c;
        seen[s] = c;
    }
    {
        std::vector<uint32_t> fill(group.begin(), group.end() - 1);
        for (uint32_t u = 0; u < size; ++u) next[fill[cell[u]]++] = static_cast<uint16_t>(size + u);
    }
    for (int s = 0; s < nsym; ++s) {
        if (norm[s] == 0) { sym[s] = make_uint2((L << 16) - size, 0u); continue; }   // never coded; keeps lookups in range
        const uint32_t c = group[s + 1] - group[s];
        const uint32_t bits = c == 1 ? L : L - top_bit(c - 1);
        sym[s] = make_uint2((bits << 16) - (c == 1 ? size : c << bits), static_cast<uint32_t>(static_cast<int32_t>(group[s]) - static_cast<int32_t>(c)));
    }
    for (uint32_t u = 0; u < size; ++u) {
        const uint16_t s = cell[u];
        const uint32_t nx = seen[s]++;
        const uint32_t nb = L - top_bit(nx);
        dec[u] = ((nx << nb) - size) | (nb << 12) | (static_cast<uint32_t>(s) << 16);
    }
    return true;
}

void free_tables(basic_tans_tables *t)
{
    if (!t) return;
    for (void *p : {static_cast<void *>(t->d_next), static_cast<void *>(t->d_sym), static_cast<void *>(t->d_dec),
                    static_cast<void *>(t->d_rowinfo), static_cast<void *>(t->d_ar)})
        if (p) (void)hipFree(p);
    delete t;
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 4); }
    template <typename T> T *as() { return static_cast<T *>(p); }
};

}  // namespace

extern "C" int basic_tans_tables_create(const int32_t *freqs, int rows, int freq_stride, const int32_t *nsym, const int32_t *offsets,
                                        int table_log, int max_symbol_value, int bypass_coding, int bypass_precision,
                                        basic_tans_tables **out)
{
    BASIC_REQUIRE(freqs && nsym && offsets && out && rows >= 1 && freq_stride >= 1, "init_params: null/empty argument");
    BASIC_REQUIRE(table_log >= 5 && table_log <= 12, "tANS: table_log must be in [5, 12] (TANS_MAX_TABLELOG, tans.hpp:16)");
    BASIC_REQUIRE(bypass_precision >= 1 && bypass_precision <= 8 && bypass_precision < table_log, "tANS: bypass_precision out of range");
    (void)max_symbol_value;   // the reference stores it and never reads it again (tans.hpp:44-47)
    int rc = require_device();
    if (rc) return rc;
    int max_nsym = bypass_coding ? (1 << bypass_precision) : 2;
    for (int r = 0; r < rows; ++r) {
        BASIC_REQUIRE(nsym[r] >= 2 && nsym[r] <= freq_stride && nsym[r] <= 65535, "init_params: num_symbols out of range");
        max_nsym = std::max(max_nsym, nsym[r]);
    }
    auto *t = new (std::nothrow) basic_tans_tables();
    if (!t) { set_error("out of host memory"); return BASIC_ERR_INVALID; }
    t->rows = rows; t->log = table_log; t->bypass = bypass_coding ? 1 : 0; t->bypass_precision = bypass_precision;
    t->max_nsym = max_nsym;
    t->nsym.assign(nsym, nsym + rows);
    t->offsets.assign(offsets, offsets + rows);
    const unsigned L = static_cast<unsigned>(table_log);
    const size_t size = size_t{1} << L, R = static_cast<size_t>(rows) + 1;
    std::vector<uint16_t> next(R * size, 0);
    std::vector<uint2> sym(R * max_nsym, make_uint2(0, 0));
    std::vector<uint32_t> dec(R * size, 0);
    std::vector<int2> info(R, make_int2(0, 1));
    std::vector<int16_t> norm;
    std::string why;
    for (size_t r = 0; r < R; ++r) {
        if (r == static_cast<size_t>(rows) && !t->bypass) break;
        std::vector<int32_t> uniform;
        const int32_t *f = freqs + r * freq_stride;
        int n = r < static_cast<size_t>(rows) ? nsym[r] : (1 << bypass_precision);
        if (r == static_cast<size_t>(rows)) { uniform.assign(n, 1); f = uniform.data(); }
        if (!normalize_counts(f, n, L, norm, why) || !build_row(norm, L, &next[r * size], &sym[r * max_nsym], &dec[r * size], why)) {
            set_error(why);
            free_tables(t);
            return BASIC_ERR_INVALID;
        }
        info[r] = make_int2(r < static_cast<size_t>(rows) ? offsets[r] : 0, n - 1);
    }
    hipError_t e = hipMalloc(&t->d_next, next.size() * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMalloc(&t->d_sym, sym.size() * sizeof(uint2));
    if (e == hipSuccess) e = hipMalloc(&t->d_dec, dec.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&t->d_rowinfo, info.size() * sizeof(int2));
    if (e == hipSuccess) e = hipMemcpy(t->d_next, next.data(), next.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->d_sym, sym.data(), sym.size() * sizeof(uint2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->d_dec, dec.data(), dec.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t->d_rowinfo, info.data(), info.size() * sizeof(int2), hipMemcpyHostToDevice);
    if (e != hipSuccess) { free_tables(t); return hip_fail(e, "tans_tables_create", __FILE__, __LINE__); }
    *out = t;
    return BASIC_OK;
}

extern "C" int basic_tans_tables_set_ar(basic_tans_tables *t, const int32_t *ar_tab, int k, int rows, int order, int s1)
{
    BASIC_REQUIRE(t && ar_tab && k >= 1 && rows == t->rows && s1 >= 1, "init_ar_params: the table's second dimension must be the number of distributions");
    BASIC_REQUIRE(order == 1 || order == 2, "Too many dimensions!");
    const size_t n = static_cast<size_t>(k) * rows * s1 * (order == 2 ? s1 : 1);
    if (t->d_ar) { (void)hipFree(t->d_ar); t->d_ar = nullptr; }
    BASIC_HIP_TRY(hipMalloc(&t->d_ar, n * sizeof(int32_t)));
    BASIC_HIP_TRY(hipMemcpy(t->d_ar, ar_tab, n * sizeof(int32_t), hipMemcpyHostToDevice));
    t->ar_k = k; t->ar_order = order; t->ar_s1 = s1;
    return BASIC_OK;
}

extern "C" void basic_tans_tables_destroy(basic_tans_tables *t) { free_tables(t); }

// Dumps one row of the device images back to the host (parity tests at table level).
extern "C" int basic_tans_tables_get_row(const basic_tans_tables *t, int row, uint16_t *next_state, uint32_t *delta_bits,
                                         int32_t *delta_state, uint32_t *dec_packed)
{
    BASIC_REQUIRE(t && row >= 0 && row <= t->rows - (t->bypass ? 0 : 1), "tans_tables_get_row: bad row");
    const size_t size = size_t{1} << t->log;
    const int n = row < t->rows ? t->nsym[row] : (1 << t->bypass_precision);
    std::vector<uint2> sym(n);
    if (next_state) BASIC_HIP_TRY(hipMemcpy(next_state, t->d_next + row * size, size * sizeof(uint16_t), hipMemcpyDeviceToHost));
    BASIC_HIP_TRY(hipMemcpy(sym.data(), t->d_sym + static_cast<size_t>(row) * t->max_nsym, n * sizeof(uint2), hipMemcpyDeviceToHost));
    for (int s = 0; s < n; ++s) {
        if (delta_bits) delta_bits[s] = sym[s].x;
        if (delta_state) delta_state[s] = static_cast<int32_t>(sym[s].y);
    }
    if (dec_packed) BASIC_HIP_TRY(hipMemcpy(dec_packed, t->d_dec + row * size, size * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return BASIC_OK;
}

// ---------------------------------------------------------------------------------------
// Device side
// ---------------------------------------------------------------------------------------
namespace {

struct TansDev {
    const uint16_t *next;
    const uint2 *sym;
    const uint32_t *dec;
    const int2 *rowinfo;
    int rows, log, bypass, bypass_precision, max_nsym;
    int lds_words;           // > 0: the state-dependent image (encoder: next, decoder: dec) is copied to LDS by every workgroup
    const int32_t *ar_tab;   // nullptr = no AR remap
    int ar_k, ar_order, ar_s1;
    const int32_t *ar_indexes, *off0, *off1;
};

__device__ __forceinline__ int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ans_interface.hpp:89-104; every index clamped so that malformed input cannot leave the tables
__device__ __forceinline__ int32_t ar_row(const TansDev &T, int32_t a, int32_t row, int32_t v0, int32_t v1)
{
    a = clampi(a, 0, T.ar_k - 1);
    row = clampi(row, 0, T.rows - 1);
    v0 = clampi(v0, 0, T.ar_s1 - 1);
    if (T.ar_order == 1) return T.ar_tab[(static_cast<int64_t>(a) * T.rows + row) * T.ar_s1 + v0];
    v1 = clampi(v1, 0, T.ar_s1 - 1);
    return T.ar_tab[((static_cast<int64_t>(a) * T.rows + row) * T.ar_s1 + v0) * T.ar_s1 + v1];
}

__device__ __forceinline__ uint32_t bcast(uint32_t v, int lane)
{
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), lane));
}

// Bit sink of one stream: 32-bit words from the slot's start upwards, written by lane 0.
struct BitSink {
    uint32_t *words;
    int64_t cap_words, nwords;
    uint64_t acc;        // pending bits, low `fill` valid
    uint32_t fill;
    int64_t total_bits;  // set by finish()
    // append a field without flushing: the caller flushes before fill + nbits could pass 64 (fill < 32 after flush32, fields <= 16 bits:
    // two fields per flush)
    __device__ __forceinline__ void add(uint32_t value, uint32_t nbits)
    {
        acc |= static_cast<uint64_t>(value & ((1u << nbits) - 1u)) << fill;   // nbits <= 16
        fill += nbits;
    }
    __device__ __forceinline__ void flush32(int lane)
    {
        if (fill >= 32u) {
            if (lane == 0 && nwords < cap_words) words[nwords] = static_cast<uint32_t>(acc);
            ++nwords;
            acc >>= 32;
            fill -= 32u;
        }
    }
    __device__ __forceinline__ void put(uint32_t value, uint32_t nbits, int lane) { add(value, nbits); flush32(lane); }
    __device__ __forceinline__ void finish(int lane)
    {
        total_bits = nwords * 32 + fill;
        if (fill) {
            if (lane == 0 && nwords < cap_words) words[nwords] = static_cast<uint32_t>(acc);
            ++nwords;
        }
    }
};

// The state-dependent images, from the workgroup's LDS copy (LDS: a ds_read with a 32-bit address -- through a generic pointer
// it would be a flat load with 64-bit address arithmetic on the chain) or from L2
extern __shared__ uint32_t tans_lds[];
template <bool LDS> __device__ __forceinline__ uint32_t enc_next(const TansDev &T, uint32_t i)
{
    return LDS ? reinterpret_cast<const uint16_t *>(tans_lds)[i] : T.next[i];
}
template <bool LDS> __device__ __forceinline__ uint32_t dec_entry(const TansDev &T, uint32_t i) { return LDS ? tans_lds[i] : T.dec[i]; }

// tbase = (row << log) + the symbol's group start - its count: the lookup index is tbase + (state >> nb), inside the row's 2^log
// entries for every state in [2^log, 2^(log+1)) by the tables' construction (symbols without mass point at entry 1).
template <bool LDS> __device__ __forceinline__ void tans_step(const TansDev &T, BitSink &sink, uint32_t &state, uint32_t dbits, uint32_t tbase, int lane)
{   // Tans_encodeSymbol, tans.cpp:245-252
    const uint32_t nb = (state + dbits) >> 16;
    sink.put(state, nb, lane);
    state = enc_next<LDS>(T, tbase + (state >> nb));
    state = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(state)));
}

// the same step, the flush test left to the caller (BitSink::add)
template <bool LDS> __device__ __forceinline__ void tans_step_add(const TansDev &T, BitSink &sink, uint32_t &state, uint32_t dbits, uint32_t tbase)
{
    const uint32_t nb = (state + dbits) >> 16;
    sink.add(state, nb);
    state = enc_next<LDS>(T, tbase + (state >> nb));
    state = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(state)));
}

template <int kFrom, int kTo, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (kFrom < kTo) {
        f(std::integral_constant<int, kFrom>{});
        static_for<kFrom + 1, kTo>(f);
    }
}

template <bool LDS> __device__ __forceinline__ void tans_step_bypass(const TansDev &T, BitSink &sink, uint32_t &state, uint32_t v, int lane)
{
    const uint2 e = T.sym[static_cast<size_t>(T.rows) * T.max_nsym + v];
    tans_step<LDS>(T, sink, state, e.x, (static_cast<uint32_t>(T.rows) << T.log) + e.y, lane);
}

// One wavefront per stream.  out_info[stream] = { total bits incl. final state and end mark, coded symbols incl. bypass }.
template <bool LDS>
__global__ __launch_bounds__(64) void tans_encode_kernel(TansDev T, const int32_t *__restrict__ symbols, const int32_t *__restrict__ indexes,
                                                         const int64_t *__restrict__ seg, uint32_t *out_words, int64_t slot_words,
                                                         int64_t *out_info)
{
    __builtin_amdgcn_s_setprio(3);   // a serial chain: never lose the issue arbitration (see rans_encode_fast_kernel)
    if (LDS) {   // a lookup that depends on the previous state then costs an LDS access instead of an L2 round trip
        const uint32_t *src = reinterpret_cast<const uint32_t *>(T.next);
        for (int e = threadIdx.x; e < T.lds_words; e += 64) tans_lds[e] = src[e];
        __syncthreads();
    }
    const int stream = blockIdx.x, lane = threadIdx.x;
    const int64_t beg = seg[stream], n = seg[stream + 1] - beg;
    const int32_t *sym = symbols + beg, *idx = indexes + beg;
    BitSink sink{out_words + static_cast<int64_t>(stream) * slot_words, slot_words, 0, 0, 0, 0};
    uint32_t state = 1u << T.log;
    int64_t coded = 0;
    const uint32_t bprec = static_cast<uint32_t>(T.bypass_precision), maxbv = (1u << bprec) - 1u;
    // per chunk of 64 symbols (lane 63 holds the chunk's last symbol): row, offset, clamp / bypass split, the symbol's transform
    // pair -- three dependent gathers, so chunk k + 1 is prepared while chunk k runs through the chain
    struct Prepared { uint32_t dbits, tbase, raw; bool is_bypass; };
    auto prepare = [&](int64_t hi) -> Prepared {
        Prepared q{0u, 0u, 0u, false};
        const int64_t i = hi - 64 + lane;
        if (hi > 0 && i >= 0) {
            int32_t row = idx[i];
            if (T.ar_tab) {
                const int64_t g = beg + i;
                const int32_t a = T.ar_indexes ? T.ar_indexes[g] : 0;
                const int32_t d0 = T.off0[g];
                const int32_t v0 = (d0 > 0 && d0 <= i) ? sym[i - d0] + 1 : 0;
                int32_t v1 = 0;
                if (T.ar_order == 2) {
                    const int32_t d1 = T.off1[g];
                    v1 = (d1 > 0 && d1 <= i) ? sym[i - d1] + 1 : 0;
                }
                row = ar_row(T, a, row, v0, v1);
            }
            row = clampi(row, 0, T.rows - 1);
            const int2 ri = T.rowinfo[row];
            const int32_t max_value = ri.y;
            int32_t value = sym[i] - ri.x;
            if (value < 0) { q.raw = static_cast<uint32_t>(-2 * value - 1); value = max_value; }
            else if (value >= max_value) { q.raw = static_cast<uint32_t>(2 * (value - max_value)); value = max_value; }
            q.is_bypass = T.bypass && value == max_value;
            const uint2 e = T.sym[static_cast<size_t>(row) * T.max_nsym + value];
            q.dbits = e.x; q.tbase = (static_cast<uint32_t>(row) << T.log) + e.y;   // e.y is an int32 offset: unsigned wrap-around adds it
        }
        return q;
    };
    Prepared nxt = prepare(n);
    for (int64_t hi = n; hi > 0; hi -= 64) {
        const Prepared cur = nxt;
        nxt = prepare(hi - 64);
        const uint32_t dbits = cur.dbits, tbase = cur.tbase, raw = cur.raw;
        const bool is_bypass = cur.is_bypass;
        const uint64_t bypass_mask = __ballot(is_bypass);
        const int j_lo = hi >= 64 ? 0 : static_cast<int>(64 - hi);
        auto slow_symbol = [&](int j) {
            if ((bypass_mask >> j) & 1ull) {
                // decode order: sentinel, digit count (unary in units of maxbv), digits low first  =>  coded reversed
                const uint32_t r = bcast(raw, j);
                int nb = 0;
                while (nb * bprec < 32u && (r >> (nb * bprec)) != 0u) ++nb;
                for (int k = nb - 1; k >= 0; --k) tans_step_bypass<LDS>(T, sink, state, (r >> (k * bprec)) & maxbv, lane);
                tans_step_bypass<LDS>(T, sink, state, static_cast<uint32_t>(nb) % maxbv, lane);
                for (uint32_t k = 0; k < static_cast<uint32_t>(nb) / maxbv; ++k) tans_step_bypass<LDS>(T, sink, state, maxbv, lane);
                coded += nb + 1 + static_cast<int>(static_cast<uint32_t>(nb) / maxbv);
            }
            tans_step<LDS>(T, sink, state, bcast(dbits, j), bcast(tbase, j), lane);
        };
        if (j_lo == 0) {
            // a full chunk, eight symbols at a time (lane 63 first): a group without a bypass symbol runs unrolled -- lane ids
            // as immediates, two bit fields per flush test; the chain is ~15 instructions + one table lookup per symbol
            static_for<0, 8>([&](auto gc) {
                constexpr int G = 7 - decltype(gc)::value;
                if (((bypass_mask >> (8 * G)) & 0xFFull) == 0ull) {
                    static_for<0, 4>([&](auto pc) {
                        constexpr int J = 8 * G + 7 - 2 * decltype(pc)::value;
                        tans_step_add<LDS>(T, sink, state, bcast(dbits, J), bcast(tbase, J));
                        tans_step_add<LDS>(T, sink, state, bcast(dbits, J - 1), bcast(tbase, J - 1));
                        sink.flush32(lane);
                    });
                } else {
                    for (int j = 8 * G + 7; j >= 8 * G; --j) slow_symbol(j);
                }
            });
            coded += 64;
        } else {
            for (int j = 63; j >= j_lo; --j) slow_symbol(j);
            coded += 64 - j_lo;
        }
    }
    sink.put(state, static_cast<uint32_t>(T.log), lane);   // Tans_flushCState, tans.cpp:254-258
    sink.put(1u, 1u, lane);                                // end mark, bitstream.h:242
    sink.finish(lane);
    if (lane == 0) {
        out_info[2 * stream] = sink.nwords <= sink.cap_words ? sink.total_bits : -1;
        out_info[2 * stream + 1] = coded;
    }
}

// Bit source of one stream: fields come off the top, below the end mark.  The stream is cut into 32-bit words counted from
// its first byte; `c` holds the unread bits [lo, lo + have) with lo a multiple of 32; below the stream's first byte it
// supplies zeros (the reference's behaviour there is undefined).  64 words live in a register window (lane k = word
// wbase + k) and are handed out by lane broadcast: no memory access on the coder's serial chain except one window load
// per 2,048 bits.
struct BitSource {
    const uint8_t *bytes;
    int64_t len;         // bytes in the stream
    int64_t lo;          // bit index of the lowest bit held (multiple of 32)
    uint64_t c;
    uint32_t have;
    int64_t wbase;       // first word of the window (multiple of 64)
    uint32_t window;     // lane k: word wbase + k (bytes past the stream read as 0)
    __device__ __forceinline__ uint32_t load_word(int64_t k) const
    {
        const int64_t b = 4 * k;
        uint32_t w = 0;
        if (b + 3 < len) {
            w = static_cast<uint32_t>(bytes[b]) | (static_cast<uint32_t>(bytes[b + 1]) << 8) | (static_cast<uint32_t>(bytes[b + 2]) << 16) |
                (static_cast<uint32_t>(bytes[b + 3]) << 24);
        } else {
            for (int i = 0; i < 4; ++i)
                if (b + i < len) w |= static_cast<uint32_t>(bytes[b + i]) << (8 * i);
        }
        return w;
    }
    __device__ __forceinline__ void fill_window(int64_t k, int lane)   // window that holds word k
    {
        wbase = k & ~int64_t{63};
        window = load_word(wbase + lane);
    }
    // data_bits = bits below the end mark
    __device__ __forceinline__ void init(const uint8_t *p, int64_t nbytes, int64_t data_bits, int lane)
    {
        bytes = p; len = nbytes;
        const int64_t wtop = data_bits > 0 ? (data_bits - 1) >> 5 : 0;
        fill_window(wtop, lane);
        lo = wtop * 32;
        have = static_cast<uint32_t>(data_bits - lo);   // 0..32
        const uint32_t w = bcast(window, static_cast<int>(wtop - wbase));
        c = have >= 32u ? w : (w & ((1u << have) - 1u));
    }
    __device__ __forceinline__ void refill(int lane)
    {
        if (have < 32u) {
            uint32_t w = 0;
            if (lo > 0) {
                const int64_t k = (lo >> 5) - 1;
                if (k < wbase) fill_window(k, lane);
                w = bcast(window, static_cast<int>(k - wbase));
                lo -= 32;
            }
            c = (c << 32) | w;
            have += 32u;
        }
    }
    __device__ __forceinline__ uint32_t take(uint32_t nbits)   // nbits <= 16, have >= nbits
    {
        have -= nbits;
        return static_cast<uint32_t>(c >> have) & ((1u << nbits) - 1u);
    }
};

template <bool LDS> __device__ __forceinline__ uint32_t tans_unstep(const TansDev &T, BitSource &src, uint32_t &state, uint32_t row)
{   // Tans_decodeSymbol, tans.cpp:338-364
    uint32_t e = dec_entry<LDS>(T, (row << T.log) + (state & ((1u << T.log) - 1u)));
    e = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(e)));
    src.refill(static_cast<int>(threadIdx.x));
    state = (e & 0xFFFu) + src.take((e >> 12) & 0xFu);
    return e >> 16;
}

// One wavefront per stream: stream s = bytes[byte_off[s] .. byte_off[s+1]).  status[s]: 0 ok, 1 empty stream / no end mark.
template <bool LDS>
__global__ __launch_bounds__(64) void tans_decode_kernel(TansDev T, const uint8_t *__restrict__ bytes_all, const int64_t *__restrict__ byte_off,
                                                         const int32_t *__restrict__ indexes, const int64_t *__restrict__ seg,
                                                         int32_t *out_symbols, int32_t *status)
{
    __builtin_amdgcn_s_setprio(3);   // a serial chain: never lose the issue arbitration (see rans_encode_fast_kernel)
    if (LDS) {
        for (int e = threadIdx.x; e < T.lds_words; e += 64) tans_lds[e] = T.dec[e];
        __syncthreads();
    }
    const int stream = blockIdx.x, lane = threadIdx.x;
    const int64_t beg = seg[stream], n = seg[stream + 1] - beg;
    const int32_t *idx = indexes + beg;
    int32_t *out = out_symbols + beg;
    const uint8_t *bytes = bytes_all + byte_off[stream];
    const int64_t len = byte_off[stream + 1] - byte_off[stream];
    const uint32_t last = len >= 1 ? bytes[len - 1] : 0u;
    if (last == 0u) {   // BIT_initDStream: srcSize_wrong / end mark not present (bitstream.h:262,270)
        if (lane == 0) status[stream] = 1;
        return;
    }
    const uint32_t top = 31u - static_cast<uint32_t>(__builtin_clz(last));   // the end mark's bit in the last byte
    BitSource src;
    src.init(bytes, len, (len - 1) * 8 + top, lane);
    src.refill(lane);
    uint32_t state = src.take(static_cast<uint32_t>(T.log));   // Tans_initDState, tans.cpp:330-335
    const uint32_t bprec = static_cast<uint32_t>(T.bypass_precision), maxbv = (1u << bprec) - 1u;
    // a chunk's rows and row info are two dependent gathers: requested one chunk ahead
    struct Rows { uint32_t row; int2 ri; };
    auto rows_of = [&](int64_t c0) -> Rows {
        Rows r{0u, make_int2(0, 1)};
        const int64_t i = c0 + lane;
        if (i < n) {
            r.row = static_cast<uint32_t>(T.ar_tab ? idx[i] : clampi(idx[i], 0, T.rows - 1));
            if (!T.ar_tab) r.ri = T.rowinfo[r.row];
        }
        return r;
    };
    Rows nxt = rows_of(0);
    for (int64_t c0 = 0; c0 < n; c0 += 64) {
        const int64_t i = c0 + lane;
        const Rows cur = nxt;
        nxt = rows_of(c0 + 64);
        const uint32_t row_l = cur.row;
        const int2 ri_l = cur.ri;
        int32_t result = 0;
        const int cnt = (n - c0) < 64 ? static_cast<int>(n - c0) : 64;
        if (!T.ar_tab && cnt == 64) {
            // a full chunk without AR remap, unrolled: lane ids as immediates, the row's table base prepared per lane, the
            // offset added lane-parallel after the chunk -- what stays per symbol is the lookup, the refill test, the field
            // and the bypass test
            const uint32_t base_l = row_l << T.log, esc_l = T.bypass ? static_cast<uint32_t>(ri_l.y) : 0xFFFFFFFFu;
            const uint32_t smask = (1u << T.log) - 1u;
            static_for<0, 64>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                int32_t &res = result;   // (generic lambda: the asm operand below needs an odr-use to capture it)
                uint32_t e = dec_entry<LDS>(T, bcast(base_l, J) + (state & smask));
                e = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(e)));
                src.refill(lane);
                state = (e & 0xFFFu) + src.take((e >> 12) & 0xFu);
                int32_t value = static_cast<int32_t>(e >> 16);
                if (__builtin_expect(static_cast<uint32_t>(value) == bcast(esc_l, J), 0)) {
                    const int32_t max_value = value;
                    uint32_t v = tans_unstep<LDS>(T, src, state, static_cast<uint32_t>(T.rows)), nb = v;
                    while (v == maxbv && nb < 64u * maxbv) { v = tans_unstep<LDS>(T, src, state, static_cast<uint32_t>(T.rows)); nb += v; }
                    uint32_t raw = 0;
                    for (uint32_t k = 0; k < nb; ++k) {
                        const uint32_t d = tans_unstep<LDS>(T, src, state, static_cast<uint32_t>(T.rows));
                        if (k * bprec < 32u) raw |= d << (k * bprec);
                    }
                    value = static_cast<int32_t>(raw >> 1);
                    if (raw & 1u) value = -value - 1; else value += max_value;
                }
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(res) : "s"(value), "n"(J));
            });
            out[i] = result + ri_l.x;
            continue;
        }
        for (int j = 0; j < cnt; ++j) {
            int32_t row = static_cast<int32_t>(bcast(row_l, j));
            int32_t offset, max_value;
            if (T.ar_tab) {
                const int64_t e = c0 + j, g = beg + e;
                const int32_t a = T.ar_indexes ? T.ar_indexes[g] : 0;
                const int32_t d0 = T.off0[g];
                const int32_t v0 = (d0 > 0 && d0 <= e) ? __hip_atomic_load(out + (e - d0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 : 0;
                int32_t v1 = 0;
                if (T.ar_order == 2) {
                    const int32_t d1 = T.off1[g];
                    v1 = (d1 > 0 && d1 <= e) ? __hip_atomic_load(out + (e - d1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 : 0;
                }
                row = __builtin_amdgcn_readfirstlane(clampi(ar_row(T, a, row, v0, v1), 0, T.rows - 1));
                const int2 ri = T.rowinfo[row];
                offset = __builtin_amdgcn_readfirstlane(ri.x);
                max_value = __builtin_amdgcn_readfirstlane(ri.y);
            } else {
                offset = static_cast<int32_t>(bcast(static_cast<uint32_t>(ri_l.x), j));
                max_value = static_cast<int32_t>(bcast(static_cast<uint32_t>(ri_l.y), j));
            }
            int32_t value = static_cast<int32_t>(tans_unstep<LDS>(T, src, state, static_cast<uint32_t>(row)));
            if (T.bypass && value == max_value) {
                uint32_t v = tans_unstep<LDS>(T, src, state, static_cast<uint32_t>(T.rows)), nb = v;
                while (v == maxbv && nb < 64u * maxbv) { v = tans_unstep<LDS>(T, src, state, static_cast<uint32_t>(T.rows)); nb += v; }
                uint32_t raw = 0;
                for (uint32_t k = 0; k < nb; ++k) {
                    const uint32_t d = tans_unstep<LDS>(T, src, state, static_cast<uint32_t>(T.rows));
                    if (k * bprec < 32u) raw |= d << (k * bprec);
                }
                value = static_cast<int32_t>(raw >> 1);
                if (raw & 1u) value = -value - 1; else value += max_value;
            }
            value += offset;
            if (T.ar_tab) { if (lane == 0) __hip_atomic_store(out + c0 + j, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            else if (lane == j) result = value;
        }
        if (!T.ar_tab && i < n) out[i] = result;
    }
    if (lane == 0) status[stream] = 0;
}

// bytes of the state-dependent image when it fits a workgroup's LDS (else 0): encoder 2 bytes, decoder 4 bytes per state and row
size_t tans_lds_bytes(const basic_tans_tables *t, bool decoder)
{
    const size_t rows = static_cast<size_t>(t->rows) + (t->bypass ? 1 : 0);
    const size_t bytes = rows * (size_t{1} << t->log) * (decoder ? 4 : 2);
    return bytes <= 144 * 1024 && !std::getenv("BASIC_TANS_NO_LDS") ? bytes : 0;
}

TansDev dev_view(const basic_tans_tables *t)
{
    TansDev T{};
    T.next = t->d_next; T.sym = t->d_sym; T.dec = t->d_dec; T.rowinfo = t->d_rowinfo;
    T.rows = t->rows; T.log = t->log; T.bypass = t->bypass; T.bypass_precision = t->bypass_precision; T.max_nsym = t->max_nsym;
    return T;
}

}  // namespace

// Words a slot must hold for a stream of n symbols in the worst case (every symbol a bypass escape with eight digits).
extern "C" int64_t basic_tans_encode_bound_words(const basic_tans_tables *t, int64_t n)
{
    if (!t || n < 0) return -1;
    const int64_t digits = (32 + t->bypass_precision - 1) / t->bypass_precision, maxbv = (int64_t{1} << t->bypass_precision) - 1;
    const int64_t per_symbol = t->bypass ? t->log * (2 + digits + digits / maxbv) : t->log;
    return (n * per_symbol + t->log + 1 + 31) / 32 + 1;
}

extern "C" int basic_tans_encode_batch_dev(const basic_tans_tables *t, const int32_t *d_symbols, const int32_t *d_indexes,
                                           const int64_t *d_seg, int nstreams, uint32_t *d_out_words, int64_t slot_words,
                                           int64_t *d_out_info, void *hip_stream)
{
    if (!t) { set_error("ANS not initialized!"); return BASIC_ERR_NOT_INIT; }
    BASIC_REQUIRE(!t->d_ar, "tans_encode_batch_dev: AR remap needs the host entry point (per-element AR arrays)");
    BASIC_REQUIRE(nstreams >= 0 && slot_words >= 1 && d_seg && d_out_words && d_out_info, "tans_encode_batch_dev: bad argument");
    if (nstreams == 0) return BASIC_OK;
    TansDev T = dev_view(t);
    const size_t lds = tans_lds_bytes(t, false);
    T.lds_words = static_cast<int>(lds / 4);
    if (lds > 48 * 1024) BASIC_HIP_TRY(ensure_max_lds(reinterpret_cast<const void *>(tans_encode_kernel<true>)));
    if (lds) hipLaunchKernelGGL(tans_encode_kernel<true>, dim3(nstreams), dim3(64), lds, as_stream(hip_stream), T, d_symbols, d_indexes, d_seg,
                                d_out_words, slot_words, d_out_info);
    else hipLaunchKernelGGL(tans_encode_kernel<false>, dim3(nstreams), dim3(64), 0, as_stream(hip_stream), T, d_symbols, d_indexes, d_seg,
                            d_out_words, slot_words, d_out_info);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}

extern "C" int basic_tans_decode_batch_dev(const basic_tans_tables *t, const uint8_t *d_bytes, const int64_t *d_byte_off,
                                           const int32_t *d_indexes, const int64_t *d_seg, int nstreams, int32_t *d_out_symbols,
                                           int32_t *d_status, void *hip_stream)
{
    if (!t) { set_error("ANS not initialized!"); return BASIC_ERR_NOT_INIT; }
    BASIC_REQUIRE(!t->d_ar, "tans_decode_batch_dev: AR remap needs the host entry point (per-element AR arrays)");
    BASIC_REQUIRE(nstreams >= 0 && d_bytes && d_byte_off && d_seg && d_out_symbols && d_status, "tans_decode_batch_dev: bad argument");
    if (nstreams == 0) return BASIC_OK;
    TansDev T = dev_view(t);
    const size_t lds = tans_lds_bytes(t, true);
    T.lds_words = static_cast<int>(lds / 4);
    if (lds > 48 * 1024) BASIC_HIP_TRY(ensure_max_lds(reinterpret_cast<const void *>(tans_decode_kernel<true>)));
    if (lds) hipLaunchKernelGGL(tans_decode_kernel<true>, dim3(nstreams), dim3(64), lds, as_stream(hip_stream), T, d_bytes, d_byte_off, d_indexes,
                                d_seg, d_out_symbols, d_status);
    else hipLaunchKernelGGL(tans_decode_kernel<false>, dim3(nstreams), dim3(64), 0, as_stream(hip_stream), T, d_bytes, d_byte_off, d_indexes,
                            d_seg, d_out_symbols, d_status);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}

// ---------------------------------------------------------------------------------------
// Host-buffer drop-ins (single stream): stage -> kernel -> copy back.
// ---------------------------------------------------------------------------------------
namespace {

int stage_ar(const basic_tans_tables *t, TansDev &T, int64_t n, const int32_t *ar_indexes, const int32_t *ar_off0, const int32_t *ar_off1,
             DevBuf &b_ai, DevBuf &b_o0, DevBuf &b_o1)
{
    if (!t->d_ar) return BASIC_OK;
    if (!ar_off0 || (t->ar_order == 2 && !ar_off1)) { set_error("ar_offsets is required for ar coding!"); return BASIC_ERR_INVALID; }
    const size_t bytes = static_cast<size_t>(n) * sizeof(int32_t);
    T.ar_tab = t->d_ar; T.ar_k = t->ar_k; T.ar_order = t->ar_order; T.ar_s1 = t->ar_s1;
    if (ar_indexes) {
        BASIC_HIP_TRY(b_ai.alloc(bytes));
        BASIC_HIP_TRY(hipMemcpy(b_ai.p, ar_indexes, bytes, hipMemcpyHostToDevice));
        T.ar_indexes = b_ai.as<int32_t>();
    }
    BASIC_HIP_TRY(b_o0.alloc(bytes));
    BASIC_HIP_TRY(hipMemcpy(b_o0.p, ar_off0, bytes, hipMemcpyHostToDevice));
    T.off0 = b_o0.as<int32_t>();
    if (t->ar_order == 2) {
        BASIC_HIP_TRY(b_o1.alloc(bytes));
        BASIC_HIP_TRY(hipMemcpy(b_o1.p, ar_off1, bytes, hipMemcpyHostToDevice));
        T.off1 = b_o1.as<int32_t>();
    }
    return BASIC_OK;
}

}  // namespace

// TansEncoder::encode_with_indexes (tans.cpp:527-680).  capacity_syms = the symbol count the reference sizes its output
// buffer with (-1: n; flush(): the cached count incl. bypass digits, tans.cpp:686): a capacity of 8 bytes or less is the
// reference's "Destination buffer is too small" error, and a stream of capacity - 8 whole bytes or more is "not storable"
// there and comes back EMPTY (bitstream.h:192,245) -- *out_len = 0 here too.  *coded_syms (optional) = symbols coded incl.
// bypass digits.
extern "C" int basic_tans_encode_host(const basic_tans_tables *t, const int32_t *symbols, const int32_t *indexes, int64_t n,
                                      const int32_t *ar_indexes, const int32_t *ar_off0, const int32_t *ar_off1,
                                      int64_t capacity_syms, uint8_t *out, int64_t out_capacity, int64_t *out_len, int64_t *coded_syms)
{
    if (!t) { set_error("ANS not initialized!"); return BASIC_ERR_NOT_INIT; }
    BASIC_REQUIRE(n >= 0 && out && out_len && (n == 0 || (symbols && indexes)), "encode_with_indexes: bad argument");
    const int64_t cap = (capacity_syms < 0 ? n : capacity_syms) * t->log / 8;
    BASIC_REQUIRE(cap > 8, "Destination buffer is too small");
    const int64_t slot_words = basic_tans_encode_bound_words(t, n);
    DevBuf b_sym, b_idx, b_seg, b_out, b_info, b_ai, b_o0, b_o1;
    const size_t bytes = static_cast<size_t>(n) * sizeof(int32_t);
    BASIC_HIP_TRY(b_sym.alloc(bytes));
    BASIC_HIP_TRY(b_idx.alloc(bytes));
    BASIC_HIP_TRY(b_seg.alloc(2 * sizeof(int64_t)));
    BASIC_HIP_TRY(b_out.alloc(static_cast<size_t>(slot_words) * 4));
    BASIC_HIP_TRY(b_info.alloc(2 * sizeof(int64_t)));
    if (n) {
        BASIC_HIP_TRY(hipMemcpy(b_sym.p, symbols, bytes, hipMemcpyHostToDevice));
        BASIC_HIP_TRY(hipMemcpy(b_idx.p, indexes, bytes, hipMemcpyHostToDevice));
    }
    const int64_t seg[2] = {0, n};
    BASIC_HIP_TRY(hipMemcpy(b_seg.p, seg, sizeof(seg), hipMemcpyHostToDevice));
    TansDev T = dev_view(t);
    int rc = stage_ar(t, T, n, ar_indexes, ar_off0, ar_off1, b_ai, b_o0, b_o1);
    if (rc) return rc;
    const size_t lds = n >= 2048 ? tans_lds_bytes(t, false) : 0;   // short streams: the copy would cost more than it saves
    T.lds_words = static_cast<int>(lds / 4);
    if (lds > 48 * 1024) BASIC_HIP_TRY(ensure_max_lds(reinterpret_cast<const void *>(tans_encode_kernel<true>)));
    if (lds) hipLaunchKernelGGL(tans_encode_kernel<true>, dim3(1), dim3(64), lds, nullptr, T, b_sym.as<int32_t>(), b_idx.as<int32_t>(), b_seg.as<int64_t>(),
                                b_out.as<uint32_t>(), slot_words, b_info.as<int64_t>());
    else hipLaunchKernelGGL(tans_encode_kernel<false>, dim3(1), dim3(64), 0, nullptr, T, b_sym.as<int32_t>(), b_idx.as<int32_t>(), b_seg.as<int64_t>(),
                            b_out.as<uint32_t>(), slot_words, b_info.as<int64_t>());
    BASIC_HIP_TRY(hipGetLastError());
    int64_t info[2] = {0, 0};
    BASIC_HIP_TRY(hipMemcpy(info, b_info.p, sizeof(info), hipMemcpyDeviceToHost));
    if (info[0] < 0) { set_error("tans encoder: slot overflow"); return BASIC_ERR_OVERFLOW; }
    if (coded_syms) *coded_syms = info[1];
    if ((info[0] >> 3) >= cap - 8) { *out_len = 0; return BASIC_OK; }
    const int64_t nbytes = (info[0] + 7) >> 3;
    *out_len = nbytes;
    if (nbytes > out_capacity) { set_error("encode_with_indexes: output buffer too small"); return BASIC_ERR_OVERFLOW; }
    BASIC_HIP_TRY(hipMemcpy(out, b_out.p, nbytes, hipMemcpyDeviceToHost));
    return BASIC_OK;
}

// TansDecoder::decode_with_indexes (tans.cpp:722-815).
extern "C" int basic_tans_decode_host(const basic_tans_tables *t, const uint8_t *stream, int64_t stream_len, const int32_t *indexes,
                                      int64_t n, const int32_t *ar_indexes, const int32_t *ar_off0, const int32_t *ar_off1,
                                      int32_t *out_symbols)
{
    if (!t) { set_error("ANS not initialized!"); return BASIC_ERR_NOT_INIT; }
    BASIC_REQUIRE(n >= 0 && (n == 0 || (indexes && out_symbols)), "decode_with_indexes: bad argument");
    BASIC_REQUIRE(stream && stream_len >= 1, "Src size is incorrect");
    BASIC_REQUIRE(stream[stream_len - 1] != 0, "Error (generic)");   // end mark not present
    DevBuf b_bytes, b_boff, b_idx, b_seg, b_out, b_status, b_ai, b_o0, b_o1;
    const size_t bytes = static_cast<size_t>(n) * sizeof(int32_t);
    BASIC_HIP_TRY(b_bytes.alloc(static_cast<size_t>(stream_len)));
    BASIC_HIP_TRY(b_boff.alloc(2 * sizeof(int64_t)));
    BASIC_HIP_TRY(b_idx.alloc(bytes));
    BASIC_HIP_TRY(b_seg.alloc(2 * sizeof(int64_t)));
    BASIC_HIP_TRY(b_out.alloc(bytes));
    BASIC_HIP_TRY(b_status.alloc(sizeof(int32_t)));
    BASIC_HIP_TRY(hipMemcpy(b_bytes.p, stream, stream_len, hipMemcpyHostToDevice));
    const int64_t boff[2] = {0, stream_len}, seg[2] = {0, n};
    BASIC_HIP_TRY(hipMemcpy(b_boff.p, boff, sizeof(boff), hipMemcpyHostToDevice));
    BASIC_HIP_TRY(hipMemcpy(b_seg.p, seg, sizeof(seg), hipMemcpyHostToDevice));
    if (n) BASIC_HIP_TRY(hipMemcpy(b_idx.p, indexes, bytes, hipMemcpyHostToDevice));
    TansDev T = dev_view(t);
    int rc = stage_ar(t, T, n, ar_indexes, ar_off0, ar_off1, b_ai, b_o0, b_o1);
    if (rc) return rc;
    const size_t lds = n >= 2048 ? tans_lds_bytes(t, true) : 0;
    T.lds_words = static_cast<int>(lds / 4);
    if (lds > 48 * 1024) BASIC_HIP_TRY(ensure_max_lds(reinterpret_cast<const void *>(tans_decode_kernel<true>)));
    if (lds) hipLaunchKernelGGL(tans_decode_kernel<true>, dim3(1), dim3(64), lds, nullptr, T, b_bytes.as<uint8_t>(), b_boff.as<int64_t>(), b_idx.as<int32_t>(),
                                b_seg.as<int64_t>(), b_out.as<int32_t>(), b_status.as<int32_t>());
    else hipLaunchKernelGGL(tans_decode_kernel<false>, dim3(1), dim3(64), 0, nullptr, T, b_bytes.as<uint8_t>(), b_boff.as<int64_t>(), b_idx.as<int32_t>(),
                            b_seg.as<int64_t>(), b_out.as<int32_t>(), b_status.as<int32_t>());
    BASIC_HIP_TRY(hipGetLastError());
    int32_t status = 0;
    BASIC_HIP_TRY(hipMemcpy(&status, b_status.p, sizeof(status), hipMemcpyDeviceToHost));
    if (status) { set_error("Error (generic)"); return BASIC_ERR_INVALID; }
    if (n) BASIC_HIP_TRY(hipMemcpy(out_symbols, b_out.p, bytes, hipMemcpyDeviceToHost));
    return BASIC_OK;
}
