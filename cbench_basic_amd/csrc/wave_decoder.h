// One wavefront decodes one rANS stream against the LDS-resident search image (see basic_rans_tables in rans.hip): the
// serial chain shared by rans_decode_fast_kernel (rans.hip) and the decoder wavefronts of the persistent scan-line launches
// (scanline.hip).  Reference: Rans64Decoder::decode_with_indexes, rans64.cpp:389-499; state update rans64.h:107-142.
//
// A lone wavefront issues one instruction per ~8.5 shader clocks whatever the instruction is (scripts/r04_chain_probe.hip,
// profiles/r04_decoder_chain.txt), so a symbol costs its INSTRUCTION COUNT:
//   * lane l holds {cdf entry l, start and frequency of symbol l-1}: every lane advances the state for its own candidate
//     while one compare + ballot finds the lane that is right; two broadcasts fetch the new state;
//   * the only test on the common path is "high word of the new state == 0" (one scalar compare), behind it "< 2^31";
//   * a PLAIN renormalisation (the selected lane's image frequency is not 0) is three instructions behind that branch: the
//     next 64 stream words of a chunk sit in one register (`cur`), the k-th of them is one broadcast away and needs no range
//     test, because the 64 symbols of a chunk read at most one word each this way;
//   * the rare symbols (bypass sentinel, rows wider than 64 entries: image frequency 0) take the long way, and the bypass
//     path -- the only one that may read several words -- lines `cur` up again when it is done.
// Stream words are kept three 64-word blocks ahead (wa, wb, wc): `cur` is cut from the first two, the third is the one that
// may still be on its way from memory.
#pragma once
#include <cstdint>
#include <type_traits>

namespace wavedec {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint64_t kLow = 1ull << 31;

// compile-time loop over J = kBegin, kBegin + 2, ... < kEnd
template <int kBegin, int kEnd, class F> __device__ __forceinline__ void static_pairs(F &&f)
{
    if constexpr (kBegin < kEnd) {
        f(std::integral_constant<int, kBegin>{});
        static_pairs<kBegin + 2, kEnd>(f);
    }
}


// ---- the common path of one symbol at precision 16, spelled out (see WaveDecoder::decode_chunk) ----
// Registers are named literally so that halves of pairs and single entries of a 16-byte row can be addressed: the state
// lives in s[52:53], x >> 16 of the previous state in s[54:55], the ballot's lane in s57; four row buffers v[10:13] ..
// v[22:25] in rotation (symbol J reads buffer J & 3 and fetches symbol J + 3's row into the buffer symbol J - 1 is done
// with); v26-v28, s56 are scratch.  13 instructions (12 for odd J: one LDS wait covers two symbols) + the compiler's
// scalar compare and branch on the new state's high word:
//   x >> 16; candidate = freq * (x >> 16) + (x & 0xffff) - start on every lane (the low 16 bits come straight from the state
//   register: src0_sel:WORD_0); compare + ballot; the row address of symbol J + 3; two broadcasts; the row fetch; the
//   provisional result into lane J.
#ifndef WD_NO_SDWA
#define WD_STEP_HEAD(WAIT, KEY, START, FREQ)                                                                              \
    WAIT                                                                                                                  \
    "s_lshr_b64 s[54:55], s[52:53], 16\n\t"                                                                               \
    "v_sub_u32_sdwa v26, s52, " START " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"           \
    "v_mul_u32_u24 v27, s55, " FREQ "\n\t"                                                                                \
    "v_mad_u64_u32 v[26:27], vcc, " FREQ ", s54, v[26:27]\n\t"                                                            \
    "v_cmp_lt_u32_sdwa vcc, s52, " KEY " src0_sel:WORD_0 src1_sel:DWORD\n\t"
#else
#define WD_STEP_HEAD(WAIT, KEY, START, FREQ)                                                                              \
    WAIT                                                                                                                  \
    "s_and_b32 s56, s52, 0xffff\n\t"                                                                                      \
    "s_lshr_b64 s[54:55], s[52:53], 16\n\t"                                                                               \
    "v_sub_u32 v26, s56, " START "\n\t"                                                                                   \
    "v_mul_u32_u24 v27, s55, " FREQ "\n\t"                                                                                \
    "v_mad_u64_u32 v[26:27], vcc, " FREQ ", s54, v[26:27]\n\t"                                                            \
    "v_cmp_lt_u32 vcc, s56, " KEY "\n\t"
#endif
#define WD_STEP_FETCH(NEXT)                                                                                               \
    "v_readlane_b32 s56, %[meta], %[jf]\n\t"                                                                              \
    "s_ff1_i32_b64 s57, vcc\n\t"                                                                                          \
    "v_readlane_b32 s53, v27, s57\n\t"                                                                                    \
    "v_readlane_b32 s52, v26, s57\n\t"                                                                                    \
    "v_add_u32 v28, s56, %[lane16]\n\t"                                                                                   \
    "ds_read_b128 " NEXT ", v28\n\t"                                                                                      \
    "v_writelane_b32 %[res], s57, %[jw]"
#define WD_STEP_TAIL                                                                                                      \
    "s_ff1_i32_b64 s57, vcc\n\t"                                                                                          \
    "v_readlane_b32 s53, v27, s57\n\t"                                                                                    \
    "v_readlane_b32 s52, v26, s57\n\t"                                                                                    \
    "v_writelane_b32 %[res], s57, %[jw]"
#define WD_STEP_OPERANDS(J)                                                                                               \
    : "={s52}"(xlo), "={s53}"(xhi), "={s[54:55]}"(t), "={s57}"(first), [res] "+v"(res), "={v[10:13]}"(b0),                \
      "={v[14:17]}"(b1), "={v[18:21]}"(b2), "={v[22:25]}"(b3)                                                             \
    : "0"(xlo), "1"(xhi), "5"(b0), "6"(b1), "7"(b2), "8"(b3), [meta] "v"(meta_l), [lane16] "v"(lane16),                   \
      [jf] "n"((J) + 3 < 64 ? (J) + 3 : 0), [jw] "n"(J)                                                                   \
    : "s56", "vcc", "scc", "v26", "v27", "v28"

template <int J>
__device__ __forceinline__ void step_precision16(uint32_t &xlo, uint32_t &xhi, uint64_t &t, int32_t &first, int32_t &res, u32x4 &b0, u32x4 &b1,
                                                 u32x4 &b2, u32x4 &b3, uint32_t meta_l, uint32_t lane16)
{
    constexpr int R = J & 3;
    constexpr bool kFetch = J + 3 < 64;
    if constexpr (R == 0) {
        if constexpr (kFetch) asm volatile(WD_STEP_HEAD("s_waitcnt lgkmcnt(1)\n\t", "v10", "v11", "v12") WD_STEP_FETCH("v[22:25]") WD_STEP_OPERANDS(J));
        else                  asm volatile(WD_STEP_HEAD("s_waitcnt lgkmcnt(0)\n\t", "v10", "v11", "v12") WD_STEP_TAIL WD_STEP_OPERANDS(J));
    } else if constexpr (R == 1) {
        if constexpr (kFetch) asm volatile(WD_STEP_HEAD("", "v14", "v15", "v16") WD_STEP_FETCH("v[10:13]") WD_STEP_OPERANDS(J));
        else                  asm volatile(WD_STEP_HEAD("s_waitcnt lgkmcnt(0)\n\t", "v14", "v15", "v16") WD_STEP_TAIL WD_STEP_OPERANDS(J));
    } else if constexpr (R == 2) {
        if constexpr (kFetch) asm volatile(WD_STEP_HEAD("s_waitcnt lgkmcnt(1)\n\t", "v18", "v19", "v20") WD_STEP_FETCH("v[14:17]") WD_STEP_OPERANDS(J));
        else                  asm volatile(WD_STEP_HEAD("s_waitcnt lgkmcnt(0)\n\t", "v18", "v19", "v20") WD_STEP_TAIL WD_STEP_OPERANDS(J));
    } else {
        if constexpr (kFetch) asm volatile(WD_STEP_HEAD("", "v22", "v23", "v24") WD_STEP_FETCH("v[18:21]") WD_STEP_OPERANDS(J));
        else                  asm volatile(WD_STEP_HEAD("s_waitcnt lgkmcnt(0)\n\t", "v22", "v23", "v24") WD_STEP_TAIL WD_STEP_OPERANDS(J));
    }
}

// compile-time loop over J = kFrom, kFrom - 1, ..., 0
template <int kFrom, class F> __device__ __forceinline__ void static_down(F &&f)
{
    f(std::integral_constant<int, kFrom>{});
    if constexpr (kFrom > 0) static_down<kFrom - 1>(f);
}

// compile-time loop over J = kBegin .. kEnd - 1
template <int kBegin, int kEnd, class F> __device__ __forceinline__ void static_each(F &&f)
{
    if constexpr (kBegin < kEnd) {
        f(std::integral_constant<int, kBegin>{});
        static_each<kBegin + 1, kEnd>(f);
    }
}

struct WaveDecoder {
    const uint32_t *img;     // LDS
    const uint32_t *words;
    int limit;               // words in the stream
    int wbase;               // stream index of lane 0 of block wa
    uint32_t wa, wb, wc;     // words wbase + lane, + 64, + 128 (0 past the end of the stream)
    uint32_t cur;            // words pos0 + lane
    int pos0, k;             // the next unread word is pos0 + k
    uint64_t x;              // coder state (uniform)
    uint32_t prec, mask, bprec, maxbv;
    bool bypass;

    __device__ __forceinline__ static uint32_t bc32(uint32_t v, int l) { return __builtin_amdgcn_readlane(v, l); }
    __device__ __forceinline__ static uint64_t bc64(uint64_t v, int l)
    {
        return static_cast<uint64_t>(bc32(static_cast<uint32_t>(v), l)) | (static_cast<uint64_t>(bc32(static_cast<uint32_t>(v >> 32), l)) << 32);
    }
    __device__ __forceinline__ uint32_t load_block(int first, int lane) const
    {
        const int i = first + lane;
        return i < limit ? words[i] : 0u;
    }
    // `start` < 0: a fresh stream (its first two words are the state); otherwise resume at word `start` with state `x_saved`
    __device__ __forceinline__ void init(const uint32_t *image_lds, const uint32_t *w, int nwords, int precision, int bypass_precision, bool has_bypass,
                                         int start, uint64_t x_saved, int lane)
    {
        img = image_lds; words = w; limit = nwords;
        prec = static_cast<uint32_t>(precision); mask = (1u << prec) - 1u;
        bprec = static_cast<uint32_t>(bypass_precision); maxbv = (1u << bprec) - 1u; bypass = has_bypass;
        wbase = start < 0 ? 0 : start;
        wa = load_block(wbase, lane); wb = load_block(wbase + 64, lane); wc = load_block(wbase + 128, lane);
        pos0 = wbase; k = start < 0 ? 2 : 0;
        cur = wa;
        x = start < 0 ? (static_cast<uint64_t>(bc32(wa, 0)) | (static_cast<uint64_t>(bc32(wa, 1)) << 32)) : x_saved;
    }
    __device__ __forceinline__ int position() const { return pos0 + k; }
    // `cur` := the 64 words from the next unread one on (k := 0).  Before every chunk and after a bypass value.
    __device__ __forceinline__ void line_up(int lane)
    {
        int off = pos0 + k - wbase;
        while (off >= 64) {
            wa = wb; wb = wc; wbase += 64; off -= 64;
            wc = load_block(wbase + 128, lane);
        }
        if (off == 0) {
            cur = wa;
        } else {
            const int src = ((lane + off) & 63) << 2;
            const uint32_t a = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(src, static_cast<int>(wa)));
            const uint32_t b = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(src, static_cast<int>(wb)));
            cur = (lane + off < 64) ? a : b;
        }
        pos0 = wbase + off; k = 0;
    }
    // any word from the next unread one on, wherever it lies (bypass values only)
    __device__ __forceinline__ uint32_t next_word_anywhere()
    {
        const int p = pos0 + k, i = p - wbase;
        ++k;
        if (i < 64) return bc32(wa, i);
        if (i < 128) return bc32(wb, i - 64);
        if (i < 192) return bc32(wc, i - 128);
        return p < limit ? __builtin_amdgcn_readfirstlane(words[p]) : 0u;
    }
    __device__ __forceinline__ uint32_t get_bits(uint32_t nbits)   // Rans64DecGetBits, rans64.cpp:49-65
    {
        const uint32_t v = static_cast<uint32_t>(x) & ((1u << nbits) - 1u);
        x >>= nbits;
        if (x < kLow) x = (x << 32) | next_word_anywhere();
        return v;
    }
    // the selected lane's image frequency is 0: a row wider than 64 entries (two-level search) or the bypass sentinel.
    // `first` = the ballot's lane; returns the symbol + 1
    __device__ __forceinline__ int32_t rare_symbol(const u32x4 &e, int32_t first, uint32_t meta, int32_t size, uint32_t cf, uint64_t t, int lane)
    {
        const uint32_t base = meta >> 2;
        int32_t sym = first - 1;
        if (size > 64) {   // wide row: 64 block-end probes after the dummy lane, then the row
            const uint32_t pr = img[base + 4 + lane];
            const int blk = __builtin_ctzll(__ballot(pr > cf));
            const int32_t step = (size + 63) >> 6;
            const int32_t lo = blk * step;
            const int32_t span = (lo + step <= size) ? step : (size - lo);
            const uint32_t va = (lane < span) ? img[base + 68 + lo + lane] : 0x7FFFFFFFu;
            const int tl = __builtin_ctzll(__ballot(va > cf));
            const uint32_t c_t = bc32(va, tl);
            const uint32_t c_s = tl > 0 ? bc32(va, tl - 1) : bc32(pr, blk - 1);   // blk, tl == 0 together never happens: cdf[0] = 0 <= cf
            sym = lo + tl - 1;
            x = static_cast<uint64_t>(c_t - c_s) * t + (cf - c_s);
        } else {           // the sentinel: redo its update with the true frequency
            const uint32_t c_t = bc32(e[0], first), c_s = bc32(e[1], first);
            x = static_cast<uint64_t>(c_t - c_s) * t + (cf - c_s);
        }
        if (x < kLow) { x = (x << 32) | bc32(cur, k); ++k; }   // at most one word, as for a plain symbol
        if (bypass && sym == size - 2) {   // bypass value: count nibbles, then the payload low-first (rans64.cpp:466-487)
            uint32_t v = get_bits(bprec);
            uint32_t nb = v;
            while (v == maxbv) { v = get_bits(bprec); nb += v; }
            uint32_t raw = 0;
            for (uint32_t i = 0; i < nb; ++i) {
                const uint32_t nib = get_bits(bprec);
                if (i * bprec < 32u) raw |= nib << (i * bprec);
            }
            sym = static_cast<int32_t>(raw >> 1);
            if (raw & 1u) sym = -sym - 1; else sym += size - 2;
            line_up(lane);
        }
        return sym + 1;
    }
    // `cnt` symbols (1..64); lane j holds symbol j's image offset (meta, bytes) and row size.  Returns symbol j + 1 on lane j.
    // line_up() must have been called since the last chunk.
    __device__ __forceinline__ int32_t decode_chunk(uint32_t meta_l, int32_t size_l, int cnt, int lane)
    {
        int32_t result = 1;
        if (cnt == 64 && prec == 16u) {   // a full chunk at the usual precision: the spelled-out common path
            const uint32_t lane16 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(img)) + static_cast<uint32_t>(lane) * 16u;
            auto row = [&](int jj) { return *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(img) + bc32(meta_l, jj) + lane * 16); };
            u32x4 b0 = row(0), b1 = row(1), b2 = row(2), b3 = b2;
            uint32_t xlo = static_cast<uint32_t>(x), xhi = static_cast<uint32_t>(x >> 32);
            uint64_t t = 0;
            int32_t first = 0;
            static_each<0, 64>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                step_precision16<J>(xlo, xhi, t, first, result, b0, b1, b2, b3, meta_l, lane16);
                if (__builtin_expect(xhi == 0u, 0)) {
                    // rows in flight land before any compiler-generated code may touch their registers
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
                    if (xlo < static_cast<uint32_t>(kLow)) {
                        const u32x4 &e = (J & 3) == 0 ? b0 : (J & 3) == 1 ? b1 : (J & 3) == 2 ? b2 : b3;
                        if (__builtin_expect(bc32(e[2], first) != 0u, 1)) {
                            // x = (x << 32) | word k  (the high half is early-clobber: no input may share its register)
                            asm volatile("s_mov_b32 s53, s52\n\tv_readlane_b32 s52, %2, %3" : "={s52}"(xlo), "=&{s53}"(xhi) : "v"(cur), "s"(k), "0"(xlo));
                            ++k;
                        } else {
                            // image frequency 0: the lane's candidate is the coded value minus its start (nothing was multiplied)
                            const uint32_t cf = xlo + bc32(e[1], first);
                            first = rare_symbol(e, first, bc32(meta_l, J), static_cast<int32_t>(bc32(static_cast<uint32_t>(size_l), J)), cf, t, lane);
                            xlo = static_cast<uint32_t>(x); xhi = static_cast<uint32_t>(x >> 32);
                            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(result) : "s"(first), "n"(J));
                        }
                    }
                    asm volatile("" : "={s52}"(xlo), "={s53}"(xhi) : "0"(xlo), "1"(xhi));   // (the state stays in its registers on every path)
                }
            });
            x = static_cast<uint64_t>(xlo) | (static_cast<uint64_t>(xhi) << 32);
            return result;
        }
        auto fetch = [&](int jj, u32x4 &e) {
            const uint32_t m = bc32(meta_l, jj);
            e = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(img) + m + lane * 16);
        };
        auto decode_one = [&](auto jc, const u32x4 &e) {   // jc: int, or std::integral_constant (lane ids become immediates)
            const int j = jc;
            int32_t &res = result;
            const uint32_t cf = static_cast<uint32_t>(x) & mask;
            const uint64_t t = x >> prec;
            // x = freq * (x >> prec) + (cf - start)   (rans64.h:128-142), per lane for its own candidate
            const uint64_t addend = static_cast<uint64_t>(cf - e[1]) |
                                    (static_cast<uint64_t>(__umul24(e[2], static_cast<uint32_t>(t >> 32))) << 32);   // freq <= 2^16, t_hi < 2^15
            uint64_t cand;
            asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(cand) : "v"(e[2]), "s"(static_cast<uint32_t>(t)), "v"(addend) : "vcc");
            int32_t first = __builtin_ctzll(__ballot(e[0] > cf));   // symbol + 1
            x = bc64(cand, first);
            if (__builtin_expect(x < kLow, 0)) {
                if (__builtin_expect(bc32(e[2], first) != 0u, 1)) {
                    x = (x << 32) | bc32(cur, k);
                    ++k;
                } else {
                    first = rare_symbol(e, first, bc32(meta_l, j), static_cast<int32_t>(bc32(static_cast<uint32_t>(size_l), j)), cf, t, lane);
                }
            }
            if constexpr (std::is_integral<decltype(jc)>::value) {
                // a run-time lane select goes through m0, saved and restored inside the statement (a reserved register on a
                // clobber list is not honoured reliably)
                uint32_t m0_save;
                asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                             : "+v"(res), "=&s"(m0_save) : "s"(first), "s"(j));
            } else {
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(res) : "s"(first), "n"(decltype(jc)::value));
            }
        };
        u32x4 ea, eb;
        fetch(0, ea);
        fetch(cnt > 1 ? 1 : 0, eb);
        int j = 0;
        for (; j + 17 < cnt; j += 16) {   // sixteen symbols per loop trip, no clamping of the prefetch index in here
#pragma unroll
            for (int u = 0; u < 16; u += 2) {
                decode_one(j + u, ea);
                fetch(j + u + 2, ea);
                decode_one(j + u + 1, eb);
                fetch(j + u + 3, eb);
            }
        }
        for (; j + 1 < cnt; j += 2) {
            decode_one(j, ea);
            fetch(j + 2 < cnt ? j + 2 : cnt - 1, ea);
            decode_one(j + 1, eb);
            fetch(j + 3 < cnt ? j + 3 : cnt - 1, eb);
        }
        if (j < cnt) decode_one(j, ea);
        return result;
    }
};

}  // namespace wavedec
