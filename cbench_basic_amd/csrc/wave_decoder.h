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
        if (cnt == 64) {   // a full chunk, fully unrolled: every lane id is an immediate
            static_pairs<0, 64>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                decode_one(std::integral_constant<int, J>{}, ea);
                if constexpr (J + 2 < 64) fetch(J + 2, ea);
                decode_one(std::integral_constant<int, J + 1>{}, eb);
                if constexpr (J + 3 < 64) fetch(J + 3, eb);
            });
            return result;
        }
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
