// Round-4 measurement aid (not part of the library): dependent-issue latencies of the instructions on the rANS decoder's
// serial chain, for ONE wavefront alone on a compute unit (the in-loop decoder's situation).  Build and run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 scripts/r04_chain_probe.hip -o /tmp/chain_probe && /tmp/chain_probe
// Prints shader clocks per dependent link of each pattern.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kIters = 4096, kUnroll = 8;
constexpr uint64_t kLow = 1ull << 31;

#define LOOP(...)                                             \
    const long long t0 = clock64();                           \
    for (int it = 0; it < kIters; ++it) {                     \
        _Pragma("unroll") for (int u = 0; u < kUnroll; ++u) { __VA_ARGS__ }  \
    }                                                         \
    const long long t1 = clock64();

__device__ __forceinline__ uint32_t rl(uint32_t v, int l) { return __builtin_amdgcn_readlane(v, l); }

__global__ void k_valu(long long *out, uint32_t *sink)
{
    uint32_t v = threadIdx.x;
    LOOP(asm volatile("v_add_u32 %0, %0, 1" : "+v"(v));)
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[0] = v; }
}
__global__ void k_salu(long long *out, uint32_t *sink)
{
    uint32_t s = 1;
    LOOP(asm volatile("s_add_u32 %0, %0, 1" : "+s"(s)::"scc");)
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[0] = s; }
}
__global__ void k_readlane_valu(long long *out, uint32_t *sink)   // VALU -> SGPR (readlane) -> VALU
{
    uint32_t v = threadIdx.x;
    LOOP(uint32_t s = rl(v, 3); asm volatile("v_add_u32 %0, %1, %0" : "+v"(v) : "s"(s));)
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[0] = v; }
}
__global__ void k_readlane_salu_valu(long long *out, uint32_t *sink)   // readlane -> s_and -> v_sub
{
    uint32_t v = threadIdx.x;
    LOOP(uint32_t s = rl(v, 3); uint32_t s2; asm volatile("s_and_b32 %0, %1, 0xffff" : "=s"(s2) : "s"(s) : "scc"); asm volatile("v_sub_u32 %0, %1, %0" : "+v"(v) : "s"(s2));)
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[0] = v; }
}
__global__ void k_mad64(long long *out, uint32_t *sink)
{
    uint64_t c = threadIdx.x;
    uint32_t f = threadIdx.x + 3, s = 12345;
    LOOP(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(f), "s"(s) : "vcc");)
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[0] = static_cast<uint32_t>(c); }
}
__global__ void k_mullo(long long *out, uint32_t *sink)
{
    uint32_t v = threadIdx.x + 3;
    LOOP(asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(v));)
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[0] = v; }
}
__global__ void k_mad24(long long *out, uint32_t *sink)
{
    uint32_t v = threadIdx.x + 3;
    LOOP(asm volatile("v_mad_u32_u24 %0, %0, %0, %0" : "+v"(v));)
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[0] = v; }
}
__global__ void k_cmp_ff1_readlane(long long *out, uint32_t *sink)   // v_cmp -> s_ff1 -> v_readlane(lane select) -> (next cmp's scalar)
{
    uint32_t v = threadIdx.x * 64u + 1u;
    uint32_t s = 777;
    LOOP(const int first = __builtin_ctzll(__ballot(v > s) | (1ull << 63)); s = rl(v, first) + 100u; asm volatile("" : "+s"(s));)
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[0] = s; }
}
__global__ void k_branch(long long *out, uint32_t *sink)   // readlane -> 64-bit compare -> branch never taken -> v_add
{
    uint64_t v = (static_cast<uint64_t>(threadIdx.x + 1) << 33);
    uint32_t extra = 0;
    LOOP(uint64_t x = static_cast<uint64_t>(rl(static_cast<uint32_t>(v), 5)) | (static_cast<uint64_t>(rl(static_cast<uint32_t>(v >> 32), 5)) << 32);
         if (__builtin_expect(x < kLow, 0)) { extra += sink[x & 63]; x = (x << 32) | extra; }
         v += x & 1;  asm volatile("" : "+v"(v));)
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[0] = static_cast<uint32_t>(v) + extra; }
}
__global__ void k_nobranch(long long *out, uint32_t *sink)   // the same without the compare + branch
{
    uint64_t v = (static_cast<uint64_t>(threadIdx.x + 1) << 33);
    LOOP(uint64_t x = static_cast<uint64_t>(rl(static_cast<uint32_t>(v), 5)) | (static_cast<uint64_t>(rl(static_cast<uint32_t>(v >> 32), 5)) << 32);
         v += x & 1;  asm volatile("" : "+v"(v));)
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[0] = static_cast<uint32_t>(v); }
}
__global__ void k_lds(long long *out, uint32_t *sink)   // dependent ds_read_b128
{
    __shared__ u32x4 tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = u32x4{static_cast<uint32_t>((i * 17 + 64) & 1023), 0, 0, 0};
    __syncthreads();
    uint32_t a = threadIdx.x;
    LOOP(a = tab[a & 1023][0];)
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[0] = a; }
}
__global__ void k_memtime(long long *out, uint32_t *sink)
{
    long long acc = 0;
    LOOP(acc += clock64();)
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[0] = static_cast<uint32_t>(acc); }
}

// ---- the decoder's per-symbol step on a synthetic row: every lane l holds {cdf end, start, freq} of symbol l of a 64-symbol
// uniform row (precision 16: freq 1024); words come from a register (never exhausted: the position wraps) ----
struct Row { uint32_t end, start, freq; };
__device__ __forceinline__ Row make_row(int lane) { return Row{static_cast<uint32_t>(lane + 1) * 1024u, static_cast<uint32_t>(lane) * 1024u, 1024u}; }

// as csrc/scanline.hip WaveDecoder::decode_one today (branch on renormalisation)
template <bool with_lds> __global__ void k_chain_now(long long *out, uint32_t *sink)
{
    __shared__ u32x4 img[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) { const Row r = make_row(i & 63); img[i] = u32x4{r.end, r.start, r.freq, 0}; }
    __syncthreads();
    uint32_t cache = lane * 2654435761u + 12345u;
    uint64_t x = (1ull << 40) + 98765;
    uint32_t pos = 0, res = 0;
    const uint32_t mask = 0xffffu, prec = 16;
    u32x4 ea = img[lane], eb = img[64 + lane];
    uint32_t rowsel = 0;
    LOOP(
        u32x4 &e = (u & 1) ? eb : ea;
        const uint32_t cf = static_cast<uint32_t>(x) & mask;
        const uint64_t t = x >> prec;
        const uint64_t addend = static_cast<uint64_t>(cf - e[1]) | (static_cast<uint64_t>(__umul24(e[2], static_cast<uint32_t>(t >> 32))) << 32);
        uint64_t cand;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(cand) : "v"(e[2]), "s"(static_cast<uint32_t>(t)), "v"(addend) : "vcc");
        const int first = __builtin_ctzll(__ballot(e[0] > cf));
        x = static_cast<uint64_t>(rl(static_cast<uint32_t>(cand), first)) | (static_cast<uint64_t>(rl(static_cast<uint32_t>(cand >> 32), first)) << 32);
        if (__builtin_expect(x < kLow, 0)) { x = (x << 32) | rl(cache, pos & 63); ++pos; }
        asm volatile("v_writelane_b32 %0, %1, 7" : "+v"(res) : "s"(first));
        if constexpr (with_lds) { rowsel = (rowsel + 64) & 192; e = img[rowsel + lane]; }
    )
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[0] = static_cast<uint32_t>(x) + res + pos; }
}

// variant: the renormalisation folded into the lanes' candidates (no branch on the chain): every lane selects between its
// candidate and the renormalised one with the NEXT word, which is known before the symbol starts
template <bool with_lds> __global__ void k_chain_select(long long *out, uint32_t *sink)
{
    __shared__ u32x4 img[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) { const Row r = make_row(i & 63); img[i] = u32x4{r.end, r.start, r.freq, 0}; }
    __syncthreads();
    uint32_t cache = lane * 2654435761u + 12345u;
    uint64_t x = (1ull << 40) + 98765;
    uint32_t pos = 0, res = 0;
    const uint32_t mask = 0xffffu, prec = 16;
    u32x4 ea = img[lane], eb = img[64 + lane];
    uint32_t rowsel = 0;
    uint32_t w = rl(cache, 0);
    LOOP(
        u32x4 &e = (u & 1) ? eb : ea;
        const uint32_t cf = static_cast<uint32_t>(x) & mask;
        const uint64_t t = x >> prec;
        const uint64_t addend = static_cast<uint64_t>(cf - e[1]) | (static_cast<uint64_t>(__umul24(e[2], static_cast<uint32_t>(t >> 32))) << 32);
        uint64_t cand;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(cand) : "v"(e[2]), "s"(static_cast<uint32_t>(t)), "v"(addend) : "vcc");
        const uint64_t below = __ballot(cand < kLow);
        const uint32_t lo = static_cast<uint32_t>(cand), hi = static_cast<uint32_t>(cand >> 32);
        const bool r = cand < kLow;
        const uint32_t nlo = r ? w : lo, nhi = r ? lo : hi;
        const int first = __builtin_ctzll(__ballot(e[0] > cf));
        x = static_cast<uint64_t>(rl(nlo, first)) | (static_cast<uint64_t>(rl(nhi, first)) << 32);
        const uint32_t took = static_cast<uint32_t>(below >> first) & 1u;
        pos += took;
        w = rl(cache, pos & 63);
        asm volatile("v_writelane_b32 %0, %1, 7" : "+v"(res) : "s"(first));
        if constexpr (with_lds) { rowsel = (rowsel + 64) & 192; e = img[rowsel + lane]; }
    )
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[0] = static_cast<uint32_t>(x) + res + pos; }
}

// variant: `+ cf` moved behind the readlane (scalar add with carry), the lanes' addend (-start, freq * t_hi) does not wait for cf
template <bool with_lds> __global__ void k_chain_late_cf(long long *out, uint32_t *sink)
{
    __shared__ u32x4 img[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) { const Row r = make_row(i & 63); img[i] = u32x4{r.end, r.start, r.freq, 0}; }
    __syncthreads();
    uint32_t cache = lane * 2654435761u + 12345u;
    uint64_t x = (1ull << 40) + 98765;
    uint32_t pos = 0, res = 0;
    const uint32_t mask = 0xffffu, prec = 16;
    u32x4 ea = img[lane], eb = img[64 + lane];
    uint32_t rowsel = 0;
    LOOP(
        u32x4 &e = (u & 1) ? eb : ea;
        const uint32_t cf = static_cast<uint32_t>(x) & mask;
        const uint64_t t = x >> prec;
        // freq * t - start: t >= 2^15 and freq >= 1 so the product is >= start (start < 2^16 <= ... holds for t >= 2^16; the probe ignores the corner)
        const uint64_t addend = (static_cast<uint64_t>(__umul24(e[2], static_cast<uint32_t>(t >> 32))) << 32) - e[1];
        uint64_t cand;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(cand) : "v"(e[2]), "s"(static_cast<uint32_t>(t)), "v"(addend) : "vcc");
        const int first = __builtin_ctzll(__ballot(e[0] > cf));
        x = (static_cast<uint64_t>(rl(static_cast<uint32_t>(cand), first)) | (static_cast<uint64_t>(rl(static_cast<uint32_t>(cand >> 32), first)) << 32)) + cf;
        if (__builtin_expect(x < kLow, 0)) { x = (x << 32) | rl(cache, pos & 63); ++pos; }
        asm volatile("v_writelane_b32 %0, %1, 7" : "+v"(res) : "s"(first));
        if constexpr (with_lds) { rowsel = (rowsel + 64) & 192; e = img[rowsel + lane]; }
    )
    if (threadIdx.x == 0) { out[0] = t1 - t0; sink[0] = static_cast<uint32_t>(x) + res + pos; }
}

template <class K, class... A> static void run(const char *name, int links, K k, A... args)
{
    long long *d_out; uint32_t *d_sink;
    (void)hipMalloc(&d_out, 8); (void)hipMalloc(&d_sink, 4096);
    (void)hipMemset(d_sink, 0, 4096);
    long long best = 1ll << 62;
    for (int r = 0; r < 5; ++r) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_sink, args...);
        long long h = 0;
        (void)hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
        if (h < best) best = h;
    }
    const double per = static_cast<double>(best) / (static_cast<double>(kIters) * kUnroll);
    printf("%-44s %8.1f clocks per iteration", name, per);
    if (links > 1) printf("  (%d dependent links: %.1f each)", links, per / links);
    printf("\n");
    fflush(stdout);
    (void)hipFree(d_out); (void)hipFree(d_sink);
}

int main()
{
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    printf("device %s, %d CUs, shader clock %.0f MHz; s_memtime counts at 100 MHz on gfx950? -> see k_valu (4-cycle op)\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate / 1e3);
    run("v_add dependent", 1, k_valu);
    run("s_add dependent", 1, k_salu);
    run("v_readlane -> v_add", 2, k_readlane_valu);
    run("v_readlane -> s_and -> v_sub", 3, k_readlane_salu_valu);
    run("v_mad_u64_u32 dependent", 1, k_mad64);
    run("v_mul_lo_u32 dependent", 1, k_mullo);
    run("v_mad_u32_u24 dependent", 1, k_mad24);
    run("v_cmp -> s_ff1 -> v_readlane(sel) -> s_add", 4, k_cmp_ff1_readlane);
    run("readlane x2 -> cmp64 -> branch -> v_add", 4, k_branch);
    run("readlane x2 -> v_add (no branch)", 2, k_nobranch);
    run("ds_read_b128 dependent", 1, k_lds);
    run("s_memtime", 1, k_memtime);
    run("decode step, as today, row in registers", 1, k_chain_now<false>);
    run("decode step, as today, row from LDS", 1, k_chain_now<true>);
    run("decode step, renorm by select, registers", 1, k_chain_select<false>);
    run("decode step, renorm by select, LDS", 1, k_chain_select<true>);
    run("decode step, cf added after readlane, regs", 1, k_chain_late_cf<false>);
    run("decode step, cf added after readlane, LDS", 1, k_chain_late_cf<true>);
    return 0;
}
