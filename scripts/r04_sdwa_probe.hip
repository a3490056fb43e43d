// Does SDWA apply src0_sel to an SGPR operand on gfx950?  (round-4 decoder experiment)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void k(uint32_t *out, uint32_t sval)
{
    uint32_t s = __builtin_amdgcn_readfirstlane(sval);
    uint32_t v = threadIdx.x, d, m_lo, m_hi;
    asm volatile("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(d) : "s"(s), "v"(v));
    uint64_t vcc_out;
    asm volatile("v_cmp_lt_u32_sdwa vcc, %1, %2 src0_sel:WORD_0 src1_sel:DWORD\n\ts_mov_b64 %0, vcc" : "=s"(vcc_out) : "s"(s), "v"(v * 1000u) : "vcc");
    out[threadIdx.x] = d;
    if (threadIdx.x == 0) { out[64] = static_cast<uint32_t>(vcc_out); out[65] = static_cast<uint32_t>(vcc_out >> 32); }
}
int main()
{
    uint32_t *d; (void)hipMalloc(&d, 4 * 66);
    const uint32_t s = 0xABCD1234u;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s);
    uint32_t h[66]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("s = %08x: lane 5 sub -> %08x (WORD_0 applied: %08x, ignored: %08x)\n", s, h[5], (s & 0xffffu) - 5u, s - 5u);
    uint64_t want_sel = 0, want_full = 0;
    for (int l = 0; l < 64; ++l) { if ((s & 0xffffu) < l * 1000u) want_sel |= 1ull << l; if (s < l * 1000u) want_full |= 1ull << l; }
    printf("cmp mask %08x%08x (WORD_0 applied: %016llx, ignored: %016llx)\n", h[65], h[64], (unsigned long long)want_sel, (unsigned long long)want_full);
    return 0;
}
