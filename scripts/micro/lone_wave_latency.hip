// Per-instruction cost of dependent chains in a LONE wavefront (one wave per CU: the rANS coders' regime).
// Each kernel repeats a short dependent pattern; cycles per pattern = s_memtime delta / iterations.
// Build: hipcc -O3 --offload-arch=gfx950 lone_wave_latency.hip -o lone_wave_latency
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X

template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned long long *out, int iters, unsigned seed)
{
    unsigned s = seed, v = seed + threadIdx.x, lanev = threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // dependent SALU adds
            REP8(asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) :: "scc");)
        } else if (MODE == 1) {  // dependent SALU multiplies
            REP8(asm volatile("s_mul_i32 %0, %0, 3" : "+s"(s) :: "scc");)
        } else if (MODE == 2) {  // dependent VALU adds
            REP8(asm volatile("v_add_u32 %0, %0, 1" : "+v"(v) :: "vcc");)
        } else if (MODE == 3) {  // SALU -> v_readlane (lane select) -> SALU
            REP8(asm volatile("s_and_b32 %0, %0, 63\n\tv_readlane_b32 %0, %1, %0" : "+s"(s) : "v"(lanev) : "scc");)
        } else if (MODE == 4) {  // SALU -> v_cmp -> s_ff1 -> v_readlane -> SALU   (the decoder's search)
            REP8(asm volatile("s_and_b32 %0, %0, 63\n\tv_cmp_lt_u32 vcc, %0, %1\n\ts_ff1_i32_b64 %0, vcc\n\tv_readlane_b32 %0, %1, %0"
                              : "+s"(s) : "v"(lanev) : "vcc", "scc");)
        } else if (MODE == 5) {  // v_mad_u64_u32 dependent chain
            unsigned long long a = v;
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a) : "v"(lanev) : "vcc");)
            v = (unsigned)a;
        } else if (MODE == 6) {  // s_mul_hi dependent
            REP8(asm volatile("s_mul_hi_u32 %0, %0, 0x7fffffff" : "+s"(s) :: "scc");)
        } else if (MODE == 7) {  // taken branch per pattern
            REP8(asm volatile("s_add_u32 %0, %0, 1\n\ts_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" : "+s"(s) :: "scc");)
        } else if (MODE == 8) {  // not-taken branch per pattern
            REP8(asm volatile("s_add_u32 %0, %0, 1\n\ts_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" : "+s"(s) :: "scc");)
        } else if (MODE == 9) {  // independent SALU adds (two chains)
            unsigned s2 = s + 7;
            REP8(asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1" : "+s"(s), "+s"(s2) :: "scc");)
            s += s2;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = s + v; }
}

template <int MODE>
void run(const char *what, int per)
{
    unsigned long long *out, h[2];
    (void)hipMalloc(&out, 16);
    const int iters = 200000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<1, 64>>>(out, iters, 5);
    (void)hipEventRecord(e0);
    k<MODE><<<1, 64>>>(out, iters, 5);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("%-50s %6.1f counter ticks, %6.2f ns per pattern (%d instr)\n", what, double(h[0]) / (iters * 8.0),
           ms * 1e6 / (iters * 8.0), per);
    (void)hipFree(out);
}

int main()
{
    run<0>("dependent s_add_u32", 1);
    run<9>("two independent s_add_u32 chains", 2);
    run<1>("dependent s_mul_i32", 1);
    run<6>("dependent s_mul_hi_u32", 1);
    run<2>("dependent v_add_u32", 1);
    run<5>("dependent v_mad_u64_u32", 1);
    run<3>("s_and -> v_readlane(lane select) -> ...", 2);
    run<4>("s_and -> v_cmp -> s_ff1 -> v_readlane -> ...", 4);
    run<7>("s_add, s_cmp, TAKEN s_cbranch", 3);
    run<8>("s_add, s_cmp, not-taken s_cbranch, s_nop", 4);
    return 0;
}
