// Micro-benchmark: issue cost of single VALU instruction forms on gfx950 at 8 waves per SIMD (marginal cost between
// 4 and 8 waves per SIMD removes the launch overhead).  build: hipcc -O3 --offload-arch=gfx950 -o valu_forms valu_forms.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    float a[8]; for (int j = 0; j < 8; ++j) a[j] = threadIdx.x + j;
    const float m = 1.0001f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %8, %8, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory");
        if (MODE == 1) asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %8, %8, %8\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %8, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory");
        if (MODE == 2) asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %8, %8, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory");
        if (MODE == 3) asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %8, %8, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory");
        if (MODE == 4) asm volatile("v_min3_f32 %0, %0, %8, %8\n v_min3_f32 %8, %8, %8, %8\n v_min3_f32 %2, %2, %8, %8\n v_min3_f32 %3, %3, %8, %8\n v_min3_f32 %4, %4, %8, %8\n v_min3_f32 %5, %5, %8, %8\n v_min3_f32 %6, %6, %8, %8\n v_min3_f32 %7, %7, %8, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory");
        if (MODE == 5) asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %8, %8, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory");
        if (MODE == 6) asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %8, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory", "vcc");
        if (MODE == 7) asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[20:21], %8, %8\n v_cmp_lt_f32 s[20:21], %2, %8\n v_cmp_lt_f32 s[20:21], %3, %8\n v_cmp_lt_f32 s[20:21], %4, %8\n v_cmp_lt_f32 s[20:21], %5, %8\n v_cmp_lt_f32 s[20:21], %6, %8\n v_cmp_lt_f32 s[20:21], %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory", "s20", "s21");
        if (MODE == 8) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %8, %8, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory", "vcc");
        if (MODE == 9) asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %8, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory");
        if (MODE == 10) asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %8, %8, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory");
        if (MODE == 11) asm volatile("v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %8, %8, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory");
        if (MODE == 12) asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %8, %8\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory");
        if (MODE == 13) asm volatile("v_mul_f32 %0, s20, %0\n v_mul_f32 %8, s20, %8\n v_mul_f32 %2, s20, %2\n v_mul_f32 %3, s20, %3\n v_mul_f32 %4, s20, %4\n v_mul_f32 %5, s20, %5\n v_mul_f32 %6, s20, %6\n v_mul_f32 %7, s20, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory");
        if (MODE == 14) asm volatile("v_sub_f32 %0, s20, %0\n v_sub_f32 %8, s20, %8\n v_sub_f32 %2, s20, %2\n v_sub_f32 %3, s20, %3\n v_sub_f32 %4, s20, %4\n v_sub_f32 %5, s20, %5\n v_sub_f32 %6, s20, %6\n v_sub_f32 %7, s20, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory");
        if (MODE == 15) asm volatile("v_mul_f32 %0, 0x3f800008, %0\n v_mul_f32 %8, 0x3f800008, %8\n v_mul_f32 %2, 0x3f800008, %2\n v_mul_f32 %3, 0x3f800008, %3\n v_mul_f32 %4, 0x3f800008, %4\n v_mul_f32 %5, 0x3f800008, %5\n v_mul_f32 %6, 0x3f800008, %6\n v_mul_f32 %7, 0x3f800008, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory");
        if (MODE == 16) asm volatile("v_med3_f32 %0, %0, %8, %8\n v_med3_f32 %8, %8, %8, %8\n v_med3_f32 %2, %2, %8, %8\n v_med3_f32 %3, %3, %8, %8\n v_med3_f32 %4, %4, %8, %8\n v_med3_f32 %5, %5, %8, %8\n v_med3_f32 %6, %6, %8, %8\n v_med3_f32 %7, %7, %8, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory");
        if (MODE == 17) asm volatile("v_max3_f32 %0, %0, %8, %0\n v_max3_f32 %8, %8, %8, %8\n v_max3_f32 %2, %2, %8, %2\n v_max3_f32 %3, %3, %8, %3\n v_max3_f32 %4, %4, %8, %4\n v_max3_f32 %5, %5, %8, %5\n v_max3_f32 %6, %6, %8, %6\n v_max3_f32 %7, %7, %8, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory");
    }
    float s = 0; for (int j = 0; j < 8; ++j) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static float once(float* d, int bpc, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256 * bpc, 256>>>(d, 100);
    hipEventRecord(e0);
    k<MODE><<<256 * bpc, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
template <int MODE>
static void run(const char* name, float* d) {
    const int iters = 20000;
    const float m4 = once<MODE>(d, 4, iters), m8 = once<MODE>(d, 8, iters);
    const double instr = 4.0 * iters * 8;    // extra wave-instructions per SIMD between the two runs
    printf("%-14s 4w %.3f ms  8w %.3f ms  marginal %.2f cyc/instr at 2.4 GHz\n", name, m4, m8, (m8 - m4) * 1e-3 * 2.4e9 / instr);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run<0>("mul", d);
    run<1>("sub", d);
    run<2>("min", d);
    run<3>("max", d);
    run<4>("min3", d);
    run<5>("fma", d);
    run<6>("cmp_vcc", d);
    run<7>("cmp_sgpr_e64", d);
    run<8>("cndmask", d);
    run<9>("mov", d);
    run<10>("add_u32", d);
    run<11>("lshl_add", d);
    run<12>("rcp", d);
    run<13>("mul_sgpr", d);
    run<14>("sub_sgpr", d);
    run<15>("mul_literal", d);
    run<16>("med3", d);
    run<17>("max3", d);
    return 0;
}
