// Micro-benchmark: issue cost of single VALU instruction forms on gfx950 at 8 waves per SIMD (marginal cost between
// 4 and 8 waves per SIMD removes the launch overhead).  build: hipcc -O3 --offload-arch=gfx950 -o valu_forms valu_forms.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    float a[8]; for (int j = 0; j < 8; ++j) a[j] = threadIdx.x + j;
    const float m = 1.0001f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile("s_mov_b64 vcc, 0x55555555\n s_mov_b64 s[20:21], 0x33333333\n v_mul_f32 %0, %0, %8\n v_mul_f32 %8, %8, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory", "vcc", "s20", "s21");
        if (MODE == 1) asm volatile("s_mov_b64 vcc, 0x55555555\n s_mov_b64 s[20:21], 0x33333333\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %8, %8, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory", "vcc", "s20", "s21");
        if (MODE == 2) asm volatile("s_mov_b64 vcc, 0x55555555\n s_mov_b64 s[20:21], 0x33333333\n v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %8, %8, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory", "vcc", "s20", "s21");
        if (MODE == 3) asm volatile("s_mov_b64 vcc, 0x55555555\n s_mov_b64 s[20:21], 0x33333333\n v_mov_b32 %0, s20\n v_mov_b32 %8, s20\n v_mov_b32 %2, s20\n v_mov_b32 %3, s20\n v_mov_b32 %4, s20\n v_mov_b32 %5, s20\n v_mov_b32 %6, s20\n v_mov_b32 %7, s20" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory", "vcc", "s20", "s21");
        if (MODE == 4) asm volatile("s_mov_b64 vcc, 0x55555555\n s_mov_b64 s[20:21], 0x33333333\n v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_f32 vcc, %8, %8\n v_cndmask_b32 %8, %8, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_cmp_lt_f32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %8, vcc\n v_cmp_lt_f32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory", "vcc", "s20", "s21");
        if (MODE == 5) asm volatile("s_mov_b64 vcc, 0x55555555\n s_mov_b64 s[20:21], 0x33333333\n v_max_f32 %0, %0, %8\n v_mul_f32 %0, %0, %8\n v_max_f32 %8, %8, %8\n v_mul_f32 %8, %8, %8\n v_max_f32 %2, %2, %8\n v_mul_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_mul_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_mul_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_mul_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_mul_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n v_mul_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory", "vcc", "s20", "s21");
        if (MODE == 6) asm volatile("s_mov_b64 vcc, 0x55555555\n s_mov_b64 s[20:21], 0x33333333\n v_fma_f32 %0, s20, %8, %0\n v_fma_f32 %8, s20, %8, %8\n v_fma_f32 %2, s20, %8, %2\n v_fma_f32 %3, s20, %8, %3\n v_fma_f32 %4, s20, %8, %4\n v_fma_f32 %5, s20, %8, %5\n v_fma_f32 %6, s20, %8, %6\n v_fma_f32 %7, s20, %8, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory", "vcc", "s20", "s21");
        if (MODE == 7) asm volatile("s_mov_b64 vcc, 0x55555555\n s_mov_b64 s[20:21], 0x33333333\n v_add_f32 %0, %0, %8\n v_add_f32 %8, %8, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory", "vcc", "s20", "s21");
        if (MODE == 8) asm volatile("s_mov_b64 vcc, 0x55555555\n s_mov_b64 s[20:21], 0x33333333\n v_mul_f32 %0, -%0, |%8|\n v_mul_f32 %8, -%8, |%8|\n v_mul_f32 %2, -%2, |%8|\n v_mul_f32 %3, -%3, |%8|\n v_mul_f32 %4, -%4, |%8|\n v_mul_f32 %5, -%5, |%8|\n v_mul_f32 %6, -%6, |%8|\n v_mul_f32 %7, -%7, |%8|" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory", "vcc", "s20", "s21");
        if (MODE == 9) asm volatile("s_mov_b64 vcc, 0x55555555\n s_mov_b64 s[20:21], 0x33333333\n v_and_b32 %0, %0, %8\n v_and_b32 %8, %8, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory", "vcc", "s20", "s21");
        if (MODE == 10) asm volatile("s_mov_b64 vcc, 0x55555555\n s_mov_b64 s[20:21], 0x33333333\n v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %8, %8, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory", "vcc", "s20", "s21");
        if (MODE == 11) asm volatile("s_mov_b64 vcc, 0x55555555\n s_mov_b64 s[20:21], 0x33333333\n v_mul_f32 %0, %0, %8\n v_sub_f32 %0, %8, %0\n v_mul_f32 %8, %8, %8\n v_sub_f32 %8, %8, %8\n v_mul_f32 %2, %2, %8\n v_sub_f32 %2, %8, %2\n v_mul_f32 %3, %3, %8\n v_sub_f32 %3, %8, %3\n v_mul_f32 %4, %4, %8\n v_sub_f32 %4, %8, %4\n v_mul_f32 %5, %5, %8\n v_sub_f32 %5, %8, %5\n v_mul_f32 %6, %6, %8\n v_sub_f32 %6, %8, %6\n v_mul_f32 %7, %7, %8\n v_sub_f32 %7, %8, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m) : "memory", "vcc", "s20", "s21");
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
    run<1>("cndmask_vcc", d);
    run<2>("cndmask_sgpr", d);
    run<3>("mov_from_sgpr", d);
    run<4>("cmp+cndmask pair", d);
    run<5>("max+mul pair", d);
    run<6>("fma_sgpr", d);
    run<7>("add_f32", d);
    run<8>("mul_e64_modifiers", d);
    run<9>("and_b32", d);
    run<10>("mul_lo_u32", d);
    run<11>("mul+sub dependent pair", d);
    return 0;
}
