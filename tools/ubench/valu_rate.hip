// Micro-benchmark: issue rate of scalar-form and packed f32 VALU instructions with 1..8 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float m = 1.0001f; const f2 pm = {m, m};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (MODE == 1) {
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
        } else if (MODE == 2) {
            asm volatile("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                         "v_min3_f32 %4, %4, %8, %0\n v_max3_f32 %5, %5, %8, %1\n v_min3_f32 %6, %6, %8, %2\n v_max3_f32 %7, %7, %8, %3\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else {
            asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         "v_pk_fma_f32 %0, %0, %4, %1\n v_pk_fma_f32 %1, %1, %4, %2\n v_pk_fma_f32 %2, %2, %4, %3\n v_pk_fma_f32 %3, %3, %4, %0\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
static void run(const char* name, float* d, int blocks_per_cu) {
    const int iters = 20000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 100);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winst = double(blocks) * 4 * iters * 8;          // wave-instructions
    const double per_simd_clk = winst / 1024.0 / (ms * 1e-3 * 2.4e9);
    printf("%-28s waves/SIMD %d  %.3f ms  wave-instr per SIMD per clk(2.4GHz) %.3f  => %.2f cyc/instr\n", name, blocks_per_cu, ms,
           per_simd_clk, 1.0 / per_simd_clk);
}

int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int b : {1, 2, 4, 8}) {
        run<0>("v_mul_f32", d, b);
        run<1>("v_pk_mul_f32", d, b);
        run<2>("v_min/max/min3/max3_f32", d, b);
        run<3>("v_pk_add/mul/fma_f32", d, b);
    }
    return 0;
}
