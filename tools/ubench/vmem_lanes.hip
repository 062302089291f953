// What a 16-byte-per-lane vector load costs the CU's vector-memory path as a function of how many lanes of the wave
// take part, for data that sits in the L1 cache: the traversal kernels' node fetch is four such loads per lane and step.
//   hipcc --offload-arch=gfx950 -O3 -o vmem_lanes vmem_lanes.hip && ./vmem_lanes
// Every wave issues LOADS loads (8 independent address chains per lane, each to a pseudo-random 64-byte-aligned record of
// a 256 KiB table, so nearly all hit L1/L2 and lanes never share a line) with lanes >= K switched off.
// Printed: cycles per wave-instruction and CU at 8 waves per SIMD (all CUs busy), and the same with all lanes reading ONE
// record (what neighbouring rays at the same node do).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int LOADS = 4096, CHAINS = 8, TABLE = 4096;      // 4096 records of 64 B

__global__ __launch_bounds__(256) void k(const float4* table, int K, int same, float* sink) {
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    if (lane < K) {
        uint32_t idx[CHAINS];
        for (int c = 0; c < CHAINS; ++c) idx[c] = (uint32_t)((threadIdx.x + blockIdx.x * 256) * 2654435761u + c * 40503u);
        if (same == 1) for (int c = 0; c < CHAINS; ++c) idx[c] = (uint32_t)(blockIdx.x * 97u + (threadIdx.x >> 6) * 13u + c * 40503u);
        // modes 2 and 3: the four lanes of a quad read one record (3: each lane its own 16-byte quarter of it)
        if (same >= 2) for (int c = 0; c < CHAINS; ++c) idx[c] = (uint32_t)(((threadIdx.x + blockIdx.x * 256) >> 2) * 2654435761u + c * 40503u);
        for (int i = 0; i < LOADS / CHAINS; ++i) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                const float4 v = table[(size_t)(idx[c] & (TABLE - 1)) * 4 + (same == 3 ? (lane & 3) : (i & 3))];
                acc += v.x;
                idx[c] = idx[c] * 1664525u + 1013904223u + (uint32_t)(v.y != 12345.f ? 0 : 1);   // next address needs the data
            }
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    float4* table; float* sink;
    CHECK(hipMalloc(&table, TABLE * 64)); CHECK(hipMemset(table, 0, TABLE * 64)); CHECK(hipMalloc(&sink, 4));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, blocks = cus * 8;      // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int same = 0; same < 4; ++same)
        for (int K : {64, 48, 32, 16, 8, 4, 1}) {
            k<<<blocks, 256>>>(table, K, same, sink);
            CHECK(hipEventRecord(a));
            k<<<blocks, 256>>>(table, K, same, sink);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            const double insts_per_cu = 32.0 * LOADS;             // 32 waves per CU
            printf("%s lanes %2d: %.3f ms, %.1f cycles per wave-load and CU (2.4 GHz), %.1f B/clk/CU returned\n", same == 0 ? "one record per lane " : same == 1 ? "one record per wave " : same == 2 ? "one record per quad " : "one quarter per lane",
                   K, ms, ms * 1e-3 * 2.4e9 / insts_per_cu, insts_per_cu * K * 16 / (ms * 1e-3 * 2.4e9));
        }
    return 0;
}
