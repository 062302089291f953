// Calibration of the TCC FETCH_SIZE counter on gfx950 against known read volumes: each kernel streams the same 256 MiB
// buffer once from HBM (the buffer is larger than the caches and a different 1 GiB buffer is written between kernels),
// with 16-byte, 4-byte and 1-byte loads per lane, all coalesced.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o fc -- ./fetch_calib
// FETCH_SIZE is in KiB; compare 1024 * FETCH_SIZE per kernel with 268435456 bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void read16(const float4* p, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; acc += v.x + v.w; }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void read4(const float* p, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void read1(const uint8_t* p, size_t n, float* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 0xFFFFFFFFu) sink[0] = (float)acc;
}
__global__ __launch_bounds__(256) void flush(float4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_float4(1, 2, 3, 4);
}

int main() {
    const size_t bytes = 256ull << 20, fbytes = 1ull << 30;
    void* buf; void* fl; float* sink;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&fl, fbytes)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 1, bytes));
    const int grid = 256 * 16;
    for (int rep = 0; rep < 3; ++rep) {
        flush<<<grid, 256>>>((float4*)fl, fbytes / 16);
        read16<<<grid, 256>>>((const float4*)buf, bytes / 16, sink);
        flush<<<grid, 256>>>((float4*)fl, fbytes / 16);
        read4<<<grid, 256>>>((const float*)buf, bytes / 4, sink);
        flush<<<grid, 256>>>((float4*)fl, fbytes / 16);
        read1<<<grid, 256>>>((const uint8_t*)buf, bytes, sink);
    }
    CHECK(hipDeviceSynchronize());
    printf("each read kernel streams %zu bytes\n", bytes);
    return 0;
}
