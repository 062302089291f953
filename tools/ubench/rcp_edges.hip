// What v_rcp_f32 returns at the edges of the float range (build: hipcc --offload-arch=gfx950 -O2 -o rcp_edges rcp_edges.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(const float* in, float* out, int n) {
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_rcpf(in[i]);
}
int main() {
    const float in[] = {1e-39f, 1e-44f, 1.17549435e-38f, 8.5e37f, 1.0e38f, 1.7e38f, 3.0e38f, 3.4e38f, -2.0e38f, 4.0f};
    const int n = sizeof(in) / 4;
    float *di, *dout, out[16];
    hipMalloc(&di, 64); hipMalloc(&dout, 64);
    hipMemcpy(di, in, n * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, dout, n);
    hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) { unsigned b; memcpy(&b, &out[i], 4); printf("rcp(%g) = %g (0x%08x)   exact %g\n", in[i], out[i], b, 1.0 / in[i]); }
    return 0;
}
