// How fast can ONE workgroup pull a 512 KB weight panel (L2 / MALL resident) through its CU?  (The fat persistent-LSTM
// idea: 4 workgroups, each streaming Wh = 256 x 1024 bf16 once per time step.)
//   hipcc --offload-arch=gfx950 -O3 -o cu_stream cu_stream.hip && ./cu_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(1024) void stream_kernel(const uint4* __restrict__ w, uint4* out, long long* cyc, int n16, int passes) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    long long t0 = wall_clock64();
    for (int p = 0; p < passes; ++p) {
        // 8 loads in flight per lane, lane-linear
        for (int i = threadIdx.x; i < n16; i += blockDim.x * 8) {
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int j = i + k * blockDim.x; v[k] = j < n16 ? w[j] : make_uint4(0, 0, 0, 0); }
#pragma unroll
            for (int k = 0; k < 8; ++k) { acc.x ^= v[k].x; acc.y += v[k].y; acc.z ^= v[k].z; acc.w += v[k].w; }
        }
        __syncthreads();
    }
    long long t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    const int bytes = 512 * 1024, n16 = bytes / 16;
    uint4* w; uint4* out; long long* cyc;
    hipMalloc(&w, bytes); hipMalloc(&out, 64 * 1024 * 16); hipMalloc(&cyc, 64 * 8);
    hipMemset(w, 1, bytes);
    for (int threads : {256, 512, 1024})
        for (int wgs : {1, 4, 16}) {
            const int passes = 20;
            hipLaunchKernelGGL(stream_kernel, dim3(wgs), dim3(threads), 0, 0, w, out, cyc, n16, passes);
            hipLaunchKernelGGL(stream_kernel, dim3(wgs), dim3(threads), 0, 0, w, out, cyc, n16, passes);
            std::vector<long long> c(wgs);
            hipMemcpy(c.data(), cyc, wgs * 8, hipMemcpyDeviceToHost);
            long long mx = 0; for (auto v : c) mx = v > mx ? v : mx;
            printf("%4d threads x %2d workgroups: %.2f us per 512 KB pass (%.1f GB/s per workgroup)\n", threads, wgs,
                   mx / 100.0 / passes, bytes / (mx / 100.0 / passes) * 1e-3);
        }
    return 0;
}
