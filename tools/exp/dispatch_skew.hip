// How far apart do the workgroups of ONE launch start?  Every workgroup stamps wall_clock64 (100 MHz) on entry; skew =
// last start - first start, for grids of G workgroups x T threads with L bytes of dynamic LDS and V VGPRs' worth of
// register footprint.  (The sampler kernels launch 192 workgroups and show 1.7 - 2.2 us of start skew.)
//   hipcc --offload-arch=gfx950 -O3 -o dispatch_skew dispatch_skew.hip && ./dispatch_skew
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int T>
__global__ __launch_bounds__(T) void k(unsigned long long* st, int spin) {
    extern __shared__ float lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) st[blockIdx.x] = t0;
    // a few us of work so that the workgroups overlap in time
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    if (a == 123.f) lds[threadIdx.x] = a;
}
template <int T> void run(int G, int L, unsigned long long* d) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<unsigned long long> h(G);
    double sk[5];
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(k<T>, dim3(G), dim3(T), L, 0, d, 2000);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
        sk[rep] = (*std::max_element(h.begin(), h.end()) - *std::min_element(h.begin(), h.end())) / 100.0;
    }
    std::sort(sk, sk + 5);
    printf("G %5d  T %4d  LDS %6d B: start skew median %.2f us (min %.2f)\n", G, T, L, sk[2], sk[0]);
}
int main() {
    unsigned long long* d; hipMalloc(&d, 8192 * 8);
    for (int G : {48, 64, 96, 192, 256, 384, 512, 1024}) {
        run<256>(G, 0, d); run<256>(G, 40 * 1024, d); run<1024>(G, 0, d); run<1024>(G, 56 * 1024, d);
    }
    return 0;
}
