// Experiment: time per dependent kernel inside a hipGraph for (a) an empty kernel, (b) a kernel with a large
// by-value argument struct, (c) one that also reads its arguments and does one global load + store per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { float* p[8]; int v[120]; };
__global__ void k_empty() {}
__global__ void k_big(Big b) { if (b.v[119] == 12345 && threadIdx.x == 0) b.p[0][0] = 1.f; }
__global__ void k_touch(Big b) { float* p = b.p[blockIdx.x & 7]; const int i = blockIdx.x * blockDim.x + threadIdx.x; p[i] = p[i] + (float)b.v[threadIdx.x & 63]; }
template <typename F> float run(const char* name, int nk, F launch) {
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < nk; ++i) launch(s);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s: %.2f us per kernel\n", name, ms * 1e3f / (20.f * nk));
    return ms;
}
int main() {
    Big b{}; for (int i = 0; i < 8; ++i) hipMalloc(&b.p[i], 1 << 22);
    for (int wgs : {1, 64, 588, 2048}) {
        char nm[96];
        snprintf(nm, 96, "empty kernel, %4d WGs x 256", wgs); run(nm, 50, [&](hipStream_t s) { k_empty<<<wgs, 256, 0, s>>>(); });
        snprintf(nm, 96, "544-byte by-value args, %4d WGs x 256", wgs); run(nm, 50, [&](hipStream_t s) { k_big<<<wgs, 256, 0, s>>>(b); });
        snprintf(nm, 96, "args + 1 load/store per thread, %4d WGs", wgs); run(nm, 50, [&](hipStream_t s) { k_touch<<<wgs, 256, 0, s>>>(b); });
    }
    return 0;
}
