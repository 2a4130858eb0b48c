// Experiment: what a consumer kernel pays for its first batch of loads inside a hipGraph chain of
// dependent kernels.  Each workgroup (256 threads) issues NL independent 16-byte loads per thread
// (rows of `ld` floats, lanes along the row: the GEMM operand pattern), then waits for all of them.
// wall_clock64 stamps of workgroup 0: start -> all loads issued -> all data arrived -> stored.
// Variants: the buffer was (a) last read by the same kernel shape (L2-warm), (b) just rewritten by a
// producer kernel running on all CUs (the producer -> consumer hand-off of the train step),
// (c) a different buffer every launch out of 64 (cold for the L2 and the TLBs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int NL>
__global__ __launch_bounds__(256) void consumer(const float* __restrict__ x, int ld, float* __restrict__ out, unsigned long long* st) {
    const int tid = threadIdx.x;
    unsigned long long t0 = wall_clock64();
    float4 v[NL];
    const float* base = x + (size_t)blockIdx.x * 16 * ld + (tid >> 4) * ld + (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = *reinterpret_cast<const float4*>(base + i * 64);
    unsigned long long t1 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t2 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
    out[blockIdx.x * 256 + tid] = s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t3 = wall_clock64();
    if (blockIdx.x == 0 && tid == 0) { st[0] += t1 - t0; st[1] += t2 - t0; st[2] += t3 - t0; st[3] += 1; }
}
__global__ void producer(float* x, long n, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n / 4; i += (long)gridDim.x * blockDim.x)
        reinterpret_cast<float4*>(x)[i] = make_float4(v, v, v, v);
}
__global__ void other(float* y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = y[i] * 1.0001f;
}
template <typename F> int run(const char* name, int nk, unsigned long long* st, F launch) {
    hipStream_t s; CK(hipStreamCreate(&s));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < nk; ++i) launch(s, i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipMemset(st, 0, 64));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
    hipEventRecord(e1, s); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4]; CK(hipMemcpy(h, st, 32, hipMemcpyDeviceToHost));
    const double n = h[3] ? (double)h[3] : 1.0;
    printf("%-58s: %6.2f us/kernel-in-chain | wg0: issued %5.2f us, data %5.2f us, stored %5.2f us\n", name,
           ms * 1e3f / (10.f * nk), h[0] / n / 100.0, h[1] / n / 100.0, h[2] / n / 100.0);
    return 0;
}
int main() {
    const int ld = 1024; const int WG = 48;                    // 48 workgroups x 16 rows x 1024 floats = 3 MB window per buffer
    const long nbuf = (long)WG * 16 * ld;
    std::vector<float*> bufs(64);
    for (auto& b : bufs) { CK(hipMalloc(&b, nbuf * 4)); CK(hipMemset(b, 0, nbuf * 4)); }
    float* out; CK(hipMalloc(&out, 1 << 22));
    float* big; CK(hipMalloc(&big, 64 << 20));
    unsigned long long* st; CK(hipMalloc(&st, 64));
    for (int wgs : {48, 384}) {
        char nm[128];
        const int rows = wgs * 16;                             // 384 WGs re-read rows modulo the buffer (nbuf covers 768 rows)
        (void)rows;
        auto cons16 = [&](hipStream_t s, const float* x) { consumer<16><<<wgs > 48 ? 48 : wgs, 256, 0, s>>>(x, ld, out, st); };
        if (wgs == 48) {
            snprintf(nm, 128, "16 loads/thread, 48 WGs, same buffer every kernel");
            run(nm, 40, st, [&](hipStream_t s, int) { cons16(s, bufs[0]); });
            snprintf(nm, 128, "16 loads/thread, 48 WGs, producer (256 WGs) rewrites it first");
            run(nm, 40, st, [&](hipStream_t s, int i) { if (i & 1) cons16(s, bufs[0]); else producer<<<256, 256, 0, s>>>(bufs[0], nbuf, (float)i); });
            snprintf(nm, 128, "16 loads/thread, 48 WGs, another of 64 buffers every kernel");
            run(nm, 64, st, [&](hipStream_t s, int i) { cons16(s, bufs[i & 63]); });
            snprintf(nm, 128, "16 loads/thread, 48 WGs, a 64 MB stream kernel in between");
            run(nm, 40, st, [&](hipStream_t s, int i) { if (i & 1) cons16(s, bufs[0]); else other<<<2048, 256, 0, s>>>(big, 16 << 20); });
            snprintf(nm, 128, " 4 loads/thread, 48 WGs, producer rewrites it first");
            run(nm, 40, st, [&](hipStream_t s, int i) { if (i & 1) consumer<4><<<48, 256, 0, s>>>(bufs[0], ld, out, st); else producer<<<256, 256, 0, s>>>(bufs[0], nbuf, (float)i); });
            snprintf(nm, 128, " 1 load /thread, 48 WGs, producer rewrites it first");
            run(nm, 40, st, [&](hipStream_t s, int i) { if (i & 1) consumer<1><<<48, 256, 0, s>>>(bufs[0], ld, out, st); else producer<<<256, 256, 0, s>>>(bufs[0], nbuf, (float)i); });
        }
    }
    return 0;
}
