// Experiment: cost of ds_add_f32 instructions under the patterns the corner feeder produces.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void lds_fadd(float* p, float v) {
    asm volatile("ds_add_f32 %0, %1" :: "v"((unsigned)(size_t)p), "v"(v) : "memory");
}
// mode 0: same address every instruction; 1: address cycles over 4 words per instruction;
// 2: like 0 but 52 of 64 lanes add -0.0f; 3: like 0 but only 12 lanes active (EXEC); 4: latency: wait after each
__global__ void k(float* out, long long* cyc, int mode, int ninstr) {
    __shared__ float acc[64];
    if (threadIdx.x < 64) acc[threadIdx.x] = 0.f;
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = 1.0f + threadIdx.x * 1e-3f;
        if (mode == 2 && threadIdx.x >= 12) v = -0.0f;
        long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < ninstr; ++i) {
            float* p = acc + ((mode == 1) ? (i & 3) : 0);
            if (mode == 3) { if (threadIdx.x < 12) lds_fadd(p, v); }
            else lds_fadd(p, v);
            if (mode == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) cyc[0] = t1 - t0;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 64); hipMalloc(&cyc, 64);
    const char* names[] = {"same address", "4 addresses cycling per instruction", "52 lanes add -0.0f", "12 lanes active (EXEC)", "wait after each"};
    for (int n : {1, 16, 256})
        for (int mode = 0; mode < 5; ++mode) {
            k<<<1, 256>>>(out, cyc, mode, n);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%4d x ds_add_f32, %-38s: %8lld cycles (%.0f per instruction)\n", n, names[mode], c, (double)c / n);
        }
    return 0;
}
