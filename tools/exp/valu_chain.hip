// Experiment: issue rate of a lone wave's v_add_f32 (dependent chain vs independent), core clock.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, long long* cyc, float v, int mode) {
    float a = threadIdx.x, b = 1.f, c = 2.f, d = 3.f;
    long long w0 = wall_clock64();
    long long t0 = __builtin_readcyclecounter();
    if (mode == 0) {
#pragma unroll 1
        for (int i = 0; i < 256; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(v));
        }
    } else {
#pragma unroll 1
        for (int i = 0; i < 256; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(v));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(b) : "v"(v));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(c) : "v"(v));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(d) : "v"(v));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
    out[threadIdx.x + blockDim.x * blockIdx.x] = a + b + c + d;
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
    for (int threads : {64, 256, 1024})
        for (int mode = 0; mode < 2; ++mode) {
            k<<<1, threads>>>(out, cyc, 1.0f, mode);
            long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
            printf("%4d threads, %s: %lld cycles for 16384 adds per wave -> %.2f cycles/add; wall %.2f us -> counter %.0f MHz\n", threads,
                   mode ? "4 independent chains" : "1 dependent chain   ", c[0], c[0] / 16384.0, c[1] / 100.0, c[0] / (c[1] / 100.0));
        }
    return 0;
}
