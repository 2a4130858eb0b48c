// Experiment: cost of executing straight-line code once (cold instruction cache) vs the same
// instruction count in a loop.  N_INSTR v_add_f32 per wave; kernels alternate so that each launch
// starts with another kernel's code in the instruction cache.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int UNROLL, int ITERS>
__global__ void k(float* out, float v, long long* cyc) {
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; u += 4) asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(v));
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
template <int UNROLL, int ITERS>
void run(const char* name, float* out, long long* cyc) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // evict: run the other shape in between
    float ms_total = 0; long long c = 0, csum = 0;
    for (int rep = 0; rep < 20; ++rep) {
        k<4096, 1><<<256, 64>>>(out, 1.0f, cyc + 8);            // different code, 16 KB straight line
        k<4088, 1><<<256, 64>>>(out, 1.0f, cyc + 8);
        k<4084, 1><<<256, 64>>>(out, 1.0f, cyc + 8);
        k<4092, 1><<<256, 64>>>(out, 1.0f, cyc + 8);
        hipEventRecord(e0);
        k<UNROLL, ITERS><<<256, 64>>>(out, 1.0f, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms_total += ms;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); csum += c;
    }
    printf("%-34s: %.2f us per launch (event), %lld cycles in-kernel = %.2f per instruction\n", name, ms_total / 20 * 1e3, csum / 20,
           (double)csum / 20 / (UNROLL * (double)ITERS));
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 256);
    run<64, 64>("4096 adds as 64 x loop of 64", out, cyc);
    run<4096, 1>("4096 adds straight-line", out, cyc);
    run<64, 16>("1024 adds as 16 x loop of 64", out, cyc);
    run<1024, 1>("1024 adds straight-line", out, cyc);
    run<2048, 1>("2048 adds straight-line", out, cyc);
    run<8192, 1>("8192 adds straight-line", out, cyc);
    run<64, 128>("8192 adds as 128 x loop of 64", out, cyc);
    return 0;
}
