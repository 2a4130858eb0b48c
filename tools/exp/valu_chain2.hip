// Does a dependent v_add_f32 chain run faster when EXEC covers only part of the wave (pass skipping)?  And what does a
// broadcast-read chain (ds_read_b128 ahead of 4 dependent adds, scalar loop control) reach per term?
//   hipcc --offload-arch=gfx950 -O3 -o valu_chain2 valu_chain2.hip && ./valu_chain2
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, long long* cyc, float v, int nact) {
    float a = threadIdx.x;
    long long t0 = 0, t1 = 0;
    if ((int)(threadIdx.x & 63) < nact) {
        t0 = __builtin_readcyclecounter();
#pragma unroll 1
        for (int i = 0; i < 256; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(v));
        }
        t1 = __builtin_readcyclecounter();
    }
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x + blockDim.x * blockIdx.x] = a;
}
// one wave per corner stream: uniform addresses (LDS broadcast), 16 terms per iteration, reads two iterations ahead
__global__ void chain_lds(const float* src, float* out, long long* cyc, int n, int nwaves_active) {
    extern __shared__ float T[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) T[i] = src[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    float acc = 0.0f;
    long long t0 = __builtin_readcyclecounter();
    if (wave < nwaves_active) {
        const float4* p4 = reinterpret_cast<const float4*>(T) + __builtin_amdgcn_readfirstlane(wave) * 0;
        const int nb = n / 16;
        // three register slots in rotation, no moves (a v_mov of a slot would wait for its loads): slot s holds block b with b % 3 == s
        float4 r[3][4];
        auto load = [&](float4 (&dst)[4], int b) {
            const int nx = min(b, nb - 1) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[q] = p4[nx + q];
        };
        auto sum = [&](const float4 (&c)[4]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc += c[q].x; acc += c[q].y; acc += c[q].z; acc += c[q].w; }
        };
        load(r[0], 0); load(r[1], 1);
        int b = 0;
        for (; b + 3 <= nb; b += 3) {
            load(r[2], b + 2); sum(r[0]);
            load(r[0], b + 3); sum(r[1]);
            load(r[1], b + 4); sum(r[2]);
        }
        if (b < nb) { sum(r[0]); ++b; }
        if (b < nb) { sum(r[1]); ++b; }
    }
    long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    out[threadIdx.x] = acc;
}
int main() {
    float* out; long long* cyc; float* src; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 256); hipMalloc(&src, 1 << 20);
    hipMemset(src, 0, 1 << 20);
    for (int nact : {1, 16, 32, 64}) {
        k<<<1, 64>>>(out, cyc, 1.0f, nact);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("dependent chain, %2d active lanes: %.2f cycles/add\n", nact, c / 16384.0);
    }
    const int n = 16384;
    for (int threads : {256, 1024})
        for (int nw : {1, 4, 8, 16}) {
            if (nw * 64 > threads) continue;
            chain_lds<<<1, threads, n * 4>>>(src, out, cyc, n, nw);
            long long c[16]; hipMemcpy(c, cyc, 128, hipMemcpyDeviceToHost);
            long long mx = 0; for (int i = 0; i < nw; ++i) mx = c[i] > mx ? c[i] : mx;
            printf("LDS broadcast chain, %4d threads, %2d waves each summing %d terms: %.2f cycles/term (slowest wave)\n", threads, nw, n, mx / (double)n);
        }
    return 0;
}
