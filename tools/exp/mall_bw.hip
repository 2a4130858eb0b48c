// What does an ApplyAdam-shaped stream (read p, m, v, g; write p, m, v) reach when its whole footprint fits the 256 MiB
// Infinity Cache, and do non-temporal hints change it?  The step's optimizer launch moves ~120 MB in 21.5 us (5.6 TB/s);
// the question is whether that is the HBM ceiling or whether repeated launches over the same 64 MB can run from the MALL.
//   hipcc --offload-arch=gfx950 -O3 -o mall_bw mall_bw.hip && ./mall_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include "air_common.h"
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0 plain, 1 nt loads of g only, 2 nt everything, 3 read-only (p, m, v, g), 4 nt stores only
__global__ __launch_bounds__(256) void adam_like(f4* __restrict__ p, f4* __restrict__ m, f4* __restrict__ v, const f4* __restrict__ g, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += gridDim.x * 256L) {
        f4 a, b, c, d;
        if (MODE == 2) { a = __builtin_nontemporal_load(p + i); b = __builtin_nontemporal_load(m + i); c = __builtin_nontemporal_load(v + i); }
        else { a = p[i]; b = m[i]; c = v[i]; }
        d = (MODE == 1 || MODE == 2) ? __builtin_nontemporal_load(g + i) : g[i];
        b = b + (d - b) * 0.1f; c = c + (d * d - c) * 0.001f; a = a - b * 1e-4f;
        if (MODE == 3) { if (a.x == 123.456f) p[i] = a + b + c; }
        else if (MODE == 2 || MODE == 4) { __builtin_nontemporal_store(a, p + i); __builtin_nontemporal_store(b, m + i); __builtin_nontemporal_store(c, v + i); }
        else { p[i] = a; m[i] = b; v[i] = c; }
    }
}
// the step's optimizer kernel with pieces removed: PRE = re-reduce the norm partials first, SH = write the bf16 shadow,
// EXACT = IEEE sqrt / divide update (else the cheap one above), U = quads per thread per iteration
template <bool PRE, bool SH, bool EXACT, int U>
__global__ __launch_bounds__(256) void adam_real(float4* __restrict__ p4, float4* __restrict__ m4, float4* __restrict__ v4, const float4* __restrict__ g4, long n4,
                                                 const float* __restrict__ partials, int npartials, const float* __restrict__ dyn, const int32_t* __restrict__ istate,
                                                 uint2* __restrict__ shadow) {
    __shared__ float red[4];
    AirAdamCoef cf{1.0f, 1e-4f, 1.0f};
    if (PRE) cf = air_adam_coef(partials, npartials, dyn, istate, 1.0f, 0.9f, 0.999f, red);
    const float omb1 = 0.1f, omb2 = 0.001f, eps = 1e-8f;
    const long stride = gridDim.x * 256L;
    for (long i0 = blockIdx.x * 256L + threadIdx.x; i0 < n4; i0 += stride * U) {
        float4 pp[U], mm[U], vv[U], gg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long i = i0 + u * stride; if (i < n4) { pp[u] = p4[i]; mm[u] = m4[i]; vv[u] = v4[i]; gg[u] = g4[i]; } }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * stride;
            if (i >= n4) continue;
            float* pa = &pp[u].x; float* ma = &mm[u].x; float* va = &vv[u].x; const float* ga = &gg[u].x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (EXACT) air_adam_update(pa[k], ma[k], va[k], ga[k], cf, omb1, omb2, eps);
                else { ma[k] = ma[k] + (ga[k] - ma[k]) * omb1; va[k] = va[k] + (ga[k] * ga[k] - va[k]) * omb2; pa[k] = pa[k] - ma[k] * cf.lr_t; }
            }
            p4[i] = pp[u]; m4[i] = mm[u]; v4[i] = vv[u];
            if (SH) shadow[i] = make_uint2(air_pack_bf16(pp[u].x, pp[u].y), air_pack_bf16(pp[u].z, pp[u].w));
        }
    }
}
// the same kernel with the first quad's loads issued between the partial-sum loads and their reduction (vmcnt retires in
// order: the partials must be the OLDER loads or waiting for them waits for everything)
template <bool SH>
__global__ __launch_bounds__(256) void adam_early(float4* __restrict__ p4, float4* __restrict__ m4, float4* __restrict__ v4, const float4* __restrict__ g4, long n4,
                                                  const float* __restrict__ partials, int npartials, const float* __restrict__ dyn, const int32_t* __restrict__ istate,
                                                  uint2* __restrict__ shadow) {
    __shared__ float red[4];
    float s = 0.0f;
    for (int i = threadIdx.x; i < npartials; i += 256) s += partials[i];
    const long stride = gridDim.x * 256L;
    long i = blockIdx.x * 256L + threadIdx.x;
    float4 pp, mm, vv, gg;
    const bool first = i < n4;
    if (first) { pp = p4[i]; mm = m4[i]; vv = v4[i]; gg = g4[i]; }
    s = air_block_sum_256(s, red);
    AirAdamCoef cf;
    cf.gnorm = sqrtf(s);
    const float clip = dyn[AIR_DYN_CLIP_NORM];
    cf.scale = (clip > 0.0f ? clip * fminf(1.0f / cf.gnorm, 1.0f / clip) : 1.0f);
    const float t = (float)istate[AIR_IST_GLOBAL_STEP];
    cf.lr_t = dyn[AIR_DYN_LEARNING_RATE] * sqrtf(1.0f - powf(0.999f, t)) / (1.0f - powf(0.9f, t));
    const float omb1 = 0.1f, omb2 = 0.001f, eps = 1e-8f;
    for (; i < n4; i += stride) {
        float4 pn, mn, vn, gn;
        const bool more = i + stride < n4;
        if (more) { pn = p4[i + stride]; mn = m4[i + stride]; vn = v4[i + stride]; gn = g4[i + stride]; }
        float* pa = &pp.x; float* ma = &mm.x; float* va = &vv.x; const float* ga = &gg.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) air_adam_update(pa[k], ma[k], va[k], ga[k], cf, omb1, omb2, eps);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (SH) shadow[i] = make_uint2(air_pack_bf16(pp.x, pp.y), air_pack_bf16(pp.z, pp.w));
        if (more) { pp = pn; mm = mn; vv = vn; gg = gn; }
    }
}
float run_early(f4* p, f4* m, f4* v, f4* g, long n4, int grid, int reps, float* partials, float* dyn, int32_t* ist, uint2* sh) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&]() { hipLaunchKernelGGL((adam_early<true>), dim3(grid), dim3(256), 0, 0, (float4*)p, (float4*)m, (float4*)v, (const float4*)g, n4, partials, 512, dyn, ist, sh); };
    for (int i = 0; i < 3; ++i) go();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) go();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}
template <bool PRE, bool SH, bool EXACT, int U> float run_real(f4* p, f4* m, f4* v, f4* g, long n4, int grid, int reps, float* partials, float* dyn, int32_t* ist, uint2* sh) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto go = [&]() { hipLaunchKernelGGL((adam_real<PRE, SH, EXACT, U>), dim3(grid), dim3(256), 0, 0, (float4*)p, (float4*)m, (float4*)v, (const float4*)g, n4, partials, 512, dyn, ist, sh); };
    for (int i = 0; i < 3; ++i) go();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) go();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}
template <int MODE> float run(f4* p, f4* m, f4* v, f4* g, long n4, int grid, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(adam_like<MODE>, dim3(grid), dim3(256), 0, 0, p, m, v, g, n4);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(adam_like<MODE>, dim3(grid), dim3(256), 0, 0, p, m, v, g, n4);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}
int main() {
    const long nmax = 64L << 20;   // floats per array
    f4 *p, *m, *v, *g;
    hipMalloc(&p, nmax * 4); hipMalloc(&m, nmax * 4); hipMalloc(&v, nmax * 4); hipMalloc(&g, nmax * 4);
    hipMemset(p, 0, nmax * 4); hipMemset(m, 0, nmax * 4); hipMemset(v, 0, nmax * 4); hipMemset(g, 0, nmax * 4);
    const char* names[5] = {"plain", "nt-g", "nt-all", "read-only", "nt-stores"};
    for (long n : {1L << 20, 2L << 20, 4L << 20, 8L << 20, 16L << 20, 64L << 20}) {
        for (int grid : {1024, 2048, 4096}) {
            float us[5];
            us[0] = run<0>(p, m, v, g, n / 4, grid, 20); us[1] = run<1>(p, m, v, g, n / 4, grid, 20);
            us[2] = run<2>(p, m, v, g, n / 4, grid, 20); us[3] = run<3>(p, m, v, g, n / 4, grid, 20);
            us[4] = run<4>(p, m, v, g, n / 4, grid, 20);
            printf("n = %3ld M floats (footprint %4ld MB) grid %4d:", n >> 20, n * 16 >> 20, grid);
            for (int k = 0; k < 5; ++k) {
                const double bytes = (k == 3 ? 16.0 : 28.0) * n;
                printf("  %s %.1f us %.2f TB/s", names[k], us[k], bytes / us[k] * 1e-6);
            }
            printf("\n");
        }
    }
    // ---- the real kernel, 4 M floats (the step's variable buffer is 4.0 M), pieces removed one at a time
    float *partials, *dyn; int32_t* ist; uint2* sh;
    hipMalloc(&partials, 4096); hipMalloc(&dyn, 256); hipMalloc(&ist, 256); hipMalloc(&sh, nmax * 2);
    { float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 1.0f; hipMemcpy(partials, h, 4096, hipMemcpyHostToDevice);
      float d[64]; for (int i = 0; i < 64; ++i) d[i] = 1e-4f; hipMemcpy(dyn, d, 256, hipMemcpyHostToDevice);
      int32_t is[64]; for (int i = 0; i < 64; ++i) is[i] = 100; hipMemcpy(ist, is, 256, hipMemcpyHostToDevice); }
    hipMemset(v, 0x3c, nmax * 4);
    const long n = 4L << 20;
    for (int grid : {512, 1024, 2048, 4096, 8192})
        printf("early-load kernel, grid %d: %.1f us (full, same grid: %.1f)\n", grid, run_early(p, m, v, g, n / 4, grid, 20, partials, dyn, ist, sh),
               run_real<true, true, true, 1>(p, m, v, g, n / 4, grid, 20, partials, dyn, ist, sh));
    for (int grid : {1024, 2048, 4096}) {
        printf("real kernel, 4 M floats, grid %d: full %.1f | no preamble %.1f | no shadow %.1f | cheap math %.1f | none of them %.1f | full U=2 %.1f | full U=4 %.1f | U=2 no preamble %.1f us\n", grid,
               run_real<true, true, true, 1>(p, m, v, g, n / 4, grid, 20, partials, dyn, ist, sh),
               run_real<false, true, true, 1>(p, m, v, g, n / 4, grid, 20, partials, dyn, ist, sh),
               run_real<true, false, true, 1>(p, m, v, g, n / 4, grid, 20, partials, dyn, ist, sh),
               run_real<true, true, false, 1>(p, m, v, g, n / 4, grid, 20, partials, dyn, ist, sh),
               run_real<false, false, false, 1>(p, m, v, g, n / 4, grid, 20, partials, dyn, ist, sh),
               run_real<true, true, true, 2>(p, m, v, g, n / 4, grid, 20, partials, dyn, ist, sh),
               run_real<true, true, true, 4>(p, m, v, g, n / 4, grid, 20, partials, dyn, ist, sh),
               run_real<false, true, true, 2>(p, m, v, g, n / 4, grid, 20, partials, dyn, ist, sh));
    }
    return 0;
}
