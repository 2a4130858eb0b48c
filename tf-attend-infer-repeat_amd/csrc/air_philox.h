// Step prologue shared by step_begin_kernel and the GEMM kernels (which can carry it as extra
// workgroups of the hoisted x.Wx launch): annealing schedules (air_model.py:94-121) + Philox4x32-10
// noise keyed by (index, global_step).
#pragma once
#include "air_common.h"

struct AirStepJob {
    const air_schedule_t* sched; int nsched; float* dyn; const int32_t* istate;
    float* normals; long n_normal; float* uniforms; long n_uniform; uint32_t seed_lo, seed_hi;
    // optional: twin_dst[i] = bf16(twin_src[i]) for i < twin_n -- the bf16 twin of the image batch (the caller's
    // fp32 tensor), read by the input-weight gradient at the end of the step
    const float* twin_src; unsigned short* twin_dst; long twin_n;
};

__device__ __forceinline__ void air_philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void air_philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        air_philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float air_u01_open_low(uint32_t x) { return ((float)(x >> 8) + 1.0f) * 5.9604644775390625e-8f; }  // (0,1]
__device__ __forceinline__ float air_u01_half_open(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }          // [0,1)

// air_model.py:94-121: exponential_decay(init, step, iters, factor, staircase) -> max(min) -> min(max) -> log(.+eps)
__device__ __forceinline__ float air_eval_schedule(const air_schedule_t& s, int step) {
    float p = (float)step / s.iters;
    if (s.flags & 1) p = floorf(p);
    float v = s.init * powf(s.factor, p);
    if (s.flags & 2) v = fmaxf(v, s.vmin);
    if (s.flags & 4) v = fminf(v, s.vmax);
    if (s.flags & 8) v = logf(v + AIR_EPS);
    return v;
}

// workgroup `wg` of `nwg` (256 threads each) of the job
__device__ __forceinline__ void air_step_job_run(const AirStepJob& j, long wg, long nwg) {
    const int step = j.istate[AIR_IST_GLOBAL_STEP];
    if (wg == 0 && (int)threadIdx.x < j.nsched) {
        const air_schedule_t s = j.sched[threadIdx.x];
        j.dyn[s.slot] = air_eval_schedule(s, step);
    }
    const long quads_n = (j.n_normal + 3) / 4, quads_u = (j.n_uniform + 3) / 4, quads_t = (j.twin_n + 3) / 4;
    for (long q = wg * 256 + threadIdx.x; q < quads_n + quads_u + quads_t; q += nwg * 256) {
        if (q >= quads_n + quads_u) {
            const long base = (q - quads_n - quads_u) * 4;
            if (base + 3 < j.twin_n) {
                const float4 v = *reinterpret_cast<const float4*>(j.twin_src + base);
                *reinterpret_cast<uint2*>(j.twin_dst + base) = make_uint2(air_pack_bf16(v.x, v.y), air_pack_bf16(v.z, v.w));
            } else
                for (int k = 0; k < 4; ++k) if (base + k < j.twin_n) j.twin_dst[base + k] = air_bf16_of(j.twin_src[base + k]);
            continue;
        }
        uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)step, 0x41495221u};
        air_philox4x32_10(c, j.seed_lo, j.seed_hi);
        float v[4];
        if (q < quads_n) {
            // Box-Muller on two pairs, on the hardware transcendentals: v_log_f32 (2^-23-level relative error away from 1),
            // v_sin_f32 / v_cos_f32 take their argument in REVOLUTIONS, i.e. the uniform itself -- no 2*pi range reduction.
            // (The library sinf / cosf / logf cost ~400 instructions per quad: at the 128x128 configuration the noise planes
            // -- 1.07 M normals per step -- were ~20 us of the x.Wx launch that carries them.  The reference's RNG ops are
            // unseeded; only the distribution matters, tests/test_gpu_kernels.py::test_step_begin_schedule_and_noise.)
            const float r0 = sqrtf(-2.0f * __logf(air_u01_open_low(c[0]))), a0 = air_u01_half_open(c[1]);
            const float r1 = sqrtf(-2.0f * __logf(air_u01_open_low(c[2]))), a1 = air_u01_half_open(c[3]);
            v[0] = r0 * __builtin_amdgcn_cosf(a0); v[1] = r0 * __builtin_amdgcn_sinf(a0);
            v[2] = r1 * __builtin_amdgcn_cosf(a1); v[3] = r1 * __builtin_amdgcn_sinf(a1);
            const long base = q * 4;
            for (int k = 0; k < 4; ++k) if (base + k < j.n_normal) j.normals[base + k] = v[k];
        } else {
            const long base = (q - quads_n) * 4;
            for (int k = 0; k < 4; ++k) if (base + k < j.n_uniform) j.uniforms[base + k] = air_u01_half_open(c[k]);
        }
    }
}
