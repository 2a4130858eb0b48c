// Step prologue (annealing schedules + Philox noise) and the optimizer
// (clip_by_global_norm + TF-1.3 ApplyAdam) over ONE flat parameter buffer.
// Both are pure HBM streams: float4 accesses, grid sized to fill 256 CUs,
// deterministic two-level reduction for the global norm.
#include "air_common.h"
#include "air_philox.h"

namespace {

constexpr int THREADS = 256;
constexpr int NORM_BLOCKS = 1024;      // partial sums; reduced again inside the Adam kernel

__global__ __launch_bounds__(THREADS) void step_begin_kernel(AirStepJob job)
{
    air_step_job_run(job, blockIdx.x, gridDim.x);
}

// sum of squares, NORM_BLOCKS partials (tf.clip_by_global_norm's 2*l2_loss terms)
__global__ __launch_bounds__(THREADS) void grad_sqnorm_kernel(
    const float* __restrict__ g, long n, float* __restrict__ partials, int32_t* __restrict__ istate)
{
    __shared__ float red[4];
    float acc = 0.0f;
    const long n4 = n / 4;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long i = (long)blockIdx.x * THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * THREADS) {
        const float4 v = g4[i];
        acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) { const float v = g[n4 * 4 + threadIdx.x]; acc += v * v; }
    acc = air_block_sum_256(acc, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = acc;
        if (blockIdx.x == 0) istate[AIR_IST_GLOBAL_STEP] += 1;   // apply_gradients(global_step=...) :692-694
    }
}

// ApplyAdam (TF 1.3 training_ops): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
// m += (g-m)(1-b1); v += (g^2-v)(1-b2); var -= lr_t*m/(sqrt(v)+eps)
__global__ __launch_bounds__(THREADS) void adam_clip_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
    const float* __restrict__ partials, int npartials, const float* __restrict__ dyn, const int32_t* __restrict__ istate,
    float prescale, float b1, float b2, float eps, uint16_t* __restrict__ shadow, float* __restrict__ gnorm_out,
    float* __restrict__ coef_out)
{
    __shared__ float red[4];
    // The first quad's loads go out BETWEEN the partial-sum loads and their reduction (vmcnt retires in order, so the
    // partials have to be the older loads), and every later quad's loads before the previous quad's arithmetic: the
    // coefficient prologue (one round trip, two barriers, sqrt / pow: ~2 us) no longer delays the stream.
    const float share = air_adam_partial_share(partials, npartials);
    const long n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    const long stride = (long)gridDim.x * THREADS;
    long i = (long)blockIdx.x * THREADS + threadIdx.x;
    float4 pp, mm, vv, gg;
    if (i < n4) { pp = p4[i]; mm = m4[i]; vv = v4[i]; gg = g4[i]; }
    const AirAdamCoef cf = air_adam_coef_from_share(share, dyn, istate, prescale, b1, b2, red);
    if (gnorm_out && blockIdx.x == 0 && threadIdx.x == 0) *gnorm_out = cf.gnorm;
    // the record deferred slices of this step's update read (air_step_job_t.ad_coef): dyn / istate / partials may have
    // moved on to the next step by the time they run
    if (coef_out && blockIdx.x == 0 && threadIdx.x == 0) { coef_out[0] = cf.scale; coef_out[1] = cf.lr_t; coef_out[2] = cf.gnorm; }
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;

    for (; i < n4; i += stride) {
        float4 pn, mn, vn, gn;
        const bool more = i + stride < n4;
        if (more) { pn = p4[i + stride]; mn = m4[i + stride]; vn = v4[i + stride]; gn = g4[i + stride]; }
        float* pa = &pp.x; float* ma = &mm.x; float* va = &vv.x; const float* ga = &gg.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) air_adam_update(pa[k], ma[k], va[k], ga[k], cf, omb1, omb2, eps);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (shadow) reinterpret_cast<uint2*>(shadow)[i] = make_uint2(air_pack_bf16(pp.x, pp.y), air_pack_bf16(pp.z, pp.w));
        if (more) { pp = pn; mm = mn; vv = vn; gg = gn; }
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) {
        const long i = n4 * 4 + threadIdx.x;
        float pk = p[i], mk = m[i], vk = v[i];
        air_adam_update(pk, mk, vk, g[i], cf, omb1, omb2, eps);
        p[i] = pk; m[i] = mk; v[i] = vk;
        if (shadow) shadow[i] = air_bf16_of(pk);
    }
}

}  // namespace

extern "C" int air_abi_version(void) { return AIR_ABI_VERSION; }

extern "C" const char* air_strerror(int code) {
    switch (code) {
        case 0: return "success";
        case AIR_EINVAL: return "air: invalid argument (null pointer or non-positive dimension)";
        case AIR_ELIMIT: return "air: size exceeds kernel limit";
        case AIR_EALIGN: return "air: pointer or leading dimension misaligned";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "air: unknown error";
    }
}

extern "C" int air_step_begin(const air_schedule_t* sched, int nsched, float* dyn, const int32_t* istate,
                              float* normals, int64_t n_normal, float* uniforms, int64_t n_uniform,
                              uint64_t seed, const float* twin_src, uint16_t* twin_dst, int64_t twin_n, void* stream) {
    if (!dyn || !istate || nsched < 0 || nsched > THREADS || (nsched > 0 && !sched)) return AIR_EINVAL;
    if (n_normal < 0 || n_uniform < 0 || (n_normal > 0 && !normals) || (n_uniform > 0 && !uniforms)) return AIR_EINVAL;
    if (twin_n < 0 || (twin_n > 0 && (!twin_src || !twin_dst))) return AIR_EINVAL;
    if (twin_n > 0 && ((((uintptr_t)twin_src) & 15) != 0 || (((uintptr_t)twin_dst) & 7) != 0)) return AIR_EALIGN;
    const long quads = (n_normal + 3) / 4 + (n_uniform + 3) / 4 + (twin_n + 3) / 4;
    long blocks = (quads + THREADS - 1) / THREADS;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    AirStepJob job{sched, nsched, dyn, istate, normals, (long)n_normal, uniforms, (long)n_uniform,
                   (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), twin_src, twin_dst, (long)twin_n,
                   nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0.f, 0.f, 0.f};
    hipLaunchKernelGGL(step_begin_kernel, dim3((int)blocks), dim3(THREADS), 0, air_stream(stream), job);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_optim_num_partials(int64_t n) { (void)n; return NORM_BLOCKS; }

extern "C" int air_grad_sqnorm(const float* grads, int64_t n, float* partials, int32_t* istate, void* stream) {
    if (!grads || !partials || !istate || n <= 0) return AIR_EINVAL;
    if (((uintptr_t)grads & 15) != 0) return AIR_EALIGN;
    hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(NORM_BLOCKS), dim3(THREADS), 0, air_stream(stream),
                       grads, (long)n, partials, istate);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_adam_clip_step_blocks(float* params, const float* grads, float* m, float* v, int64_t n,
                                         const float* partials, int npartials, const float* dyn, const int32_t* istate,
                                         float grad_prescale, float beta1, float beta2, float epsilon,
                                         uint16_t* bf16_shadow, float* gnorm_out, int max_blocks, float* coef_out, void* stream);

extern "C" int air_adam_clip_step(float* params, const float* grads, float* m, float* v, int64_t n,
                                  const float* partials, int npartials, const float* dyn, const int32_t* istate,
                                  float grad_prescale, float beta1, float beta2, float epsilon,
                                  uint16_t* bf16_shadow, float* gnorm_out, void* stream) {
    return air_adam_clip_step_blocks(params, grads, m, v, n, partials, npartials, dyn, istate, grad_prescale, beta1, beta2,
                                     epsilon, bf16_shadow, gnorm_out, 0, nullptr, stream);
}

extern "C" int air_adam_clip_step_blocks(float* params, const float* grads, float* m, float* v, int64_t n,
                                         const float* partials, int npartials, const float* dyn, const int32_t* istate,
                                         float grad_prescale, float beta1, float beta2, float epsilon,
                                         uint16_t* bf16_shadow, float* gnorm_out, int max_blocks, float* coef_out, void* stream) {
    if (!params || !grads || !m || !v || !partials || npartials <= 0 || !dyn || !istate || n <= 0) return AIR_EINVAL;
    if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return AIR_EALIGN;
    if (((uintptr_t)bf16_shadow & 7) != 0) return AIR_EALIGN;
    long blocks = (n / 4 + THREADS - 1) / THREADS;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;        // (512 .. 8192 measured: 20.3 .. 23.8 us in isolation, no difference inside the step)
    if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL(adam_clip_kernel, dim3((int)blocks), dim3(THREADS), 0, air_stream(stream),
                       params, grads, m, v, (long)n, partials, npartials, dyn, istate, grad_prescale, beta1, beta2,
                       epsilon, bf16_shadow, gnorm_out, coef_out);
    AIR_CHECK_LAUNCH();
    return 0;
}
