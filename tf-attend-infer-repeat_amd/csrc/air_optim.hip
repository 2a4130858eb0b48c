// Step prologue (annealing schedules + Philox noise) and the optimizer
// (clip_by_global_norm + TF-1.3 ApplyAdam) over ONE flat parameter buffer.
// Both are pure HBM streams: float4 accesses, grid sized to fill 256 CUs,
// deterministic two-level reduction for the global norm.
#include "air_common.h"
#include "air_philox.h"
#include <cstdlib>

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

// Panel-blocked bf16 shadows (air_panel_t): where quad `i` (elements 4i .. 4i+3 of the flat buffer; a quad never
// straddles a row: N % 4 == 0) of a described [K, N] matrix goes.  All in units of quads / 32-bit: n / 4 < 2^31.
// row = rel / N4 through a multiply-shift (magic = ceil(2^40 / N4): exact for rel < 2^40 / N4, i.e. always here).
struct PanelTab {
    int count;
    unsigned first4[AIR_MAX_PANELS];              // first quad of matrix j (0xffffffff past `count`): ascending
    unsigned len4[AIR_MAX_PANELS];                // K * N / 4
    unsigned N4[AIR_MAX_PANELS], K16[AIR_MAX_PANELS];          // N / 4, K * 16
    unsigned R4[AIR_MAX_PANELS];                  // gates == 4: R / 4 (quads per gate block), else 0
    unsigned excl[AIR_MAX_PANELS];
    unsigned long long magic[AIR_MAX_PANELS], dst[AIR_MAX_PANELS];
};
// returns true when the quad lies in a described matrix; `at` = its element offset in the panel shadow,
// `keep_flat` = whether the row-major shadow is maintained for it as well
__device__ __forceinline__ bool panel_of(const PanelTab& t, unsigned i, unsigned long long& at, bool& keep_flat) {
    int pi = -1;
#pragma unroll
    for (int j = 0; j < AIR_MAX_PANELS; ++j) pi += (i >= t.first4[j]) ? 1 : 0;
    keep_flat = true;
    if (pi < 0) return false;
    const unsigned rel = i - t.first4[pi];
    if (rel >= t.len4[pi]) return false;
    const unsigned N4 = t.N4[pi];
    const unsigned k = (unsigned)(((unsigned long long)rel * t.magic[pi]) >> 40);
    const unsigned c4 = rel - k * N4;                          // quad within the row: columns 4 c4 .. 4 c4 + 3
    const unsigned R4 = t.R4[pi];
    unsigned panel, within;
    if (R4) {                                                  // gate-interleaved: quad = 4 units of one gate
        const unsigned gate = (c4 >= R4 ? 1u : 0u) + (c4 >= 2u * R4 ? 1u : 0u) + (c4 >= 3u * R4 ? 1u : 0u);
        panel = c4 - gate * R4; within = gate * 4u;
    } else { panel = c4 >> 2; within = (c4 & 3u) * 4u; }
    at = t.dst[pi] + (unsigned long long)panel * t.K16[pi] + k * 16u + within;
    keep_flat = t.excl[pi] == 0u;
    return true;
}

// The same map with the matrix found by a caller that knows it block-uniformly (adam_panels_kernel): `pi` is uniform,
// so the table fields are scalar loads and the per-quad cost is one multiply-high and a few adds.
__device__ __forceinline__ unsigned long long panel_at(const PanelTab& t, int pi, unsigned rel) {
    const unsigned N4 = t.N4[pi];
    const unsigned k = (unsigned)(((unsigned long long)rel * t.magic[pi]) >> 40);
    const unsigned c4 = rel - k * N4;
    const unsigned R4 = t.R4[pi];
    unsigned panel, within;
    if (R4) {
        const unsigned gate = (c4 >= R4 ? 1u : 0u) + (c4 >= 2u * R4 ? 1u : 0u) + (c4 >= 3u * R4 ? 1u : 0u);
        panel = c4 - gate * R4; within = gate * 4u;
    } else { panel = c4 >> 2; within = (c4 & 3u) * 4u; }
    return t.dst[pi] + (unsigned long long)panel * t.K16[pi] + k * 16u + within;
}

__global__ __launch_bounds__(THREADS) void panel_shadow_kernel(const float* __restrict__ p, uint16_t* __restrict__ panels, PanelTab tab)
{
    const unsigned lo = tab.first4[0], hi = tab.first4[tab.count - 1] + tab.len4[tab.count - 1];
    for (unsigned i = lo + blockIdx.x * THREADS + threadIdx.x; i < hi; i += gridDim.x * THREADS) {
        unsigned long long at; bool keep;
        if (!panel_of(tab, i, at, keep)) continue;
        const float4 v = reinterpret_cast<const float4*>(p)[i];
        *reinterpret_cast<uint2*>(panels + at) = make_uint2(air_pack_bf16(v.x, v.y), air_pack_bf16(v.z, v.w));
    }
}

// ApplyAdam (TF 1.3 training_ops): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
// m += (g-m)(1-b1); v += (g^2-v)(1-b2); var -= lr_t*m/(sqrt(v)+eps)
__global__ __launch_bounds__(THREADS) void adam_clip_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
    const float* __restrict__ partials, int npartials, const float* __restrict__ dyn, const int32_t* __restrict__ istate,
    float prescale, float b1, float b2, float eps, uint16_t* __restrict__ shadow, float* __restrict__ gnorm_out)
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
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;

    for (; i < n4; i += stride) {
        float4 pn, mn, vn, gn;
        const bool more = i + stride < n4;
        if (more) { pn = p4[i + stride]; mn = m4[i + stride]; vn = v4[i + stride]; gn = g4[i + stride]; }
        float* pa = &pp.x; float* ma = &mm.x; float* va = &vv.x; const float* ga = &gg.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) air_adam_update(pa[k], ma[k], va[k], ga[k], cf, omb1, omb2, eps);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        const uint2 tw = make_uint2(air_pack_bf16(pp.x, pp.y), air_pack_bf16(pp.z, pp.w));
        if (shadow) reinterpret_cast<uint2*>(shadow)[i] = tw;
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

// clip + ApplyAdam that also maintains the panel-blocked twins.  Same arithmetic per element as adam_clip_kernel; the
// WORK MAP differs: a workgroup owns a contiguous chunk of CHUNK quads (4 per thread, all 16 operand loads of a thread
// issued before the coefficient reduction) instead of a grid-stride sweep, because
//   * the matrix a quad belongs to is then (nearly) block-uniform: the table lookup is scalar work, and
//   * a chunk of 4 096 consecutive variables covers whole rows of a matrix, so the 8-byte pieces it scatters into the
//     panel twin (the four gates of an LSTM row land in four different panels' rows; a plain row in N / 16 panels)
//     complete whole 128-byte lines from ONE workgroup -- one L2 merges them -- instead of leaving every line to be
//     assembled from partial writes of sixteen workgroups on eight XCDs.
constexpr int CHUNK = 4 * THREADS;
__global__ __launch_bounds__(THREADS) void adam_panels_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
    const float* __restrict__ partials, int npartials, const float* __restrict__ dyn, const int32_t* __restrict__ istate,
    float prescale, float b1, float b2, float eps, uint16_t* __restrict__ shadow, float* __restrict__ gnorm_out,
    uint16_t* __restrict__ panels, PanelTab tab)
{
    __shared__ float red[4];
    const float share = air_adam_partial_share(partials, npartials);
    const long n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    const long lo = (long)blockIdx.x * CHUNK;
    float4 pp[4], mm[4], vv[4], gg[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long i = lo + threadIdx.x + THREADS * r;
        const long j = i < n4 ? i : 0;                       // (clamped: loads stay unconditional, stores are guarded)
        pp[r] = p4[j]; mm[r] = m4[j]; vv[r] = v4[j]; gg[r] = g4[j];
    }
    const AirAdamCoef cf = air_adam_coef_from_share(share, dyn, istate, prescale, b1, b2, red);
    if (gnorm_out && blockIdx.x == 0 && threadIdx.x == 0) *gnorm_out = cf.gnorm;
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    // the matrices that intersect this chunk: [j0, j1) -- block-uniform, almost always one
    const unsigned ulo = (unsigned)lo, uhi = (unsigned)(lo + CHUNK);
    int j0 = 0, j1 = 0;
    for (int j = 0; j < tab.count; ++j) {
        if (tab.first4[j] + tab.len4[j] <= ulo) j0 = j + 1;
        if (tab.first4[j] < uhi) j1 = j + 1;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long i = lo + threadIdx.x + THREADS * r;
        if (i >= n4) continue;
        float* pa = &pp[r].x; float* ma = &mm[r].x; float* va = &vv[r].x; const float* ga = &gg[r].x;
#pragma unroll
        for (int k = 0; k < 4; ++k) air_adam_update(pa[k], ma[k], va[k], ga[k], cf, omb1, omb2, eps);
        p4[i] = pp[r]; m4[i] = mm[r]; v4[i] = vv[r];
        const uint2 tw = make_uint2(air_pack_bf16(pp[r].x, pp[r].y), air_pack_bf16(pp[r].z, pp[r].w));
        bool keep_flat = true;
        for (int j = j0; j < j1; ++j) {                      // (uniform bounds)
            const unsigned rel = (unsigned)i - tab.first4[j];
            if ((unsigned)i >= tab.first4[j] && rel < tab.len4[j]) {
                *reinterpret_cast<uint2*>(panels + panel_at(tab, j, rel)) = tw;
                keep_flat = tab.excl[j] == 0u;
            }
        }
        if (shadow && keep_flat) reinterpret_cast<uint2*>(shadow)[i] = tw;
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
                   (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), twin_src, twin_dst, (long)twin_n};
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

static int fill_panels(const air_panel_t* panels, int count, int64_t n, PanelTab& tab) {
    if (!panels || count <= 0 || count > AIR_MAX_PANELS) return count > AIR_MAX_PANELS ? AIR_ELIMIT : AIR_EINVAL;
    tab.count = count;
    int64_t prev_end = 0;
    for (int j = 0; j < AIR_MAX_PANELS; ++j) {
        tab.first4[j] = 0xffffffffu; tab.len4[j] = 0; tab.N4[j] = 1; tab.K16[j] = 0; tab.R4[j] = 0; tab.excl[j] = 0;
        tab.magic[j] = 0; tab.dst[j] = 0;
        if (j >= count) continue;
        const air_panel_t& q = panels[j];
        if (q.K <= 0 || q.N <= 0 || q.src_off < prev_end || q.src_off + (int64_t)q.K * q.N > n || q.dst_off < 0) return AIR_EINVAL;
        if ((q.N & 3) || (q.src_off & 3) || (q.dst_off & 3)) return AIR_EALIGN;
        if (q.gates != 0 && (q.gates != 4 || (q.N & 15))) return AIR_EINVAL;
        if ((int64_t)q.K * 16 >= (1ll << 31) || n / 4 >= (1ll << 31)) return AIR_ELIMIT;
        // the multiply-shift row division (magic, below) is exact while rel * (magic * N4 - 2^40) < 2^40; rel < K * N4 and
        // the bracket is < N4, so K * N4 * N4 < 2^40 suffices (16 384 x 1 024: 2^30)
        if ((unsigned __int128)q.K * (unsigned)(q.N / 4) * (unsigned)(q.N / 4) >= ((unsigned __int128)1 << 40)) return AIR_ELIMIT;
        prev_end = q.src_off + (int64_t)q.K * q.N;
        tab.first4[j] = (unsigned)(q.src_off / 4); tab.len4[j] = (unsigned)((int64_t)q.K * q.N / 4);
        tab.N4[j] = (unsigned)(q.N / 4); tab.K16[j] = (unsigned)q.K * 16u;
        tab.R4[j] = q.gates == 4 ? (unsigned)(q.N / 16) : 0u;
        tab.excl[j] = q.exclusive ? 1u : 0u;
        tab.magic[j] = ((1ull << 40) + tab.N4[j] - 1) / tab.N4[j];
        tab.dst[j] = (unsigned long long)q.dst_off;
    }
    return 0;
}

extern "C" int air_panel_shadow(const float* params, uint16_t* panel_shadow, const air_panel_t* panels, int count, void* stream) {
    if (!params || !panel_shadow) return AIR_EINVAL;
    if (((uintptr_t)params & 15) != 0 || ((uintptr_t)panel_shadow & 7) != 0) return AIR_EALIGN;
    PanelTab tab;
    const int rc = fill_panels(panels, count, (int64_t)1 << 32, tab);   // (the flat buffer's length is not known here: offsets only have to fit 32-bit quad indices)
    if (rc) return rc;
    const long quads = (long)(tab.first4[count - 1] + tab.len4[count - 1]) - tab.first4[0];
    long blocks = (quads + THREADS - 1) / THREADS;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(panel_shadow_kernel, dim3((int)blocks), dim3(THREADS), 0, air_stream(stream), params, panel_shadow, tab);
    AIR_CHECK_LAUNCH();
    return 0;
}

static long adam_blocks(int64_t n) {
    long blocks = (n / 4 + THREADS - 1) / THREADS;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;        // (512 .. 8192 measured: 20.3 .. 23.8 us in isolation, no difference inside the step)
    return blocks;
}

extern "C" int air_adam_clip_step_panels(float* params, const float* grads, float* m, float* v, int64_t n,
                                         const float* partials, int npartials, const float* dyn, const int32_t* istate,
                                         float grad_prescale, float beta1, float beta2, float epsilon,
                                         uint16_t* bf16_shadow, const air_panel_t* panels, int npanels, uint16_t* panel_shadow,
                                         float* gnorm_out, void* stream) {
    if (!params || !grads || !m || !v || !partials || npartials <= 0 || !dyn || !istate || n <= 0 || !panel_shadow) return AIR_EINVAL;
    if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return AIR_EALIGN;
    if (((uintptr_t)bf16_shadow & 7) != 0 || ((uintptr_t)panel_shadow & 7) != 0) return AIR_EALIGN;
    PanelTab tab;
    const int rc = fill_panels(panels, npanels, n, tab);
    if (rc) return rc;
    const long chunks = (n / 4 + CHUNK - 1) / CHUNK;
    hipLaunchKernelGGL(adam_panels_kernel, dim3((int)(chunks < 1 ? 1 : chunks)), dim3(THREADS), 0, air_stream(stream),
                       params, grads, m, v, (long)n, partials, npartials, dyn, istate, grad_prescale, beta1, beta2,
                       epsilon, bf16_shadow, gnorm_out, panel_shadow, tab);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_adam_clip_step(float* params, const float* grads, float* m, float* v, int64_t n,
                                  const float* partials, int npartials, const float* dyn, const int32_t* istate,
                                  float grad_prescale, float beta1, float beta2, float epsilon,
                                  uint16_t* bf16_shadow, float* gnorm_out, void* stream) {
    if (!params || !grads || !m || !v || !partials || npartials <= 0 || !dyn || !istate || n <= 0) return AIR_EINVAL;
    if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return AIR_EALIGN;
    if (((uintptr_t)bf16_shadow & 7) != 0) return AIR_EALIGN;
    hipLaunchKernelGGL(adam_clip_kernel, dim3((int)adam_blocks(n)), dim3(THREADS), 0, air_stream(stream),
                       params, grads, m, v, (long)n, partials, npartials, dyn, istate, grad_prescale, beta1, beta2,
                       epsilon, bf16_shadow, gnorm_out);
    AIR_CHECK_LAUNCH();
    return 0;
}
