// Fused pointwise / reduction kernels of the AIR loop: LSTM gate math, VAE
// re-parameterisation, Bernoulli cross-entropy, batch means, bias gradients.
// All are HBM/latency bound; each reads and writes every byte once with
// coalesced accesses and reduces in a fixed order (deterministic).
#include "air_common.h"

namespace {

constexpr int THREADS = 256;

// BasicLSTMCell (TF 1.3): i, j, f, o = split(gates, 4, 1);
// c' = c*sigmoid(f + 1) + sigmoid(i)*tanh(j); h' = tanh(c')*sigmoid(o)   (air_model.py:286)
__global__ __launch_bounds__(THREADS) void lstm_gates_fwd_kernel(
    const float* __restrict__ gp, const float* __restrict__ c_prev, float* __restrict__ acts,
    float* __restrict__ c, float* __restrict__ h, int B, int R)
{
    const int idx = blockIdx.x * THREADS + threadIdx.x;
    if (idx >= B * R) return;
    const int b = idx / R, u = idx % R;
    const float* g = gp + (size_t)b * 4 * R;
    const float si = air_sigmoid(g[u]);
    const float tj = tanhf(g[R + u]);
    const float sf = air_sigmoid(g[2 * R + u] + 1.0f);
    const float so = air_sigmoid(g[3 * R + u]);
    const float cn = c_prev[idx] * sf + si * tj;
    float* a = acts + (size_t)b * 4 * R;
    a[u] = si; a[R + u] = tj; a[2 * R + u] = sf; a[3 * R + u] = so;
    c[idx] = cn;
    h[idx] = tanhf(cn) * so;
}

// The FIRST step of the recurrence: h_0 = c_0 = 0 (zero_state, air_model.py:540), so [x, h].K reduces to
// the hoisted x.Wx and no MatMul is needed.  Pre-activation = ((0 + slab_0) + slab_1 ...) + bias -- the
// summation order of the fused GEMM epilogue (AIR_EPI_LSTM_FWD) with a zero accumulator, bit for bit.
// NS > 0: the slab count is a compile-time constant (4 in the train step); NS == 0: run-time count up to 8
template <int NS>
__global__ __launch_bounds__(THREADS) void lstm_first_step_kernel(
    const float* __restrict__ slabs, int nslabs, long slab_stride, const float* __restrict__ bias,
    float* __restrict__ acts, float* __restrict__ c, float* __restrict__ h, unsigned short* __restrict__ h16, int B, int R)
{
    const int idx = blockIdx.x * THREADS + threadIdx.x;
    if (idx >= B * R) return;
    const int b = idx / R, u = idx % R;
    // all 4 x nslabs loads in flight before the first add (one memory round trip).  No control flow between them: with
    // `k < nslabs ? load : 0` and `bias ? load : 0` the compiler laid a branch around every load and an s_waitcnt
    // vmcnt(0) at several of the joins -- six round trips in a row in a kernel that has nothing else to do.  An absent slab /
    // bias is read from a valid stand-in address and masked to +0.0f by bit operations.
    float v[4][8], bj[4], g[4];
    const unsigned bmask = bias ? 0xffffffffu : 0u;
    const float* bsrc = bias ? bias : slabs;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const size_t o = (size_t)b * 4 * R + j * R + u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (NS > 0) v[j][k] = k < NS ? slabs[k * slab_stride + o] : 0.0f;
            else {
                const unsigned m = k < nslabs ? 0xffffffffu : 0u;
                v[j][k] = __uint_as_float(__float_as_uint(slabs[(k < nslabs ? k : 0) * slab_stride + o]) & m);
            }
        }
        bj[j] = __uint_as_float(__float_as_uint(bsrc[bias ? j * R + u : 0]) & bmask);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[j][k];          // slabs past nslabs add +0.0f: the fused epilogue does the same
        if (bias) s += bj[j];
        g[j] = s;
    }
    const float si = air_sigmoid(g[0]), tj = tanhf(g[1]);
    const float sf = air_sigmoid(g[2] + 1.0f), so = air_sigmoid(g[3]);
    const float cn = 0.0f * sf + si * tj;
    float* a = acts + (size_t)b * 4 * R;
    a[u] = si; a[R + u] = tj; a[2 * R + u] = sf; a[3 * R + u] = so;
    c[idx] = cn;
    const float hn = tanhf(cn) * so;
    h[idx] = hn;
    if (h16) h16[idx] = air_bf16_of(hn);
}

__global__ __launch_bounds__(THREADS) void lstm_gates_bwd_kernel(
    const float* __restrict__ dh, const float* __restrict__ dc_in, const float* __restrict__ acts,
    const float* __restrict__ c_prev, const float* __restrict__ c, float* __restrict__ dgates,
    float* __restrict__ dc_prev, float* __restrict__ dgsum, int dgsum_acc, int B, int R)
{
    const int idx = blockIdx.x * THREADS + threadIdx.x;
    if (idx >= B * R) return;
    const int b = idx / R, u = idx % R;
    const float* a = acts + (size_t)b * 4 * R;
    const float si = a[u], tj = a[R + u], sf = a[2 * R + u], so = a[3 * R + u];
    const float tc = tanhf(c[idx]);
    const float dhv = dh[idx];
    const float dc = (dc_in ? dc_in[idx] : 0.0f) + dhv * so * (1.0f - tc * tc);
    const float dgi = dc * tj * si * (1.0f - si);
    const float dgj = dc * si * (1.0f - tj * tj);
    const float dgf = dc * c_prev[idx] * sf * (1.0f - sf);
    const float dgo = dhv * tc * so * (1.0f - so);
    const size_t base = (size_t)b * 4 * R;
    dgates[base + u] = dgi; dgates[base + R + u] = dgj; dgates[base + 2 * R + u] = dgf; dgates[base + 3 * R + u] = dgo;
    dc_prev[idx] = dc * sf;
    if (dgsum) {
        if (dgsum_acc) {
            dgsum[base + u] += dgi; dgsum[base + R + u] += dgj; dgsum[base + 2 * R + u] += dgf; dgsum[base + 3 * R + u] += dgo;
        } else {
            dgsum[base + u] = dgi; dgsum[base + R + u] = dgj; dgsum[base + 2 * R + u] = dgf; dgsum[base + 3 * R + u] = dgo;
        }
    }
}

// vae.py:22-24: sample = mean + eps * sqrt(exp(log_var))
__global__ __launch_bounds__(THREADS) void reparam_fwd_kernel(
    const float* __restrict__ ml, const float* __restrict__ eps, float* __restrict__ zs, int B, int Z)
{
    const int idx = blockIdx.x * THREADS + threadIdx.x;
    if (idx >= B * Z) return;
    const int b = idx / Z, j = idx % Z;
    const float* r = ml + (size_t)b * 2 * Z;
    zs[idx] = r[j] + eps[idx] * sqrtf(expf(r[Z + j]));
}

// gradient of the sample + masked VAE KL (air_model.py:481-493) wrt mean | log_var
__global__ __launch_bounds__(THREADS) void reparam_bwd_kernel(
    const float* __restrict__ dzs, const float* __restrict__ ml, const float* __restrict__ eps,
    const float* __restrict__ att, const float* __restrict__ dyn, float* __restrict__ dml, int B, int Z)
{
    const int idx = blockIdx.x * THREADS + threadIdx.x;
    if (idx >= B * Z) return;
    const int b = idx / Z, j = idx % Z;
    const float* r = ml + (size_t)b * 2 * Z;
    const float klg = att[(size_t)b * AIR_ATT_STRIDE + AIR_ATT_MASK] * dyn[AIR_DYN_GRAD_SCALE];
    const float pv = dyn[AIR_DYN_VAE_PV], pm = dyn[AIR_DYN_VAE_PM];
    const float var = expf(r[Z + j]);
    const float sd = sqrtf(var);
    const float d = dzs[idx];
    dml[(size_t)b * 2 * Z + j] = d + klg * (r[j] - pm) / pv;
    dml[(size_t)b * 2 * Z + Z + j] = d * eps[idx] * 0.5f * sd + klg * 0.5f * (var / pv - 1.0f);
}

// air_model.py:580-593 + d loss / d running_recon; one workgroup per image
__global__ __launch_bounds__(THREADS) void bce_kernel(
    const float* __restrict__ images, const float* __restrict__ R, const float* __restrict__ dyn,
    float* __restrict__ recon, float* __restrict__ rec_loss, float* __restrict__ dR, int B, int D)
{
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float gsc = dyn[AIR_DYN_GRAD_SCALE];
    const size_t base = (size_t)b * D;
    float acc = 0.0f;
    for (int k = threadIdx.x; k < D; k += THREADS) {
        const float x = images[base + k], rr = R[base + k];
        const float r = fmaxf(fminf(rr, 1.0f), 0.0f);                   // clipped_rec :582
        const float p1 = r + AIR_EPS, p0 = (1.0f - r) + AIR_EPS;
        acc += x * logf(p1) + (1.0f - x) * logf(p0);                    // :586-589
        recon[base + k] = r;
        if (dR) {
            // Minimum/Maximum gradients pass at ties (LessEqual / GreaterEqual)
            const bool pass = (rr <= 1.0f) && (fminf(rr, 1.0f) >= 0.0f);
            dR[base + k] = pass ? -gsc * (x / p1 - (1.0f - x) / p0) : 0.0f;
        }
    }
    acc = air_block_sum_256(acc, red);
    if (threadIdx.x == 0) rec_loss[b] = -acc;
}

// loss = mean(L + rec_loss) :593,610; accuracy = mean(target == digits) :597-611
__global__ __launch_bounds__(THREADS) void finalize_kernel(
    const float* __restrict__ L, const float* __restrict__ rec_loss, const int32_t* __restrict__ targets,
    const int32_t* __restrict__ digits, float* __restrict__ loss_item, float* __restrict__ scalars, int B)
{
    __shared__ float red[4];
    float sl = 0.0f, sa = 0.0f;
    for (int b = threadIdx.x; b < B; b += THREADS) {
        const float l = L[b] + rec_loss[b];
        loss_item[b] = l;
        sl += l;
        sa += (targets[b] == digits[b]) ? 1.0f : 0.0f;
    }
    sl = air_block_sum_256(sl, red);
    sa = air_block_sum_256(sa, red);
    if (threadIdx.x == 0) { scalars[0] = sl / (float)B; scalars[1] = sa / (float)B; }
}

// column sums (BiasAdd_grad): blockIdx.y = problem, blockIdx.x = 64-column strip.
// 4 waves take interleaved rows; 8 independent loads in flight per thread.
struct ColsumTable { air_colsum_t p[16]; };
__global__ __launch_bounds__(THREADS) void colsum_kernel(ColsumTable tab)
{
    __shared__ float part[4][64];
    const air_colsum_t pr = tab.p[blockIdx.y];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int wave = threadIdx.x >> 6;
    if (blockIdx.x * 64 >= pr.cols) return;
    float acc = 0.0f;
    if (col < pr.cols) {
        int r = wave;
        for (; r + 28 < pr.rows; r += 32) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = pr.src[(size_t)(r + 4 * k) * pr.ld + col];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k];
        }
        for (; r < pr.rows; r += 4) acc += pr.src[(size_t)r * pr.ld + col];
    }
    part[wave][threadIdx.x & 63] = acc;
    __syncthreads();
    if (wave == 0 && col < pr.cols) {
        float v = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
        if (pr.accumulate) v += pr.dst[col];
        pr.dst[col] = v;
    }
}

// gradients of the 7 head output units; one workgroup per output unit, 4 waves on interleaved
// rows, 8 independent loads in flight per thread, fixed-order LDS reduction
__global__ __launch_bounds__(THREADS) void heads_out_wgrad_kernel(
    const float* __restrict__ d7, const float* __restrict__ hid, float* __restrict__ dw,
    float* __restrict__ db, int rows, int Hs, int Hh, int Hz, int ld)
{
    __shared__ float part[4][64];
    __shared__ float red[4];
    const int o = blockIdx.x;
    const int head[7] = {0, 1, 2, 2, 3, 3, 4};
    const int wid[5] = {Hs, Hs, Hh, Hh, Hz};
    int off = 0;
    for (int i = 0; i < head[o]; ++i) off += wid[i];
    const int HT = 2 * Hs + 2 * Hh + Hz;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j0 = 0; j0 < wid[head[o]]; j0 += 64) {
        const int j = j0 + lane;
        float acc = 0.0f;
        if (j < wid[head[o]]) {
            int r = wave;
            for (; r + 28 < rows; r += 32) {
                float a[8], h[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    a[k] = d7[(size_t)(r + 4 * k) * AIR_OUT_STRIDE + o];
                    h[k] = hid[(size_t)(r + 4 * k) * HT + off + j];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += a[k] * h[k];
            }
            for (; r < rows; r += 4) acc += d7[(size_t)r * AIR_OUT_STRIDE + o] * hid[(size_t)r * HT + off + j];
        }
        __syncthreads();
        part[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && j < wid[head[o]])
            dw[o * ld + j] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    }
    float s = 0.0f;
    for (int r = threadIdx.x; r < rows; r += THREADS) s += d7[(size_t)r * AIR_OUT_STRIDE + o];
    s = air_block_sum_256(s, red);
    if (threadIdx.x == 0) db[o] = s;
}

}  // namespace

extern "C" int air_lstm_gates_fwd(const float* gates_pre, const float* c_prev, float* acts,
                                  float* c, float* h, int B, int R, void* stream) {
    if (!gates_pre || !c_prev || !acts || !c || !h || B <= 0 || R <= 0) return AIR_EINVAL;
    hipLaunchKernelGGL(lstm_gates_fwd_kernel, dim3((B * R + THREADS - 1) / THREADS), dim3(THREADS), 0,
                       air_stream(stream), gates_pre, c_prev, acts, c, h, B, R);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_lstm_first_step(const float* xw_slabs, int nslabs, const float* bias, float* acts,
                                   float* c, float* h, uint16_t* h16, int B, int R, void* stream) {
    if (!xw_slabs || !acts || !c || !h || B <= 0 || R <= 0 || nslabs <= 0) return AIR_EINVAL;
    if (nslabs > 8) return AIR_ELIMIT;
    if (nslabs == 4)
        hipLaunchKernelGGL(lstm_first_step_kernel<4>, dim3((B * R + THREADS - 1) / THREADS), dim3(THREADS), 0,
                           air_stream(stream), xw_slabs, nslabs, (long)B * 4 * R, bias, acts, c, h, h16, B, R);
    else
        hipLaunchKernelGGL(lstm_first_step_kernel<0>, dim3((B * R + THREADS - 1) / THREADS), dim3(THREADS), 0,
                           air_stream(stream), xw_slabs, nslabs, (long)B * 4 * R, bias, acts, c, h, h16, B, R);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_lstm_gates_bwd(const float* dh, const float* dc_in, const float* acts,
                                  const float* c_prev, const float* c, float* dgates, float* dc_prev,
                                  float* dgsum, int dgsum_accumulate, int B, int R, void* stream) {
    if (!dh || !acts || !c_prev || !c || !dgates || !dc_prev || B <= 0 || R <= 0) return AIR_EINVAL;
    hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3((B * R + THREADS - 1) / THREADS), dim3(THREADS), 0,
                       air_stream(stream), dh, dc_in, acts, c_prev, c, dgates, dc_prev, dgsum,
                       dgsum_accumulate, B, R);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_reparam_fwd(const float* ml, const float* eps_z, float* zs, int B, int Z, void* stream) {
    if (!ml || !eps_z || !zs || B <= 0 || Z <= 0) return AIR_EINVAL;
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3((B * Z + THREADS - 1) / THREADS), dim3(THREADS), 0,
                       air_stream(stream), ml, eps_z, zs, B, Z);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_reparam_bwd(const float* d_zs, const float* ml, const float* eps_z, const float* att,
                               const float* dyn, float* d_ml, int B, int Z, void* stream) {
    if (!d_zs || !ml || !eps_z || !att || !dyn || !d_ml || B <= 0 || Z <= 0) return AIR_EINVAL;
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3((B * Z + THREADS - 1) / THREADS), dim3(THREADS), 0,
                       air_stream(stream), d_zs, ml, eps_z, att, dyn, d_ml, B, Z);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_bce_fwd_bwd(const float* images, const float* run_recon, const float* dyn,
                               float* recon, float* rec_loss, float* d_recon, int B, int D, void* stream) {
    if (!images || !run_recon || !dyn || !recon || !rec_loss || B <= 0 || D <= 0) return AIR_EINVAL;
    hipLaunchKernelGGL(bce_kernel, dim3(B), dim3(THREADS), 0, air_stream(stream),
                       images, run_recon, dyn, recon, rec_loss, d_recon, B, D);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_finalize(const float* run_loss, const float* rec_loss, const int32_t* targets,
                            const int32_t* digits, float* loss_per_item, float* scalars, int B, void* stream) {
    if (!run_loss || !rec_loss || !targets || !digits || !loss_per_item || !scalars || B <= 0) return AIR_EINVAL;
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(THREADS), 0, air_stream(stream),
                       run_loss, rec_loss, targets, digits, loss_per_item, scalars, B);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_colsum(const air_colsum_t* probs, int count, void* stream) {
    if (!probs || count <= 0) return AIR_EINVAL;
    if (count > 16) return AIR_ELIMIT;
    ColsumTable tab;
    int maxc = 0;
    for (int i = 0; i < count; ++i) {
        if (!probs[i].src || !probs[i].dst || probs[i].rows <= 0 || probs[i].cols <= 0) return AIR_EINVAL;
        tab.p[i] = probs[i];
        if (probs[i].cols > maxc) maxc = probs[i].cols;
    }
    for (int i = count; i < 16; ++i) tab.p[i] = probs[0];
    hipLaunchKernelGGL(colsum_kernel, dim3((maxc + 63) / 64, count), dim3(THREADS), 0, air_stream(stream), tab);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_heads_out_wgrad(const float* d_out7, const float* hid, float* dwout, float* dbout,
                                   int rows, int Hs, int Hh, int Hz, int wout_ld, void* stream) {
    if (!d_out7 || !hid || !dwout || !dbout || rows <= 0 || Hs <= 0 || Hh <= 0 || Hz <= 0) return AIR_EINVAL;
    hipLaunchKernelGGL(heads_out_wgrad_kernel, dim3(7), dim3(THREADS), 0, air_stream(stream),
                       d_out7, hid, dwout, dbout, rows, Hs, Hh, Hz, wout_ld);
    AIR_CHECK_LAUNCH();
    return 0;
}
