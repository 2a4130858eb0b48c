// Spatial-transformer kernels of the AIR loop: glimpse read (canvas -> window),
// canvas write (window -> canvas), their gradients, fused with the per-item
// head / sampling / KL / stop logic that surrounds them in the reference body
// (air_model.py:288-333, 351-439, 441-496).
//
// Geometry: theta is axis-aligned on this path (air_model.py:324-327, 353-356:
// the off-diagonals are zeros_like(s)), so source coordinates are separable:
// X depends only on the output column, Y only on the output row.  Each
// workgroup (one image) builds two small per-axis tap tables in LDS and then
// evaluates the reference's literal 4-product / add_n expression per pixel
// (transformer.py:108-116) -- same op order, no FMA contraction -- so that the
// out-of-range residues that later pass through log(r + 1e-9) match the
// reference arithmetic.  The backward of the write uses
// the exact adjoint in separable form (R_y^T g R_x) as a deterministic gather:
// no atomics anywhere.
#include "air_sampler_common.h"

AIR_STAMPS_READER(air_debug_stamps)

namespace {

// ---------------------------------------------------------------------------
// generic transformer (any theta): one thread per output pixel
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void transformer_fwd_kernel(
    const float* __restrict__ U, const float* __restrict__ theta, float* __restrict__ out,
    int B, int Hi, int Wi, int Ho, int Wo)
{
    const long total = (long)B * Ho * Wo;
    for (long p = (long)blockIdx.x * THREADS + threadIdx.x; p < total; p += (long)gridDim.x * THREADS) {
        const int b = (int)(p / (Ho * Wo));
        const int r = (int)(p % (Ho * Wo));
        const int i = r / Wo, j = r % Wo;
        const float* th = theta + (size_t)b * 6;
        const float xt = (Wo > 1) ? (-1.0f + (2.0f / (float)(Wo - 1)) * (float)j) : -1.0f;
        const float yt = (Ho > 1) ? (-1.0f + (2.0f / (float)(Ho - 1)) * (float)i) : -1.0f;
        const float xs = (th[0] * xt + th[1] * yt) + th[2] * 1.0f;
        const float ys = (th[3] * xt + th[4] * yt) + th[5] * 1.0f;
        const float X = ((xs + 1.0f) * ((float)Wi - 1.001f)) / 2.0f;
        const float Y = ((ys + 1.0f) * ((float)Hi - 1.001f)) / 2.0f;
        const float fx = floorf(X), fy = floorf(Y);
        const float x0 = fminf(fmaxf(fx, 0.f), (float)(Wi - 1)), x1 = fminf(fmaxf(fx + 1.f, 0.f), (float)(Wi - 1));
        const float y0 = fminf(fmaxf(fy, 0.f), (float)(Hi - 1)), y1 = fminf(fmaxf(fy + 1.f, 0.f), (float)(Hi - 1));
        const float* img = U + (size_t)b * Hi * Wi;
        const float Ia = img[(int)y0 * Wi + (int)x0], Ib = img[(int)y1 * Wi + (int)x0];
        const float Ic = img[(int)y0 * Wi + (int)x1], Id = img[(int)y1 * Wi + (int)x1];
        Tap tx{x1 - X, X - x0, 0, 0}, ty{y1 - Y, Y - y0, 0, 0};
        out[p] = bilinear4(tx, ty, Ia, Ib, Ic, Id);
    }
}

// head index -> (offset, width) inside the concatenated hidden vector
struct HeadSeg { int off[5]; int wid[5]; };
__device__ __forceinline__ HeadSeg head_segments(int Hs, int Hh, int Hz) {
    HeadSeg h;
    h.wid[0] = Hs; h.wid[1] = Hs; h.wid[2] = Hh; h.wid[3] = Hh; h.wid[4] = Hz;
    h.off[0] = 0;
    for (int i = 1; i < 5; ++i) h.off[i] = h.off[i - 1] + h.wid[i - 1];
    return h;
}
// output unit -> head, head -> (offset, width), as arithmetic: a table in constant memory indexed per lane, then a
// dynamically indexed HeadSeg (which the compiler keeps in scratch memory) were two dependent memory round trips on
// attend_fwd's critical path
__device__ __forceinline__ constexpr int out_head(int o) { return o < 2 ? o : (o < 4 ? 2 : (o < 6 ? 3 : 4)); }
__device__ __forceinline__ int head_wid(int Hs, int Hh, int Hz, int h) { return h < 2 ? Hs : (h < 4 ? Hh : Hz); }
__device__ __forceinline__ int head_off(int Hs, int Hh, int h) {
    return h == 0 ? 0 : h == 1 ? Hs : h == 2 ? 2 * Hs : h == 3 ? 2 * Hs + Hh : 2 * Hs + 2 * Hh;
}

// air_model.py:443-447 for one element
__device__ __forceinline__ float gauss_kl_term(float plv, float lv, float var, float pv, float mean, float pm) {
    const float d = mean - pm;
    return (((plv - lv) - 1.0f) + var / pv) + (d * d) / pv;
}

// ---------------------------------------------------------------------------
// attend forward, TIME-BATCHED: one workgroup per (image, time step).
// The LSTM input is the same image at every step (air_model.py:286, 535), so h'
// of all N steps exists before any head runs; the only cross-step dependence
// left is the scalar stopping sum S, which block (b, t) re-derives from the
// z_pres of the earlier steps with the identical op sequence (bitwise equal).
// ---------------------------------------------------------------------------
constexpr int MAX_STEPS = 16;

// concrete.py:20-27 + air_model.py:385-390: pre-sigmoid sample and z_pres
__device__ __forceinline__ float concrete_presigmoid(float lo, float u, float T) {
    const float noise = logf(u + AIR_EPS) - logf((1.0f - u) + AIR_EPS);
    return (lo + noise) / T;
}

__global__ __launch_bounds__(THREADS) void attend_fwd_kernel(air_attend_fwd_t a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, t = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.C, w = a.w, B = a.B;
    const HeadSeg hs = head_segments(a.Hs, a.Hh, a.Hz);
    const int HT = hs.off[4] + hs.wid[4];

    float* sh_out = smem;                       // [8]
    float* sh_sc = smem + 8;                    // [8]: s, x, y
    float* sh_zlo = smem + 16;                  // [MAX_STEPS] z log-odds of the earlier steps
    Tap* sh_tx = reinterpret_cast<Tap*>(smem + 16 + MAX_STEPS);   // [w]
    Tap* sh_ty = sh_tx + w;                              // [w]
    int* sh_box = reinterpret_cast<int*>(sh_ty + w);     // [4]: x_lo, x_hi, y_lo, y_hi
    float* sh_hid = reinterpret_cast<float*>(sh_box + 4);   // [HT]
    float* sh_wout = sh_hid + ((HT + 3) & ~3);              // [7][wout_ld] output-unit weights
    float* sh_hprev = sh_wout + 7 * a.wout_ld;              // [t][Hz] z_pres hidden segment of the earlier steps
    float* sh_img = sh_hprev + MAX_STEPS * hs.wid[4];       // [C*C] canvas

    const size_t row = (size_t)t * B + b;
    AIR_STAMP(10);
    // every global load that does not depend on the sampled (s, x, y) is issued up front: the
    // canvas (consumed last) rides under the head computation instead of following it
    const float* img = a.canvas + (size_t)b * C * C;
    constexpr int PF = 10;                                   // 2560 floats per pass: the 50x50 canvas in one
    // a canvas beyond one pass (stress config: 128x128 = 64 KB per workgroup) is not prefetched whole:
    // only the bounding box of the glimpse is read, once (s, x, y) are known
    const bool whole = C * C <= PF * THREADS;
    float pf[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) { const int p = tid + k * THREADS; pf[k] = (whole && p < C * C) ? img[p] : 0.0f; }
    // the sampling / KL section below is spread over the lanes of wave 0: lanes 0..2 take the three Gaussian
    // heads (scale, shift x, shift y), lane 3 the Concrete z_pres of this step, lanes 8.. the z_pres of the
    // earlier steps (for the stopping sum); each lane fetches its own noise
    // (ONE load through a selected address: an if / else-if chain made wave 0 walk five branches, each with its own load
    // and s_waitcnt -- five memory round trips in front of the workgroup's first barrier)
    float in_eps = 0.0f, in_u = 0.5f;
    {
        const bool prev = tid >= 8 && tid < 8 + t;
        const float* nsrc = tid == 0 ? a.eps_scale + row : tid == 1 ? a.eps_shift + 2 * row : tid == 2 ? a.eps_shift + 2 * row + 1
                          : prev ? a.u + (size_t)(tid - 8) * B + b : a.u + row;
        const float nv = *nsrc;
        in_eps = tid < 3 ? nv : 0.0f;
        in_u = (tid == 3 || prev) ? nv : 0.5f;
    }
    for (int j = tid; j < HT; j += THREADS) sh_hid[j] = a.hid[row * HT + j];
    for (int j = tid; j < 7 * a.wout_ld; j += THREADS) sh_wout[j] = a.wout[j];
    for (int j = tid; j < t * hs.wid[4]; j += THREADS) {
        const int tp = j / hs.wid[4], jj = j % hs.wid[4];
        sh_hprev[j] = a.hid[((size_t)tp * B + b) * HT + hs.off[4] + jj];
    }
    // one 16-lane group per dot product: group (wave, lane >> 4) owns dots grp0, grp0 + 16, ...
    const int gl = lane & 15, grp0 = wave * 4 + (lane >> 4);
    const float bo = (gl == 0 && grp0 < 7) ? a.bout[grp0] : 0.0f;
    const float bz = a.bout[6];
    if (whole) {
#pragma unroll
        for (int k = 0; k < PF; ++k) { const int p = tid + k * THREADS; if (p < C * C) sh_img[p] = pf[k]; }
    }
    __syncthreads();

    AIR_STAMP(11);
    // 7 output units (air_model.py:294,299,311,316,376): x.W + b
    // 7 output units (air_model.py:294,299,311,316,376: x.W + b) and the z log-odds of the earlier
    // steps t' < t, ALL in one pass: 16 lanes per dot product, 4-step butterfly.  The z unit of this
    // step and of the earlier steps use the same lane assignment and reduction order, so a step's
    // z is bit-identical wherever it is recomputed.
    for (int dot = grp0; dot < 7 + t; dot += 16) {
        float p = 0.0f;
        const int tp = dot - 7;
        if (dot < 7) {
            const int h = out_head(dot);
            const int hw = head_wid(a.Hs, a.Hh, a.Hz, h), ho = head_off(a.Hs, a.Hh, h);
            for (int j = gl; j < hw; j += 16) p += sh_hid[ho + j] * sh_wout[dot * a.wout_ld + j];
        } else {
            const float* hp = sh_hprev + tp * hs.wid[4];
            for (int j = gl; j < hs.wid[4]; j += 16) p += hp[j] * sh_wout[6 * a.wout_ld + j];
        }
        p += __shfl_xor(p, 8, 64); p += __shfl_xor(p, 4, 64); p += __shfl_xor(p, 2, 64); p += __shfl_xor(p, 1, 64);
        if (gl == 0) {
            if (dot < 7) sh_out[dot] = p + bo;
            else sh_zlo[tp] = p + bz;
        }
    }
    __syncthreads();

    AIR_STAMP(12);
    if (tid < 64) {
        const float* dyn = a.dyn;
        const float T = dyn[AIR_DYN_TEMPERATURE], thr = dyn[AIR_DYN_STOP_THRESHOLD];
        // lanes 0..2 -- scale :300-303, shift :317-320 (_sample_from_mvn :123-128) and their KL terms :441-477:
        // one instruction stream, per-lane operands
        const float mu = lane == 0 ? sh_out[0] : lane == 1 ? sh_out[2] : sh_out[3];
        const float lv = lane == 0 ? sh_out[1] : lane == 1 ? sh_out[4] : sh_out[5];
        const float var = expf(lv);
        const float pre_act = mu + in_eps * sqrtf(var);
        const float act = lane == 0 ? air_sigmoid(pre_act) : tanhf(pre_act);
        const float term = lane == 0
            ? gauss_kl_term(dyn[AIR_DYN_SCALE_PLV], lv, var, dyn[AIR_DYN_SCALE_PV], mu, dyn[AIR_DYN_SCALE_PM])
            : gauss_kl_term(dyn[AIR_DYN_SHIFT_PLV], lv, var, dyn[AIR_DYN_SHIFT_PV], mu, dyn[AIR_DYN_SHIFT_PM]);
        // lane 3 (this step) and lanes 8 + t' (earlier steps) -- Concrete sample :377-390, concrete.py:20-27
        const float z_lo = sh_out[6];
        const float lo = lane == 3 ? z_lo : (lane >= 8 && lane - 8 < t) ? sh_zlo[lane - 8] : 0.0f;
        const float ypre_l = concrete_presigmoid(lo, in_u, T);
        float z_l = air_sigmoid(ypre_l);
        if (!a.train) z_l = rintf(z_l);                   // tf.round (half-to-even) :389-390
        // concrete.py:30-43 (prior and posterior temperatures are both T :403-407); meaningful on lane 3
        const float plo = dyn[AIR_DYN_PRIOR_LOG_ODDS];
        const float yT = ypre_l * T;
        const float log_prior = ((logf(T + AIR_EPS) - yT) + plo) - 2.0f * logf((1.0f + expf(-yT + plo)) + AIR_EPS);
        const float log_post = ((logf(T + AIR_EPS) - yT) + lo) - 2.0f * logf((1.0f + expf(-yT + lo)) + AIR_EPS);
        const float kl_z_l = log_post - log_prior;
        // gather (uniform broadcasts)
        const float s = __shfl(act, 0, 64), x = __shfl(act, 1, 64), y = __shfl(act, 2, 64);
        const float kl_s = 0.5f * __shfl(term, 0, 64);
        const float kl_h = 0.5f * (__shfl(term, 1, 64) + __shfl(term, 2, 64));
        const float ypre = __shfl(ypre_l, 3, 64), z = __shfl(z_l, 3, 64), kl_z = __shfl(kl_z_l, 3, 64);
        // stopping sum on entry to step t (air_model.py:424): S += 1 - z_pres, in step order
        float S = 0.0f;
        for (int tp = 0; tp < t; ++tp) S = S + (1.0f - __shfl(z_l, 8 + tp, 64));
        // stop logic :409-427
        const bool mask_prev = S < thr;
        S = S + (1.0f - z);
        const bool mask = S < thr;
        if (tid == 0) {
            const float zprob = air_sigmoid(z_lo);
            float* o7 = a.out7 + row * AIR_OUT_STRIDE;
            for (int o = 0; o < 7; ++o) o7[o] = sh_out[o];
            o7[7] = 0.0f;
            float* at = a.att + row * AIR_ATT_STRIDE;
            at[AIR_ATT_S] = s; at[AIR_ATT_X] = x; at[AIR_ATT_Y] = y;
            at[AIR_ATT_ZPRE] = ypre; at[AIR_ATT_Z] = z; at[AIR_ATT_ZPROB] = zprob;
            at[AIR_ATT_KL_Z] = kl_z; at[AIR_ATT_KL_SCALE] = kl_s; at[AIR_ATT_KL_SHIFT] = kl_h;
            at[AIR_ATT_KL_VAE] = 0.0f;
            at[AIR_ATT_MASK_PREV] = mask_prev ? 1.0f : 0.0f;
            at[AIR_ATT_MASK] = mask ? 1.0f : 0.0f;
            // theta_recon :353-356
            at[AIR_ATT_ST_BACK + 0] = 1.0f / s;
            at[AIR_ATT_ST_BACK + 1] = (-x) / s;
            at[AIR_ATT_ST_BACK + 2] = (-y) / s;
            at[AIR_ATT_ST_BACK + 3] = 0.0f;
            sh_sc[0] = s; sh_sc[1] = x; sh_sc[2] = y;
        }
    }
    __syncthreads();

    AIR_STAMP(13);
    // ST read :322-333 -- theta = [[s,0,x],[0,s,y]]
    const float s = sh_sc[0], sx = sh_sc[1], sy = sh_sc[2];
    if (tid < w) sh_tx[tid] = axis_tap(tid, w, C, s, sx);
    else if (tid >= 64 && tid < 64 + w) sh_ty[tid - 64] = axis_tap(tid - 64, w, C, s, sy);
    __syncthreads();
    // a canvas beyond one prefetch pass (128x128 = 64 KB) is NOT staged in LDS: the glimpse's 4 x w*w taps are
    // gathered straight from memory -- they fall into a box of ~(s*C)^2 pixels that stays in the CU's L1, and without the
    // C*C floats of LDS five workgroups share a CU instead of one (stress configuration: 48 -> 12 us for 1280 of them)
    AIR_STAMP(14);
    float* win = a.window + row * w * w;
    unsigned short* win16 = a.window16 ? a.window16 + row * w * w : nullptr;      // bf16 twin: A operand of the first recognition GEMM
    // (the loop is instantiated per address space behind ONE uniform branch: an LDS and a global pointer must not meet
    // in a flat pointer)
    auto read_loop = [&](auto staged_t) __attribute__((always_inline)) {
        constexpr bool STAGED = decltype(staged_t)::value;
        for (int p = tid; p < w * w; p += THREADS) {
            const int i = p / w, j = p % w;
            const Tap tx = sh_tx[j], ty = sh_ty[i];
            const int r0 = ty.i0 * C, r1 = ty.i1 * C;
            const float v = STAGED ? bilinear4(tx, ty, sh_img[r0 + tx.i0], sh_img[r1 + tx.i0], sh_img[r0 + tx.i1], sh_img[r1 + tx.i1])
                                   : bilinear4(tx, ty, img[r0 + tx.i0], img[r1 + tx.i0], img[r0 + tx.i1], img[r1 + tx.i1]);
            win[p] = v;
            if (win16) win16[p] = air_bf16_of(v);
        }
    };
    if (whole) read_loop(std::true_type{}); else read_loop(std::false_type{});
    AIR_STAMP(15);
}

// ---------------------------------------------------------------------------
// attend backward: ST-read gradient wrt (s,x,y) + sampling / KL / head-output
// gradients.  One workgroup per (image, time step).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS) void attend_bwd_kernel(air_attend_bwd_t a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    const int C = a.C, w = a.w;
    const HeadSeg hs = head_segments(a.Hs, a.Hh, a.Hz);
    const int HT = hs.off[4] + hs.wid[4];
    const size_t row = (size_t)t * a.B + b;

    float* sh_red = smem;                                // [16]
    float* sh_d = smem + 16;                             // [8] d_out7
    Tap* sh_tx = reinterpret_cast<Tap*>(smem + 24);
    Tap* sh_ty = sh_tx + w;
    float* sh_t = reinterpret_cast<float*>(sh_ty + w);   // [w] linspace values
    int* sh_box = reinterpret_cast<int*>(sh_t + w);
    float* sh_img = reinterpret_cast<float*>(sh_box + 4);

    // all global loads up front (one memory round trip): the canvas, this thread's share of
    // d_window, and thread 0's scalars for the head-output gradients at the end
    const float* at = a.att + row * AIR_ATT_STRIDE;
    const float* img = a.canvas + (size_t)b * C * C;
    const float* g = a.d_window + row * w * w;
    constexpr int PF = 10, GF = 4;
    const bool whole = C * C <= PF * THREADS;            // larger canvases: bounding box only (see attend_fwd)
    float pf[PF], gf[GF];
#pragma unroll
    for (int k = 0; k < PF; ++k) { const int p = tid + k * THREADS; pf[k] = (whole && p < C * C) ? img[p] : 0.0f; }
#pragma unroll
    for (int k = 0; k < GF; ++k) { const int p = tid + k * THREADS; gf[k] = p < w * w ? g[p] : 0.0f; }
    float pre_o7[8], pre_dw[4], pre_e[3], pre_dyn[AIR_DYN_COUNT], pre_at[AIR_ATT_STRIDE];
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) pre_o7[k] = a.out7[row * AIR_OUT_STRIDE + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) pre_dw[k] = a.d_sxy_write[row * 4 + k];
        pre_e[0] = a.eps_scale[row]; pre_e[1] = a.eps_shift[2 * row]; pre_e[2] = a.eps_shift[2 * row + 1];
#pragma unroll
        for (int k = 0; k < AIR_DYN_COUNT; ++k) pre_dyn[k] = a.dyn[k];
#pragma unroll
        for (int k = 0; k < AIR_ATT_STRIDE; ++k) pre_at[k] = at[k];
    }
    const float s = at[AIR_ATT_S], sx = at[AIR_ATT_X], sy = at[AIR_ATT_Y];
    if (tid < w) { float tv; sh_tx[tid] = axis_tap(tid, w, C, s, sx, &tv); sh_t[tid] = tv; }
    else if (tid >= 64 && tid < 64 + w) sh_ty[tid - 64] = axis_tap(tid - 64, w, C, s, sy);
    if (whole) {
#pragma unroll
        for (int k = 0; k < PF; ++k) { const int p = tid + k * THREADS; if (p < C * C) sh_img[p] = pf[k]; }
    }
    __syncthreads();
    // d out / dX = (Ic-Ia)(y1-Y) + (Id-Ib)(Y-y0);  d out / dY = (Ib-Ia)(x1-X) + (Id-Ic)(X-x0)
    const float half_c = ((float)C - 1.001f) / 2.0f;     // dX/dx_s
    float ds = 0.f, dx = 0.f, dy = 0.f;
    // large canvases: the taps are gathered from memory (see attend_fwd).  The loop is instantiated per address space
    // behind ONE uniform branch: an LDS and a global pointer must not meet in a flat pointer
    auto pixel_loop = [&](auto staged_t) __attribute__((always_inline)) {
        constexpr bool STAGED = decltype(staged_t)::value;
        for (int p = tid, k = 0; p < w * w; p += THREADS, ++k) {
            const int i = p / w, j = p % w;
            const Tap tx = sh_tx[j], ty = sh_ty[i];
            const int r0 = ty.i0 * C, r1 = ty.i1 * C;
            float Ia, Ib, Ic, Id;
            if (STAGED) { Ia = sh_img[r0 + tx.i0]; Ib = sh_img[r1 + tx.i0]; Ic = sh_img[r0 + tx.i1]; Id = sh_img[r1 + tx.i1]; }
            else { Ia = img[r0 + tx.i0]; Ib = img[r1 + tx.i0]; Ic = img[r0 + tx.i1]; Id = img[r1 + tx.i1]; }
            const float gv = k < GF ? gf[k < GF ? k : 0] : g[p];
            float gX, gY;
            if (a.literal) graph_dxy(gv, Ia, Ib, Ic, Id, tx, ty, (float)C - 1.001f, gX, gY);
            else {
                gX = gv * ((Ic - Ia) * ty.w0 + (Id - Ib) * ty.w1) * half_c;
                gY = gv * ((Ib - Ia) * tx.w0 + (Id - Ic) * tx.w1) * half_c;
            }
            ds += gX * sh_t[j] + gY * sh_t[i];
            dx += gX;
            dy += gY;
        }
    };
    if (whole) pixel_loop(std::true_type{}); else pixel_loop(std::false_type{});
    {
        float r4[4] = {ds, dx, dy, 0.0f};
        air_block_sum4<THREADS / 64>(r4, sh_red);
        ds = r4[0]; dx = r4[1]; dy = r4[2];
    }

    if (tid == 0) {
        const float* dyn = pre_dyn;                              // preloaded at kernel entry
        const float gsc = dyn[AIR_DYN_GRAD_SCALE];               // d loss / d per-item loss
        const float T = dyn[AIR_DYN_TEMPERATURE];
        const float* o7 = pre_o7;
        const float* dw = pre_dw;
        const float* at = pre_at;
        const float mask = at[AIR_ATT_MASK], mask_prev = at[AIR_ATT_MASK_PREV];
        const float d_s = ds + dw[0], d_x = dx + dw[1], d_y = dy + dw[2], d_z = dw[3];
        const float lv_s = o7[1], lv_x = o7[4], lv_y = o7[5];
        const float sd_s = sqrtf(expf(lv_s)), sd_x = sqrtf(expf(lv_x)), sd_y = sqrtf(expf(lv_y));
        const float pv_s = dyn[AIR_DYN_SCALE_PV], pm_s = dyn[AIR_DYN_SCALE_PM];
        const float pv_h = dyn[AIR_DYN_SHIFT_PV], pm_h = dyn[AIR_DYN_SHIFT_PM];
        const float klg = mask * gsc;
        // s = sigmoid(mu + eps*sd), (x,y) = tanh(mu + eps*sd), sd = sqrt(exp(lv))
        const float da_s = d_s * s * (1.0f - s);
        const float da_x = d_x * (1.0f - sx * sx), da_y = d_y * (1.0f - sy * sy);
        const float e_s = pre_e[0], e_x = pre_e[1], e_y = pre_e[2];
        sh_d[0] = da_s + klg * (o7[0] - pm_s) / pv_s;
        sh_d[1] = da_s * e_s * 0.5f * sd_s + klg * 0.5f * (sd_s * sd_s / pv_s - 1.0f);
        sh_d[2] = da_x + klg * (o7[2] - pm_h) / pv_h;
        sh_d[3] = da_y + klg * (o7[3] - pm_h) / pv_h;
        sh_d[4] = da_x * e_x * 0.5f * sd_x + klg * 0.5f * (sd_x * sd_x / pv_h - 1.0f);
        sh_d[5] = da_y * e_y * 0.5f * sd_y + klg * 0.5f * (sd_y * sd_y / pv_h - 1.0f);
        // z = sigmoid(ypre), ypre = (lo + noise)/T; kl_z = log q(ypre; lo) - log p(ypre; plo)
        const float z = at[AIR_ATT_Z], ypre = at[AIR_ATT_ZPRE], z_lo = o7[6];
        const float plo = dyn[AIR_DYN_PRIOR_LOG_ODDS];
        const float eq = expf(-ypre * T + z_lo), ep = expf(-ypre * T + plo);
        const float rq = eq / ((1.0f + eq) + AIR_EPS), rp = ep / ((1.0f + ep) + AIR_EPS);
        const float dkl = mask_prev * gsc;
        // d logq/dy = -T + 2T rq ; d logp/dy = -T + 2T rp ; d logq/dlo = 1 - 2 rq
        const float d_ypre = d_z * z * (1.0f - z) + dkl * (2.0f * T * (rq - rp));
        sh_d[6] = d_ypre / T + dkl * (1.0f - 2.0f * rq);
        sh_d[7] = 0.0f;
        float* d7 = a.d_out7 + row * AIR_OUT_STRIDE;
        for (int o = 0; o < 8; ++o) d7[o] = sh_d[o];
    }
    __syncthreads();
    // back through the 7 output units and the hidden ReLU
    for (int j = tid; j < HT; j += THREADS) {
        const int h = (j >= a.Hs) + (j >= 2 * a.Hs) + (j >= 2 * a.Hs + a.Hh) + (j >= 2 * a.Hs + 2 * a.Hh);
        const int jj = j - head_off(a.Hs, a.Hh, h);
        // all seven weights and the activation fetched unconditionally (jj < wout_ld for every unit), the units of other
        // heads masked to an exact +0 term: the conditional form was one load + s_waitcnt per unit, one after the other
        float wv[7];
#pragma unroll
        for (int o = 0; o < 7; ++o) wv[o] = a.wout[o * a.wout_ld + jj];
        const float hv = a.hid[row * HT + j];
        float v = 0.0f;
#pragma unroll
        for (int o = 0; o < 7; ++o) {
            const float term = sh_d[o] * wv[o];
            v += (out_head(o) == h) ? term : 0.0f;
        }
        const float dv = (hv > 0.0f) ? v : 0.0f;
        a.d_hid[row * HT + j] = dv;
        if (a.d_hid16) a.d_hid16[row * HT + j] = air_bf16_of(dv);
    }
}

// ---------------------------------------------------------------------------
// compose ("write" forward, time-batched): for one image, all N steps of
// window -> canvas (x z_pres, masked, accumulated IN STEP ORDER in registers --
// the canvas never round-trips through HBM), the per-step VAE KL, the running
// loss in the reference's exact summation order, the digit count, and the
// reconstruction loss + its gradient (air_model.py:351-366, 409-439, 479-496,
// 580-593).  One workgroup per image.
// ---------------------------------------------------------------------------
// 16 waves per image: the pixel loop is a chain of dependent LDS gathers and two logf per pixel; 4 waves per SIMD hide
// that latency.  ONE workgroup of 1024 threads per image.  (An image as 2 or 4 workgroups -- 256 workgroups at B = 64,
// the per-image sums finished by the next launch, bit-identical -- was built and measured in round 4: 7.7 us as one
// workgroup, 8.5 in 4 bands, 7.7 in 2; 31 -> 35.5 us at 128 x 128: every band repeats the set-up loads.  Removed.)
constexpr int CF_THREADS = 1024;

// LONGEST-FIRST ORDER of the (image, step) items for the graph-order write backward (air_write_fwd_t.wb_order), computed
// by ONE extra workgroup of the compose launch.  Why: the backward's workgroups differ by two orders of magnitude in work
// -- an inactive item returns at once, an item whose glimpse is small pushes up to 4 C^2 corner terms through its CU's
// LDS atomic pipe -- and with several workgroups per CU (128 x 128: 1280 items on 256 CUs) the hardware hands them out
// in launch order: per-CU stamps (tools/wb_wg_stamps.py --stress) showed 68 k corner terms per CU on average but 218 k on
// the busiest, which then IS the launch (282 us against a mean CU busy time of 138).  Greedy dispatch of a list sorted
// by decreasing cost is the classic 4/3-optimal schedule.  Cost of an item = the corner terms it will accumulate (exact:
// the out-of-range columns x rows of its write transformer, x 4 taps) + a constant for its fixed phases; 0 when
// inactive.  The order is a permutation: which workgroup computes an item changes nothing in the results.
constexpr int WB_ORDER_MAX = 4096;
constexpr int WB_COST_SHIFT = 10, WB_BUCKETS = 96;          // cost classes of 1024 terms (~2 us of atomic pipe each)
__device__ __forceinline__ void wb_order_block(const air_write_fwd_t& a, unsigned* lds /* >= 4096 words */) {
    // A COUNTING sort by cost class (a full bitonic sort of 2048 keys took this one workgroup 40 us -- longer than the
    // compose launch it rides in): class histogram with LDS atomics, prefix over the classes from the most expensive
    // down, scatter.  Within a class the order is whatever the atomics deliver -- it only decides which workgroup
    // computes which item, never a result.
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int NB = a.N * a.B, C = a.C, w = a.w;
    int* hist = reinterpret_cast<int*>(lds);                // [WB_BUCKETS]
    int* basep = hist + WB_BUCKETS;                         // [WB_BUCKETS]
    for (int i = tid; i < 2 * WB_BUCKETS; i += nthreads) hist[i] = 0;
    __syncthreads();
    constexpr int PER = (WB_ORDER_MAX + CF_THREADS - 1) / CF_THREADS;      // items per thread, at most
    int cls[PER], pos[PER];
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        const int it = tid + r * nthreads;
        cls[r] = -1; pos[r] = 0;
        if (it < NB) {
            const float* at = a.att + (size_t)it * AIR_ATT_STRIDE;
            unsigned cost = 0u;
            if (at[AIR_ATT_MASK] != 0.0f) {
                const float s = at[AIR_ATT_S], x = at[AIR_ATT_X], y = at[AIR_ATT_Y];
                const float ia = 1.0f / s, bx = (-x) / s, by = (-y) / s;
                // canvas columns / rows whose two taps clip to one index: X < 0 or X >= w - 1 (axis_tap).  X is monotone in
                // the canvas coordinate (1 / s > 0): both counts are binary searches with axis_tap's own arithmetic
                auto oob = [&](float bb) {
                    auto Xof = [&](int j) {
                        const float step = 2.0f / (float)(C - 1);
                        const float tt = -1.0f + step * (float)j;
                        const float xs = ia * tt + bb;
                        return ((xs + 1.0f) * ((float)w - 1.001f)) / 2.0f;
                    };
                    int lo = 0, hi = C;                         // first j with X(j) >= 0
                    while (lo < hi) { const int m = (lo + hi) >> 1; if (Xof(m) >= 0.0f) hi = m; else lo = m + 1; }
                    const int below = lo;
                    lo = 0; hi = C;                             // first j with X(j) >= w - 1
                    while (lo < hi) { const int m = (lo + hi) >> 1; if (Xof(m) >= (float)(w - 1)) hi = m; else lo = m + 1; }
                    return below + (C - lo);
                };
                cost = 4u * (unsigned)oob(bx) * (unsigned)oob(by) + 12000u;   // + ~25 us of term / chain / set-up phases at 4 cycles per term
            }
            // class 0 = the most expensive; inactive items (cost 0) in the last class
            const int c = cost ? max(0, WB_BUCKETS - 2 - (int)(cost >> WB_COST_SHIFT)) : WB_BUCKETS - 1;
            cls[r] = c;
            pos[r] = atomicAdd(&hist[c], 1);
        }
    }
    __syncthreads();
    if (tid == 0) { int run = 0; for (int c = 0; c < WB_BUCKETS; ++c) { basep[c] = run; run += hist[c]; } }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER; ++r)
        if (cls[r] >= 0) a.wb_order[basep[cls[r]] + pos[r]] = tid + r * nthreads;
}

template <int NT>
__global__ __launch_bounds__(NT) void write_fwd_kernel(air_write_fwd_t a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (a.wb_order && (int)blockIdx.x == a.B) {                  // the one extra workgroup of the launch (block-uniform)
        wb_order_block(a, reinterpret_cast<unsigned*>(smem));
        return;
    }
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int vt = tid;
    const int C = a.C, w = a.w, Z = a.Z, N = a.N, B = a.B;
    float* sh_red = smem;                                    // [16]
    float* sh_z = smem + 16;                                 // [MAX_STEPS] z_pres
    int* sh_act = reinterpret_cast<int*>(smem + 16 + MAX_STEPS);         // [MAX_STEPS]
    float* sh_kl = smem + 16 + 2 * MAX_STEPS;                            // [MAX_STEPS] VAE KL per step
    float* sh_rec = smem + 16 + 3 * MAX_STEPS;                           // [MAX_STEPS][4]: mask_prev, KL z, KL scale, KL shift
    Tap* sh_tx = reinterpret_cast<Tap*>(smem + 16 + 7 * MAX_STEPS);      // [N][C]
    Tap* sh_ty = sh_tx + (size_t)N * C;                                    // [N][C]
    float* sh_win = reinterpret_cast<float*>(sh_ty + (size_t)N * C);      // [N][w*w]
    const float* dyn = a.dyn;
    const size_t base = (size_t)b * C * C;
    const int CC = C * C;

    // the image pixels of this thread do not depend on anything computed here: on small canvases (<= 4 per thread) their
    // loads go out with phase A's -- one memory round trip instead of two
    constexpr int XPRE = 4;
    const bool xpre = CC <= XPRE * CF_THREADS;
    float xs[XPRE];
#pragma unroll
    for (int k = 0; k < XPRE; ++k) { const int p = vt + k * CF_THREADS; xs[k] = a.images[base + ((xpre && p < CC) ? p : 0)]; }

    // phase A -- everything that only needs the per-step records, for all steps at once
    // (independent loads: one memory round trip instead of one per step)
    {
        for (int t = wave; t < N; t += NT / 64) {
            // VAE KL :479-493 (a public output whether or not the item is still active): one wave per step
            const float* ml = a.ml + ((size_t)t * B + b) * 2 * Z;
            const float pv = dyn[AIR_DYN_VAE_PV], pm = dyn[AIR_DYN_VAE_PM], plv = dyn[AIR_DYN_VAE_PLV];
            float klt = 0.0f;
            for (int j = lane; j < Z; j += 64) {
                const float lv = ml[Z + j];
                klt += gauss_kl_term(plv, lv, expf(lv), pv, ml[j], pm);
            }
            klt = air_wave_sum(klt);
            if (lane == 0) sh_kl[t] = 0.5f * klt;
        }
    }
    for (int it = tid; it < N * C; it += NT) {
        const int t = it / C, j = it % C;
        const float* at = a.att + ((size_t)t * B + b) * AIR_ATT_STRIDE;
        // theta_recon :353-356
        const float s = at[AIR_ATT_S], x = at[AIR_ATT_X], y = at[AIR_ATT_Y];
        const float ia = 1.0f / s, bx = (-x) / s, by = (-y) / s;
        sh_tx[it] = axis_tap(j, C, w, ia, bx);
        sh_ty[it] = axis_tap(j, C, w, ia, by);
    }
    for (int it = tid; it < N * w * w; it += NT) {
        const int t = it / (w * w);
        sh_win[it] = a.vrec[((size_t)t * B + b) * w * w + (it - t * w * w)];
    }
    if (tid < N) {
        // one thread per step fetches its record (six independent loads); thread 0 below then sums from LDS -- as a loop
        // of conditional loads on one thread it was a dozen memory round trips in a row
        const float* at = a.att + ((size_t)tid * B + b) * AIR_ATT_STRIDE;
        const float zv = at[AIR_ATT_Z], mk = at[AIR_ATT_MASK], mp = at[AIR_ATT_MASK_PREV];
        const float kz = at[AIR_ATT_KL_Z], ks = at[AIR_ATT_KL_SCALE], kh = at[AIR_ATT_KL_SHIFT];
        sh_z[tid] = zv;
        sh_act[tid] = mk != 0.0f ? 1 : 0;
        sh_rec[4 * tid] = mp; sh_rec[4 * tid + 1] = kz; sh_rec[4 * tid + 2] = ks; sh_rec[4 * tid + 3] = kh;
    }
    __syncthreads();
    float Lkeep = 0.0f;
    if (tid == 0) {
        // running loss in the reference order: z KL (old mask), scale, shift, VAE KL (new mask) :411-493
        float L = 0.0f;
        int digits = 0;
        for (int t = 0; t < N; ++t) {
            float* at = a.att + ((size_t)t * B + b) * AIR_ATT_STRIDE;
            const bool mask = sh_act[t] != 0;
            L = L + (sh_rec[4 * t] != 0.0f ? sh_rec[4 * t + 1] : 0.0f);
            L = L + (mask ? sh_rec[4 * t + 2] : 0.0f);
            L = L + (mask ? sh_rec[4 * t + 3] : 0.0f);
            L = L + (mask ? sh_kl[t] : 0.0f);
            digits += mask ? 1 : 0;
            at[AIR_ATT_KL_VAE] = sh_kl[t];
        }
        a.run_loss[b] = L;
        a.run_digits[b] = digits;
        Lkeep = L;
    }

    // phase B -- canvas + Bernoulli cross-entropy, pixel by pixel
    const float gsc = dyn[AIR_DYN_GRAD_SCALE];
    float acc = 0.0f;
    const int di = CF_THREADS / C, dj = CF_THREADS % C;
    int i = vt / C, j = vt % C;
    int k = 0;
    for (int p = vt; p < CC; p += CF_THREADS, ++k) {
        float x;
        if (xpre) {                                                 // (k < XPRE whenever xpre: a compile-time-indexed pick)
            x = xs[0];
#pragma unroll
            for (int q = 1; q < XPRE; ++q) x = (k == q) ? xs[q] : x;
        } else x = a.images[base + p];
        float R = 0.0f;                                             // running_recon :552
        for (int t = 0; t < N; ++t) {
            if (!sh_act[t]) continue;                               // where(active, z*w, 0) :433-439
            const Tap tx = sh_tx[(size_t)t * C + j], ty = sh_ty[(size_t)t * C + i];
            const float* win = sh_win + (size_t)t * w * w;
            const float wr = bilinear4(tx, ty, win[ty.i0 * w + tx.i0], win[ty.i1 * w + tx.i0],
                                       win[ty.i0 * w + tx.i1], win[ty.i1 * w + tx.i1]);
            R = R + sh_z[t] * wr;
        }
        i += di; j += dj;
        if (j >= C) { j -= C; ++i; }
        const float r = fmaxf(fminf(R, 1.0f), 0.0f);                // clipped_rec :582
        const float p1 = r + AIR_EPS, p0 = (1.0f - r) + AIR_EPS;
        acc += x * logf(p1) + (1.0f - x) * logf(p0);                // :586-589
        a.recon[base + p] = r;
        if (a.d_recon) {
            const bool pass = (R <= 1.0f) && (fminf(R, 1.0f) >= 0.0f);   // Minimum/Maximum grads pass at ties
            a.d_recon[base + p] = pass ? -gsc * (x / p1 - (1.0f - x) / p0) : 0.0f;
        }
    }
    acc = air_block_sum_n<NT / 64>(acc, sh_red);
    if (tid == 0) {
        const float rl = -acc;
        a.rec_loss[b] = rl;
        a.loss_item[b] = Lkeep + rl;                                // loss += reconstruction_loss :593
    }
}

// the canvas is staged in LDS only when one prefetch pass covers it (PF * THREADS floats, see the kernels)
size_t attend_canvas_floats(int C) { return (size_t)C * C <= 10 * (size_t)THREADS ? (size_t)C * C : 0; }
size_t attend_smem(int C, int w, int HT) {
    return (16 + MAX_STEPS + 8 * w + 4 + ((HT + 3) & ~3) + 7 * (size_t)HT + MAX_STEPS * (size_t)HT + attend_canvas_floats(C)) * sizeof(float);
}
size_t attend_bwd_smem(int C, int w) {
    return (24 + 8 * w + w + 4 + attend_canvas_floats(C)) * sizeof(float);
}
size_t write_smem(int N, int C, int w) { return (16 + 7 * MAX_STEPS + (size_t)N * (8 * C + (size_t)w * w)) * sizeof(float); }

}  // namespace

extern "C" int air_transformer_fwd(const float* U, const float* theta, float* out,
                                   int B, int Hi, int Wi, int Ho, int Wo, void* stream) {
    if (!U || !theta || !out || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return AIR_EINVAL;
    const long total = (long)B * Ho * Wo;
    const int blocks = (int)((total + THREADS - 1) / THREADS < 2048 ? (total + THREADS - 1) / THREADS : 2048);
    hipLaunchKernelGGL(transformer_fwd_kernel, dim3(blocks), dim3(THREADS), 0, air_stream(stream),
                       U, theta, out, B, Hi, Wi, Ho, Wo);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_attend_fwd(const air_attend_fwd_t* a, void* stream) {
    if (!a || !a->hid || !a->wout || !a->bout || !a->canvas || !a->eps_scale || !a->eps_shift || !a->u ||
        !a->dyn || !a->out7 || !a->att || !a->window)
        return AIR_EINVAL;
    if (a->B <= 0 || a->N <= 0 || a->C < 2 || a->w < 2 || a->Hs <= 0 || a->Hh <= 0 || a->Hz <= 0) return AIR_EINVAL;
    if (a->w > 64 || a->N > MAX_STEPS) return AIR_ELIMIT;
    const size_t lds = attend_smem(a->C, a->w, 2 * a->Hs + 2 * a->Hh + a->Hz);
    int rc = ensure_lds(attend_fwd_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(attend_fwd_kernel, dim3(a->B, a->N), dim3(THREADS), lds, air_stream(stream), *a);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_attend_bwd(const air_attend_bwd_t* a, void* stream) {
    if (!a || !a->hid || !a->wout || !a->canvas || !a->eps_scale || !a->eps_shift || !a->dyn || !a->out7 ||
        !a->att || !a->d_window || !a->d_sxy_write || !a->d_hid || !a->d_out7)
        return AIR_EINVAL;
    if (a->B <= 0 || a->N <= 0 || a->C < 2 || a->w < 2 || a->Hs <= 0 || a->Hh <= 0 || a->Hz <= 0) return AIR_EINVAL;
    if (a->w > 64) return AIR_ELIMIT;
    const size_t lds = attend_bwd_smem(a->C, a->w);
    int rc = ensure_lds(attend_bwd_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(attend_bwd_kernel, dim3(a->B, a->N), dim3(THREADS), lds, air_stream(stream), *a);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_write_fwd(const air_write_fwd_t* a, void* stream) {
    if (!a || !a->vrec || !a->ml || !a->images || !a->dyn || !a->att || !a->recon || !a->rec_loss ||
        !a->run_loss || !a->run_digits || !a->loss_item)
        return AIR_EINVAL;
    if (a->B <= 0 || a->N <= 0 || a->C < 2 || a->w < 2 || a->Z <= 0) return AIR_EINVAL;
    if (a->N > MAX_STEPS) return AIR_ELIMIT;
    size_t lds = write_smem(a->N, a->C, a->w);
    const unsigned extra = a->wb_order ? 1u : 0u;                 // + the workgroup that sorts the backward's items
    if (a->wb_order) {
        if ((long)a->N * a->B > WB_ORDER_MAX) return AIR_ELIMIT;
        if (lds < WB_ORDER_MAX * sizeof(unsigned)) lds = WB_ORDER_MAX * sizeof(unsigned);
    }
    {
        int rc = ensure_lds(write_fwd_kernel<CF_THREADS>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((write_fwd_kernel<CF_THREADS>), dim3(a->B + extra), dim3(CF_THREADS), lds, air_stream(stream), *a);
    }
    AIR_CHECK_LAUNCH();
    return 0;
}
