// The reference's numeric summaries as ONE launch (air_model.py:160-209 _summarize_by_digit_count / _summarize_by_step,
// :608-632 the list training.py:169-200 evaluates on the 1 000 test images every 50 iterations): see air_summaries_t in
// air_hip.h.  One workgroup per summary row (4 post-loop quantities + 6 per-step quantities x N steps), each producing the
// row's max_digits + 2 masked means.  Sums are taken in fp64 over a fixed thread / wave order (deterministic; the result
// is the correctly rounded fp32 mean, within 1e-7 relative of any fp32 summation order).
#include "air_common.h"

constexpr int SM_THREADS = 256;
constexpr int SM_GROUPS = 8;                 // max_digits + 2 <= 8

__global__ __launch_bounds__(SM_THREADS) void summaries_kernel(air_summaries_t a) {
    __shared__ int sh_alive[AIR_MAX_STEPS_SUMMARY];
    __shared__ double sh_sum[SM_THREADS / 64][SM_GROUPS];
    __shared__ int sh_cnt[SM_THREADS / 64][SM_GROUPS];
    const int r = blockIdx.x, tid = threadIdx.x, B = a.B, N = a.N, G = a.max_digits + 1;
    if (r == 0 && tid < 2) a.out[tid] = a.scalars[tid];                 // loss, accuracy (:610-611), already batch means
    int q = -1, i = 0;
    if (r >= 4) { q = (r - 4) / N; i = (r - 4) % N; }
    // T' of the while_loop (cond :271-275): the loop ran step t + 1 iff some image was still active after step t; a column
    // i >= T' does not exist in the reference's [B, T'] stacks and is zero-padded (:187)
    bool col = true;
    if (q >= 0 && i > 0) {
        if (tid < N) sh_alive[tid] = 0;
        __syncthreads();
        for (int t = 0; t < i; ++t) {
            int any = 0;
            for (int b = tid; b < B; b += SM_THREADS) any |= a.att[((size_t)t * B + b) * AIR_ATT_STRIDE + AIR_ATT_MASK] > 0.0f;
            if (any) sh_alive[t] = 1;                                    // (benign: every writer stores 1)
        }
        __syncthreads();
        for (int t = 0; t < i; ++t) col = col && sh_alive[t];
    }
    static const int slot_of[6] = {AIR_ATT_S, AIR_ATT_ZPROB, AIR_ATT_KL_Z, AIR_ATT_KL_SCALE, AIR_ATT_KL_SHIFT, AIR_ATT_KL_VAE};
    const int slot = q >= 0 ? slot_of[q] : 0;
    const bool all_steps = q == 1;
    const int thr = i - (q == 2 ? 1 : 0);                               // one_more_step: z_pres_kl (:623)
    double sum[SM_GROUPS];
    int cnt[SM_GROUPS];
#pragma unroll
    for (int g = 0; g < SM_GROUPS; ++g) { sum[g] = 0.0; cnt[g] = 0; }
    for (int b = tid; b < B; b += SM_THREADS) {
        const int tg = a.targets[b], dg = a.digits[b];
        float v;
        bool m = true;
        if (r == 0) v = (float)dg;
        else if (r == 1) v = a.rec_loss[b];
        else if (r == 2) v = dg == tg ? 1.0f : 0.0f;
        else if (r == 3) v = a.loss_item[b];
        else {
            v = col ? a.att[((size_t)i * B + b) * AIR_ATT_STRIDE + slot] : 0.0f;
            m = all_steps || dg > thr;
        }
        if (!m) continue;
#pragma unroll
        for (int g = 0; g < SM_GROUPS - 1; ++g)
            if (g < G && tg == g) { sum[g] += (double)v; ++cnt[g]; }
        sum[SM_GROUPS - 1] += (double)v; ++cnt[SM_GROUPS - 1];         // "_all_dig"
    }
#pragma unroll
    for (int g = 0; g < SM_GROUPS; ++g) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { sum[g] += __shfl_xor(sum[g], off, 64); cnt[g] += __shfl_xor(cnt[g], off, 64); }
    }
    if ((tid & 63) == 0) {
#pragma unroll
        for (int g = 0; g < SM_GROUPS; ++g) { sh_sum[tid >> 6][g] = sum[g]; sh_cnt[tid >> 6][g] = cnt[g]; }
    }
    __syncthreads();
    if (tid <= G) {
        const int g = tid < G ? tid : SM_GROUPS - 1;
        double s = 0.0;
        int c = 0;
        for (int wv = 0; wv < SM_THREADS / 64; ++wv) { s += sh_sum[wv][g]; c += sh_cnt[wv][g]; }
        a.out[2 + r * (G + 1) + tid] = (float)(s / (double)c);          // empty group: 0 / 0 = NaN, as tf.reduce_mean of nothing
    }
}

extern "C" int air_summaries_count(int N, int max_digits) { return 2 + (4 + 6 * N) * (max_digits + 2); }

extern "C" int air_summaries(const air_summaries_t* a, void* stream) {
    if (!a || !a->att || !a->targets || !a->digits || !a->rec_loss || !a->loss_item || !a->scalars || !a->out) return AIR_EINVAL;
    if (a->B <= 0 || a->N <= 0 || a->max_digits < 0) return AIR_EINVAL;
    if (a->N > AIR_MAX_STEPS_SUMMARY || a->max_digits + 2 > SM_GROUPS) return AIR_ELIMIT;
    hipLaunchKernelGGL(summaries_kernel, dim3(4 + 6 * a->N), dim3(SM_THREADS), 0, air_stream(stream), *a);
    AIR_CHECK_LAUNCH();
    return 0;
}
