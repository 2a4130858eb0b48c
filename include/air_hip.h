/*
 * air_hip.h -- C ABI of libair_hip.so, the MI355X (gfx950) implementation of the
 * Attend-Infer-Repeat hot path (aakhundov/tf-attend-infer-repeat).
 *
 * The reference has no FFI: its seam is the TensorFlow graph that
 * air/air_model.py builds.  Every entry point below replaces a group of TF ops
 * the reference instantiates per time step; the reference file:line each one
 * replaces is cited.  INTEGRATION.md shows the binding a maintainer of the
 * reference would add.
 *
 * Conventions (all entry points):
 *   - every pointer is a CALLER-OWNED DEVICE pointer, contiguous row-major fp32
 *     unless stated; the library never allocates or frees, keeps no state but two per-device caches (the LDS grant of a
 *     kernel function, the lane-order probe below), is re-entrant, and does not synchronise -- with ONE exception: the
 *     first EAGER air_write_bwd(literal 2) / air_transformer_bwd of a process on a device probes the lane order of the LDS
 *     atomic pipe (one tiny kernel, an 8-byte copy back, one stream synchronise).  Under stream capture nothing is probed:
 *     a capture-first caller takes the register-chain accumulator (same bits, slower; a message on stderr).  Every other
 *     call is hipGraph-capture safe from the first call;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*);
 *   - return value: 0 on success, a positive hipError_t from the launch, or a
 *     negative AIR_E* argument error; no C++ exception crosses the boundary.
 */
#ifndef AIR_HIP_H
#define AIR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: air_gemm_t / air_wgrad_t carry bf16 twins; `literal` of the sampler backward: 0 exact, 1 per-tap order, 2 the
 * reference graph's order (version 1 headers of round 1 called the per-tap order "reference")
 * 3: panel-blocked bf16 shadows of the weights (air_panel_t, air_gemm_t.B16p, air_panel_shadow,
 *    air_adam_clip_step_panels)
 * 4: measured-negative paths removed (DESIGN.md sections 8-10 keep the record): the deferred Adam slices
 *    (air_step_job_t.ad_*, air_adam_clip_step_blocks), the banded compose (air_write_fwd_t.rec_part / bands,
 *    air_finalize_parts, air_write_bwd_t.fin_rec_part ..), air_adam_clip_step_factored; added: `literal` 3 / 4 of air_write_bwd, air_shuffle_batch_*
 * 5: added air_shuffle_batch_dequeue_many, air_batch_gather, air_summaries; removed `literal` 1 and 3 of the sampler
 *    backward (backward="taps" / "reference_blocked") */
#define AIR_ABI_VERSION 5

#define AIR_EINVAL   (-1)   /* bad dimension / null pointer            */
#define AIR_ELIMIT   (-2)   /* size exceeds what the kernel supports    */
#define AIR_EALIGN   (-3)   /* pointer / leading dimension misaligned   */

int air_abi_version(void);
const char* air_strerror(int code);

/* ---- dynamic scalars --------------------------------------------------
 * Kernels read the float hyper-parameters that the reference allows to be
 * annealed (air_model.py:76-82) from a device array `dyn` so that a captured
 * hipGraph sees new values without re-capture. */
enum {
    AIR_DYN_PRIOR_LOG_ODDS = 0,  /* z_pres_prior_log_odds          air_model.py:50,405 */
    AIR_DYN_TEMPERATURE    = 1,  /* z_pres_temperature             :51,381             */
    AIR_DYN_STOP_THRESHOLD = 2,  /* stopping_threshold             :52,274,412         */
    AIR_DYN_LEARNING_RATE  = 3,  /* learning_rate                  :54,654             */
    AIR_DYN_CLIP_NORM      = 4,  /* gradient_clipping_norm         :55,673             */
    AIR_DYN_SCALE_PM = 5, AIR_DYN_SCALE_PV = 6,   /* scale prior mean / variance  :38-39 */
    AIR_DYN_SHIFT_PM = 7, AIR_DYN_SHIFT_PV = 8,   /* shift prior                  :40-41 */
    AIR_DYN_VAE_PM   = 9, AIR_DYN_VAE_PV   = 10,  /* vae prior                    :42-43 */
    AIR_DYN_LIK_STD  = 11,                        /* vae_likelihood_std           :44    */
    AIR_DYN_GRAD_SCALE = 12,     /* d(loss)/d(per-item loss) = 1/B (reduce_mean, :610) */
    /* log of the prior variances AS PASSED TO THE CONSTRUCTOR (air_model.py:72-74: tf.log(...) is taken
     * before any annealing schedule replaces the attribute, :76-82) -- never written by a schedule */
    AIR_DYN_SCALE_PLV = 13, AIR_DYN_SHIFT_PLV = 14, AIR_DYN_VAE_PLV = 15,
    AIR_DYN_COUNT    = 16
};

/* int state words (device int32 array `istate`) */
enum {
    AIR_IST_GLOBAL_STEP = 0,     /* air/global_step, air_model.py:69 */
    AIR_IST_COUNT = 4
};

/* one annealing schedule (air_model.py:94-121), evaluated on device each step */
typedef struct {
    int32_t slot;        /* AIR_DYN_* written                               */
    int32_t flags;       /* bit0 staircase, bit1 has_min, bit2 has_max, bit3 log */
    float init, iters, factor, vmin, vmax;
} air_schedule_t;

/* ---- per-step records ---------------------------------------------------
 * att[t][b][AIR_ATT_STRIDE] -- per-item scalars of one time step. */
enum {
    AIR_ATT_S = 0, AIR_ATT_X = 1, AIR_ATT_Y = 2,     /* scale, shift  :301,318      */
    AIR_ATT_ZPRE = 3,                                /* z_pres_pre_sigmoid :380     */
    AIR_ATT_Z = 4,                                   /* z_pres (rounded if !train)  */
    AIR_ATT_ZPROB = 5,                               /* sigmoid(log_odds) :395      */
    AIR_ATT_KL_Z = 6, AIR_ATT_KL_SCALE = 7, AIR_ATT_KL_SHIFT = 8, AIR_ATT_KL_VAE = 9,
    AIR_ATT_MASK_PREV = 10,                          /* stopping_sum(old) < thr :412 */
    AIR_ATT_MASK = 11,                               /* stopping_sum(new) < thr :427 */
    AIR_ATT_ST_BACK = 12,                            /* 1/s, -x/s, -y/s  :353-356   */
    AIR_ATT_STRIDE = 16
};
/* out7[t][b][8]: scale_mean, scale_lv, shift_mean.x, .y, shift_lv.x, .y, z_log_odds, pad */
#define AIR_OUT_STRIDE 8

/* ---- GEMM (K1/K2/K5: LSTM, head and VAE MatMul + BiasAdd + activation) --------
 * C[M,N] (+)= epilogue( op(A)[M,K] . op(B)[K,N] )
 *   op(A)(m,k) = transA ? A[k*lda+m] : A[m*lda+k];  op(B)(k,n) = transB ? B[n*ldb+k] : B[k*ldb+n]
 * epilogue, in this order:  v = acc; v += bias[n]; v += addend[m*ldadd+n];
 *   v = act(v); v *= actgrad(aux[m*ldaux+n]); if (ACCUM) v += C[m*ldc+n].
 * Replaces tf.matmul + BiasAdd + Relu/Softplus/Sigmoid nodes of
 * layers.fully_connected (air_model.py:292-316, 374-376; vae.py:13-34), the
 * BasicLSTMCell MatMul (:286) and their MatMul_grad nodes.
 * precision: 0 = exact-fp32 MFMA (v_mfma_f32_16x16x4_f32);
 *            1 = bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16). */
enum {
    AIR_ACT_NONE = 0, AIR_ACT_RELU = 1, AIR_ACT_SOFTPLUS = 2,
    AIR_ACT_SIGMOID_NOISE = 3   /* sigmoid(v + aux*aux_scale): vae.py:36-41 */
};
enum {
    AIR_GRAD_NONE = 0,
    AIR_GRAD_RELU = 1,          /* v *= (aux > 0)                        */
    AIR_GRAD_SOFTPLUS = 2       /* v *= 1 - exp(-aux)  (aux = softplus output) */
};
/* fused epilogues (the pointwise ops the reference runs right after the MatMul) */
enum {
    AIR_EPI_GENERIC = 0,
    /* BasicLSTMCell pointwise part (air_model.py:286): N = 4R gate columns i,j,f,o (groups of R);
     * pre-activation = acc + sum(addend slabs) + bias.  p0 = c_prev [M,R];
     * q0 = acts [M,4R] (sigmoid i, tanh j, sigmoid(f+1), sigmoid o), q1 = c [M,R], q2 = h [M,R]. */
    AIR_EPI_LSTM_FWD = 1,
    /* vae.py:16-24: N = 2Z (mean | log_var); C = ml [M,2Z]; p0 = eps [M,Z]; q0 = sample [M,Z]. */
    AIR_EPI_REPARAM_FWD = 2,
    /* LSTM backward: acc (+ addend) = d loss/d h' [M,R]; p0 = acts, p1 = c_prev, p2 = c,
     * p3 = d c' from the next step (nullable); q0 = dgates [M,4R], q1 = d c_prev [M,R],
     * q2 = running sum of dgates (nullable), i0 = accumulate into q2. */
    AIR_EPI_LSTM_BWD = 3,
    /* re-parameterisation backward + VAE-KL gradient (air_model.py:481-493): acc = d loss/d z;
     * p0 = ml [M,2Z], p1 = eps [M,Z], p2 = att, p3 = dyn; C = d_ml [M,2Z]. */
    AIR_EPI_REPARAM_BWD = 4,
    /* rows [0, i0): plain store of acc into C; rows [i0, M): AIR_EPI_LSTM_BWD of the LAST time step
     * (no d c' from a later step, q2 is stored not accumulated), with p0..p2 / q0..q2 indexed by
     * (row - i0).  Lets the GEMM that produces d h' of all steps also start the BPTT chain. */
    AIR_EPI_LSTM_BWD_TAIL = 5,
    /* The FIRST step of the recurrence in the GEMM that hoists x.Wx (zero_state, air_model.py:540: h_0 = c_0 = 0,
     * so [x, h].kernel = x.Wx): N = 4R; C [M,4R] receives the raw product (the addend of the later steps,
     * one slab), and q0 = acts, q1 = c, q2 = h of step 0 from acc + bias.  16-column tiles of four units x
     * four gates (untransposed operands, no split-K, no addend). */
    AIR_EPI_LSTM_FWD0 = 6
};
/* air_step_begin's work as a descriptor, so that a GEMM launch can carry it (air_gemm_t.step_job) */
typedef struct {
    const air_schedule_t* sched /*device*/; int32_t nsched; float* dyn; const int32_t* istate;
    float* normals; int64_t n_normal; float* uniforms; int64_t n_uniform; uint64_t seed;
    const float* twin_src; uint16_t* twin_dst; int64_t twin_n;   /* see air_step_begin */
} air_step_job_t;
typedef struct {
    const float* A; const float* B; float* C;
    int32_t M, N, K, lda, ldb, ldc;
    int32_t transA, transB;
    const float* bias;             /* [N] or NULL                        */
    const float* addend; int32_t ldadd;   /* [M,N] or NULL               */
    const float* aux;    int32_t ldaux;   /* [M,N] or NULL               */
    float   aux_scale;
    int32_t act;                   /* AIR_ACT_*                          */
    int32_t actgrad;               /* AIR_GRAD_*                         */
    int32_t accumulate;            /* C += result                        */
    int32_t precision;             /* 0 fp32, 1 bf16                     */
    int32_t epi;                   /* AIR_EPI_*                          */
    int32_t tile_m, tile_n;        /* output tile in units of 16 (0 = auto): (1,1)(1,2)(1,4)(2,2)(2,4)(4,1)(4,2)(4,4); (8,4) = the
                                      throughput tiling for a deep split-K product of an fp32 A with a bf16 B16 (M % 128, N % 64, K-slab % 64 == 0,
                                      ksplit > 1, generic epilogue; AIR_EINVAL / AIR_EALIGN otherwise -- callers fall back to (4,4)) */
    int32_t ksplit;                /* > 1: split K over grid.z; C receives `air_gemm_slabs()` slabs of
                                      [M,ldc] (plain stores, generic epilogue skipped)            */
    int32_t addend_slabs;          /* addend is that many [M,ldadd] slabs (0/1 = one)            */
    int32_t i0;
    const float* p0; const float* p1; const float* p2; const float* p3;
    float* q0; float* q1; float* q2;
    /* optional (HOST pointer, read during the call): the step prologue of air_step_begin is run by
     * an extra plane of workgroups of THIS launch.  Only for a GEMM that reads neither the noise nor
     * dyn (the hoisted x.Wx): the train step then has no separate prologue launch. */
    const air_step_job_t* step_job;
    /* bf16 TWINS (precision 1 only; each nullable).  A twin is the RNE-rounded bf16 copy of an fp32 array with the
     * same shape and leading dimension -- exactly what the kernel would have produced on the operand's way into
     * LDS, so results are bit-identical with and without them.  A16 / B16: twins of A / B; when B16 (and A16, or an
     * fp32 A) are given and aligned to 16 bytes the operands are read as bf16 (half the bytes through the CU,
     * no conversion; row-major weights through the LDS transpose read).  C16 / q0_16 / q2_16: twins the epilogue
     * writes next to C / q0 / q2 for the next consumer (q2_16 of AIR_EPI_LSTM_BWD: pass it with the LAST
     * accumulation into q2 only). */
    const uint16_t* A16; const uint16_t* B16;
    uint16_t* C16; uint16_t* q0_16; uint16_t* q2_16;
    /* PANEL-BLOCKED bf16 twin of an untransposed B = [K, N] (precision 1, nullable; air_panel_t below describes the
     * layout and who writes it): the same bf16 values as B16, stored as N/16 panels of [K rows][16 columns], so that a
     * 16-column output tile reads ONE contiguous 32*K-byte block instead of a 32-byte piece of every 2*ldb-byte row
     * (a quarter of each cache line it pulls through the CU).  For AIR_EPI_LSTM_FWD / AIR_EPI_LSTM_FWD0 (N = 4R gate
     * columns) the panels are the gate-interleaved ones (air_panel_t.gates = 4): panel p holds units 4p..4p+3, column
     * gate*4 + u%4 -- the four gates of four units in 32 contiguous bytes per row.  Read instead of B16 by the 16- and
     * 32-column tiles; results are bit-identical.  B16 may be NULL when B16p is given and the tile supports it. */
    const uint16_t* B16p;
} air_gemm_t;
/* number of K-slabs a ksplit request produces for contraction depth K */
int air_gemm_slabs(int K, int ksplit);
int air_gemm(const air_gemm_t* g, void* stream);
/* name of the kernel function `g` dispatches to, as rocprofv3 prints it (profiling aid) */
int air_gemm_kernel_name(const air_gemm_t* g, char* buf, int n);

/* dst[i] = bf16(src[i]), round to nearest even: the twin of an array whose producer could not write it (the flat
 * variable buffer after a host-side load -- air_adam_clip_step keeps its shadow fresh afterwards). */
int air_bf16_twin(const float* src, uint16_t* dst, int64_t n, void* stream);

/* ---- panel-blocked bf16 shadows of row-major [K, N] weight matrices that live in a flat fp32 buffer ----------
 * Element (k, n) of the matrix at flat offset src_off goes to
 *   gates == 0:  dst_off + (n / 16) * (K * 16) + k * 16 + n % 16                  (ceil(N / 16) * K * 16 elements)
 *   gates == 4:  N = 4R gate columns (BasicLSTMCell's i, j, f, o blocks, air_model.py:286), gate = n / R, u = n % R:
 *                dst_off + (u / 4) * (K * 16) + k * 16 + gate * 4 + u % 4          (K * N elements, R % 4 == 0)
 * of the panel shadow (what air_gemm_t.B16p points into).  N % 4 == 0, src_off % 4 == 0, dst_off % 4 == 0.
 * `exclusive` != 0: the row-major bf16 shadow of this range is NOT maintained by air_adam_clip_step_panels (no kernel
 * reads it: the hoisted x.Wx is the only consumer of Wx and takes the panels). */
typedef struct {
    int64_t src_off, dst_off;
    int32_t K, N, gates, exclusive;
} air_panel_t;
#define AIR_MAX_PANELS 16
/* panel_shadow <- bf16(params), for every described matrix (after a host-side change of the variables) */
int air_panel_shadow(const float* params, uint16_t* panel_shadow, const air_panel_t* panels /*HOST array, <= 16, ascending src_off*/,
                     int count, void* stream);

/* ---- grouped weight gradients: every dW = A^T . dY (+ db = column sums of dY) of the
 * step in ONE launch (MatMul_grad / BiasAdd_grad nodes of all variables; weights are shared
 * across time steps so K = N_steps * B rows).  head_pack = 1 handles the 7 head output units:
 * A = d_out7 [K,8], dY = hid [K,HT], dW = wout [7][ldc] (unit o keeps only its head's hidden
 * segment), db = bout [7]. */
typedef struct {
    const float* A; const float* dY; float* dW; float* db /*nullable*/;
    int32_t M, N, K, lda, ldb, ldc;
    int32_t head_pack, Hs, Hh, Hz;
    /* bf16 twins of A / dY (precision 1, nullable, same leading dimensions; see air_gemm_t): a problem that has both
     * (8-byte aligned, lda/ldb/N multiples of 4, M a multiple of 4 or lda >= M rounded up to 4 -- padded rows --, not
     * head_pack) reads its operands as bf16 -- bit-identical dW.  Padded rows (M % 4 != 0, lda >= M rounded up to 4): the pad
     * columns of A and A16 must be READABLE but may hold anything, NaN included -- a piece that straddles M only feeds output
     * rows >= M, which are neither stored nor counted in sq_partials (tests/test_gpu_kernels.py::
     * test_wgrad_twins_of_padded_rows_ignore_what_the_pad_holds).
     * db is always summed from the fp32 dY. */
    const uint16_t* A16; const uint16_t* dY16;
} air_wgrad_t;
/* precision: 0 = fp32 MFMA (exact fp32 products), 1 = operands rounded to bf16, fp32 accumulate.
 * sq_partials (nullable): [air_wgrad_num_blocks()] floats receiving the sum of squares of every
 * gradient element each workgroup stored -- the tf.global_norm terms (air_model.py:673), handed to
 * air_adam_clip_step instead of a separate air_grad_sqnorm pass (single-GPU path; with data
 * parallelism the norm is taken after the all-reduce).  istate (nullable, only with sq_partials):
 * istate[GLOBAL_STEP] += 1, as air_grad_sqnorm does.  A plain problem may have dW == NULL when
 * sq_partials is given: its tiles are computed and squared but not stored (a caller that rebuilds the
 * gradient from its factors elsewhere). */
/* air_wgrad_num_blocks(): the number of global-norm partials = 64 x 64 tiles of all problems + one per bias column tile of
 * the problems with K >= 384 (their db is summed by workgroups of their own, the launch's first, instead of by a tile). */
int air_wgrad_num_blocks(const air_wgrad_t* probs, int count);
/* Workgroups air_wgrad_grouped(precision) launches for these problems: air_wgrad_num_blocks() unless a problem runs in
 * STRIPS -- at precision 1 a twin problem of >= 512 tiles (K = 64, 128, 192 or 256; 16-byte dY rows; M % 4 == 0) gives each
 * workgroup 2 (from 2048 tiles: 4) consecutive column tiles of a block-row, its A block fetched once and its MFMA fragments
 * kept in registers.  Values, db and the partials (still one per tile, at the same index) are bit-identical either way.
 * For a problem of >= 512 tiles (any precision) the tile that sums db of column tile nt is the one in block-row nt % 16,
 * not block-row 0.  Diagnostic; AIR_WGRAD_STRIP=<tiles per workgroup, 0 = off> overrides. */
int air_wgrad_num_workgroups(const air_wgrad_t* probs, int count, int precision);
int air_wgrad_grouped(const air_wgrad_t* probs /*HOST array, <= 16*/, int count, int precision,
                      float* sq_partials, int32_t* istate, void* stream);

/* column sums db[n] = sum_r dY[r*ld + n]  (BiasAdd_grad nodes) for `count` problems */
typedef struct { const float* src; float* dst; int32_t rows, cols, ld, accumulate; } air_colsum_t;
int air_colsum(const air_colsum_t* probs /*HOST array*/, int count, void* stream);

/* ---- LSTM gates (K1c) -- BasicLSTMCell pointwise part, air_model.py:286 -------
 * gates_pre [B,4R] = [x,h].kernel + bias, split order i,j,f,o; forget_bias 1.0.
 * acts [B,4R] receives sigmoid(i), tanh(j), sigmoid(f+1), sigmoid(o). */
int air_lstm_gates_fwd(const float* gates_pre, const float* c_prev, float* acts,
                       float* c, float* h, int B, int R, void* stream);
/* The first step of the loop: the LSTM starts from zero_state (air_model.py:540), so [x, h].kernel is the
 * hoisted x.Wx alone -- no MatMul.  xw_slabs: `nslabs` split-K slabs [B,4R] of x.Wx (air_gemm ksplit);
 * pre-activation = slab sum (in slab order) + bias, i.e. AIR_EPI_LSTM_FWD with a zero accumulator. */
int air_lstm_first_step(const float* xw_slabs, int nslabs, const float* bias /*nullable*/, float* acts,
                        float* c, float* h, uint16_t* h16 /*bf16 twin of h, nullable*/, int B, int R, void* stream);
/* dgates [B,4R] (pre-activation grads), dc_prev [B,R]; dgsum (+)= dgates when given */
int air_lstm_gates_bwd(const float* dh, const float* dc_in /*nullable*/, const float* acts,
                       const float* c_prev, const float* c, float* dgates, float* dc_prev,
                       float* dgsum /*nullable*/, int dgsum_accumulate, int B, int R, void* stream);

/* ---- spatial transformer, generic (transformer.py:18-175) ---------------------
 * out[B,Ho,Wo] = transformer(U[B,Hi,Wi,1], theta[B,2,3], (Ho,Wo)); literal
 * 4-product / add_n op order, indices clipped before the weights. */
int air_transformer_fwd(const float* U, const float* theta, float* out,
                        int B, int Hi, int Wi, int Ho, int Wo, void* stream);
/* its gradient (what tf.gradients builds for transformer.py:56-171): d_U [B,Hi,Wi] and / or d_theta [B,2,3]
 * (either may be NULL) from d_out [B,Ho,Wo].
 *   d_U: ONE fp32 accumulator per input pixel that receives the a-, b-, c-, d-tap terms in output-pixel order -- the
 *        graph's single UnsortedSegmentSum over the concatenated Gather gradients -- BIT-IDENTICAL to the executed
 *        reference graph (oracle.transformer_backward, tests/test_gpu_kernels.py).
 *   d_theta: per output pixel the graph's AddN order for the coordinate gradients; the contraction over the output
 *        pixels with (x_t, y_t, 1) (MatMul_grad) is a per-thread strided sum followed by a fixed-order block reduction,
 *        i.e. it matches the graph up to the reduction order of that one sum: tested to 2e-5 of the largest element.
 * Hi*Wi <= ~20 000 (U and d_U of one image live in LDS). */
int air_transformer_bwd(const float* U, const float* theta, const float* d_out, float* d_U, float* d_theta,
                        int B, int Hi, int Wi, int Ho, int Wo, void* stream);

/* ---- "attend": heads output layer + sampling + KLs + stop logic + ST read -----
 * air_model.py:288-333 (scale/shift heads, theta, transformer canvas->window) and
 * :368-427 (z_pres head, Concrete sample, z KL, stopping_sum masks), :441-477
 * (scale/shift KL) for ALL N time steps at once: one workgroup per (image, step).
 * The LSTM input is the same image every step (:286, :535), so h' of every step
 * exists before the heads run; the stopping sum is re-derived per block in step
 * order (bitwise equal to the sequential loop).
 * hid [N,B,HT] = ReLU hidden activations of the 5 heads, concatenated in the order
 * scale/mean, scale/log_variance, shift/mean, shift/log_variance, z_pres/log_odds
 * with widths Hs,Hs,Hh,Hh,Hz (HT = 2Hs+2Hh+Hz).  wout [7][wout_ld] holds one row
 * per output unit, bout [7]. */
typedef struct {
    const float* hid; const float* wout; const float* bout;
    const float* canvas;                 /* input_images [B,C*C]            */
    const float* eps_scale; const float* eps_shift; const float* u;   /* [N,B,1],[N,B,2],[N,B] */
    const float* dyn;                    /* AIR_DYN_* device array          */
    float* out7;                         /* [N,B,8]                         */
    float* att;                          /* [N,B,AIR_ATT_STRIDE]            */
    float* window;                       /* [N,B,w*w]                       */
    int32_t B, N, C, w, Hs, Hh, Hz, wout_ld, train;
    uint16_t* window16;                  /* bf16 twin of window (nullable): A operand of the first recognition GEMM */
} air_attend_fwd_t;
int air_attend_fwd(const air_attend_fwd_t* a, void* stream);

typedef struct {
    const float* hid; const float* wout;
    const float* canvas; const float* eps_scale; const float* eps_shift;
    const float* dyn; const float* out7; const float* att;
    const float* d_window;               /* [N,B,w*w] grad wrt the glimpse  */
    const float* d_sxy_write;            /* [N,B,4]: ds,dx,dy,dz from the write path */
    float* d_hid;                        /* [N,B,HT] grad wrt pre-ReLU hidden */
    float* d_out7;                       /* [N,B,8]                         */
    int32_t B, N, C, w, Hs, Hh, Hz, wout_ld;
    int32_t literal;                     /* 0: exact adjoint.
                                            2: the op order of the reference's saved graph (model/air-model.meta):
                                               AddN_10/11 (AddN_20/21) input order for the coordinate gradients and,
                                               in air_write_bwd, ONE fp32 accumulator per window pixel through the four
                                               concatenated Gather gradients (the graph's single UnsortedSegmentSum) --
                                               keeps the out-of-range rounding residue, bit-identical to the executed graph.
                                            (air_attend_bwd: any non-zero value selects the graph order)               */
    uint16_t* d_hid16;                   /* bf16 twin of d_hid (nullable)   */
} air_attend_bwd_t;
int air_attend_bwd(const air_attend_bwd_t* a, void* stream);

/* grads of the 7 output units: dwout[o][j] = sum_{r} d_out7[r][o]*hid[r][seg(o)+j],
 * dbout[o] = sum_r d_out7[r][o]; rows = N*B */
int air_heads_out_wgrad(const float* d_out7, const float* hid, float* dwout, float* dbout,
                        int rows, int Hs, int Hh, int Hz, int wout_ld, void* stream);

/* ---- VAE re-parameterisation (vae.py:22-24) ---------------------------------- */
int air_reparam_fwd(const float* ml /*[B,2Z] mean|log_var*/, const float* eps_z, float* zs,
                    int B, int Z, void* stream);
/* d_ml [B,2Z] from d_zs [B,Z] plus the VAE-KL gradient (air_model.py:481-493) */
int air_reparam_bwd(const float* d_zs, const float* ml, const float* eps_z, const float* att,
                    const float* dyn, float* d_ml, int B, int Z, void* stream);

/* ---- "write"/compose: ST window->canvas for all N steps + z_pres scaling + masked
 * accumulate (in step order, in registers) + VAE KL + running loss + digit count +
 * reconstruction loss and its gradient.  air_model.py:351-366 (theta_recon,
 * transformer), :409-439 (masks, running_recon), :479-496 (VAE KL), :580-593 (loss).
 * One workgroup per image. */
typedef struct {
    const float* vrec;                   /* vae reconstructions [N,B,w*w]   */
    const float* ml;                     /* [N,B,2Z] mean | log_var         */
    const float* images;                 /* [B,C*C]                         */
    const float* dyn;
    float* att;                          /* [N,B,16]: reads s,x,y,z,masks,KLs; writes KL_VAE */
    float* recon;                        /* [B,C*C] clipped reconstruction  */
    float* rec_loss;                     /* [B]                             */
    float* d_recon;                      /* [B,C*C] d loss/d running_recon, nullable */
    float* run_loss;                     /* [B] sum of masked KLs (running_loss) */
    int32_t* run_digits;                 /* [B] rec_num_digits              */
    float* loss_item;                    /* [B] run_loss + rec_loss         */
    int32_t B, N, C, w, Z;
    /* nullable, [N*B] (N*B <= 4096): one extra workgroup of this launch sorts the (image, step) items by the work the
     * graph-order write backward will have with them (its corner terms; inactive items last) and leaves the permutation
     * here -- air_write_bwd_t.order.  Worth it when there are more items than CUs (128 x 128, b = 256: 1280 items). */
    int32_t* wb_order;
} air_write_fwd_t;
int air_write_fwd(const air_write_fwd_t* a, void* stream);

/* one workgroup per (image, step): all steps see the same d loss / d canvas */
typedef struct {
    const float* d_recon;                /* [B,C*C]                         */
    const float* vrec; const float* att; /* [N,B,w*w], [N,B,16]             */
    float* d_gen_pre;                    /* [N,B,w*w] grad wrt gen_mean pre-sigmoid input */
    float* d_sxy_write;                  /* [N,B,4]: ds,dx,dy (via theta_recon), dz */
    int32_t B, N, C, w;
    int32_t literal;                     /* 0, 2: see air_attend_bwd_t.  air_write_bwd only:
                                            4: as 2 for every window pixel whose four term streams have at most 64 terms each
                                               (all but the four corners and a few long borders); a pixel with a longer stream
                                               has every tap's stream cut into at most 16 chunks of at least 64 terms and walks
                                               each chunk TWICE -- C_k from +0.0, Q_k from P_k = the running sum of the earlier
                                               C's -- and returns Q_last + sum_{k < last} (Q_k - P_k+1), the (exact) differences
                                               added left to right: every add of the Q chains rounds at the magnitude it has in
                                               the one long chain of 2, and nothing else does (backward="reference_carried"; no
                                               LDS atomics, no probe).  The coordinate / z gradients d_sxy_write of 4 are
                                               per-column thread sums in the graph's AddN order per pixel (rows whose y taps clip
                                               to one index are exact zeros and skipped): equal to 2's to ~1e-7 relative, not bit
                                               for bit -- pinned against the executed graph's tensors by
                                               tests/test_gpu_graph_golden.py.
                                            (1, the per-tap order, and 3, chunks from +0.0 added left to right, were removed
                                             with ABI 5: measured to train worse, DESIGN.md section 10.1.  AIR_EINVAL.)      */
    /* optional: workgroup (0,0) also does air_finalize's batch means (train step: saves a launch).
     * fin_scalars == NULL disables; loss_item is the [B] output of air_write_fwd */
    const float* fin_loss_item; const int32_t* fin_targets; const int32_t* fin_digits;
    float* fin_scalars;
    uint16_t* d_gen_pre16;               /* bf16 twin of d_gen_pre (nullable) */
    /* nullable (literal 2 only): workgroup i of the launch computes item order[i] = t * B + b instead of item i -- the
     * longest-first permutation of air_write_fwd_t.wb_order (the hardware hands workgroups out in launch order).  Results
     * do not depend on it.  With `order` the batch-mean finisher is the LAST workgroup of the launch. */
    const int32_t* order;
} air_write_bwd_t;
int air_write_bwd(const air_write_bwd_t* a, void* stream);
/* the kernel function air_write_bwd dispatches this descriptor to, as rocprofv3 prints it (profiling tools) */
int air_write_bwd_kernel_name(const air_write_bwd_t* a, char* buf, int n);

/* ---- VAE bottleneck, one launch per direction (vae.py:16-34) ------------------
 * Forward = the last recognition product and the first generative layer:
 *   ml [M,2Z] = X [M,K1].Wml [K1,2Z] + bml  (mean | log-variance, vae.py:16-20)
 *   z  [M,Z]  = mean + eps * sqrt(exp(lv))                          (vae.py:22-24)
 *   g  [M,H]  = softplus(z.Wg [Z,H] + bg)                           (vae.py:26-30)
 * i.e. air_gemm(AIR_EPI_REPARAM_FWD) followed by air_gemm(bias, AIR_ACT_SOFTPLUS) without the second
 * launch and without z's round trip through memory.  Operands are rounded to bf16 for the MFMAs
 * (as air_gemm precision 1 does), accumulation is fp32.
 * Backward = their data gradients:
 *   d_z = dG [M,H].Wg^T; d_ml [M,2Z] as AIR_EPI_REPARAM_BWD writes it (ml, eps, att mask, dyn);
 *   d_x [M,K1] = (d_ml.Wml^T) * softplus'(x)    (x = the saved activation X, AIR_GRAD_SOFTPLUS)
 * Limits: K1 == 256 (forward) / H == 256 (backward), Z <= 64 and even, 16-byte aligned operands;
 * AIR_ELIMIT / AIR_EALIGN otherwise (callers fall back to the two air_gemm launches). */
typedef struct {
    const float* X; const float* Wml; const float* bml; const float* eps; const float* Wg; const float* bg;
    float* ml; float* z; float* g;
    int32_t M, K1, Z, H, ldx;
    uint16_t* z16; uint16_t* g16;        /* bf16 twins of z / g written next to them (nullable) */
    const uint16_t* X16; const uint16_t* Wml16; const uint16_t* Wg16;   /* bf16 twins of the three operands (all or none;
                                          * X16 16-byte aligned with ldx % 8 == 0, the weights 8-byte aligned): read instead of
                                          * the fp32 arrays, same results */
    int32_t exact_fp32;                  /* 1: exact fp32 products (v_mfma_f32_16x16x4_f32) on the fp32 operands -- the fp32
                                          * path; twins ignored / not written; additionally 2 Z <= 104 */
    int32_t ldz;                         /* row stride of z and z16 in elements; 0 = Z.  A stride that is a multiple of 4
                                          * (Z = 50 -> 52) lets the weight gradient of the first generative layer read z's
                                          * twin in 8-byte pieces (air_wgrad_t: lda % 4 == 0, lda >= M rounded up to 4).  Only
                                          * columns < Z of a row are written: the pad keeps what the caller put there */
} air_bottleneck_fwd_t;
typedef struct {
    const float* dG; const float* Wg; const float* ml; const float* eps;
    const float* att;                    /* [M, AIR_ATT_STRIDE]: AIR_ATT_MASK gates the KL gradient */
    const float* dyn; const float* Wml; const float* x;
    float* d_ml; float* d_x;
    int32_t M, K1, Z, H;
    uint16_t* d_ml16; uint16_t* d_x16;   /* bf16 twins of d_ml / d_x written next to them (nullable) */
    const uint16_t* dG16; const uint16_t* Wg16; const uint16_t* Wml16;  /* bf16 twins of the three operands (all or none) */
    int32_t exact_fp32;                  /* as in air_bottleneck_fwd_t */
} air_bottleneck_bwd_t;
int air_vae_bottleneck_fwd(const air_bottleneck_fwd_t* a, void* stream);
int air_vae_bottleneck_bwd(const air_bottleneck_bwd_t* a, void* stream);

/* ---- reconstruction loss (air_model.py:580-593) + its gradient ---------------- */
int air_bce_fwd_bwd(const float* images, const float* run_recon, const float* dyn,
                    float* recon /*clipped [B,D]*/, float* rec_loss /*[B]*/,
                    float* d_recon /*[B,D] nullable*/, int B, int D, void* stream);
/* loss = mean(run_loss + rec_loss), accuracy = mean(target == digits)  (:597-611)
 * scalars[0] = loss, scalars[1] = accuracy */
int air_finalize(const float* run_loss, const float* rec_loss, const int32_t* targets,
                 const int32_t* digits, float* loss_per_item, float* scalars, int B, void* stream);

/* ---- step prologue: annealing schedules + Philox noise ------------------------
 * Evaluates `nsched` schedules at istate[GLOBAL_STEP] into dyn, and fills
 * normals[n_normal] ~ N(0,1), uniforms[n_uniform] ~ U[0,1) from
 * Philox4x32-10(key = seed, counter = (index, global_step + call_salt)).  Optionally also writes
 * twin_dst[i] = bf16(twin_src[i]), i < twin_n: the bf16 twin of the image batch (the caller's fp32 tensor; it has no
 * producing kernel of ours) that the input-weight gradient reads at the end of the step (air_wgrad_t.A16). */
int air_step_begin(const air_schedule_t* sched /*device*/, int nsched, float* dyn,
                   const int32_t* istate, float* normals, int64_t n_normal,
                   float* uniforms, int64_t n_uniform, uint64_t seed,
                   const float* twin_src /*nullable*/, uint16_t* twin_dst, int64_t twin_n, void* stream);

/* ---- optimizer (air_model.py:651-694): clip_by_global_norm + TF1.3 ApplyAdam --
 * partials: scratch [>= air_optim_num_partials(n)] floats.
 * air_grad_sqnorm also increments istate[GLOBAL_STEP] (apply_gradients, :692). */
int air_optim_num_partials(int64_t n);
int air_grad_sqnorm(const float* grads, int64_t n, float* partials, int32_t* istate, void* stream);
int air_adam_clip_step(float* params, const float* grads, float* m, float* v, int64_t n,
                       const float* partials, int npartials, const float* dyn, const int32_t* istate,
                       float grad_prescale /* e.g. 1/world_size */, float beta1, float beta2,
                       float epsilon, uint16_t* bf16_shadow /*nullable*/, float* gnorm_out /*nullable*/,
                       void* stream);
/* air_adam_clip_step that ALSO maintains the panel-blocked shadows: every updated variable that lies
 * in a described matrix is written to its panel position (and to bf16_shadow unless the matrix is `exclusive`). */
int air_adam_clip_step_panels(float* params, const float* grads, float* m, float* v, int64_t n,
                              const float* partials, int npartials, const float* dyn, const int32_t* istate,
                              float grad_prescale, float beta1, float beta2, float epsilon,
                              uint16_t* bf16_shadow /*nullable*/, const air_panel_t* panels /*HOST array*/, int npanels,
                              uint16_t* panel_shadow, float* gnorm_out /*nullable*/, void* stream);

/* ---- input pipeline: tf.train.shuffle_batch on the device (the reference's multi_mnist.py:228-249) -------------
 * A RandomShuffleQueue of `capacity` record indices resident in HBM over the epoch-repeating record stream
 * 0, 1, .., n_records-1, 0, 1, ..  (string_input_producer([file], num_epochs) + ONE TFRecordReader: file order, every
 * epoch the same -- training.py:76-81).  The reader threads are far faster than a train step, so the queue is modelled AT
 * CAPACITY whenever a batch is taken.  air_shuffle_batch_dequeue then does what RandomShuffleQueue::TryDequeueMany does
 * for each of the `batch` elements in turn:  index = r_k mod size;  emit queue[index];  queue[index] = queue[size-1];
 * size -= 1  (uniform pick, swap with the back, pop) -- and the `batch` freed slots at the back are refilled from the
 * stream in order (enqueue appends at the back).  r_k = word k mod 4 of Philox4x32-10(key = seed, counter = (dequeue
 * number lo, hi, k / 4, 0x53485546)).  TF's own queue is unseeded here (seed 0 -> nondeterministic): the distribution is
 * the contract, and tests/test_shuffle_queue.py holds a numpy model of the queue that this kernel matches pick for pick.
 * state[0] = records enqueued so far (the stream position), state[1] = batches dequeued so far.
 * One workgroup per call; the picks of a batch are resolved in parallel (the content of a slot at pick k is found by walking
 * the earlier picks of the batch backwards: csrc/air_input.hip), the same picks as the sequential queue.  Limits:
 * capacity * 4 + batch * 8 <= 48 KB (capacity 10 640, batch 64: 43 KB -- air_shuffle_batch_dequeue_many stages the queue
 * in LDS), batch a multiple of 4 and <= 1024, capacity - batch >= min_after_dequeue >= 0 (the reference: 10 000).  No
 * allocation, no synchronisation;
 * both calls are stream work and capturable. */
typedef struct {
    int32_t* queue;            /* [capacity] record indices in the queue */
    int64_t* state;            /* [2] */
    int32_t* picks;            /* [batch] out: the records of this batch, in dequeue order */
    int32_t capacity, batch, min_after_dequeue, n_records;
    uint64_t seed;
} air_shuffle_batch_t;
/* queue <- the first `capacity` records of the stream, state <- (capacity, 0) */
int air_shuffle_batch_init(const air_shuffle_batch_t* q, void* stream);
int air_shuffle_batch_dequeue(const air_shuffle_batch_t* q, void* stream);
/* `n_batches` consecutive dequeues in one launch, pick for pick what n_batches calls of air_shuffle_batch_dequeue make:
 * picks_out[k * batch + i] = pick i of the k-th of them (q->picks is not written).  In training.py it is the first launch
 * of every hipGraph replay of n_batches train steps, whose row gathers read picks_out (multi_mnist.ShuffleBatchQueue). */
int air_shuffle_batch_dequeue_many(const air_shuffle_batch_t* q, int n_batches, int32_t* picks_out, void* stream);
/* the decoded tensors of a dequeued batch (read_and_decode, multi_mnist.py:228-238, after the queue):
 * out_images[i] = images[picks[i]] ([batch, D] fp32 rows, D % 4 == 0, 16-byte aligned), out_digits[i] = digits[picks[i]]
 * (out_digits / digits nullable together). */
int air_batch_gather(const float* images, const int32_t* digits, const int32_t* picks, float* out_images,
                     int32_t* out_digits, int batch, int D, void* stream);

/* ---- numeric summaries (the reference's air_model.py:160-209, 608-632; evaluated by training.py:169-200 on the test
 * model every 50 iterations) as one launch over the outputs of a forward pass --------------------------------------
 * out[0] = loss, out[1] = accuracy (copies of `scalars`), then one ROW of max_digits + 2 masked means per summarised
 * quantity -- the images with exactly 0 .. max_digits target digits, then all images ("_<i>_dig", "_all_dig"; an empty
 * group is NaN as tf.reduce_mean of an empty tensor) -- in the order the reference appends them:
 *   rows 0..3          steps (= rec_num_digits as float), rec_loss, digit_acc (digits == targets), total_loss (loss_item)
 *   rows 4 + q * N + i quantity q of step i + 1, q = scale, z_pres_prob (all_steps=True), z_pres_kl (one_more_step=True),
 *                      scale_kl, shift_kl, vae_kl; masked to the images with rec_num_digits > i (> i - 1: one_more_step; no
 *                      mask: all_steps); a step the while_loop did not reach (i >= T', derived from AIR_ATT_MASK) is the
 *                      zero padding of :187.
 * air_summaries_count(N, max_digits) = 2 + (4 + 6 N)(max_digits + 2) floats.  N <= 16, max_digits <= 6.  Stream work,
 * capturable, deterministic (fp64 sums in a fixed order, rounded once). */
#define AIR_MAX_STEPS_SUMMARY 16
typedef struct {
    const float* att;            /* [N,B,AIR_ATT_STRIDE] of the pass */
    const int32_t* targets;      /* [B] target_num_digits */
    const int32_t* digits;       /* [B] rec_num_digits */
    const float* rec_loss;       /* [B] reconstruction_loss */
    const float* loss_item;      /* [B] loss before the batch mean (:598-600) */
    const float* scalars;        /* [2] loss, accuracy (air_finalize) */
    float* out;                  /* [air_summaries_count(N, max_digits)] */
    int32_t B, N, max_digits;
} air_summaries_t;
int air_summaries_count(int N, int max_digits);
int air_summaries(const air_summaries_t* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AIR_HIP_H */
