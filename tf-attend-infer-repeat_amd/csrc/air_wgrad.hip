// Grouped weight-gradient launch: every dW = A^T . dY of the train step (the
// MatMul_grad/BiasAdd_grad nodes of all 36 variables) in ONE kernel.
//
// Weights are shared across the N time steps, so each dW contracts over all
// N*B (step, image) rows: K is tiny (64..192) while M x N is the size of the
// weight matrix -- an outer-product-shaped GEMM.  A 64 x 64 output tile is owned
// by one workgroup; its 4 waves own disjoint 32 x 32 quadrants and run the full
// K loop themselves (no split-K, no cross-wave reduction).  Operand chunks of
// 32 rows are staged through double-buffered LDS with the next chunk's 16-byte
// loads in flight under the MFMAs.  The bias gradient (column sums of dY) is
// taken from the dY chunk already in LDS by the tiles of block-row 0, so no
// separate reduction launch exists.  The ten problems of the step are
// independent: one launch fills the chip instead of ten latency-bound ones.
#include "air_common.h"
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int THREADS = 256;
constexpr int BT = 64;          // output tile (both dims)
constexpr int KC = 32;          // rows per staged chunk
constexpr int LS = BT + 16;     // LDS row stride: = 16 (mod 32) dwords -> conflict-free fragment reads, 16-B aligned
constexpr int MAXP = 12;

struct Prob {
    const float* A; const float* dY; float* dW; float* db;
    const unsigned short* A16; const unsigned short* dY16;      // bf16 twins of A / dY (nullable): see tile_bf16_tw
    int M, N, K, lda, ldb, ldc;
    int head_pack, Hs, Hh, Hz;
    int tiles_n, first_block;
};
// first[i] = first workgroup of problem i (INT_MAX past `count`): kept apart from the descriptors so that
// ONE wide scalar load fetches all of them and the owner is found without a chain of dependent loads
struct Table { int count; int total_blocks; int first[MAXP]; Prob p[MAXP]; };

// One float4 of a row-major operand, zero outside [rows x cols], in two branch-free halves: fetch4
// issues the load(s) with out-of-range accesses redirected to element 0, mask4 zeroes what was out of
// range.  Callers issue ALL fetches of a batch before the first mask: a run-time branch around a
// load (or a consumer right behind it) makes the compiler wait on the spot, which serialises the
// 16-24 loads a thread should have in flight (12 us instead of 3 for the ragged-edge tiles).
// VEC: base 16-byte aligned and ld % 4 == 0 -- a quad that starts inside a row then lies inside
// the padded row, so the 16-byte load is issued even when its last columns are past `cols`.
template <bool VEC>
__device__ __forceinline__ float4 fetch4(const float* __restrict__ base, int ld, int row, int col, int rows, int cols) {
    const bool okr = row < rows;
    const unsigned at = (unsigned)row * (unsigned)ld + (unsigned)col;
    if (VEC) return *reinterpret_cast<const float4*>(base + ((okr && col < cols) ? at : 0u));
    float4 v;
    v.x = base[(okr && col < cols) ? at : 0u];
    v.y = base[(okr && col + 1 < cols) ? at + 1u : 0u];
    v.z = base[(okr && col + 2 < cols) ? at + 2u : 0u];
    v.w = base[(okr && col + 3 < cols) ? at + 3u : 0u];
    return v;
}
__device__ __forceinline__ float4 mask4(float4 t, int row, int col, int rows, int cols) {
    const bool okr = row < rows;
    float4 v;
    v.x = (okr && col < cols) ? t.x : 0.f;
    v.y = (okr && col + 1 < cols) ? t.y : 0.f;
    v.z = (okr && col + 2 < cols) ? t.z : 0.f;
    v.w = (okr && col + 3 < cols) ? t.w : 0.f;
    return v;
}


// sum of squares of everything this workgroup stored (tf.global_norm terms): one partial per
// workgroup, reduced again in fixed order by the Adam kernel; workgroup 0 also counts the step
// (apply_gradients(global_step=...), air_model.py:692-694)
__device__ __forceinline__ void publish_sq(float sq, float* sq_partials, int32_t* istate) {
    __shared__ float sq_red[4];
    sq = air_block_sum_256(sq, sq_red);
    if (threadIdx.x == 0) {
        sq_partials[blockIdx.x] = sq;
        if (blockIdx.x == 0 && istate) istate[AIR_IST_GLOBAL_STEP] += 1;
    }
}

// fp32 tile: the 64 x 64 block (m0, n0) of pr.A^T . pr.dY, left as 64 rows of LS floats at &As[0][0]
// (barrier-synchronised).  Returns this thread's share of the squared bias gradient it stored.
template <int NCH>      // chunks of 32 rows in flight per round trip (8: K <= 256 in one)
__device__ __forceinline__ float tile_f32(const Prob& pr, int m0, int n0, float (*As)[KC * LS], float (*Bs)[KC * LS])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;        // this wave's quadrant
    const bool vecA = ((pr.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(pr.A) & 15) == 0);
    const bool vecB = ((pr.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(pr.dY) & 15) == 0);

    // copy the problem descriptor out of the kernel-argument table once
    const float* __restrict__ Ap = pr.A;
    const float* __restrict__ Yp = pr.dY;
    const int M = pr.M, N = pr.N, K = pr.K, lda = pr.lda, ldb = pr.ldb;
    float* db = pr.db;
    const int head_pack = pr.head_pack;

    // staging map: 32 rows x 16 float4 columns = 512 float4 per operand chunk, 2 per thread
    const int srow = tid >> 4, scol = (tid & 15) * 4;             // rows srow and srow + 16
    float4 ra[NCH][2], rb[NCH][2];
    auto fetch = [&](int set, int k0, auto vec_t) __attribute__((always_inline)) {
        constexpr bool V = decltype(vec_t)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + srow + 16 * h;
            ra[set][h] = fetch4<V>(Ap, lda, k, m0 + scol, K, M);
            rb[set][h] = fetch4<V>(Yp, ldb, k, n0 + scol, K, N);
        }
    };
    // interior tile, whole chunks: nothing to clamp or mask
    const bool plain = vecA && vecB && m0 + BT <= M && n0 + BT <= N && (K % KC) == 0;
    auto fetch_plain = [&](int set, int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned k = (unsigned)(k0 + srow + 16 * h);
            ra[set][h] = *reinterpret_cast<const float4*>(Ap + (k * (unsigned)lda + (unsigned)(m0 + scol)));
            rb[set][h] = *reinterpret_cast<const float4*>(Yp + (k * (unsigned)ldb + (unsigned)(n0 + scol)));
        }
    };
    auto stage = [&](int set, int buf, int k0, auto plain_t) __attribute__((always_inline)) {
        constexpr bool PL = decltype(plain_t)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + srow + 16 * h;
            *reinterpret_cast<float4*>(&As[buf][(srow + 16 * h) * LS + scol]) = PL ? ra[set][h] : mask4(ra[set][h], k, m0 + scol, K, M);
            *reinterpret_cast<float4*>(&Bs[buf][(srow + 16 * h) * LS + scol]) = PL ? rb[set][h] : mask4(rb[set][h], k, n0 + scol, K, N);
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float colsum = 0.0f;        // wave 0 of block-row 0: column n0 + lane of dY (or of A for the head units)
    const bool do_bias = (db != nullptr) && (wave == 0) && (head_pack ? (n0 == 0) : (m0 == 0));

    auto compute = [&](int buf) {
        const float* as = As[buf];
        const float* bs = Bs[buf];
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            const int kk = ks * 4 + (lane >> 4);
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = as[kk * LS + wm + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = bs[kk * LS + wn + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (do_bias) {
            const float* cs = head_pack ? as : bs;                 // zero rows beyond K
#pragma unroll 8
            for (int k = 0; k < KC; ++k) colsum += cs[k * LS + lane];
        }
    };
    // ALL loads of up to NCH chunks are issued before the first is consumed (one memory round
    // trip), then one barrier per chunk: stage(c+1) only overwrites the buffer every wave finished
    // reading before it passed the barrier of iteration c.
    for (int ks0 = 0; ks0 < K; ks0 += NCH * KC) {
        if (ks0 > 0) __syncthreads();
        if (plain) {                            // ONE uniform branch around the whole batch of loads
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (ks0 + c * KC < K) fetch_plain(c, ks0 + c * KC);
        } else if (vecA && vecB) {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (ks0 + c * KC < K) fetch(c, ks0 + c * KC, std::true_type{});
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (ks0 + c * KC < K) fetch(c, ks0 + c * KC, std::false_type{});
        }
        if (plain) {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (ks0 + c * KC < K) {
                    stage(c, c & 1, ks0 + c * KC, std::true_type{});
                    __syncthreads();
                    compute(c & 1);
                }
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (ks0 + c * KC < K) {
                    stage(c, c & 1, ks0 + c * KC, std::false_type{});
                    __syncthreads();
                    compute(c & 1);
                }
        }
    }

    // C/D map row = (lane>>4)*4 + q, col = lane&15.  The tile goes through LDS so that every store
    // instruction of the epilogue touches whole 256-byte rows (full cache lines).
    __syncthreads();
    float* Ct = &As[0][0];                      // 64 x LS floats = 20 KB: spans As[0..1]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                Ct[(wm + i * 16 + (lane >> 4) * 4 + q) * LS + wn + j * 16 + (lane & 15)] = acc[i][j][q];
    __syncthreads();
    float bias_sq = 0.0f;
    if (do_bias) {
        if (!head_pack) { if (n0 + lane < N) { db[n0 + lane] = colsum; bias_sq = colsum * colsum; } }
        else if (lane < 7) { db[lane] = colsum; bias_sq = colsum * colsum; }
    }
    return bias_sq;
}

// Epilogue of the weight-gradient kernels: the tile in Ct goes to pr.dW (whole rows per store
// instruction) and its sum of squares to the caller.  pr.dW == NULL: nothing is stored, only the
// sum of squares is taken (a gradient that air_adam_clip_step_factored rebuilds from its factors).
__device__ __forceinline__ float store_tile(const Prob& pr, int m0, int n0, const float* Ct, float sq)
{
    const int tid = threadIdx.x;
    const int M = pr.M, N = pr.N, ldc = pr.ldc;
    float* dW = pr.dW;
    if (!pr.head_pack) {
        const bool vecC = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(dW) & 15) == 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (tid >> 4) + 16 * r, col = (tid & 15) * 4;
            const int m = m0 + row, n = n0 + col;
            if (m >= M) continue;
            const float4 t = *reinterpret_cast<const float4*>(&Ct[row * LS + col]);
            float* dst = dW + (size_t)m * ldc + n;
            if (vecC && n + 3 < N) { if (dW) *reinterpret_cast<float4*>(dst) = t; sq += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w); }
            else {
                if (n < N) { if (dW) dst[0] = t.x; sq += t.x * t.x; }
                if (n + 1 < N) { if (dW) dst[1] = t.y; sq += t.y * t.y; }
                if (n + 2 < N) { if (dW) dst[2] = t.z; sq += t.z * t.z; }
                if (n + 3 < N) { if (dW) dst[3] = t.w; sq += t.w * t.w; }
            }
        }
    } else {
        // head output units (air_model.py:294-316, 376): A = d_out7 [K,8], dY = hid [K,HT];
        // unit o only owns the hidden segment of its head: dW = wout[o][n - off], db = bout[o] = sum_k d_out7[k][o]
        const int wid[5] = {pr.Hs, pr.Hs, pr.Hh, pr.Hh, pr.Hz};
        const int head[7] = {0, 1, 2, 2, 3, 3, 4};
        for (int it = tid; it < 7 * BT; it += THREADS) {
            const int o = it / BT, col = it % BT, n = n0 + col;
            int off = 0;
            for (int h = 0; h < head[o]; ++h) off += wid[h];
            if (m0 == 0 && n < N && n >= off && n < off + wid[head[o]]) {
                const float t = Ct[o * LS + col];
                dW[(size_t)o * ldc + (n - off)] = t;
                sq += t * t;
            }
        }
    }
    return sq;
}

__device__ __forceinline__ const Prob& find_tile(const Table& tab, int block, int& m0, int& n0) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAXP; ++i) pi += (block >= tab.first[i]) ? 1 : 0;
    const Prob& pr = tab.p[pi];
    const int local = block - pr.first_block;
    m0 = (local / pr.tiles_n) * BT; n0 = (local % pr.tiles_n) * BT;
    return pr;
}

__global__ __launch_bounds__(THREADS) void wgrad_grouped_kernel(Table tab, float* __restrict__ sq_partials, int32_t* __restrict__ istate)
{
    __shared__ __attribute__((aligned(16))) float As[2][KC * LS];
    __shared__ __attribute__((aligned(16))) float Bs[2][KC * LS];
    int m0, n0;
    const Prob& pr = find_tile(tab, blockIdx.x, m0, n0);
    const float bias_sq = tile_f32<6>(pr, m0, n0, As, Bs);      // K = 192 (3 steps x 64 images) in one round trip
    const float sq = store_tile(pr, m0, n0, &As[0][0], 0.0f) + bias_sq;
    if (sq_partials) publish_sq(sq, sq_partials, istate);
}


// ---------------------------------------------------------------------------
// bf16-operand variant (precision 1): operands are rounded to bf16 (RNE,
// v_cvt_pk_bf16_f32) on their way into LDS, products run on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  K sits on the slow (row)
// axis of both operands, the MFMA wants 8 consecutive k per lane: a thread
// loads the same 4 columns of 8 consecutive rows (8 x 16-B loads), packs one
// 16-B k-run per column and stores it into a [column][k] image whose 16-B slots
// are XOR-swizzled by the column index -- conflict-free for the 8-lane
// ds_write_b128 groups and for the 16-lane ds_read_b128 fragment groups.  All
// loads of up to 192 rows are in flight at once, one barrier before the MFMAs
// (no per-chunk barriers); the bias column sums are taken from the fp32
// registers before rounding.
// ---------------------------------------------------------------------------
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr int KB = 64;          // rows per LDS image
constexpr int NIMG_W = 3;       // images resident per round in the weight-gradient kernel: K <= 192 needs one round

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// bf16 tile: as tile_f32, operands rounded to bf16; Img = [operand][image][column][k] shorts of LDS,
// the fp32 tile is left at its start (64 rows of LS floats, barrier-synchronised).
template <int NIMG>     // LDS images (64 rows each) resident per round; Img holds max(2 * NIMG * 8 KB, 20 KB)
__device__ __forceinline__ float tile_bf16(const Prob& pr, int m0, int n0, unsigned short* Img)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const float* __restrict__ Ap = pr.A;
    const float* __restrict__ Yp = pr.dY;
    const int M = pr.M, N = pr.N, K = pr.K, lda = pr.lda, ldb = pr.ldb;
    float* db = pr.db;
    const int head_pack = pr.head_pack;

    // staging role: threads 0..127 own operand A, 128..255 own dY; g = k-run (8 rows), q = column quad
    const int op = tid >> 7, g = tid & 7, q = (tid & 127) >> 3;
    const float* src = op ? Yp : Ap;
    const int ld = op ? ldb : lda, cols = op ? N : M, c0 = (op ? n0 : m0) + 4 * q;
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    unsigned short* img = Img + (size_t)op * NIMG * BT * KB;
    const bool bias_block = (db != nullptr) && (head_pack ? (n0 == 0) : (m0 == 0));
    const bool bias_thread = bias_block && (op == (head_pack ? 0 : 1));
    float csum[4] = {0.f, 0.f, 0.f, 0.f};

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    AIR_STAMP(1);
    for (int kr = 0; kr < K; kr += NIMG * KB) {
        if (kr > 0) __syncthreads();                     // every wave is done reading the previous images
        float4 v[NIMG][8];
        // wave-uniform choices, each ONE branch around the whole batch of loads
        const bool inside = (op ? n0 : m0) + BT <= cols;                  // no ragged column edge
        if (vec && inside && (K % KB) == 0) {
            // interior tile, whole images: uniform base + 32-bit byte offsets, nothing to mask
            const char* base = reinterpret_cast<const char*>(src);
            const unsigned step = (unsigned)ld * 4u;
            const unsigned off0 = (unsigned)(kr + g * 8) * step + (unsigned)c0 * 4u;
#pragma unroll
            for (int c = 0; c < NIMG; ++c)
                if (kr + c * KB < K) {
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        v[c][r] = *reinterpret_cast<const float4*>(base + (off0 + (unsigned)(c * KB + r) * step));
                }
        } else if (vec) {
#pragma unroll
            for (int c = 0; c < NIMG; ++c)
                if (kr + c * KB < K) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[c][r] = fetch4<true>(src, ld, kr + c * KB + g * 8 + r, c0, K, cols);
                }
        } else {
#pragma unroll
            for (int c = 0; c < NIMG; ++c)
                if (kr + c * KB < K) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[c][r] = fetch4<false>(src, ld, kr + c * KB + g * 8 + r, c0, K, cols);
                }
        }
        AIR_STAMP(2);
        if (!(vec && inside && (K % KB) == 0)) {
#pragma unroll
            for (int c = 0; c < NIMG; ++c)
                if (kr + c * KB < K) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[c][r] = mask4(v[c][r], kr + c * KB + g * 8 + r, c0, K, cols);
                }
        }
#pragma unroll
        for (int c = 0; c < NIMG; ++c)
            if (kr + c * KB < K) {
                if (bias_thread) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) { csum[0] += v[c][r].x; csum[1] += v[c][r].y; csum[2] += v[c][r].z; csum[3] += v[c][r].w; }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = 4 * q + j;
                    auto e = [&](int r) { const float4& t = v[c][r]; return j == 0 ? t.x : j == 1 ? t.y : j == 2 ? t.z : t.w; };
                    uint4 w;
                    w.x = pack_bf16(e(0), e(1)); w.y = pack_bf16(e(2), e(3));
                    w.z = pack_bf16(e(4), e(5)); w.w = pack_bf16(e(6), e(7));
                    *reinterpret_cast<uint4*>(&img[(size_t)c * BT * KB + col * KB + ((g ^ (col & 7)) << 3)]) = w;
                }
            }
        AIR_STAMP(3);
        __syncthreads();
        AIR_STAMP(4);
#pragma unroll
        for (int c = 0; c < NIMG; ++c)
            if (kr + c * KB < K) {
                const unsigned short* ai = Img + (size_t)c * BT * KB;
                const unsigned short* bi = Img + (size_t)(NIMG + c) * BT * KB;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int slot = ks * 4 + (lane >> 4);
                    bf16x8 av[2], bv[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int col = wm + i * 16 + (lane & 15);
                        av[i] = *reinterpret_cast<const bf16x8*>(&ai[col * KB + ((slot ^ (col & 7)) << 3)]);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int col = wn + j * 16 + (lane & 15);
                        bv[j] = *reinterpret_cast<const bf16x8*>(&bi[col * KB + ((slot ^ (col & 7)) << 3)]);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
                }
            }
    }

    float sq = 0.0f;
    // bias: reduce the 8 k-runs (lanes g = 0..7 are contiguous) of each column quad
    if (bias_block) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = csum[j];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
            csum[j] = s;
        }
        if (bias_thread && g == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = 4 * q + j;
                if (!head_pack) { if (n0 + col < N) { db[n0 + col] = csum[j]; sq += csum[j] * csum[j]; } }
                else if (col < 7) { db[col] = csum[j]; sq += csum[j] * csum[j]; }
            }
        }
    }

    // epilogue through LDS: whole 256-byte rows per store instruction
    AIR_STAMP(5);
    __syncthreads();
    float* Ct = reinterpret_cast<float*>(Img);           // 64 x LS floats = 20 KB <= 48 KB
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
                Ct[(wm + i * 16 + (lane >> 4) * 4 + qq) * LS + wn + j * 16 + (lane & 15)] = acc[i][j][qq];
    __syncthreads();
    return sq;
}

// bf16-TWIN tile: both operands are read from the bf16 twins their producers wrote (air_wgrad_t.A16 / dY16).
// K is the slow axis of both, i.e. both are "n-contiguous" for the MFMA: a 64-row image of an operand is
// copied as it lies into a [k][64 columns] LDS image (16- or 8-byte pieces, lane-linear rows of 128 bytes,
// no conversion, no register transpose) and the MFMA fragments -- 8 consecutive k per lane -- come out of
// gfx950's transpose read ds_read_b64_tr_b16 (tools/exp/tr_read.hip; the same scheme as the forward GEMM's
// row-major weights, air_gemm_bf16.hip).  Same bf16 values, same k order per wave as tile_bf16: the tile is
// bit-identical.  The bias gradient still comes from the fp32 dY (column sums before rounding, in tile_bf16's
// order) -- only the m0 == 0 tiles pay those loads.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bool twin_ok(const Prob& pr) {
    auto a8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; };
    return pr.A16 && pr.dY16 && !pr.head_pack && a8(pr.A16) && a8(pr.dY16) && (pr.lda & 3) == 0 && (pr.ldb & 3) == 0 &&
           (pr.M & 3) == 0 && (pr.N & 3) == 0;
}

template <int NIMG>
__device__ __forceinline__ float tile_bf16_tw(const Prob& pr, int m0, int n0, unsigned short* Img)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int M = pr.M, N = pr.N, K = pr.K, lda = pr.lda, ldb = pr.ldb;
    float* db = pr.db;
    const char* Ab = reinterpret_cast<const char*>(pr.A16);
    const char* Yb = reinterpret_cast<const char*>(pr.dY16);
    unsigned short* ImgA = Img;                              // [NIMG][64 k][64 m]
    unsigned short* ImgB = Img + NIMG * KB * BT;             // [NIMG][64 k][64 n]
    // 16-byte pieces when rows start 16-byte aligned (ld % 8 == 0), else 8-byte pieces (ld % 4 == 0)
    const bool a16 = (lda & 7) == 0 && (reinterpret_cast<uintptr_t>(pr.A16) & 15) == 0 && (M & 7) == 0;
    const bool b16 = (ldb & 7) == 0 && (reinterpret_cast<uintptr_t>(pr.dY16) & 15) == 0 && (N & 7) == 0;
    const bool bias_block = (db != nullptr) && (m0 == 0);
    // the bias threads' fp32 view of dY: g = k-run (8 rows), q = column quad -- tile_bf16's staging role of operand dY
    const int bg = tid & 7, bq = (tid & 127) >> 3;
    const bool bias_thread = bias_block && tid >= 128;
    float csum[4] = {0.f, 0.f, 0.f, 0.f};

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int T8 = NIMG * KB * 16 / THREADS;             // 8-byte pieces per thread per operand (two of them = one 16-byte piece)
    uint2 ra[T8], rb[T8];
    // all operand loads of one round, one wave-uniform branch per operand.  Rounds are software-pipelined: the loads of
    // round r+1 are issued as soon as round r's registers are in LDS, i.e. under its barrier and MFMAs (K > 192: the
    // 128x128 configuration contracts over 256 rows)
    auto issue_round = [&](int kr) __attribute__((always_inline)) {
        if (a16) {
#pragma unroll
            for (int i = 0; i < T8 / 2; ++i) {
                const int t = tid + THREADS * i, k = kr + (t >> 3), col = m0 + (t & 7) * 8;
                const bool ok = k < K && col < M;
                const uint4 x = *reinterpret_cast<const uint4*>(Ab + (ok ? ((unsigned)k * (unsigned)lda + (unsigned)col) * 2u : 0u));
                ra[2 * i] = ok ? make_uint2(x.x, x.y) : make_uint2(0u, 0u);
                ra[2 * i + 1] = ok ? make_uint2(x.z, x.w) : make_uint2(0u, 0u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < T8; ++i) {
                const int t = tid + THREADS * i, k = kr + (t >> 4), col = m0 + (t & 15) * 4;
                const bool ok = k < K && col < M;
                const uint2 x = *reinterpret_cast<const uint2*>(Ab + (ok ? ((unsigned)k * (unsigned)lda + (unsigned)col) * 2u : 0u));
                ra[i] = ok ? x : make_uint2(0u, 0u);
            }
        }
        if (b16) {
#pragma unroll
            for (int i = 0; i < T8 / 2; ++i) {
                const int t = tid + THREADS * i, k = kr + (t >> 3), col = n0 + (t & 7) * 8;
                const bool ok = k < K && col < N;
                const uint4 x = *reinterpret_cast<const uint4*>(Yb + (ok ? ((unsigned)k * (unsigned)ldb + (unsigned)col) * 2u : 0u));
                rb[2 * i] = ok ? make_uint2(x.x, x.y) : make_uint2(0u, 0u);
                rb[2 * i + 1] = ok ? make_uint2(x.z, x.w) : make_uint2(0u, 0u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < T8; ++i) {
                const int t = tid + THREADS * i, k = kr + (t >> 4), col = n0 + (t & 15) * 4;
                const bool ok = k < K && col < N;
                const uint2 x = *reinterpret_cast<const uint2*>(Yb + (ok ? ((unsigned)k * (unsigned)ldb + (unsigned)col) * 2u : 0u));
                rb[i] = ok ? x : make_uint2(0u, 0u);
            }
        }
    };
    AIR_STAMP(1);
    issue_round(0);
    AIR_STAMP(2);
    for (int kr = 0; kr < K; kr += NIMG * KB) {
        if (kr > 0) __syncthreads();
        // ---- straight into the [k][64] images (lane-linear rows)
        if (a16) {
#pragma unroll
            for (int i = 0; i < T8 / 2; ++i)
                *reinterpret_cast<uint4*>(&ImgA[(tid + THREADS * i) * 8]) = make_uint4(ra[2 * i].x, ra[2 * i].y, ra[2 * i + 1].x, ra[2 * i + 1].y);
        } else {
#pragma unroll
            for (int i = 0; i < T8; ++i) *reinterpret_cast<uint2*>(&ImgA[(tid + THREADS * i) * 4]) = ra[i];
        }
        if (b16) {
#pragma unroll
            for (int i = 0; i < T8 / 2; ++i)
                *reinterpret_cast<uint4*>(&ImgB[(tid + THREADS * i) * 8]) = make_uint4(rb[2 * i].x, rb[2 * i].y, rb[2 * i + 1].x, rb[2 * i + 1].y);
        } else {
#pragma unroll
            for (int i = 0; i < T8; ++i) *reinterpret_cast<uint2*>(&ImgB[(tid + THREADS * i) * 4]) = rb[i];
        }
        AIR_STAMP(3);
        if (kr + NIMG * KB < K) issue_round(kr + NIMG * KB);
        __syncthreads();
        AIR_STAMP(4);
        // ---- MFMAs: fragments through the transpose read (lane i of a 16-lane group hands in row 8g + i/4 (+4), column quad i%4)
        const int il = lane & 15;
#pragma unroll
        for (int c = 0; c < NIMG; ++c)
            if (kr + c * KB < K) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int krow = c * KB + ks * 32 + (lane >> 4) * 8 + (il >> 2);
                    const unsigned short* pa = &ImgA[krow * BT + wm + (il & 3) * 4];
                    const unsigned short* pb = &ImgB[krow * BT + wn + (il & 3) * 4];
                    bf16x8 av[2], bv[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa + i * 16));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa + i * 16 + 4 * BT));
                        av[i] = bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb + j * 16));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pb + j * 16 + 4 * BT));
                        bv[j] = bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
                }
            }
    }
    AIR_STAMP(5);
    if (bias_thread) {
        // fp32 column sums of dY (m0 == 0 tiles only) in tile_bf16's order: per k-run g the rows g*8 + r of image c, images
        // in order; the xor-tree over g follows.  After the MFMAs (the operand registers are free), the next image's 8 loads
        // in flight while this one is summed: ONE exposed round trip per tile (three sequential ones per round made the
        // bias tiles the long pole of the launch: 41 us at K = 1280).
        const bool vec = ((ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(pr.dY) & 15) == 0);
        const int nimg = (K + KB - 1) / KB;
        float4 cur[8], nxt[8];
        auto fetch_img = [&](float4 (&v)[8], int c) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                v[r] = vec ? fetch4<true>(pr.dY, ldb, c * KB + bg * 8 + r, n0 + 4 * bq, K, N)
                           : fetch4<false>(pr.dY, ldb, c * KB + bg * 8 + r, n0 + 4 * bq, K, N);
        };
        fetch_img(cur, 0);
        for (int c = 0; c < nimg; ++c) {
            if (c + 1 < nimg) fetch_img(nxt, c + 1);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float4 t = mask4(cur[r], c * KB + bg * 8 + r, n0 + 4 * bq, K, N);
                csum[0] += t.x; csum[1] += t.y; csum[2] += t.z; csum[3] += t.w;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) cur[r] = nxt[r];
        }
    }

    float sq = 0.0f;
    if (bias_block) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = csum[j];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
            csum[j] = s;
        }
        if (bias_thread && bg == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = 4 * bq + j;
                if (n0 + col < N) { db[n0 + col] = csum[j]; sq += csum[j] * csum[j]; }
            }
        }
    }
    __syncthreads();
    float* Ct = reinterpret_cast<float*>(Img);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
                Ct[(wm + i * 16 + (lane >> 4) * 4 + qq) * LS + wn + j * 16 + (lane & 15)] = acc[i][j][qq];
    __syncthreads();
    return sq;
}

__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void wgrad_grouped_bf16_kernel(Table tab, float* __restrict__ sq_partials, int32_t* __restrict__ istate)
{
    // [operand][image][column 0..63][k 0..63] bf16 = 2 x 3 x 8 KB; reused as the fp32 output tile
    __shared__ __attribute__((aligned(16))) unsigned short Img[2 * NIMG_W * BT * KB];
    int m0, n0;
    const Prob& pr = find_tile(tab, blockIdx.x, m0, n0);
    AIR_STAMP(0);
    // block-uniform: operands from their bf16 twins where the problem supplies usable ones
    const float bias_sq = twin_ok(pr) ? tile_bf16_tw<NIMG_W>(pr, m0, n0, Img) : tile_bf16<NIMG_W>(pr, m0, n0, Img);
    AIR_STAMP(6);
    const float sq = store_tile(pr, m0, n0, reinterpret_cast<const float*>(Img), bias_sq);
    AIR_STAMP(7);
    if (sq_partials) publish_sq(sq, sq_partials, istate);
    AIR_STAMP(8);
}

// ---------------------------------------------------------------------------
// clip + Adam with ONE gradient taken from its factors instead of from memory.
// dWx = X^T . (sum_t dgates) has rank <= B and is 64 % of all gradient elements
// (air_model.py:286: the [x, h] kernel's x rows): the weight-gradient launch
// only takes its sum of squares (dW == NULL above), and the workgroups here that
// own its 64 x 64 blocks rebuild each block with the same tile function (so the
// values are the ones a stored gradient would hold) and apply ApplyAdam to the
// matching block of var / m / v.  The other workgroups stream the rest of the flat
// buffer as adam_clip_kernel does.  -10 MB written, -10 MB read per step.
// ---------------------------------------------------------------------------
struct AdamFac {
    float* p; const float* g; float* m; float* v; long n;
    long roff, rlen;                   // region [roff, roff + rlen) of the flat buffers = the factored M x N block
    const float* partials; int npartials; const float* dyn; const int32_t* istate;
    float prescale, b1, b2, eps; float* gnorm_out; int tile_blocks;
};

template <int PREC>
__device__ __forceinline__ void adam_factored_body(const Prob& pr, const AdamFac& a, unsigned char* smem)
{
    __shared__ float red[4];
    const float omb1 = 1.0f - a.b1, omb2 = 1.0f - a.b2, eps = a.eps;
    const int tid = threadIdx.x;

    if ((int)blockIdx.x < a.tile_blocks) {
        const int m0 = ((int)blockIdx.x / pr.tiles_n) * BT, n0 = ((int)blockIdx.x % pr.tiles_n) * BT;
        const int M = pr.M, N = pr.N;                      // ldc == N, N % 4 == 0 (checked on the host)
        float* P = a.p + a.roff; float* Mo = a.m + a.roff; float* V = a.v + a.roff;
        // var / m / v of this block are requested first: they arrive while the tile is rebuilt
        float4 pp[4], mm[4], vv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + (tid >> 4) + 16 * r, n = n0 + (tid & 15) * 4;
            const size_t i = (m < M && n < N) ? (size_t)m * N + n : 0;
            pp[r] = *reinterpret_cast<const float4*>(P + i);
            mm[r] = *reinterpret_cast<const float4*>(Mo + i);
            vv[r] = *reinterpret_cast<const float4*>(V + i);
        }
        const AirAdamCoef cf = air_adam_coef(a.partials, a.npartials, a.dyn, a.istate, a.prescale, a.b1, a.b2, red);
        if (a.gnorm_out && blockIdx.x == 0 && tid == 0) *a.gnorm_out = cf.gnorm;
        if (PREC) tile_bf16<1>(pr, m0, n0, reinterpret_cast<unsigned short*>(smem));
        else tile_f32<2>(pr, m0, n0, reinterpret_cast<float (*)[KC * LS]>(smem), reinterpret_cast<float (*)[KC * LS]>(smem) + 2);
        const float* Ct = reinterpret_cast<const float*>(smem);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (tid >> 4) + 16 * r, col = (tid & 15) * 4;
            const int m = m0 + row, n = n0 + col;
            if (m >= M || n >= N) continue;
            const float4 gg = *reinterpret_cast<const float4*>(&Ct[row * LS + col]);
            const size_t i = (size_t)m * N + n;
            float* pa = &pp[r].x; float* ma = &mm[r].x; float* va = &vv[r].x; const float* ga = &gg.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) air_adam_update(pa[k], ma[k], va[k], ga[k], cf, omb1, omb2, eps);
            *reinterpret_cast<float4*>(P + i) = pp[r]; *reinterpret_cast<float4*>(Mo + i) = mm[r]; *reinterpret_cast<float4*>(V + i) = vv[r];
        }
        return;
    }
    // everything outside the region: the flat stream of adam_clip_kernel with the region skipped
    const AirAdamCoef cf = air_adam_coef(a.partials, a.npartials, a.dyn, a.istate, a.prescale, a.b1, a.b2, red);
    const long bid = (long)blockIdx.x - a.tile_blocks, nb = (long)gridDim.x - a.tile_blocks;
    const long n4 = a.n / 4, roff4 = a.roff / 4, rlen4 = a.rlen / 4, plain4 = n4 - rlen4;
    float4* p4 = reinterpret_cast<float4*>(a.p);
    const float4* g4 = reinterpret_cast<const float4*>(a.g);
    float4* m4 = reinterpret_cast<float4*>(a.m);
    float4* v4 = reinterpret_cast<float4*>(a.v);
    for (long j = bid * THREADS + tid; j < plain4; j += nb * THREADS) {
        const long i = j < roff4 ? j : j + rlen4;
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i];
        float* pa = &pp.x; float* ma = &mm.x; float* va = &vv.x; const float* ga = &gg.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) air_adam_update(pa[k], ma[k], va[k], ga[k], cf, omb1, omb2, eps);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    if (bid == 0 && tid < (int)(a.n - n4 * 4)) {           // the region is 4-aligned, so the tail is outside it
        const long i = n4 * 4 + tid;
        float pk = a.p[i], mk = a.m[i], vk = a.v[i];
        air_adam_update(pk, mk, vk, a.g[i], cf, omb1, omb2, eps);
        a.p[i] = pk; a.m[i] = mk; a.v[i] = vk;
    }
}

__global__ __launch_bounds__(THREADS) void adam_factored_kernel(Prob pr, AdamFac a)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * KC * LS * sizeof(float)];
    adam_factored_body<0>(pr, a, smem);
}
__global__ __launch_bounds__(THREADS) void adam_factored_bf16_kernel(Prob pr, AdamFac a)
{
    // one 64-row image per operand (16 KB: the contraction is over B rows) or the 20 KB fp32 tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[BT * LS * sizeof(float)];
    adam_factored_body<1>(pr, a, smem);
}

}  // namespace

AIR_STAMPS_READER(air_debug_stamps_wgrad)

static int fill_table(const air_wgrad_t* probs, int count, Table& tab, bool allow_null_dw) {
    if (!probs || count <= 0) return AIR_EINVAL;
    if (count > MAXP) return AIR_ELIMIT;
    tab.count = count;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        const air_wgrad_t& g = probs[i];
        if (!g.A || !g.dY || g.M <= 0 || g.N <= 0 || g.K <= 0) return AIR_EINVAL;
        if (!g.dW && (g.head_pack || !allow_null_dw)) return AIR_EINVAL;     // norm-only problems: plain layout, and only with sq_partials
        Prob& p = tab.p[i];
        p.A = g.A; p.dY = g.dY; p.dW = g.dW; p.db = g.db;
        p.A16 = g.A16; p.dY16 = g.dY16;
        p.M = g.M; p.N = g.N; p.K = g.K; p.lda = g.lda; p.ldb = g.ldb; p.ldc = g.ldc;
        p.head_pack = g.head_pack; p.Hs = g.Hs; p.Hh = g.Hh; p.Hz = g.Hz;
        p.tiles_n = (g.N + BT - 1) / BT;
        p.first_block = blocks;
        tab.first[i] = blocks;
        blocks += p.tiles_n * ((g.M + BT - 1) / BT);
    }
    for (int i = count; i < MAXP; ++i) { tab.p[i] = tab.p[0]; tab.first[i] = 0x7fffffff; }
    tab.total_blocks = blocks;
    return 0;
}

extern "C" int air_wgrad_num_blocks(const air_wgrad_t* probs, int count) {
    Table tab;
    const int rc = fill_table(probs, count, tab, true);
    return rc ? rc : tab.total_blocks;
}

extern "C" int air_wgrad_grouped(const air_wgrad_t* probs, int count, int precision,
                                 float* sq_partials, int32_t* istate, void* stream) {
    Table tab;
    const int rc = fill_table(probs, count, tab, sq_partials != nullptr);
    if (rc) return rc;
    if (precision != 0 && precision != 1) return AIR_EINVAL;
    if (precision == 1) hipLaunchKernelGGL(wgrad_grouped_bf16_kernel, dim3(tab.total_blocks), dim3(THREADS), 0, air_stream(stream), tab, sq_partials, istate);
    else hipLaunchKernelGGL(wgrad_grouped_kernel, dim3(tab.total_blocks), dim3(THREADS), 0, air_stream(stream), tab, sq_partials, istate);
    AIR_CHECK_LAUNCH();
    return 0;
}

extern "C" int air_adam_clip_step_factored(float* params, const float* grads, float* m, float* v, int64_t n,
                                           const air_wgrad_t* factored, int precision,
                                           const float* partials, int npartials, const float* dyn, const int32_t* istate,
                                           float grad_prescale, float beta1, float beta2, float epsilon,
                                           float* gnorm_out, void* stream) {
    if (!params || !grads || !m || !v || !partials || npartials <= 0 || !dyn || !istate || n <= 0 || !factored) return AIR_EINVAL;
    if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return AIR_EALIGN;
    if (precision != 0 && precision != 1) return AIR_EINVAL;
    Table tab;
    const int rc = fill_table(factored, 1, tab, false);
    if (rc) return rc;
    const air_wgrad_t& f = *factored;
    if (f.head_pack || f.db || f.ldc != f.N) return AIR_EINVAL;
    if ((f.N & 3) != 0) return AIR_EALIGN;
    const long roff = (long)(f.dW - grads), rlen = (long)f.M * f.N;
    if (f.dW < grads || roff + rlen > n) return AIR_EINVAL;      // the block must lie inside the flat buffer
    if ((roff & 3) != 0) return AIR_EALIGN;
    long plain = ((n - rlen) / 4 + THREADS - 1) / THREADS;
    if (plain < 1) plain = 1;
    if (plain > 2048) plain = 2048;
    AdamFac a{params, grads, m, v, (long)n, roff, rlen, partials, npartials, dyn, istate,
              grad_prescale, beta1, beta2, epsilon, gnorm_out, tab.total_blocks};
    const dim3 grid((unsigned)(tab.total_blocks + plain));
    if (precision == 1) hipLaunchKernelGGL(adam_factored_bf16_kernel, grid, dim3(THREADS), 0, air_stream(stream), tab.p[0], a);
    else hipLaunchKernelGGL(adam_factored_kernel, grid, dim3(THREADS), 0, air_stream(stream), tab.p[0], a);
    AIR_CHECK_LAUNCH();
    return 0;
}
