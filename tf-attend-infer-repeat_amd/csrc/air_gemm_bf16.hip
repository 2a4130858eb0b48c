// bf16-TWIN operand GEMM (precision 1 when the caller supplies bf16 twins of the operands).
//
// The fp32-operand kernel of air_gemm.hip (gemm_bf16v2_kernel) rounds both operand panels to bf16 on
// their way into LDS: every workgroup pulls 4-byte elements through its CU's L1 and spends most of its
// VALU instructions on v_cvt_pk_bf16_f32 and on transposing row-major weights into k-contiguous
// fragments.  Here the operands ARE bf16 in memory -- the producing epilogue / the Adam step wrote an
// RNE-rounded twin next to every fp32 array (air_gemm_t.A16 / B16) -- so
//   * a k-contiguous operand (activations; weights of a data-gradient GEMM, transB) is copied
//     global -> VGPR -> LDS in 16-byte pieces of 8 k, one ds_write_b128 into the same XOR-swizzled
//     [row][64 k] image the fp32-operand kernel builds: half the bytes, half the load instructions,
//     no conversion;
//   * an n-contiguous operand (row-major [K,N] weights of a forward GEMM) is copied AS IT LIES into a
//     [64 k][BN] image (lane-linear 16-byte stores, conflict free) and its MFMA fragments are read
//     with gfx950's transpose read ds_read_b64_tr_b16: each 16-lane group fetches a [4 k][16 n] block
//     and every lane receives the 4 k of its own column (semantics and bank behaviour measured in
//     tools/exp/tr_read.hip) -- two of them give the 8 consecutive k v_mfma_f32_16x16x32_bf16 wants.
//     No VALU instruction touches the operand;
//   * the number of 64-deep images per round R is a template parameter picked from K, so a K = 256
//     product does not issue (masked) loads for images it does not have, and K <= 1024 is ONE round.
// The rounding is the RNE the other kernel applies on the way into LDS, accumulation order over k,
// cross-wave reduction and epilogues are shared code (air_gemm_common.h): results are bit-identical to
// the fp32-operand bf16 path (tests/test_gpu_kernels.py::test_gemm_bf16_twins_bit_identical).
// A may stay fp32 (AF32: the hoisted x.Wx reads the caller's fp32 image batch).
#include "air_gemm_common.h"
#include <atomic>
#include <cstdlib>
#include <cstdio>

using namespace airg;

AIR_STAMPS_READER(air_debug_stamps_gemm_tw)

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint4 ldg16u(const char* base, unsigned off, bool ok) {
    const uint4 t = *reinterpret_cast<const uint4*>(base + (ok ? off : 0u));
    return ok ? t : make_uint4(0u, 0u, 0u, 0u);
}
__device__ __forceinline__ uint2 ldg8u(const char* base, unsigned off, bool ok) {
    const uint2 t = *reinterpret_cast<const uint2*>(base + (ok ? off : 0u));
    return ok ? t : make_uint2(0u, 0u);
}
__device__ __forceinline__ float4 ldg16f(const char* base, unsigned off, bool ok) {
    const float4 t = *reinterpret_cast<const float4*>(base + (ok ? off : 0u));
    return ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int TM, int TN, int R>
struct TwCfg {
    static constexpr int BM = 16 * TM, BN = 16 * TN, KB = 64;
    static constexpr int IMG = (BM + BN) * KB * 2;                       // bytes per image pair
    static constexpr int RED = 3 * TM * TN * 4 * 64 * 4;                 // bytes of the cross-wave reduction
    static constexpr int BYTES = (R * IMG > RED) ? R * IMG : RED;
};

template <int TM, int TN, bool TB, int EPI_, bool AF32, int R>
__global__ __launch_bounds__(THREADS) void gemm_bf16tw_kernel(Args a)
{
    using Cfg = TwCfg<TM, TN, R>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, KB = Cfg::KB;
    extern __shared__ __attribute__((aligned(16))) unsigned char Lds[];   // Cfg::BYTES
    unsigned short* ImgA = reinterpret_cast<unsigned short*>(Lds);       // [R][BM][64], 16-byte slots swizzled by row
    unsigned short* ImgB = ImgA + R * BM * KB;                           // TB: [R][BN][64] swizzled; else [R][64][BN]
    float* Red = reinterpret_cast<float*>(Lds);

    if ((int)blockIdx.z < a.job_on) {                    // block-uniform: the prologue's planes of workgroups (dispatched first)
        const long plane = (long)gridDim.x * gridDim.y;
        air_step_job_run(a.job, blockIdx.z * plane + (long)blockIdx.y * gridDim.x + blockIdx.x, plane * a.job_on);
        return;
    }
    const int nslab = (int)gridDim.z - a.job_on;
    const int zslab = (int)blockIdx.z - a.job_on;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tile_m, tile_n;
    xcd_tile(tile_m, tile_n);
    constexpr bool QUAD = EPI_ == EPI_LSTM_FWD_Q || EPI_ == AIR_EPI_LSTM_FWD0;   // 16 columns = 4 gates x 4 units (TN == 1, untransposed B)
    const int m0 = tile_m * BM, n0 = QUAD ? tile_n * 4 : tile_n * BN / TN * (a.gstride == 16 ? TN : 1);
    const int kbeg = zslab * a.kslab;
    const int kend = min(a.K, kbeg + a.kslab);

    AIR_STAMP(56);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Pre<TM, TN> pre;

    const char* Ab = AF32 ? reinterpret_cast<const char*>(a.A) : reinterpret_cast<const char*>(a.A16);
    // B from its panel-blocked twin when the caller has one (untransposed B only): a 16-column tile's rows are then
    // 32 contiguous bytes each, consecutive k contiguous -- whole cache lines instead of a quarter (plain panels) or
    // a sixteenth (four 8-byte gate pieces per row of a row-major LSTM kernel) of every line pulled through the CU
    const bool pnl = !TB && a.B16p != nullptr;
    const char* Bb = reinterpret_cast<const char*>(pnl ? a.B16p : a.B16);
    const unsigned pK16 = (unsigned)a.K * 16u;
    // tasks of one round, 16 bytes each.  k-contiguous bf16 operand: (image, row, slot g of 8 k) -- eight
    // consecutive lanes read one whole 128-byte row of an image.  fp32 A: (image, row, 4 k), rounded on the
    // way into LDS as the fp32-operand kernel does.  n-contiguous B: (image, k, 8 columns).
    constexpr int TA_N = AF32 ? (R * BM * 16 + THREADS - 1) / THREADS : (R * BM * 8 + THREADS - 1) / THREADS;
    constexpr int TBK_N = (R * BN * 8 + THREADS - 1) / THREADS;
    constexpr int TBN_N = (R * KB * (BN / 8) + THREADS - 1) / THREADS;
    constexpr int TBQ_N = (R * KB * 4 + THREADS - 1) / THREADS;           // QUAD: (image, k, gate) -> 8 bytes = 4 units
    uint4 va[AF32 ? 1 : TA_N];
    float4 vaf[AF32 ? TA_N : 1];
    uint4 vb[QUAD ? 1 : (TB ? TBK_N : TBN_N)];
    uint2 vq[QUAD ? TBQ_N : 1];

    auto issue_loads = [&](int kr) __attribute__((always_inline)) {
        if (AF32) {
#pragma unroll
            for (int i = 0; i < TA_N; ++i) {
                const int u = tid + THREADS * i;
                const int c = u / (BM * 16), row = (u / 16) % BM, hh = u & 15;
                const int gm = m0 + row, gk = kr + c * KB + hh * 4;
                const bool ok = (u < R * BM * 16) && gm < a.M && gk < kend;
                vaf[i] = ldg16f(Ab, ((unsigned)gm * (unsigned)a.lda + (unsigned)gk) * 4u, ok);
            }
        } else {
#pragma unroll
            for (int i = 0; i < TA_N; ++i) {
                const int u = tid + THREADS * i;
                const int c = u / (BM * 8), row = (u / 8) % BM, g = u & 7;
                const int gm = m0 + row, gk = kr + c * KB + g * 8;
                const bool ok = (u < R * BM * 8) && gm < a.M && gk < kend;
                va[i] = ldg16u(Ab, ((unsigned)gm * (unsigned)a.lda + (unsigned)gk) * 2u, ok);
            }
        }
        if (QUAD) {
#pragma unroll
            for (int i = 0; i < TBQ_N; ++i) {
                const int t = tid + THREADS * i;
                const int c = t / (KB * 4), k = (t >> 2) % KB, gate = t & 3;
                const int gn = gate * a.gstride + n0, gk = kr + c * KB + k;
                const bool ok = (t < R * KB * 4) && n0 < a.gwidth && gk < kend;
                const unsigned off = pnl ? (unsigned)(n0 >> 2) * pK16 + (unsigned)gk * 16u + (unsigned)gate * 4u
                                         : (unsigned)gk * (unsigned)a.ldb + (unsigned)gn;
                vq[i] = ldg8u(Bb, off * 2u, ok);
            }
        } else if (TB) {
#pragma unroll
            for (int i = 0; i < TBK_N; ++i) {
                const int u = tid + THREADS * i;
                const int c = u / (BN * 8), col = (u / 8) % BN, g = u & 7;
                const int j = col >> 4, cc = col & 15;
                const int gn = n0 + j * a.gstride + cc, cg = n0 + cc + (a.gstride == 16 ? j * 16 : 0);
                const int gk = kr + c * KB + g * 8;
                const bool ok = (u < R * BN * 8) && cg < a.gwidth && gn < a.N && gk < kend;
                vb[i] = ldg16u(Bb, ((unsigned)gn * (unsigned)a.ldb + (unsigned)gk) * 2u, ok);
            }
        } else {
#pragma unroll
            for (int i = 0; i < TBN_N; ++i) {
                const int t = tid + THREADS * i;
                const int c = t / (KB * (BN / 8)), k = (t / (BN / 8)) % KB, h = t % (BN / 8);
                const int col = h * 8, j = col >> 4, cc = col & 15;
                const int gn = n0 + j * a.gstride + cc, cg = n0 + cc + (a.gstride == 16 ? j * 16 : 0);
                const int gk = kr + c * KB + k;
                const bool ok = (t < R * KB * (BN / 8)) && cg < a.gwidth && gn < a.N && gk < kend;
                const unsigned off = pnl ? (unsigned)(gn >> 4) * pK16 + (unsigned)gk * 16u + (unsigned)(gn & 15)
                                         : (unsigned)gk * (unsigned)a.ldb + (unsigned)gn;
                vb[i] = ldg16u(Bb, off * 2u, ok);
            }
        }
    };
    // (tried: a mask-free form for interior tiles / full passes -- one per-thread base address, uniform pass offsets.
    // In-kernel stamps: the issue phase of the K = 256 kernels 0.84 -> 0.56 us, nothing at K = 784, and the in-graph
    // launch times did not move (the phase is bound by the CU's vector-memory path -- 64 B/clk, and a 16-column tile
    // uses 32 bytes of every 128-byte line of a row-major weight -- not by its ~15 VALU instructions per piece))
    auto store_images = [&](int) __attribute__((always_inline)) {
        if (AF32) {
#pragma unroll
            for (int i = 0; i < TA_N; ++i) {
                const int u = tid + THREADS * i;
                const int c = u / (BM * 16), row = (u / 16) % BM, hh = u & 15;
                uint2 w;
                w.x = pack_bf16(vaf[i].x, vaf[i].y); w.y = pack_bf16(vaf[i].z, vaf[i].w);
                if (u < R * BM * 16)
                    *reinterpret_cast<uint2*>(&ImgA[(c * BM + row) * KB + (((hh >> 1) ^ (row & 7)) << 3) + (hh & 1) * 4]) = w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < TA_N; ++i) {
                const int u = tid + THREADS * i;
                const int c = u / (BM * 8), row = (u / 8) % BM, g = u & 7;
                if (u < R * BM * 8) *reinterpret_cast<uint4*>(&ImgA[(c * BM + row) * KB + ((g ^ (row & 7)) << 3)]) = va[i];
            }
        }
        if (QUAD) {
#pragma unroll
            for (int i = 0; i < TBQ_N; ++i) {
                const int t = tid + THREADS * i;
                if (t < R * KB * 4) *reinterpret_cast<uint2*>(&ImgB[t * 4]) = vq[i];           // [c][k][gate][4 units]: lane-linear
            }
        } else if (TB) {
#pragma unroll
            for (int i = 0; i < TBK_N; ++i) {
                const int u = tid + THREADS * i;
                const int c = u / (BN * 8), col = (u / 8) % BN, g = u & 7;
                if (u < R * BN * 8) *reinterpret_cast<uint4*>(&ImgB[(c * BN + col) * KB + ((g ^ (col & 7)) << 3)]) = vb[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < TBN_N; ++i) {
                const int t = tid + THREADS * i;
                if (t < R * KB * (BN / 8)) *reinterpret_cast<uint4*>(&ImgB[t * 8]) = vb[i];      // [c][k][BN]: lane-linear
            }
        }
    };

    issue_loads(kbeg);
    // the epilogue's operands ride behind the first round's panels (same memory round trip)
    if (nslab == 1) epilogue_prefetch<TM, TN, EPI_>(a, pre, m0, n0, lane, wave);
    AIR_STAMP(57);
    for (int kr = kbeg; kr < kend; kr += R * KB) {
        if (kr > kbeg) __syncthreads();                                   // images of the previous round consumed
        store_images(kr);
        if (kr + R * KB < kend) issue_loads(kr + R * KB);
        __syncthreads();
        AIR_STAMP(58);
        // ---- MFMAs: wave w owns the images whose index within the slab is w (mod 4), whatever R is
        const int cfirst = (wave - ((kr - kbeg) / KB)) & 3;
#pragma unroll
        for (int cc = 0; cc < (R + 3) / 4; ++cc) {
            const int c = cfirst + 4 * cc;
            if (c < R && kr + c * KB < kend) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int slot = ks * 4 + (lane >> 4);
                    bf16x8 av[TM], bv[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int row = i * 16 + (lane & 15);
                        av[i] = *reinterpret_cast<const bf16x8*>(&ImgA[(c * BM + row) * KB + ((slot ^ (row & 7)) << 3)]);
                    }
                    if (TB) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int col = j * 16 + (lane & 15);
                            bv[j] = *reinterpret_cast<const bf16x8*>(&ImgB[(c * BN + col) * KB + ((slot ^ (col & 7)) << 3)]);
                        }
                    } else {
                        // transpose read: lane i of a 16-lane group hands in the address of row 8g + i/4 (+4),
                        // column quad i%4 of the [k][16] block; it receives k = 8g .. 8g+3 (+4) of column i
                        const int il = lane & 15;
                        const unsigned short* blk = &ImgB[(c * KB + ks * 32 + (lane >> 4) * 8 + (il >> 2)) * BN + (il & 3) * 4];
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(blk + j * 16));
                            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(blk + j * 16 + 4 * BN));
                            bv[j] = bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                        }
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
                }
            }
        }
    }
    AIR_STAMP(59);
    __syncthreads();                                                      // Red aliases the images
    reduce_waves<TM, TN>(acc, Red, lane, wave);
    AIR_STAMP(60);
    if (nslab > 1) {
        float* Cz = a.C + (size_t)zslab * a.slab_stride;
        for (int t = wave; t < TM * TN; t += 4) {
            const int i = t / TN, j = t % TN;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + i * 16 + (lane >> 4) * 4 + q;
                const int n = n0 + j * 16 + (lane & 15);
                if (m < a.M && n < a.N) Cz[(size_t)m * a.ldc + n] = Red[(t * 4 + q) * 64 + lane];
            }
        }
        return;
    }
    epilogue<TM, TN, EPI_>(a, pre, Red, m0, n0, lane, wave);
    AIR_STAMP(61);
}


// ---------------------------------------------------------------------------
// Throughput tiling for the ONE deep contraction of a large canvas: the hoisted x.Wx of the 128x128 configuration,
// [256 x 1024 x 16384] -- fp32 image batch (the caller's tensor) x the bf16 shadow of Wx, split-K slabs out.
// The latency kernel above splits K over its four waves and re-reads the image batch once per 32/64-column tile
// (512 / 256 MB through the L1s).  Here a workgroup owns a 64 x 64 output tile, each wave a 32 x 32 quadrant of it
// over the WHOLE K slab (no cross-wave reduction), operands stream through double-buffered 64-deep LDS stages with the
// loads of FOUR stages in flight per thread and two workgroups per CU; the row panel of a (row tile, slab) pair is kept in
// one XCD's L2 for its 16 column tiles.  Interior tiles only (M % 64, N % 64, K slab % 64 == 0: the dispatcher checks).
// ---------------------------------------------------------------------------
// BN = 128 (round 4): the launch is bound by what ONE CU can pull through its L1 (~45 GB/s when every line misses it):
// with 64 x 64 tiles a CU's two workgroups read 2 x (512 KB of fp32 rows + 256 KB of bf16 columns) per slab = 1.5 MB ->
// 37 us.  A 64 x 128 tile (each wave 32 x 64) reads 512 + 512 KB for twice the outputs: 256 workgroups, one per CU, 1 MB
// each.  Same k order per accumulator: bit-identical.
template <int BN>
__global__ __launch_bounds__(THREADS) void gemm_xw_tp_kernel(Args a)
{
    constexpr int BM = 64, KB = 64, D = 4;               // D: stages of global loads in flight per thread (a memory round
                                                         // trip is ~1 us, a stage's MFMAs ~0.15 us: one stage ahead is not enough)
    constexpr int NJ = BN / 32;                          // 16-column MFMA tiles per wave (its half of the tile's columns)
    constexpr int PPR = BN / 8, RPP = THREADS / PPR, NBP = KB / RPP;    // B: 16-byte pieces per k row, rows per pass, passes
    __shared__ __attribute__((aligned(16))) unsigned short ImgA[2][BM * KB];   // [row][64 k], 16-byte slots swizzled by row
    __shared__ __attribute__((aligned(16))) unsigned short ImgB[2][KB * BN];   // [k][64 n] as it lies (transpose read)
    // the step prologue's planes of workgroups come LAST in dispatch order here: the product's 512 workgroups take their
    // CUs first and the prologue (at 128 x 128: 1.3 M quads of noise and image-twin work, 4 096 workgroups) fills in beside
    // them -- in front, it delayed the product by its whole duration (54.5 us per launch against 34 + 6 as two launches)
    const int nz = (int)gridDim.z - a.job_on;
    if ((int)blockIdx.z >= nz) {
        const long plane = (long)gridDim.x * gridDim.y;
        air_step_job_run(a.job, ((int)blockIdx.z - nz) * plane + (long)blockIdx.y * gridDim.x + blockIdx.x, plane * a.job_on);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware map (workgroup b runs on XCD b % 8, each XCD with its own 4 MB L2).  Round 3 kept all column tiles of a
    // (row tile, slab) pair on one XCD: the fp32 A panel of the pair came from memory once -- but the four row tiles that
    // share a B panel (the bf16 shadow of Wx: 33.5 MB, the BIGGER operand) then sat on four different XCDs, and the
    // counters showed it: 167 MB read for 50 MB of operands.  Now a K SLAB is an XCD's: all (row tile, column tile) pairs
    // of slab z run on XCD z % 8, stepping through k together (two workgroups per CU, 64 per XCD at the stress shape), so
    // every line of BOTH operands is fetched from memory by one L2 only and its other users hit there.
    // (The old map was an environment switch until ABI 4 removed it.)
    int tile_m = blockIdx.y, tile_n = blockIdx.x, zslab = (int)blockIdx.z;
    {
        const int nx = gridDim.x, ny = gridDim.y, pairs = ny * nz;
        const int lin = (zslab * ny + (int)blockIdx.y) * nx + (int)blockIdx.x;
        const int xcd = lin & 7, slot = lin >> 3;
        if (a.i1 == 0 && (nz & 7) == 0) {
            zslab = xcd + 8 * (slot / (nx * ny));
            const int rem = slot % (nx * ny);
            tile_n = rem % nx; tile_m = rem / nx;
        } else if ((pairs & 7) == 0) {
            const int pair = xcd + 8 * (slot / nx);
            tile_n = slot % nx; tile_m = pair % ny; zslab = pair / ny;
        }
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = zslab * a.kslab, kend = min(a.K, kbeg + a.kslab);
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * (BN / 2);   // this wave's 32 x BN/2 part, over the whole K slab

    f32x4 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging maps: A piece = float4 (4 k) of row (tid >> 4) + 16 i; B piece = 16 bytes (8 columns) of k row tid / PPR + RPP i
    const int ar = tid >> 4, ah = tid & 15, bk = tid / PPR, bh = tid % PPR;
    const float* pa = a.A + (size_t)(m0 + ar) * a.lda + ah * 4;
    const unsigned short* pb = a.B16 + (size_t)bk * a.ldb + n0 + bh * 8;
    const unsigned la = ar * KB + (((ah >> 1) ^ (ar & 7)) << 3) + (ah & 1) * 4;      // (row + 16 i) & 7 == ar & 7
    typedef float f32v4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32v4 __attribute__((ext_vector_type(4)));
    f32v4 va[D][4];                                       // (native vector types: the HIP structs kept this ring in scratch)
    u32v4 vb[D][NBP];
    // (the ring slot is a compile-time constant everywhere -- std::integral_constant -- so that the ring lives in
    // registers: with a run-time slot index the arrays went to scratch memory, 39.8 -> 54.6 us)
    auto load_stage = [&](auto dc, int k0) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) va[d][i] = *reinterpret_cast<const f32v4*>(pa + (size_t)(16 * i) * a.lda + k0);
#pragma unroll
        for (int i = 0; i < NBP; ++i) vb[d][i] = *reinterpret_cast<const u32v4*>(pb + (size_t)(k0 + RPP * i) * a.ldb);
    };
    int buf = 0;
    auto stage = [&](auto dc, int k0) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value;
        if (k0 >= kend) return;                            // (uniform)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 w;
            w.x = pack_bf16(va[d][i].x, va[d][i].y); w.y = pack_bf16(va[d][i].z, va[d][i].w);
            *reinterpret_cast<uint2*>(&ImgA[buf][la + 16 * i * KB]) = w;
        }
#pragma unroll
        for (int i = 0; i < NBP; ++i) *reinterpret_cast<u32v4*>(&ImgB[buf][(bk + RPP * i) * BN + bh * 8]) = vb[d][i];
        if (k0 + D * KB < kend) load_stage(dc, k0 + D * KB);        // refill this ring slot: D stages ahead
        __syncthreads();                                  // (also: everyone is past the MFMAs of the stage that used the other buffer)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int slot = ks * 4 + (lane >> 4), il = lane & 15;
            bf16x8 av[2], bv[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm + i * 16 + il;
                av[i] = *reinterpret_cast<const bf16x8*>(&ImgA[buf][row * KB + ((slot ^ (row & 7)) << 3)]);
            }
            const unsigned short* blk = &ImgB[buf][(ks * 32 + (lane >> 4) * 8 + (il >> 2)) * BN + wn + (il & 3) * 4];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(blk + j * 16));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(blk + j * 16 + 4 * BN));
                bv[j] = bf16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        buf ^= 1;
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    static_assert(D == 4, "ring depth");
    if (kbeg < kend) load_stage(I0{}, kbeg);
    if (kbeg + KB < kend) load_stage(I1{}, kbeg + KB);
    if (kbeg + 2 * KB < kend) load_stage(I2{}, kbeg + 2 * KB);
    if (kbeg + 3 * KB < kend) load_stage(I3{}, kbeg + 3 * KB);
    for (int kb = kbeg; kb < kend; kb += D * KB) {
        stage(I0{}, kb); stage(I1{}, kb + KB); stage(I2{}, kb + 2 * KB); stage(I3{}, kb + 3 * KB);
    }
    float* Cz = a.C + (size_t)zslab * a.slab_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                Cz[(size_t)(m0 + wm + i * 16 + (lane >> 4) * 4 + q) * a.ldc + n0 + wn + j * 16 + (lane & 15)] = acc[i][j][q];
}

template <int TM, int TN, bool TB, int EPI_, bool AF32, int R>
int launch_one(const Args& a, dim3 grid, hipStream_t s) {
    using Cfg = TwCfg<TM, TN, R>;
    auto kern = gemm_bf16tw_kernel<TM, TN, TB, EPI_, AF32, R>;
    const int rc = air_grant_lds(reinterpret_cast<const void*>(kern), Cfg::BYTES);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(THREADS), Cfg::BYTES, s, a);
    AIR_CHECK_LAUNCH();
    return 0;
}

int images_of(const Args& a) { return (a.kslab + 63) / 64; }

}  // namespace

namespace airg {

// Which (tile, epilogue, layout) combinations exist as twin kernels, and with how many images per round.
// Returns R (> 0) or 0 when this descriptor has to take the fp32-operand kernels.
int twin_rounds(const Args& a, int tm, int tn, bool ta, bool tb) {
    // (a panel-blocked B twin serves the untransposed 16- / 32-column tiles and the four-unit LSTM tiles; everything
    // else needs the row-major twin)
    const bool pnl_tile = !tb && a.B16p != nullptr && ((tm == 1 && tn == 1) || (tm == 2 && tn == 2) || (tm == 1 && tn == 4));
    if (ta || (!a.B16 && !pnl_tile)) return 0;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool af32 = a.A16 == nullptr;
    // whole 16-byte pieces only: the ragged shapes keep the fp32-operand kernels
    if (af32) { if (!al16(a.A) || (a.lda & 3) || (a.K & 3) || (a.kslab & 3)) return 0; }
    else if (!al16(a.A16) || (a.lda & 7) || (a.K & 7) || (a.kslab & 7)) return 0;
    if (pnl_tile) { if (!al16(a.B16p)) return 0; }
    else if (!al16(a.B16) || (a.ldb & 7)) return 0;
    if (tb) { if ((a.K & 7) || (a.kslab & 7)) return 0; }
    else if ((a.N & 7) || (a.gstride & 7) || (a.gwidth & 7)) return 0;
    const int nimg = images_of(a);
    const int e = a.epi;
    if (tm == 1 && tn == 1) {
        // the hoisted x.Wx carrying the first LSTM step: four-unit tiles over the WHOLE contraction, fp32 or twin A
        // (K <= 2560: all 40 images in ONE round -- one memory round trip, 160 KB of LDS, one workgroup per CU)
        if (e == AIR_EPI_LSTM_FWD0)
            return (!tb && (a.gwidth & 3) == 0 && (int)((a.K + a.kslab - 1) / a.kslab) == 1)
                       ? ((nimg <= 40 && nimg > 16) ? 40 : 16) : 0;
        if (af32) return 0;
        if (e == AIR_EPI_GENERIC || ((e == AIR_EPI_LSTM_BWD || e == AIR_EPI_LSTM_BWD_TAIL) && tb)) return nimg <= 4 ? 4 : (nimg <= 8 ? 8 : 16);
        return 0;
    }
    if (tm == 1 && tn == 4) return (!af32 && !tb && e == AIR_EPI_LSTM_FWD) ? 4 : 0;       // (quad-unit tiles when gwidth % 4 == 0: twin_launch)
    if (tm == 2 && tn == 2) return e == AIR_EPI_GENERIC ? (af32 ? (tb ? 0 : 8) : (nimg <= 4 ? 4 : 8)) : 0;
    if (tm == 4 && tn == 2) return (e == AIR_EPI_GENERIC && !tb) ? 4 : 0;
    if (tm == 4 && tn == 4) return (e == AIR_EPI_GENERIC && !tb) ? 4 : 0;      // 64 x 64: the deep x.Wx of large canvases
    return 0;
}

int twin_launch(const Args& a, int tm, int tn, bool tb, dim3 grid, hipStream_t s) {
    const int r = twin_rounds(a, tm, tn, false, tb);
    const bool af32 = a.A16 == nullptr;
    const int e = a.epi;
#define TW(TM_, TN_, TB_, EPI__, AF_, R_) return launch_one<TM_, TN_, TB_, EPI__, AF_, R_>(a, grid, s)
    if (tm == 1 && tn == 1 && e == AIR_EPI_LSTM_FWD0) {
        if (r == 40) { if (af32) TW(1, 1, false, AIR_EPI_LSTM_FWD0, true, 40); TW(1, 1, false, AIR_EPI_LSTM_FWD0, false, 40); }
        if (af32) TW(1, 1, false, AIR_EPI_LSTM_FWD0, true, 16);
        TW(1, 1, false, AIR_EPI_LSTM_FWD0, false, 16);
    }
    if (tm == 1 && tn == 1) {
        if (e == AIR_EPI_GENERIC) {
            if (tb) { if (r == 4) TW(1, 1, true, AIR_EPI_GENERIC, false, 4); if (r == 8) TW(1, 1, true, AIR_EPI_GENERIC, false, 8); TW(1, 1, true, AIR_EPI_GENERIC, false, 16); }
            if (r == 4) TW(1, 1, false, AIR_EPI_GENERIC, false, 4); if (r == 8) TW(1, 1, false, AIR_EPI_GENERIC, false, 8); TW(1, 1, false, AIR_EPI_GENERIC, false, 16);
        }
        if (e == AIR_EPI_LSTM_BWD) { if (r == 4) TW(1, 1, true, AIR_EPI_LSTM_BWD, false, 4); if (r == 8) TW(1, 1, true, AIR_EPI_LSTM_BWD, false, 8); TW(1, 1, true, AIR_EPI_LSTM_BWD, false, 16); }
        if (e == AIR_EPI_LSTM_BWD_TAIL) { if (r == 4) TW(1, 1, true, AIR_EPI_LSTM_BWD_TAIL, false, 4); if (r == 8) TW(1, 1, true, AIR_EPI_LSTM_BWD_TAIL, false, 8); TW(1, 1, true, AIR_EPI_LSTM_BWD_TAIL, false, 16); }
    }
    if (tm == 1 && tn == 4) {
        // AIR_EPI_LSTM_FWD: 16-column tiles of four units x four gates where the layout allows (8-byte pieces of 4 units)
        if ((a.gwidth & 3) == 0) {
            dim3 gq((a.gwidth + 3) / 4, grid.y, grid.z);
            Args b = a;
            if (a.job_on) {          // the carried job's planes were sized for the wide tiles' grid: same number of workgroups
                b.job_on = (int)((a.job_on * grid.x + gq.x - 1) / gq.x);
                if (b.job_on < 1) b.job_on = 1;
                gq.z = grid.z - a.job_on + b.job_on;
            }
            return launch_one<1, 1, false, EPI_LSTM_FWD_Q, false, 4>(b, gq, s);
        }
        // (the wide tiles read the row-major twin: a gate-interleaved panel twin is laid out for the four-unit tiles only)
        if (!a.B16) return AIR_EINVAL;
        Args w = a;
        w.B16p = nullptr;
        return launch_one<1, 4, false, AIR_EPI_LSTM_FWD, false, 4>(w, grid, s);
    }
    if (tm == 2 && tn == 2) {
        if (af32) TW(2, 2, false, AIR_EPI_GENERIC, true, 8);
        if (tb) { if (r == 4) TW(2, 2, true, AIR_EPI_GENERIC, false, 4); TW(2, 2, true, AIR_EPI_GENERIC, false, 8); }
        if (r == 4) TW(2, 2, false, AIR_EPI_GENERIC, false, 4); TW(2, 2, false, AIR_EPI_GENERIC, false, 8);
    }
    if (tm == 4 && tn == 2) { if (af32) TW(4, 2, false, AIR_EPI_GENERIC, true, 4); TW(4, 2, false, AIR_EPI_GENERIC, false, 4); }
    if (tm == 4 && tn == 4) { if (af32) TW(4, 4, false, AIR_EPI_GENERIC, true, 4); TW(4, 4, false, AIR_EPI_GENERIC, false, 4); }
#undef TW
    return AIR_EINVAL;
}

// tile (8, 4) of the ABI = the throughput kernel (128 x 64 per workgroup, quadrant per wave): eligibility and launch
int xw_tp_ok(const Args& a, int precision, bool ta, bool tb, int ksplit) {
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (precision != 1 || ta || tb || !a.B16 || a.epi != AIR_EPI_GENERIC || ksplit <= 1) return AIR_EINVAL;
    if ((a.M % 64) || (a.N % 64) || (a.kslab % 64) || (a.K % 64)) return AIR_EALIGN;
    if (!al16(a.A) || !al16(a.B16) || (a.lda & 3) || (a.ldb & 7)) return AIR_EALIGN;
    return 0;
}

// 64 x 128 tiles when they still give every CU a workgroup
int xw_tp_columns(const Args& a) {
    const long slabs = (a.K + a.kslab - 1) / a.kslab;
    const bool wide = (a.N % 128) == 0 && (long)(a.N / 128) * (a.M / 64) * slabs >= 256;
    return wide ? 128 : 64;
}

int xw_tp_launch(const Args& a0, int job_planes_hint, hipStream_t s) {
    Args a = a0;
    const bool wide = xw_tp_columns(a0) == 128;
    dim3 grid(a.N / (wide ? 128 : 64), a.M / 64, 1);
    if (a.job_on) {
        const long quads = (a.job.n_normal + 3) / 4 + (a.job.n_uniform + 3) / 4 + (a.job.twin_n + 3) / 4;
        const long plane = (long)grid.x * grid.y * THREADS;
        long planes = (quads + plane - 1) / plane;
        a.job_on = (int)(planes < 1 ? 1 : (planes > 64 ? 64 : planes));
    }
    (void)job_planes_hint;
    grid.z = (a.K + a.kslab - 1) / a.kslab + a.job_on;
    a.slab_stride = (long)a.M * a.ldc;
    a.i1 = 0;
    if (wide) hipLaunchKernelGGL(gemm_xw_tp_kernel<128>, grid, dim3(THREADS), 0, s, a);
    else hipLaunchKernelGGL(gemm_xw_tp_kernel<64>, grid, dim3(THREADS), 0, s, a);
    AIR_CHECK_LAUNCH();
    return 0;
}

void twin_kernel_name(const Args& a, int tm, int tn, bool tb, char* buf, int n) {
    if (tm == 1 && tn == 4 && (a.gwidth & 3) == 0) {
        snprintf(buf, n, "gemm_bf16tw_kernel<1, 1, false, %d, false, 4>", EPI_LSTM_FWD_Q);
        return;
    }
    if (tm == 1 && tn == 1 && a.epi == AIR_EPI_LSTM_FWD0) {
        snprintf(buf, n, "gemm_bf16tw_kernel<1, 1, false, %d, %s, %d>", AIR_EPI_LSTM_FWD0, a.A16 == nullptr ? "true" : "false",
                 twin_rounds(a, tm, tn, false, tb));
        return;
    }
    snprintf(buf, n, "gemm_bf16tw_kernel<%d, %d, %s, %d, %s, %d>", tm, tn, tb ? "true" : "false", a.epi,
             a.A16 == nullptr ? "true" : "false", twin_rounds(a, tm, tn, false, tb));
}

}  // namespace airg

namespace {

// fp32 -> bf16 (RNE) copy: the twin of an array its producer could not write (variables after a
// host-side load; Adam keeps the shadow fresh afterwards)
__global__ __launch_bounds__(256) void bf16_twin_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long n) {
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        reinterpret_cast<uint2*>(dst)[i] = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) dst[n4 * 4 + threadIdx.x] = bf16_of(src[n4 * 4 + threadIdx.x]);
}

}  // namespace

extern "C" int air_bf16_twin(const float* src, uint16_t* dst, int64_t n, void* stream) {
    if (!src || !dst || n <= 0) return AIR_EINVAL;
    if (((reinterpret_cast<uintptr_t>(src) & 15) != 0) || ((reinterpret_cast<uintptr_t>(dst) & 7) != 0)) return AIR_EALIGN;
    long blocks = (n / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(bf16_twin_kernel, dim3((unsigned)blocks), dim3(256), 0, air_stream(stream), src, dst, (long)n);
    AIR_CHECK_LAUNCH();
    return 0;
}
