// What does a dependent GEMM-shaped stage cost when S stages share ONE launch and hand over through device-side counters
// (consumer workgroups are dispatched behind the producers, fetch their weight panel, then wait for their row tile's
// producers) -- against S dependent kernel nodes of a hipGraph (1.7 us empty, 4.5 - 5.5 us for the step's skinny GEMMs)?
// Stage: [192 x K] bf16 activations x [K x 512] bf16 weights, 12 x 32 tiles of 16 x 16, one workgroup of 256 threads per
// tile: 16 KB of activations (written by the previous stage, other XCDs) + 16 KB of weights through registers into LDS,
// a barrier, a token amount of arithmetic, fp32 + bf16 outputs.  No MFMA: the question is the hand-over.
//   hipcc --offload-arch=gfx950 -O3 -o chain_bench chain_bench.hip && ./chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int M = 192, N = 512, K = 512, TM = 12, TN = 32, WGS = TM * TN;
struct Stage { const unsigned short* A; const unsigned short* W; unsigned short* C; };

__device__ int g_mode;   // bit 0: acquire fence after the wait, bit 1: release on the arrive, bits 4..: s_sleep length class
template <bool CHAIN, bool WT>
__device__ __forceinline__ void stage_body(const Stage& st, int tile, unsigned* wait_cnt, unsigned need, unsigned* done_cnt, unsigned char* lds) {
    const int mode = g_mode;
    const int tid = threadIdx.x;
    const int tm = tile % TM, tn = tile / TM;          // consecutive workgroups: different row tiles (as xcd_tile spreads them)
    uint4* img = reinterpret_cast<uint4*>(lds);
    // weights first: [K x 16] panel = K rows of 32 bytes -> 2 pieces of 16 B per row: 1024 pieces, 4 per thread
    uint4 vb[4], va[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = tid + 256 * i, k = p >> 1, h = p & 1;
        vb[i] = *reinterpret_cast<const uint4*>(st.W + (size_t)k * N + tn * 16 + h * 8);
    }
    if (CHAIN && wait_cnt && (mode & 4)) {
        // flags[row tile][32 column tiles] = epoch, one 128-byte line per row tile: lanes 0..31 of wave 0 poll one flag each
        if (tid < 64) {
            const unsigned* f = wait_cnt + tm * 32 + (tid & 31);
            while (true) {
                const unsigned v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_ballot_w64(v < need) == 0) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = tid + 256 * i, r = p >> 6, g = p & 63;
            const unsigned short* src = st.A + (size_t)(tm * 16 + r) * K + g * 8;
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(va[i]) : "v"(src) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (CHAIN && wait_cnt) {
        if (tid == 0) {
            // (an ACQUIRE load per poll invalidates the XCD's L2 every iteration: 63 us per stage)
            if (mode & 16) { while (__hip_atomic_load(wait_cnt + tm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(30); }
            else { while (__hip_atomic_load(wait_cnt + tm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(2); }
        }
        __syncthreads();
        if (mode & 1) __atomic_thread_fence(__ATOMIC_ACQUIRE);       // (agent scope by default for HIP device code)
    }
    // activations: 16 rows of K bf16 = 1024 B each -> 64 pieces per row, 1024 pieces, 4 per thread
    if (!(CHAIN && wait_cnt && (mode & 4))) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = tid + 256 * i, r = p >> 6, g = p & 63;
        va[i] = *reinterpret_cast<const uint4*>(st.A + (size_t)(tm * 16 + r) * K + g * 8);
    }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { img[tid + 256 * i] = va[i]; img[1024 + tid + 256 * i] = vb[i]; }
    __syncthreads();
    // token arithmetic: every thread folds 8 pieces of each image
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const uint4 x = img[(i * 256 + tid) & 1023], y = img[1024 + ((i * 256 + tid + 64) & 1023)]; acc += x.x ^ y.y; acc += x.z ^ y.w; }
    // outputs: 16 x 16 bf16 of the NEXT stage's activations [192 x 512] (N == K)
    const int r = tid >> 4, c = tid & 15;
    unsigned short* dst = st.C + (size_t)(tm * 16 + r) * N + tn * 16 + c;
    if (WT) __hip_atomic_store(dst, (unsigned short)(acc & 0x3fff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *dst = (unsigned short)(acc & 0x3fff);
    if (CHAIN && done_cnt && (mode & 4)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the write-through stores have reached memory
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done_cnt + tm * 32 + tn, need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (CHAIN && done_cnt) {
        __syncthreads();
        if (tid == 0) {
            if (mode & 2) __hip_atomic_fetch_add(done_cnt + tm, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(done_cnt + tm, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
template <bool WT>
__global__ __launch_bounds__(256) void stage_kernel(Stage st) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    stage_body<false, WT>(st, blockIdx.x, nullptr, 0, nullptr, lds);
}
struct Chain { Stage st[16]; unsigned* cnt; int nstage; unsigned epoch; };
template <bool WT>
__global__ __launch_bounds__(256) void chain_kernel(Chain ch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int s = blockIdx.x / WGS, tile = blockIdx.x % WGS;
    // counters: [stage][row tile], monotonic over launches: stage s is complete for row tile m at epoch * TN
    const bool fl = g_mode & 4;
    stage_body<true, WT>(ch.st[s], tile, s > 0 ? ch.cnt + (s - 1) * TM * (fl ? 32 : 1) : nullptr, fl ? ch.epoch : ch.epoch * TN, s + 1 < ch.nstage ? ch.cnt + s * TM * (fl ? 32 : 1) : nullptr, lds);
}
int main() {
    const int S = 8;
    unsigned short* act[2]; unsigned short* w[16]; unsigned* cnt;
    for (auto& a : act) { hipMalloc(&a, M * K * 2); hipMemset(a, 1, M * K * 2); }
    for (int s = 0; s < S; ++s) { hipMalloc(&w[s], K * N * 2); hipMemset(w[s], 2, K * N * 2); }
    hipMalloc(&cnt, 16 * TM * 32 * 4); hipMemset(cnt, 0, 16 * TM * 32 * 4);
    Chain ch; ch.cnt = cnt; ch.nstage = S;
    for (int s = 0; s < S; ++s) ch.st[s] = Stage{act[s & 1], w[s], act[(s + 1) & 1]};
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 32768;
    for (int wt = 0; wt < 2; ++wt) {
        // ---- S dependent kernel nodes in a hipGraph
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int rep = 0; rep < 10; ++rep)
            for (int s = 0; s < S; ++s) {
                if (wt) hipLaunchKernelGGL(stage_kernel<true>, dim3(WGS), dim3(256), lds, st, ch.st[s]);
                else hipLaunchKernelGGL(stage_kernel<false>, dim3(WGS), dim3(256), lds, st, ch.st[s]);
            }
        hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
        hipEventRecord(e0, st);
        for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, st);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s stores | graph of dependent kernels: %.2f us per stage\n", wt ? "write-through" : "plain        ", ms * 1e3f / (10 * 10 * S));
        // ---- the same stages in one launch per chain of S (10 chained launches per timing unit, eager: the epoch is an argument)
        unsigned epoch = 0;
        for (int mode : {0, 3, 4}) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_mode), &mode, 4);
        epoch = 0;
        auto run_chain = [&]() {
            ch.epoch = ++epoch;
            if (wt) hipLaunchKernelGGL(chain_kernel<true>, dim3(WGS * S), dim3(256), lds, st, ch);
            else hipLaunchKernelGGL(chain_kernel<false>, dim3(WGS * S), dim3(256), lds, st, ch);
        };
        hipMemsetAsync(cnt, 0, 16 * TM * 32 * 4, st);
        for (int i = 0; i < 5; ++i) run_chain();
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int i = 0; i < 100; ++i) run_chain();
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s stores | mode %2d (acquire %d release %d long-sleep %d) | one launch of %d chained stages: %.2f us per stage (%.2f us per launch)\n", wt ? "write-through" : "plain        ", mode, mode & 1, (mode >> 1) & 1, mode >> 4, S, ms * 1e3f / (100 * S), ms * 1e3f / 100);
        }
    }
    return 0;
}
