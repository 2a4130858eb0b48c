// A sequential fp32 accumulator that needs ONE LDS request per 64 terms: lane k of a wave holds term k of a batch
// (one coalesced ds_read_b32), and the running sum travels around the lanes, S[k] = S[k-1] + r[k], one
// v_add_f32 ... wave_ror:1 per term (a DPP operand reading the previous instruction's result: 2 wait states).
// After 64 steps lane 63 holds the sum of the batch on top of what lane 63 held before -- which is what lane 0 reads
// in the first step of the next batch.  Padding lanes carry -0.0f (x + -0 == x for every x).
// Questions: is it the ascending-order sequential sum (bit for bit), what does a term cost, and does it keep its rate
// while another wave of the CU saturates the LDS atomic pipe (the register chains reading 4 terms per request did not)?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o dpp_chain dpp_chain.hip && ./dpp_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((address_space(3))) float lds_f;
template <int V> __device__ __forceinline__ float ring_batch_v(float S, float r) {
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        if (V == 0) asm volatile("v_add_f32_dpp %0, %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(S) : "v"(r));
        if (V == 1) asm volatile("s_nop 0\n v_add_f32_dpp %0, %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(S) : "v"(r));
        if (V == 2) asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(S) : "v"(r));
    }
    return S;
}
__device__ __forceinline__ float ring_batch(float S, float r) {
    // 64 dependent steps; "s_nop 1" = the 2 wait states a DPP read of a just-written VGPR needs
#pragma unroll
    for (int k = 0; k < 64; ++k)
        asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(S) : "v"(r));
    return S;
}
__global__ __launch_bounds__(1024) void k(const float* src, int n, float* out, long long* cyc, int feeders, int chains, int variant) {
    extern __shared__ float T[];          // [n] terms | [16] accumulators
    float* acc = T + n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) T[i] = src[i];
    if (threadIdx.x < 16) acc[threadIdx.x] = 0.0f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long t0 = __builtin_readcyclecounter();
    if (wave < chains) {
        // the ring: lane 63 carries the running sum between batches; start with +0 in lane 63 (0 + r0 == r0)
        float S = 0.0f;
        const int nb = (n + 63) / 64;
        float r = lane < n ? T[lane] : -0.0f;
        for (int b = 0; b < nb; ++b) {
            const int nx = (b + 1) * 64 + lane;
            const float rn = (b + 1 < nb && nx < n) ? T[nx] : -0.0f;      // next batch's read in flight under this batch's adds
            if (variant == 0) S = ring_batch_v<0>(S, r);
            else if (variant == 1) S = ring_batch_v<1>(S, r);
            else if (variant == 2) { const float carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, S), 63)); if (lane == 0) r = carry + r;   /* NOT the same sum: timing only */ S = ring_batch_v<2>(S, r); }
            else S = ring_batch(S, r);
            r = rn;
        }
        if (lane == 63) out[wave] = S;
    } else if (wave < chains + feeders) {
        // the atomic pipe kept busy: this wave feeds the same stream to one LDS word, 64 terms per instruction
        const unsigned a = (unsigned)(size_t)(lds_f*)(acc + wave);
        for (int rep = 0; rep < 4; ++rep)
            for (int b = 0; b < n / 64; ++b) {
                const float v = T[b * 64 + lane];
                asm volatile("ds_add_f32 %0, %1" :: "v"(a), "v"(v) : "memory");
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[wave] = t1 - t0;
    __syncthreads();
    if (threadIdx.x < 16) out[16 + threadIdx.x] = acc[threadIdx.x];
}
int main() {
    const int n = 10000;
    std::vector<float> h(n);
    unsigned x = 12345u;
    float want = 0.0f;
    for (int i = 0; i < n; ++i) {
        x = x * 1664525u + 1013904223u;
        const float mant = 1.0f + (float)((x >> 9) & 0x3fffu) / 16384.0f;
        h[i] = ((x >> 31) ? -1.0f : 1.0f) * ldexpf(mant, (int)((x >> 24) & 31u) - 8);
        want = want + h[i];
    }
    float* src; float* out; long long* cyc;
    hipMalloc(&src, n * 4); hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 16 * 8);
    hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int variant : {3, 0, 1, 2})
    for (int feeders : {0, 1})
        for (int chains : {1, 4}) {
            printf("variant %d: ", variant);
            k<<<1, 1024, (n + 16) * 4>>>(src, n, out, cyc, feeders, chains, variant);
            float o[32]; long long c[16];
            hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost); hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
            long long mx = 0; for (int i = 0; i < chains; ++i) mx = c[i] > mx ? c[i] : mx;
            bool ok = true; for (int i = 0; i < chains; ++i) ok = ok && memcmp(&o[i], &want, 4) == 0;
            printf("%d ring chains beside %d atomic feeders: %.2f cycles per term (slowest chain), %s the sequential sum (%.9g vs %.9g); feeder: %.2f cycles per term\n",
                   chains, feeders, mx / (double)n, ok ? "==" : "!=", (double)o[0], (double)want, feeders ? c[chains] / (4.0 * n) : 0.0);
        }
    return 0;
}
