// Experiment: is a same-address ds_add_f32 from the 64 lanes of one wave applied in ascending lane
// order (=> a sequential fp32 accumulator at LDS-atomic rate), and how fast is it?
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_order lds_atomic_order.hip && ./lds_atomic_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>

__device__ __forceinline__ void lds_fadd(float* p, float v) {
    // no-return LDS float add; inline asm so that no CAS loop can be substituted
    asm volatile("ds_add_f32 %0, %1" :: "v"((unsigned)(size_t)p), "v"(v) : "memory");
}

// trial t: 64 lanes add vals[t*64 + lane] to slot[key[t*64+lane]] in ONE instruction; `reps` instructions in sequence
__global__ void order_kernel(const float* vals, const int* keys, float* out, int ninstr, int nslots) {
    extern __shared__ float slots[];
    for (int i = threadIdx.x; i < nslots; i += blockDim.x) slots[i] = 0.0f;
    __syncthreads();
    if (threadIdx.x < 64) {
        for (int k = 0; k < ninstr; ++k) {
            const float v = vals[k * 64 + threadIdx.x];
            const int key = keys[k * 64 + threadIdx.x];
            lds_fadd(&slots[key], v);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < nslots; i += blockDim.x) out[blockIdx.x * nslots + i] = slots[i];
}

__global__ void time_kernel(float* out, long long* cycles, int ninstr, int spread) {
    extern __shared__ float slots[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) slots[i] = 0.0f;
    __syncthreads();
    float v = 1.0f + threadIdx.x * 1e-3f;
    const int key = spread ? (threadIdx.x % spread) * 33 % 4096 : 0;   // spread = number of distinct addresses
    long long t0 = 0, t1 = 0;
    if (threadIdx.x < 64) {
        t0 = __builtin_readcyclecounter();
        for (int k = 0; k < ninstr; ++k) lds_fadd(&slots[key], v);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t1 = __builtin_readcyclecounter();
    }
    __syncthreads();
    if (threadIdx.x == 0) { cycles[blockIdx.x] = t1 - t0; out[0] = slots[0]; }
}

// VALU reference chain: one lane adds n values sequentially (values in LDS)
__global__ void valu_chain(const float* vals, float* out, long long* cycles, int n) {
    extern __shared__ float sh[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) sh[i] = vals[i];
    __syncthreads();
    if (threadIdx.x < 64) {
        long long t0 = __builtin_readcyclecounter();
        float acc = 0.0f;
        const float4* p = reinterpret_cast<const float4*>(sh);
#pragma unroll 4
        for (int k = 0; k < n / 4; ++k) { float4 q = p[k]; acc += q.x; acc += q.y; acc += q.z; acc += q.w; }
        long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) { out[0] = acc; cycles[0] = t1 - t0; }
    }
}

int main() {
    const int ninstr = 200, nslots = 64;
    std::vector<float> vals(ninstr * 64);
    std::vector<int> keys(ninstr * 64);
    srand(1);
    int bad_total = 0;
    for (int mode = 0; mode < 3; ++mode) {
        // mode 0: all lanes -> slot 0; mode 1: runs of equal keys (like canvas columns); mode 2: random keys
        for (int i = 0; i < ninstr * 64; ++i) {
            float mag = powf(10.0f, (float)(rand() % 12) - 3.0f);
            vals[i] = ((rand() & 1) ? 1.f : -1.f) * mag * (0.5f + (rand() % 1000) / 1000.0f);
            keys[i] = mode == 0 ? 0 : (mode == 1 ? ((i % 64) / 13) : rand() % nslots);
        }
        float *dv, *dout; int* dk;
        hipMalloc(&dv, vals.size() * 4); hipMalloc(&dk, keys.size() * 4); hipMalloc(&dout, nslots * 4);
        hipMemcpy(dv, vals.data(), vals.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dk, keys.data(), keys.size() * 4, hipMemcpyHostToDevice);
        order_kernel<<<1, 256, nslots * 4>>>(dv, dk, dout, ninstr, nslots);
        std::vector<float> got(nslots), ref(nslots, 0.0f);
        hipMemcpy(got.data(), dout, nslots * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < ninstr * 64; ++i) { volatile float s = ref[keys[i]] + vals[i]; ref[keys[i]] = s; }
        int bad = 0;
        for (int s = 0; s < nslots; ++s) if (memcmp(&got[s], &ref[s], 4)) { ++bad; if (bad < 4) printf("  mode %d slot %d got %.9g ref %.9g\n", mode, s, got[s], ref[s]); }
        printf("mode %d: %d of %d slots differ from the ascending-lane sequential sum\n", mode, bad, nslots);
        bad_total += bad;
        hipFree(dv); hipFree(dk); hipFree(dout);
    }
    float* dout; long long* dcyc;
    hipMalloc(&dout, 4096 * 4); hipMalloc(&dcyc, 64 * 8);
    for (int spread : {0, 2, 4, 16, 64}) {
        time_kernel<<<1, 256, 4096 * 4>>>(dout, dcyc, 1000, spread);
        long long c; hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
        printf("1000 x ds_add_f32, %2d distinct addresses per instruction: %lld cycles (%.2f per instruction, %.3f per lane-add)\n",
               spread ? spread : 1, c, c / 1000.0, c / 64000.0);
    }
    {
        const int n = 8192;
        std::vector<float> v(n, 1.0f);
        float* dv; hipMalloc(&dv, n * 4); hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice);
        valu_chain<<<1, 256, n * 4>>>(dv, dout, dcyc, n);
        long long c; hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
        printf("VALU chain of %d adds from LDS (one lane): %lld cycles (%.2f per add)\n", n, c, (double)c / n);
    }
    return bad_total ? 1 : 0;
}
