// Experiment: the feeder's batch pattern (16 reads | wait | 16 ds_add_f32, reads of batch i+1 queued behind
// the adds of batch i) in isolation, 1 wave alone vs 16 waves of which 15 sit at a barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) float lds_f;
__global__ void k(float* out, long long* cyc, int nbatch, int mode, int nvalid) {
    extern __shared__ float sh[];
    float* acc = sh; float* T = sh + 64;
    for (int i = threadIdx.x; i < 12000; i += blockDim.x) sh[i] = (i < 64) ? 0.f : 1.0f + (i % 97) * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        float r[16], q[16];
        const unsigned acc_addr = (unsigned)(size_t)(lds_f*)acc;
        long long t0 = __builtin_readcyclecounter();
        auto reads = [&](float (&x)[16], int b) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const unsigned addr = (unsigned)(size_t)(lds_f*)(T + ((b * 16 + u) * 64 + lane) % 10000);
                asm volatile("ds_read_b32 %0, %1" : "=v"(x[u]) : "v"(addr) : "memory");
            }
        };
        auto adds = [&](float (&x)[16]) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                         "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) :: "memory");
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (mode >= 2) { if (lane < nvalid) asm volatile("ds_add_f32 %0, %1" :: "v"(acc_addr), "v"(x[u]) : "memory"); }
                else asm volatile("ds_add_f32 %0, %1" :: "v"(acc_addr), "v"(x[u]) : "memory");
            }
        };
        reads(r, 0);
        for (int b = 0; b < nbatch; b += 2) {
            if (mode == 0 || mode == 2) { reads(q, b + 1); adds(r); reads(r, b + 2); adds(q); }           // reads BEFORE the adds they follow
            else { adds(r); reads(q, b + 1); adds(q); reads(r, b + 2); }                      // reads AFTER (queued behind the adds)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[0];
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int grid : {1, 8, 32, 64, 128, 192, 256, 512})
        for (int threads : {64, 1024}) {
            const int nb = 8, mode = 0;
            k<<<grid, threads, 12000 * 4>>>(out, cyc, nb, mode, 64);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%4d workgroups x %4d threads, %3d batches of 16 x 64 terms: %8lld cycles = %.2f per term, %.0f per instruction\n", grid, threads,
                   nb, c, (double)c / (nb * 1024), (double)c / (nb * 16));
        }
    return 0;
}
