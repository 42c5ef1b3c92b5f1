// Development: dependent-issue latency of the packed fp32 vector instructions the chain kernels are made of
// (one wavefront per SIMD, nothing else to hide it) against the unpacked ones and against independent issue.
// What bounds update_chain3_kernel on C4 (64x64x32 Median: 1024 wavefronts, one per SIMD): per sample a chain
// of 5 dependent packed operations.
//   hipcc -O3 --offload-arch=gfx950 tools/exp/pk_latency_bench.hip -o tools/exp/pk_latency_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64) void chain_kernel(int iters, float *out, unsigned long long *cycles)
{
    f2 a = {1.0f + threadIdx.x * 1e-7f, 1.0f}, b = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
    f2 a1 = a, a2 = a, a3 = a;
    float s = a.x, s1 = a.y, s2 = 1.f, s3 = 1.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {            // 1 dependent packed chain
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
            } else if (MODE == 1) {     // 4 independent packed chains
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5"
                             : "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
            } else if (MODE == 2) {     // 1 dependent unpacked chain
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s) : "v"(b.x), "v"(c.x));
            } else if (MODE == 3) {     // 4 independent unpacked chains
                asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                             : "+v"(s), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(b.x), "v"(c.x));
            } else if (MODE == 4) {     // 2 independent packed chains
                asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3" : "+v"(a), "+v"(a1) : "v"(b), "v"(c));
            } else if (MODE == 5) {     // dependent packed add (no fma)
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));
            } else if (MODE == 6) {     // dependent packed mul with clamp
                asm volatile("v_pk_mul_f32 %0, %0, %1 clamp" : "+v"(a) : "v"(b));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0)
        cycles[0] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = a.x + a1.x + a2.x + a3.x + s + s1 + s2 + s3;
}

int main()
{
    float *out;
    unsigned long long *cyc, h;
    CK(hipMalloc(&out, 1024 * 64 * 4));
    CK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 4096;
    const char *names[] = {"v_pk_fma_f32, 1 dependent chain", "v_pk_fma_f32, 4 independent chains", "v_fma_f32, 1 dependent chain",
                           "v_fma_f32, 4 independent chains", "v_pk_fma_f32, 2 independent chains", "v_pk_add_f32, 1 dependent chain",
                           "v_pk_mul_f32 clamp, 1 dependent chain"};
    const int per_iter[] = {16, 64, 16, 64, 32, 16, 16};
    void (*kern[])(int, float *, unsigned long long *) = {chain_kernel<0>, chain_kernel<1>, chain_kernel<2>, chain_kernel<3>,
                                                          chain_kernel<4>, chain_kernel<5>, chain_kernel<6>};
    for (int m = 0; m < 7; ++m) {
        for (int blocks : {1, 1024}) {      // one wavefront alone; one wavefront on every SIMD (clock under load)
            hipLaunchKernelGGL(kern[m], dim3(blocks), dim3(64), 0, 0, 16, out, cyc);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern[m], dim3(blocks), dim3(64), 0, 0, iters, out, cyc);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
            const double n = (double)iters * per_iter[m];
            printf("%-42s %4d wavefront(s): %.2f ns per instruction, %.2f s_memtime ticks per instruction\n", names[m], blocks,
                   ms * 1e6 / n, (double)h / n);
        }
    }
    return 0;
}
