// Development: cost of a grid-wide barrier inside one cooperative launch (the alternative to the two
// dependent launches per sample of the online path).
//   hipcc -O3 --offload-arch=gfx950 tools/exp/grid_sync_bench.hip -o tools/exp/grid_sync_bench
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void sync_kernel(int iters, unsigned *sink)
{
    cg::grid_group g = cg::this_grid();
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        g.sync();
        acc += i;
    }
    if (acc == 0xFFFFFFFFu)
        sink[0] = acc;
}

// hand-rolled: one counter, monotonically increasing target; bounded spin
__global__ __launch_bounds__(256) void spin_kernel(int iters, unsigned *counter, unsigned *flag, unsigned nblocks)
{
    for (int i = 0; i < iters; ++i) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned target = (unsigned)(i + 1) * nblocks;
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1u << 22)) {   // never hang the box: give up loudly
                    *flag = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
}

int main()
{
    unsigned *buf;
    CK(hipMalloc(&buf, 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int nb : {256, 512, 1024}) {
        int iters = 2000;
        void *args[] = {&iters, &buf};
        CK(hipMemset(buf, 0, 256));
        hipError_t e = hipLaunchCooperativeKernel((void *)sync_kernel, dim3(nb), dim3(256), args, 0, 0);
        if (e != hipSuccess) {
            printf("cooperative launch with %d blocks: %s\n", nb, hipGetErrorString(e));
            (void)hipGetLastError();
            continue;
        }
        CK(hipDeviceSynchronize());
        hipEventRecord(e0);
        CK(hipLaunchCooperativeKernel((void *)sync_kernel, dim3(nb), dim3(256), args, 0, 0));
        hipEventRecord(e1);
        CK(hipEventSynchronize(e1));
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("cg grid.sync, %4d blocks: %.2f us per barrier\n", nb, ms * 1e3 / iters);
        unsigned *counter = buf + 16, *flag = buf + 32;
        unsigned nblocks = (unsigned)nb;
        void *args2[] = {&iters, &counter, &flag, &nblocks};
        CK(hipMemset(buf, 0, 256));
        hipEventRecord(e0);
        CK(hipLaunchCooperativeKernel((void *)spin_kernel, dim3(nb), dim3(256), args2, 0, 0));
        hipEventRecord(e1);
        CK(hipEventSynchronize(e1));
        hipEventElapsedTime(&ms, e0, e1);
        unsigned h[64];
        CK(hipMemcpy(h, buf, 256, hipMemcpyDeviceToHost));
        printf("one-counter barrier, %4d blocks: %.2f us per barrier (gave up: %u)\n", nb, ms * 1e3 / iters, h[32]);
    }
    return 0;
}
