// Development: dependency latency of the two-stream online schedule, with dummy kernels that just
// take as long as the real ones (scan 9 us; window update 4 us; window re-scan 2 us).
//   serial    : scan -> update                      (what vsom_train_online_chunk enqueues today)
//   overlapped: A: scan_main(j)            waits for scan_win(j-1)
//               B: update(j-1) -> scan_win(j)      update waits for scan_main(j-1)
//   hipcc -O3 --offload-arch=gfx950 tools/exp/online_overlap_bench.hip -o tools/exp/online_overlap_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin_kernel(long long ticks, unsigned *sink)   // ticks of the 100 MHz constant clock
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks)
        __builtin_amdgcn_s_sleep(2);
    if (ticks < 0)
        sink[0] = 1;
}

int main()
{
    unsigned *sink;
    CK(hipMalloc(&sink, 64));
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    const int n = 2000;
    std::vector<hipEvent_t> evA(n), evB(n);
    for (int i = 0; i < n; ++i) {
        CK(hipEventCreateWithFlags(&evA[i], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&evB[i], hipEventDisableTiming));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long long us = 100;   // ticks per microsecond
    auto spin = [&](hipStream_t s, double t) { hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, (long long)(t * us), sink); };
    float ms;
    // serial
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, A));
        for (int j = 0; j < n; ++j) {
            spin(A, 9.0);
            spin(A, 4.0);
        }
        CK(hipEventRecord(e1, A));
        CK(hipEventSynchronize(e1));
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("serial (scan 9 + update 4, one stream):        %.2f us per sample\n", ms * 1e3 / n);
    // overlapped, streams + events
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, A));
        CK(hipStreamWaitEvent(B, e0, 0));
        for (int j = 0; j < n; ++j) {
            if (j > 0)
                CK(hipStreamWaitEvent(A, evB[j - 1], 0));   // scan_main(j) needs scan_win(j-1)
            spin(A, 8.0);                                    // scan_main(j)
            CK(hipEventRecord(evA[j], A));
            if (j > 0) {
                CK(hipStreamWaitEvent(B, evA[j - 1], 0));   // update(j-1) needs scan_main(j-1) (scan_win(j-1) is on B)
                spin(B, 4.0);                                // update(j-1)
            }
            spin(B, 2.0);                                    // scan_win(j)
            CK(hipEventRecord(evB[j], B));
        }
        CK(hipStreamWaitEvent(A, evB[n - 1], 0));
        CK(hipEventRecord(e1, A));
        CK(hipEventSynchronize(e1));
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("overlapped (two streams, events):              %.2f us per sample\n", ms * 1e3 / n);
    // the same captured into a graph
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(A, hipStreamCaptureModeGlobal));
    CK(hipEventRecord(e0, A));
    CK(hipStreamWaitEvent(B, e0, 0));
    for (int j = 0; j < n; ++j) {
        if (j > 0)
            CK(hipStreamWaitEvent(A, evB[j - 1], 0));
        spin(A, 8.0);
        CK(hipEventRecord(evA[j], A));
        if (j > 0) {
            CK(hipStreamWaitEvent(B, evA[j - 1], 0));
            spin(B, 4.0);
        }
        spin(B, 2.0);
        CK(hipEventRecord(evB[j], B));
    }
    CK(hipStreamWaitEvent(A, evB[n - 1], 0));
    CK(hipStreamEndCapture(A, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t g0, g1;
    CK(hipEventCreate(&g0));
    CK(hipEventCreate(&g1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(g0, A));
        CK(hipGraphLaunch(exec, A));
        CK(hipEventRecord(g1, A));
        CK(hipEventSynchronize(g1));
        hipEventElapsedTime(&ms, g0, g1);
    }
    printf("overlapped, captured into one graph:           %.2f us per sample\n", ms * 1e3 / n);
    // serial chain as a graph
    CK(hipStreamBeginCapture(A, hipStreamCaptureModeGlobal));
    for (int j = 0; j < n; ++j) {
        spin(A, 9.0);
        spin(A, 4.0);
    }
    CK(hipStreamEndCapture(A, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(g0, A));
        CK(hipGraphLaunch(exec, A));
        CK(hipEventRecord(g1, A));
        CK(hipEventSynchronize(g1));
        hipEventElapsedTime(&ms, g0, g1);
    }
    printf("serial, captured into one graph:               %.2f us per sample\n", ms * 1e3 / n);
    return 0;
}
