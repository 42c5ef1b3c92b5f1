// Micro-experiment (development only): can the fp32 matrix pipe compute the chains' delta = x - M
// EXACTLY (v_mfma_f32_32x32x1_2b_f32 with B = 1.0, C = -M is round(x*1 + (-M))) and does moving that
// sixth of the update kernel's VALU work to the MFMA pipe shorten the loop?  Layout: lane = node
// (32 per block, 2 blocks), registers = 16+16 dims per lane.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off mfma_delta_bench.hip -o mfma_delta_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

typedef float v32f __attribute__((ext_vector_type(32)));

__global__ void exact_kernel(const float *x, const float *m, float *out_mfma, float *out_valu)
{
    // one wave: x[32] dims, m[32 dims][64 nodes]; lane l <-> node (l%32 + 32*block)
    const int l = threadIdx.x;
    v32f c;
    for (int r = 0; r < 32; ++r) {
        const int blk = r >> 4, rr = r & 15;
        const int dim = (rr & 3) + 8 * (rr >> 2) + 4 * (l >> 5);
        const int node = (l & 31) + 32 * blk;
        c[r] = -m[dim * 64 + node];
    }
    const float a = x[l & 31];
    v32f d = __builtin_amdgcn_mfma_f32_32x32x1f32(a, 1.0f, c, 0, 0, 0);
    for (int r = 0; r < 32; ++r) {
        const int blk = r >> 4, rr = r & 15;
        const int dim = (rr & 3) + 8 * (rr >> 2) + 4 * (l >> 5);
        const int node = (l & 31) + 32 * blk;
        out_mfma[dim * 64 + node] = d[r];
        out_valu[dim * 64 + node] = x[dim] - m[dim * 64 + node];
    }
}

// timing loops: NS samples; every wave keeps 32 M and 32 S registers
template <bool USE_MFMA>
__global__ __launch_bounds__(256) void loop_kernel(const float *__restrict__ xs, const float2 *__restrict__ cw, int ns,
                                                    float *__restrict__ out)
{
    const int l = threadIdx.x & 63;
    const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    v32f mneg, s;
    for (int r = 0; r < 32; ++r) {
        mneg[r] = 0.f;
        s[r] = 0.f;
    }
    const float2 *cwp = cw + (size_t)(wid & 63) * 64 + l;
    for (int j = 0; j < ns; ++j) {
        const float xv = xs[j * 32 + (l & 31)];
        const float2 cw0 = cwp[(size_t)j * 8192], cw1 = cwp[(size_t)j * 8192 + 4096];
        v32f d;
        if (USE_MFMA) {
            d = __builtin_amdgcn_mfma_f32_32x32x1f32(xv, 1.0f, mneg, 0, 0, 0);
        } else {
            // VALU stand-in with the same operand pattern (x per lane): one add per element
#pragma unroll
            for (int r = 0; r < 32; ++r)
                d[r] = xv + mneg[r];
        }
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const float c = r < 16 ? cw0.x : cw1.x, w = r < 16 ? cw0.y : cw1.y;
            float t = c * d[r];
            mneg[r] = mneg[r] - t;
            float u = w * d[r];
            u = u * d[r];
            s[r] = s[r] + u;
        }
    }
    float acc = 0.f;
    for (int r = 0; r < 32; ++r)
        acc += mneg[r] + s[r];
    out[(size_t)wid * 64 + l] = acc;
}

int main()
{
    // ---- exactness ----
    std::vector<float> hx(32), hm(32 * 64);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto &v : hx) v = rnd() * 255.f;
    for (auto &v : hm) v = rnd() * 300.f;
    const float specials[] = {0.f, -0.f, 1e-40f, -1e-40f, 1.17549435e-38f, INFINITY, -INFINITY, NAN, 3.4e38f, -3.4e38f, 1e-30f, 16777216.f};
    for (int i = 0; i < 12; ++i) {
        hx[i] = specials[i];
        for (int n = 0; n < 12; ++n)
            hm[(12 + i) * 64 + n] = specials[n], hm[i * 64 + 20 + n] = specials[(n + i) % 12];
    }
    hx[13] = 1e-39f; hx[14] = -2e-39f; hx[15] = 5e-41f;
    float *dx, *dm, *o1, *o2;
    hipMalloc(&dx, 128); hipMalloc(&dm, 8192); hipMalloc(&o1, 8192); hipMalloc(&o2, 8192);
    hipMemcpy(dx, hx.data(), 128, hipMemcpyHostToDevice);
    hipMemcpy(dm, hm.data(), 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(exact_kernel, dim3(1), dim3(64), 0, 0, dx, dm, o1, o2);
    std::vector<float> r1(2048), r2(2048);
    hipMemcpy(r1.data(), o1, 8192, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), o2, 8192, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 2048; ++i) {
        unsigned a, b;
        memcpy(&a, &r1[i], 4); memcpy(&b, &r2[i], 4);
        const bool bothnan = std::isnan(r1[i]) && std::isnan(r2[i]);
        if (a != b && !bothnan) {
            if (bad < 10)
                printf("MISMATCH dim %d node %d: x=%g m=%g mfma=%g (%08x) valu=%g (%08x)\n", i / 64, i % 64, hx[i / 64], hm[i], r1[i], a, r2[i], b);
            ++bad;
        }
    }
    printf("exactness: %d mismatches of 2048\n", bad);

    // ---- timing ----
    const int ns = 4096;
    float *xs, *out;
    float2 *cw;
    hipMalloc(&xs, (size_t)ns * 32 * 4);
    hipMalloc(&cw, (size_t)ns * 8192 * 8);
    hipMemset(xs, 0, (size_t)ns * 32 * 4);
    hipMemset(cw, 0, (size_t)ns * 8192 * 8);
    for (int wps = 1; wps <= 4; ++wps) {
        const int nwaves = 1024 * wps;          // waves per SIMD = wps
        hipMalloc(&out, (size_t)nwaves * 64 * 4);
        for (int variant = 0; variant < 2; ++variant) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (variant)
                    hipLaunchKernelGGL(loop_kernel<true>, dim3(nwaves / 4), dim3(256), 0, 0, xs, cw, ns, out);
                else
                    hipLaunchKernelGGL(loop_kernel<false>, dim3(nwaves / 4), dim3(256), 0, 0, xs, cw, ns, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double elems = (double)nwaves * 2048.0 * ns;
            printf("waves/SIMD %d  %s: %.3f ms  -> %.1f Gelem-steps/s (C3 update = 52.6 Gelem-steps)\n", wps,
                   variant ? "MFMA delta" : "VALU delta", ms, elems / ms / 1e6);
        }
        hipFree(out);
    }
    return 0;
}
