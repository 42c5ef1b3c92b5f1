// vsom_xq.hip -- the staged chunk, transposed for the lane = node / four-dims-per-wavefront chain kernels
// (gen_nt_asm.py; Som::trainBatchSomEpoch phase 2, Som.cpp:840-870).
//
//   Xq[q][j] = float4 { x_j[4q], x_j[4q+1], x_j[4q+2], x_j[4q+3] }    q = column quad, j = sample
//   zq[q][j / 32] bit (j % 32) = "the four values are all +-0"          (the kernels' 5-operation step)
//
// A wavefront of those kernels owns one quad and walks the samples: with this layout ONE s_load_dwordx16
// brings its operands of four consecutive samples (a 64-byte line), as wave-uniform SGPR operands of the packed
// arithmetic.  Rows are padded to `bpad` = roundup(B, 32) + 32 samples (the kernels read one group ahead) and
// the quads to a multiple of 8 (a workgroup is 8 quads); everything outside the chunk reads as zero.
// The source is the staged row-major chunk: Xs, or Xc (the chunk gathered onto its live columns,
// vsom_compact.hip) -- one pass over ~13 MB per 4096 x 784 chunk.
#include "vsom_device.hpp"

__global__ __launch_bounds__(256) void xq_transpose_kernel(const float *__restrict__ src, int ld, int B, int bpad, int nq8, float scale,
                                                           float4 *__restrict__ xq, unsigned *__restrict__ zq)
{
    __shared__ float4 tile[16][65];
    const int j0 = blockIdx.x * 64, q0 = blockIdx.y * 16, t = threadIdx.x;
    for (int pass = 0; pass < 4; ++pass) {
        const int js = pass * 16 + (t >> 4), qq = t & 15;
        const int j = j0 + js, col = 4 * (q0 + qq);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < B && col + 3 < ld)                       // ld is a multiple of 32: whole quads
            v = *reinterpret_cast<const float4 *>(src + (size_t)j * ld + col);
        v.x *= scale;                                    // Median: x * 2^24, exact (gen_nt_asm.py); else 1
        v.y *= scale;
        v.z *= scale;
        v.w *= scale;
        tile[qq][js] = v;
    }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
        const int qq = pass * 4 + (t >> 6), js = t & 63;
        const int q = q0 + qq, j = j0 + js;
        const float4 v = tile[qq][js];
        const bool z = v.x == 0.f && v.y == 0.f && v.z == 0.f && v.w == 0.f;
        const unsigned long long m = __ballot(z);
        if (q < nq8 && j < bpad) {
            xq[(size_t)q * bpad + j] = v;
            if (js == 0)
                zq[(size_t)q * (bpad >> 5) + (j >> 5)] = (unsigned)m;
            if (js == 32)
                zq[(size_t)q * (bpad >> 5) + (j >> 5)] = (unsigned)(m >> 32);
        }
    }
}

// Xq / zq of the staged chunk (built once per chunk, on first use by a phase 2)
int vsom_xq_ensure(vsom_ctx *c)
{
    if (c->xq_valid)
        return VSOM_OK;
    const bool compact = c->cc_valid;
    const uint32_t cols = compact ? c->cpitch : c->xpitch;
    const uint32_t nq8 = (cols / 4 + 7) / 8 * 8;
    const size_t bpad_cap = (c->Bcap + 31) / 32 * 32 + 32;
    const size_t need = (size_t)nq8 * bpad_cap;                     // float4 elements
    if (need > c->Xq_cap) {
        if (c->Xq) {
            VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
            VSOM_HIP_CHECK(hipFree(c->Xq));
            VSOM_HIP_CHECK(hipFree(c->zq));
        }
        c->Xq = nullptr;
        c->zq = nullptr;
        c->Xq_cap = 0;
        VSOM_HIP_CHECK(hipMalloc(&c->Xq, need * sizeof(float4)));
        VSOM_HIP_CHECK(hipMalloc(&c->zq, need / 32 * sizeof(unsigned)));
        c->Xq_cap = need;
    }
    const uint32_t bpad = (uint32_t)((c->B + 31) / 32 * 32 + 32);
    hipLaunchKernelGGL(xq_transpose_kernel, dim3((bpad + 63) / 64, nq8 / 16 + (nq8 % 16 ? 1 : 0)), dim3(256), 0, c->stream,
                       compact ? c->Xc : c->Xs, (int)cols, (int)c->B, (int)bpad, (int)nq8,
                       c->transform == VSOM_MEDIAN ? 0x1.0p24f : 1.f, reinterpret_cast<float4 *>(c->Xq),
                       c->zq);
    VSOM_HIP_CHECK(hipGetLastError());
    c->xq_bpad = bpad;
    c->xq_quads = nq8;
    c->xq_valid = true;
    return VSOM_OK;
}
