// vsom_compact.hip -- exact retirement of the columns that are zero in EVERY row of the staged chunk.
//
// Som::trainBatchSomEpoch on such a column d (Som.cpp:840-875, Standard and Median steppers):
//   phase 2 : the chain starts at M = 0, S = 0 and every step sees x = 0: delta = 0 - M = 0 (sign(0) = 0),
//             M = M + c*0, S = S + (w*0)*0 -- M and S stay +0 for every node whose coefficients are finite, and
//             become NaN for a node whose FIRST weight underflowed to 0 (c_1 = 0/0, SURVEY Q7: NaN*0 = NaN, and
//             from the second sample on delta = 0 - NaN).  map[d] = (c_1 is NaN) ? NaN : +0, and
//             sigmaMap[d] = sqrt(S/W) = the same (W = 0 for a one-sample chunk, W > 0 otherwise).
//   phase 1 : the MFMA shortlist contracts <x, M> over the columns; a column with x = 0 adds exactly 0 to every
//             dot product (the norms |M|^2, the error bound, the exact-order refinement and the NaN / inf
//             screening all keep reading the whole rows, so a NaN or inf model value in such a column still
//             reaches the redo list through the norm).
// So the chain kernels and the contraction run on the LIVE columns only, gathered into dense scratch matrices,
// and an expansion pass writes the model rows back in the reference's layout.  Which columns are live is
// data-dependent and known on the device only: the kernels read the counts from `cc_meta`, the host launches
// full-size grids whose surplus workgroups exit at once, and nothing synchronises.  The count is also mirrored
// into pinned host memory; when the last chunks had (almost) no dead column the host skips the passes for a
// few chunks (dense data pays ~2 % for nothing otherwise).
//
// MNIST (BASELINE configs 2-3): 67 of the 784 pixels are zero in every training image and ~120 in a
// 4096-image chunk (tests/gen.py reproduces 661 live); the exact results are unchanged bit for bit.
#include "vsom_device.hpp"

#include <cstdlib>

// meta: [0] live columns Kc (the chain kernels derive their live column quads from it), [1] live column quads ceil(Kc/4),
//       [2] Kc rounded up to 32 (K of the contraction), [3] sequence number
__global__ __launch_bounds__(1024) void cc_scan_kernel(unsigned *__restrict__ flags, int D, int cpitch,
                                                       int *__restrict__ idx, int *__restrict__ inv,
                                                       unsigned *__restrict__ meta, unsigned *__restrict__ host_fb)
{
    __shared__ int wsum[16];
    __shared__ int base;
    if (threadIdx.x == 0)
        base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d0 = 0; d0 < D; d0 += 1024) {
        const int d = d0 + (int)threadIdx.x;
        const int f = d < D && flags[d] ? 1 : 0;
        if (d < D)
            flags[d] = 0u;              // cleared for the next chunk's staging pass (stage_rows_kernel sets them)
        const unsigned long long m = __ballot(f);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0)
            wsum[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w)
            off += wsum[w];
        if (d < D) {
            if (f)
                idx[off + before] = d;
            inv[d] = f ? off + before : -1;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w)
                t += wsum[w];
            base += t;
        }
        __syncthreads();
    }
    const int kc = base;
    for (int k = kc + (int)threadIdx.x; k < cpitch; k += 1024)
        idx[k] = -1;
    if (threadIdx.x == 0) {
        meta[0] = (unsigned)kc;
        meta[1] = (unsigned)((kc + 3) / 4);
        meta[2] = (unsigned)((kc + 31) / 32 * 32);
        meta[3] = meta[3] + 1u;
        host_fb[0] = (unsigned)kc;
        __threadfence_system();
        host_fb[1] = host_fb[1] + 1u;
    }
}

// dst[row][k] = src[row][idx[k]] for k < Kc, 0 up to the pitch: one workgroup per row
__global__ __launch_bounds__(256) void cc_gather_rows_kernel(const float *__restrict__ src, int lds_, float *__restrict__ dst,
                                                             int ldd, const int *__restrict__ idx, int nrows)
{
    const int row = blockIdx.x;
    if (row >= nrows)
        return;
    const float *s = src + (size_t)row * lds_;
    float *d = dst + (size_t)row * ldd;
    for (int k = threadIdx.x; k < ldd; k += 256) {
        const int c = idx[k];
        d[k] = c >= 0 ? s[c] : 0.f;
    }
}

// model rows back in the reference's layout: live column d <- compacted column inv[d] (map: the chain's M;
// sigmaMap: sqrt(S / W), Som.cpp:873), dead column <- +0, or NaN when the node's first coefficient is 0/0
// (header); the padding columns of the rows are put to zero as sigma_finalize_kernel does
__global__ __launch_bounds__(256) void cc_expand_kernel(const float *__restrict__ um, const float *__restrict__ us, int ldc,
                                                        const int *__restrict__ inv, float *__restrict__ map,
                                                        float *__restrict__ sigma, int pitch, int D, int n0, int nloc,
                                                        const float *__restrict__ weight, const float4 *__restrict__ cw2)
{
    const int nl = blockIdx.x;
    if (nl >= nloc)
        return;
    const size_t node = (size_t)n0 + nl;
    const float Wf = weight[node];
    const float c1 = cw2[nl].x;                 // pair-row 0: {c_1, w_1, c_2, w_2} of this node
    const float dead_m = c1 * 0.f;              // NaN when c_1 is NaN, else +0 (c >= 0)
    const float dead_s = sqrtf(dead_m / Wf);
    const float *pm = um + node * ldc, *ps = us + node * ldc;
    for (int d = threadIdx.x; d < pitch; d += 256) {
        float m = 0.f, s = 0.f;
        if (d < D) {
            const int k = inv[d];
            if (k >= 0) {
                m = pm[k];
                s = sqrtf(ps[k] / Wf);
            } else {
                m = dead_m;
                s = dead_s;
            }
        }
        map[node * pitch + d] = m;
        sigma[node * pitch + d] = s;
    }
}

static bool cc_env_enabled()
{
    static int v = -1;
    if (v < 0) {
        const char *e = std::getenv("VSOM_NO_COMPACT");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v != 0;
}

// whether this context's shapes can use the compaction at all
bool vsom_cc_applies(const vsom_ctx *c)
{
    if (!cc_env_enabled())
        return false;
    if (c->transform != VSOM_STANDARD && c->transform != VSOM_MEDIAN)
        return false;
    if (c->D < 64)           // the chain / tiny kernels, and nothing worth retiring
        return false;
    // only the lane = node chain kernels and the MFMA shortlist consume the compaction: maps below their
    // thresholds (launch_phase2: VSOM_CHAIN_MAX_WAVES; launch_bmu_full: 1024 nodes) skip the passes
    const size_t waves = ((size_t)c->N + 63) / 64 * ((c->D + 13) / 14);
    return waves > VSOM_CHAIN_MAX_WAVES || c->N >= 1024;
}

static int cc_ensure(vsom_ctx *c)
{
    if (!c->cpitch)
        c->cpitch = (c->D + 31) / 32 * 32;
    if (!c->cc_meta) {
        VSOM_HIP_CHECK(hipMalloc(&c->cc_flags, (size_t)c->xpitch * 4));
        VSOM_HIP_CHECK(hipMemsetAsync(c->cc_flags, 0, (size_t)c->xpitch * 4, c->stream));
        VSOM_HIP_CHECK(hipMalloc(&c->cc_idx, (size_t)c->cpitch * 4));
        VSOM_HIP_CHECK(hipMalloc(&c->cc_inv, (size_t)c->xpitch * 4));
        VSOM_HIP_CHECK(hipMalloc(&c->cc_idx_alt, (size_t)c->cpitch * 4));      // the record of a chunk staged ahead
        VSOM_HIP_CHECK(hipMalloc(&c->cc_inv_alt, (size_t)c->xpitch * 4));
        const size_t meta_bytes = 64;
        VSOM_HIP_CHECK(hipMalloc(&c->cc_meta, meta_bytes));
        VSOM_HIP_CHECK(hipMemsetAsync(c->cc_meta, 0, meta_bytes, c->stream));
        VSOM_HIP_CHECK(hipMalloc(&c->cc_meta_alt, meta_bytes));
        VSOM_HIP_CHECK(hipMemsetAsync(c->cc_meta_alt, 0, meta_bytes, c->stream));
        VSOM_HIP_CHECK(hipHostMalloc(&c->cc_fb, 64));
        c->cc_fb[0] = 0u;
        c->cc_fb[1] = 0u;
    }
    const size_t need = (c->Bcap + VSOM_ROW_PAD) * (size_t)c->cpitch;
    if (need > c->Xc_cap) {
        if (c->Xc) {
            VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
            VSOM_HIP_CHECK(hipFree(c->Xc));
        }
        c->Xc = nullptr;
        c->Xc_cap = 0;
        VSOM_HIP_CHECK(hipMalloc(&c->Xc, need * 4));
        // the spare rows behind the chunk are read ahead (never consumed) by the pipelined update kernels
        VSOM_HIP_CHECK(hipMemsetAsync(c->Xc, 0, need * 4, c->stream));
        c->Xc_cap = need;
    }
    return VSOM_OK;
}

// before the rows are staged: does this chunk get the compaction?  (then stage_rows_kernel also flags the live columns)
int vsom_cc_begin(vsom_ctx *c, size_t B, bool *on)
{
    *on = false;
    // small chunks: the passes (and the model-row gather / expansion around them) cost more than a few retired
    // column quads of a short chain save; vsom_set_column_compaction moves the threshold
    if (!vsom_cc_applies(c) || B == 0 || c->cc_min_rows < 0 || (long)B < c->cc_min_rows)
        return VSOM_OK;
    // feedback of earlier chunks (pinned memory, read without synchronising: stale values only delay the decision)
    volatile unsigned *fb = c->cc_fb;
    if (fb && fb[1] != c->cc_seen) {
        c->cc_seen = fb[1];
        const unsigned kc = fb[0];
        if ((kc + 3) / 4 >= (c->D + 3) / 4)      // not even one column quad to retire
            c->cc_skip = 8;
    }
    if (c->cc_skip > 0) {
        --c->cc_skip;
        return VSOM_OK;
    }
    int rc = cc_ensure(c);
    if (rc)
        return rc;
    *on = true;
    return VSOM_OK;
}

// after the rows are staged (and their live columns flagged): the live-column record, and the chunk gathered onto them
int vsom_cc_stage(vsom_ctx *c, size_t B, hipStream_t stream, int *idx, int *inv, unsigned *meta, bool *xi_out)
{
    hipLaunchKernelGGL(cc_scan_kernel, dim3(1), dim3(1024), 0, stream, c->cc_flags, (int)c->D, (int)c->cpitch, idx, inv, meta,
                       c->cc_fb);
    // the chunk's rows gathered onto the live columns; with the integer shortlist's buffers in place (vsom_sl_i8.hip
    // allocates them at the first search) the same pass writes the chunk's int8 images
    return launch_sl_gather_quant(c, B, stream, idx, xi_out);
}

// the model rows gathered onto the live columns, for the contraction of the shortlist search
int vsom_cc_gather_map(vsom_ctx *c)
{
    const size_t need = (size_t)c->N * c->cpitch;
    if (!c->Mc) {
        VSOM_HIP_CHECK(hipMalloc(&c->Mc, need * 4));
    }
    hipLaunchKernelGGL(cc_gather_rows_kernel, dim3((unsigned)c->N), dim3(256), 0, c->stream, c->map, (int)c->pitch, c->Mc,
                       (int)c->cpitch, c->cc_idx, (int)c->N);
    VSOM_HIP_CHECK(hipGetLastError());
    return VSOM_OK;
}

int vsom_cc_ensure_update_scratch(vsom_ctx *c)
{
    const size_t need = (size_t)c->N * c->cpitch;
    if (!c->Uc_map) {
        VSOM_HIP_CHECK(hipMalloc(&c->Uc_map, need * 4));
        VSOM_HIP_CHECK(hipMalloc(&c->Uc_S, need * 4));
    }
    return VSOM_OK;
}

int vsom_cc_expand(vsom_ctx *c, size_t n0, size_t nloc)
{
    hipLaunchKernelGGL(cc_expand_kernel, dim3((unsigned)nloc), dim3(256), 0, c->stream, c->Uc_map, c->Uc_S, (int)c->cpitch,
                       c->cc_inv, c->map, c->sigma, (int)c->pitch, (int)c->D, (int)n0, (int)nloc, c->weight,
                       reinterpret_cast<const float4 *>(c->cw));
    VSOM_HIP_CHECK(hipGetLastError());
    return VSOM_OK;
}
