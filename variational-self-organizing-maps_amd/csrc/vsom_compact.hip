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
// full-size grids whose surplus wavefronts exit at once, and nothing synchronises.  The count is also mirrored
// into pinned host memory; when the last chunks had (almost) no dead column the host skips the passes for a
// few chunks (dense data pays ~2 % for nothing otherwise).
//
// MNIST (BASELINE configs 2-3): 67 of the 784 pixels are zero in every training image and ~120 in a
// 4096-image chunk (tests/gen.py reproduces 661 live); the exact results are unchanged bit for bit.
#include "vsom_device.hpp"

#include <cstdlib>

// meta: [0] live columns Kc, [1] live 14-dim slices ceil(Kc/14), [2] Kc rounded up to 32 (K of the contraction),
//       [3] sequence number; [16 + L] the PHYSICAL slice the chain kernels' logical slice L works on (identity until
//       cc_slice_order_kernel sorts the slices by how often they are all zero)
__global__ __launch_bounds__(256) void cc_flag_kernel(const float *__restrict__ xs, int ldx, int B, int D,
                                                      unsigned *__restrict__ flags)
{
    // block = 16 rows x all columns; a column is live when any row holds something != 0 (NaN counts)
    const int r0 = blockIdx.x * 16, r1 = r0 + 16 < B ? r0 + 16 : B;
    for (int d = threadIdx.x; d < D; d += 256) {
        bool live = false;
        for (int r = r0; r < r1; ++r) {
            const float v = xs[(size_t)r * ldx + d];
            live |= !(v == 0.f);
        }
        if (live && flags[d] == 0u)
            flags[d] = 1u;          // every writer stores the same value: no atomic needed
    }
}

__global__ __launch_bounds__(1024) void cc_scan_kernel(const unsigned *__restrict__ flags, int D, int cpitch,
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
    for (int q = threadIdx.x; q < (cpitch + 13) / 14 + 4; q += 1024)
        meta[16 + q] = (unsigned)q;
    if (threadIdx.x == 0) {
        meta[0] = (unsigned)kc;
        meta[1] = (unsigned)((kc + 13) / 14);
        meta[2] = (unsigned)((kc + 31) / 32 * 32);
        meta[3] = meta[3] + 1u;
        host_fb[0] = (unsigned)kc;
        __threadfence_system();
        host_fb[1] = host_fb[1] + 1u;
    }
}

// Zero-slice mask for the Standard chain kernels' fast path (gen_update_asm.py, compute_zero_x): bit j of word
// zmask[q * ldz + i] says that the 14 values of slice q (compacted columns [14 q, 14 q + 14)) of sample 32 i + j
// are all zero (+0 or -0; NaN / inf / anything else is not).  One workgroup = 32 rows x CC_ZQ slices: a thread
// tests one (row, slice) block -- 56 contiguous, 8-byte aligned bytes; consecutive threads read consecutive
// slices of a row -- and thread q then assembles the 32-sample word of its slice.  Words past the chunk stay
// zero (the kernels read two words ahead).
#define CC_ZQ 16
__global__ __launch_bounds__(256) void cc_zmask_kernel(const float *__restrict__ xc, int ldc, int B, int nslices_max,
                                                       const unsigned *__restrict__ meta, unsigned *__restrict__ zmask, int ldz)
{
    __shared__ unsigned char zero[32][CC_ZQ];
    const int nsl = (int)meta[1] < nslices_max ? (int)meta[1] : nslices_max;
    const int q0 = blockIdx.y * CC_ZQ, r0 = blockIdx.x * 32;
    if (q0 >= nsl)
        return;                                            // block-uniform
    for (int t = threadIdx.x; t < 32 * CC_ZQ; t += 256) {
        const int rr = t / CC_ZQ, q = q0 + t % CC_ZQ, r = r0 + rr;
        bool z = false;
        if (r < B && q < nsl) {
            const float2 *p = reinterpret_cast<const float2 *>(xc + (size_t)r * ldc + 14 * q);
            float2 v[7];
#pragma unroll
            for (int u = 0; u < 7; ++u)
                v[u] = p[u];
            z = true;
#pragma unroll
            for (int u = 0; u < 7; ++u)
                z = z && (v[u].x == 0.f) && (v[u].y == 0.f);
        }
        zero[rr][t % CC_ZQ] = z ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x < CC_ZQ && q0 + (int)threadIdx.x < nsl) {
        unsigned word = 0u;
        for (int j = 0; j < 32; ++j)
            word |= (unsigned)zero[j][threadIdx.x] << j;
        zmask[(size_t)(q0 + threadIdx.x) * ldz + blockIdx.x] = word;
    }
}

// Slice order for the chain kernels: a workgroup is four consecutive LOGICAL slices, and with the (c,w) stream
// shared through LDS its barrier holds it to the slowest of them per 8 samples -- so slices that take the
// zero-slice form about equally often should sit together.  Logical slice L works on physical slice order[L]:
// the live slices sorted by their count of all-zero samples, descending (ties: lower slice first).  Any
// permutation gives the same results (it only assigns chains to wavefronts).
#define CC_ORDER_MAX 512
__global__ __launch_bounds__(512) void cc_slice_order_kernel(const unsigned *__restrict__ zmask, int ldz, int nwords,
                                                             unsigned *__restrict__ meta, int interleave)
{
    __shared__ int cnt[CC_ORDER_MAX];
    const int nsl = (int)meta[1];
    if (nsl > CC_ORDER_MAX)
        return;                                   // (identity order stays)
    const int q = threadIdx.x;
    if (q < nsl) {
        int c = 0;
        for (int i = 0; i < nwords; ++i)
            c += __popc(zmask[(size_t)q * ldz + i]);
        cnt[q] = c;
    }
    __syncthreads();
    if (q < nsl) {
        int rank = 0;
        for (int j = 0; j < nsl; ++j)
            rank += (cnt[j] > cnt[q] || (cnt[j] == cnt[q] && j < q)) ? 1 : 0;
        // interleave: workgroup k takes the slices ranked k, k + nq, k + 2 nq, k + 3 nq -- the same mix everywhere
        int pos = rank;
        if (interleave) {
            const int nq = (nsl + 3) / 4;
            pos = 4 * (rank % nq) + rank / nq;
            if (pos >= nsl)          // ragged last quad: keep it a permutation of 0 .. nsl-1
                pos = rank;
        }
        meta[16 + pos] = (unsigned)q;
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

static bool cc_zero_path_enabled()
{
    static int v = -1;
    if (v < 0) {
        const char *e = std::getenv("VSOM_NO_ZERO_PATH");     // development: time the chains without the fast path
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v != 0;
}

// whether this context's shapes can use the compaction at all
bool vsom_cc_applies(const vsom_ctx *c)
{
    if (!cc_env_enabled() || !c->use_asm)
        return false;
    if (c->transform != VSOM_STANDARD && c->transform != VSOM_MEDIAN)
        return false;
    if (c->D < 64)           // the chain / tiny kernels, and nothing worth retiring
        return false;
    // only the lane = node assembly kernels and the MFMA shortlist consume the compaction: maps below their
    // thresholds (launch_phase2: VSOM_CHAIN_MAX_WAVES wavefronts; launch_bmu_full: 1024 nodes) skip the passes
    const size_t waves = ((size_t)c->N + 63) / 64 * ((c->D + 13) / 14);
    return waves > VSOM_CHAIN_MAX_WAVES || c->N >= 1024;
}

static int cc_ensure(vsom_ctx *c)
{
    if (!c->cpitch)
        c->cpitch = ((c->D + 13) / 14 * 14 + 31) / 32 * 32;
    if (!c->cc_meta) {
        VSOM_HIP_CHECK(hipMalloc(&c->cc_flags, (size_t)c->xpitch * 4));
        VSOM_HIP_CHECK(hipMalloc(&c->cc_idx, (size_t)c->cpitch * 4));
        VSOM_HIP_CHECK(hipMalloc(&c->cc_inv, (size_t)c->xpitch * 4));
        const size_t meta_bytes = 64 + 4 * ((size_t)(c->cpitch + 13) / 14 + 8);
        VSOM_HIP_CHECK(hipMalloc(&c->cc_meta, meta_bytes));
        VSOM_HIP_CHECK(hipMemsetAsync(c->cc_meta, 0, meta_bytes, c->stream));
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

// after the rows are staged: which columns are live, and the chunk gathered onto them
int vsom_cc_stage(vsom_ctx *c)
{
    c->cc_valid = false;
    c->cc_zmask_valid = false;
    // small chunks: the passes (and the model-row gather / expansion around them) cost more than a few retired
    // slices of a short chain save; vsom_set_column_compaction moves the threshold
    if (!vsom_cc_applies(c) || c->B == 0 || c->cc_min_rows < 0 || (long)c->B < c->cc_min_rows)
        return VSOM_OK;
    // feedback of earlier chunks (pinned memory, read without synchronising: stale values only delay the decision)
    volatile unsigned *fb = c->cc_fb;
    if (fb && fb[1] != c->cc_seen) {
        c->cc_seen = fb[1];
        const unsigned kc = fb[0];
        if (kc + 14 > c->D)          // not even one 14-dim slice to retire
            c->cc_skip = 8;
    }
    if (c->cc_skip > 0) {
        --c->cc_skip;
        return VSOM_OK;
    }
    int rc = cc_ensure(c);
    if (rc)
        return rc;
    VSOM_HIP_CHECK(hipMemsetAsync(c->cc_flags, 0, (size_t)c->xpitch * 4, c->stream));
    hipLaunchKernelGGL(cc_flag_kernel, dim3((unsigned)((c->B + 15) / 16)), dim3(256), 0, c->stream, c->Xs, (int)c->xpitch,
                       (int)c->B, (int)c->D, c->cc_flags);
    hipLaunchKernelGGL(cc_scan_kernel, dim3(1), dim3(1024), 0, c->stream, c->cc_flags, (int)c->D, (int)c->cpitch, c->cc_idx,
                       c->cc_inv, c->cc_meta, c->cc_fb);
    hipLaunchKernelGGL(cc_gather_rows_kernel, dim3((unsigned)c->B), dim3(256), 0, c->stream, c->Xs, (int)c->xpitch, c->Xc,
                       (int)c->cpitch, c->cc_idx, (int)c->B);
    c->cc_zmask_valid = false;        // built on demand by vsom_cc_ensure_zmask
    VSOM_HIP_CHECK(hipGetLastError());
    c->cc_valid = true;
    return VSOM_OK;
}

// zero-slice mask of the gathered chunk, built once per chunk when a Standard phase 2 asks for it
int vsom_cc_ensure_zmask(vsom_ctx *c)
{
    if (c->cc_zmask_valid)
        return VSOM_OK;
    if (!c->cc_valid || !cc_zero_path_enabled())
        return VSOM_OK;
    const size_t nslm = (c->D + 13) / 14, ldz = (c->B + 31) / 32 + 2, need = nslm * ldz;
    if (need > c->cc_zmask_cap) {
        if (c->cc_zmask) {
            VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
            VSOM_HIP_CHECK(hipFree(c->cc_zmask));
        }
        c->cc_zmask = nullptr;
        c->cc_zmask_cap = 0;
        VSOM_HIP_CHECK(hipMalloc(&c->cc_zmask, need * 4));
        c->cc_zmask_cap = need;
    }
    VSOM_HIP_CHECK(hipMemsetAsync(c->cc_zmask, 0, need * 4, c->stream));
    hipLaunchKernelGGL(cc_zmask_kernel, dim3((unsigned)((c->B + 31) / 32), (unsigned)((nslm + CC_ZQ - 1) / CC_ZQ)), dim3(256), 0,
                       c->stream, c->Xc, (int)c->cpitch, (int)c->B, (int)nslm, c->cc_meta, c->cc_zmask, (int)ldz);
    // measured and left OFF: with the slices sorted every workgroup is homogeneous, but the workgroups then differ
    // from each other as much as they can, and the launch got 1-3 % slower in all three arithmetics (C3 update
    // strict 4.45 -> 4.51, sigma-contracted 3.81 -> 3.93, contracted 3.20 -> 3.27 ms); the opposite, every workgroup
    // the same mix (interleaved ranks), is within noise of the raster order (4.42 / 3.89 / 3.23 vs 4.47 / 3.88 / 3.22).
    // VSOM_SLICE_ORDER=1 (sorted) / 2 (interleaved) enable them
    static int order_env = -1;
    if (order_env < 0) {
        const char *e = std::getenv("VSOM_SLICE_ORDER");
        order_env = e ? std::atoi(e) : 0;
    }
    if (order_env && nslm <= CC_ORDER_MAX)
        hipLaunchKernelGGL(cc_slice_order_kernel, dim3(1), dim3(512), 0, c->stream, c->cc_zmask, (int)ldz,
                           (int)((c->B + 31) / 32), c->cc_meta, order_env == 2 ? 1 : 0);
    VSOM_HIP_CHECK(hipGetLastError());
    c->cc_zmask_valid = true;
    return VSOM_OK;
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
