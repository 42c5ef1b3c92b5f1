// vsom_bmu.hip -- BMU search kernels (gfx950).
//
//  bmu_tile_kernel   : Som::findBmu (Som.cpp:291-309) for a 64-sample x 64-node tile per
//                      workgroup; every distance is evaluated in the reference's fp32 order
//                      (8 class accumulators + Eigen's reduction tree) on the VALU.
//  bmu_reduce_kernel : per-sample argmin over the node tiles (strict <, lowest index, NaN
//                      rules of Som.cpp:293-304).
//  bmu_local_kernel  : Som::findLocalBmu (Som.cpp:335-454), one wavefront per sample,
//                      8 candidates x 8 accumulator classes across the 64 lanes.
//  pair_dist_kernel  : Som::euclidianWeightedDist for arbitrary (node,row) pairs.
//  finish_kernel     : bmuHits[idx] += 1 and the fp32 MSE running sum in sample order
//                      (Som.cpp:777-781).
//  stage kernels     : chunk re-layout (zero-padded rows; CLR x'/y' expansion).
#include "vsom_device.hpp"
#include "vsom_digits.hpp"
#include <algorithm>
#include <utility>

// ------------------------------------------------------------------------------------------
// chunk staging
// ------------------------------------------------------------------------------------------
// one workgroup = 16 rows: zero-padded copy into Xs, lastBMU of those rows zeroed (DataSet::loadNextDataFromStream,
// DataSet.cpp:136-137), and -- flags != null: the column compaction is on for this chunk (vsom_compact.hip) -- a column
// is flagged live when any of the rows holds something != 0 (NaN counts; every writer stores the same value, no
// atomic needed; cc_scan_kernel clears the flags again after reading them).  xflag: the chunk's data-kind word of the
// integer shortlist (vsom_sl_i8.hip), cleared for the new chunk.
__global__ __launch_bounds__(256) void stage_rows_kernel(const float *__restrict__ x, int J, int B, float *__restrict__ xs,
                                                         int xpitch, u64 *__restrict__ lastbmu, unsigned *__restrict__ flags,
                                                         unsigned *__restrict__ xflag)
{
    const int r0 = blockIdx.x * 16, r1 = r0 + 16 < B ? r0 + 16 : B;
    if (threadIdx.x < 16 && r0 + (int)threadIdx.x < B)
        lastbmu[r0 + threadIdx.x] = 0;
    if (xflag && blockIdx.x == 0 && threadIdx.x < 33)
        xflag[32 * threadIdx.x] = 0u;                    // the word and its 32 write slots (sl_kind_mark, vsom_sl_i8.hip)
    for (int d = threadIdx.x; d < xpitch; d += 256) {
        // the 16 loads of a column first, then the stores (a rolled row loop waits for every load before its store:
        // 64 round trips per thread, 27 us for C3's chunk with one workgroup per CU)
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            v[i] = (d < J && r0 + i < r1) ? x[(size_t)(r0 + i) * J + d] : 0.f;
        bool live = false;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (r0 + i < r1)
                xs[(size_t)(r0 + i) * xpitch + d] = v[i];
            live |= !(v[i] == 0.f);
        }
        if (flags && live && flags[d] == 0u)
            flags[d] = 1u;
    }
}

// x'_p = x[i(p)], y'_p = x[j(p)]  (Transformation.cpp:94-101)
__global__ void stage_pairs_kernel(const float *__restrict__ x, int J, int B, int P,
                                   const int *__restrict__ pi, const int *__restrict__ pj,
                                   float *__restrict__ xp, float *__restrict__ yp, int ppitch)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)B * ppitch;
    if (i >= total)
        return;
    int s = (int)(i / ppitch), p = (int)(i % ppitch);
    float a = 0.f, b = 0.f;
    if (p < P) {
        a = x[(size_t)s * J + pi[p]];
        b = x[(size_t)s * J + pj[p]];
    }
    xp[i] = a;
    yp[i] = b;
}

// the staging kernels of a chunk of B rows on `stream`: rows -> Xs (+ lastBMU := 0 in `lastbmu`), CLR pair rows, and with
// the column compaction its record (idx / inv / meta) and the gathered rows / int8 images
static int stage_chunk_on(vsom_ctx *c, const float *x_dev, size_t B, hipStream_t stream, u64 *lastbmu, int *idx, int *inv,
                          unsigned *meta, bool *cc_out, bool *xi_out)
{
    *cc_out = false;
    *xi_out = false;
    if (B == 0)
        return VSOM_OK;
    bool cc = false;
    if (int rc = vsom_cc_begin(c, B, &cc))      // does this chunk get the column compaction? (buffers, skip counters)
        return rc;
    if (cc && !idx) {                            // (buffers allocated by vsom_cc_begin just now)
        idx = c->cc_idx;
        inv = c->cc_inv;
        meta = c->cc_meta;
    }
    hipLaunchKernelGGL(stage_rows_kernel, dim3((unsigned)((B + 15) / 16)), dim3(256), 0, stream, x_dev, (int)c->J,
                       (int)B, c->Xs, (int)c->xpitch, lastbmu, cc ? c->cc_flags : (unsigned *)nullptr,
                       c->sl_scal ? c->sl_scal + 8192 : (unsigned *)nullptr);
    if (c->transform == VSOM_CLR) {
        size_t total = B * c->part_pitch;
        hipLaunchKernelGGL(stage_pairs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           stream, x_dev, (int)c->J, (int)B, (int)c->part_len, c->pair_i,
                           c->pair_j, c->XP, c->YP, (int)c->part_pitch);
    }
    VSOM_HIP_CHECK(hipGetLastError());
    if (cc) {                                    // live-column record of this chunk (vsom_compact.hip)
        if (int rc = vsom_cc_stage(c, B, stream, idx, inv, meta, xi_out))
            return rc;
        *cc_out = true;
    }
    return VSOM_OK;
}

int launch_stage_chunk(vsom_ctx *c, const float *x_dev, size_t B)
{
    TimerScope ts(c, VSOM_T_STAGE);
    c->cc_valid = false;
    c->xq_valid = false;
    c->xi_valid = false;
    c->rows_free_valid = false;
    if (c->ahead_rows) {
        // staging kernels of a chunk staged ahead (adopted or abandoned since) may still be writing the buffers this
        // staging writes: order behind them.  A chunk staged ahead is overwritten by this one; its rows are staged anew
        // when it is committed.
        VSOM_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_ahead, 0));
        c->ahead_rows = false;
    }
    c->ahead_valid = false;
    bool cc = false, xi = false;
    int rc = stage_chunk_on(c, x_dev, B, c->stream, c->lastbmu, c->cc_idx, c->cc_inv, c->cc_meta, &cc, &xi);
    c->cc_valid = cc;
    c->xi_valid = xi;
    return rc;
}

// Can the next chunk be staged NOW, beside whatever the context's stream is running?  Needs: every buffer in place
// (no allocation: that would synchronise), a transformation whose phase 2 does not read the staged rows (CLR reads its
// pair rows, the small-map chain kernel the rows themselves), and an epoch enqueued on the current chunk whose
// rows-are-free event is still the last word on the context (launch_phase2 records it).
bool vsom_can_stage_ahead(const vsom_ctx *c, size_t B)
{
    if (c->transform == VSOM_CLR || !c->rows_free_valid || !c->ev_rows_free || !c->lastbmu_alt || B > c->Bcap)
        return false;
    if (vsom_cc_applies(c) && B > 0 && c->cc_min_rows >= 0 && (long)B >= c->cc_min_rows &&
        (!c->cc_meta_alt || (c->Bcap + VSOM_ROW_PAD) * (size_t)c->cpitch > c->Xc_cap))
        return false;
    return true;
}

int launch_stage_chunk_ahead(vsom_ctx *c, const float *x_dev, size_t B)
{
    // after the epoch's last reader of the staged rows (xq_transpose in launch_phase2) -- NOT after its chains
    VSOM_HIP_CHECK(hipStreamWaitEvent(c->copy_stream, c->ev_rows_free, 0));
    bool cc = false, xi = false;
    int rc = stage_chunk_on(c, x_dev, B, c->copy_stream, c->lastbmu_alt, c->cc_idx_alt, c->cc_inv_alt, c->cc_meta_alt, &cc, &xi);
    if (rc)
        return rc;
    VSOM_HIP_CHECK(hipEventRecord(c->ev_ahead, c->copy_stream));
    c->ahead_rows = true;     // until c->stream has waited for ev_ahead (adoption, or the next staging on c->stream)
    c->ahead_valid = true;
    c->ahead_B = B;
    c->ahead_cc = cc;
    c->ahead_xi = xi;
    return VSOM_OK;
}

// vsom_commit_chunk on a chunk staged ahead: the compute stream waits for the staging, the double-buffered pieces swap
int vsom_adopt_ahead(vsom_ctx *c)
{
    VSOM_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_ahead, 0));
    c->ahead_rows = false;
    std::swap(c->lastbmu, c->lastbmu_alt);
    std::swap(c->cc_idx, c->cc_idx_alt);
    std::swap(c->cc_inv, c->cc_inv_alt);
    std::swap(c->cc_meta, c->cc_meta_alt);
    c->B = c->ahead_B;
    c->chunk_loaded = true;
    c->cc_valid = c->ahead_cc;
    c->xi_valid = c->ahead_xi;
    c->xq_valid = false;
    c->rows_free_valid = false;
    c->ahead_valid = false;
    return VSOM_OK;
}

static DistArgs make_dist_args(const vsom_ctx *c)
{
    DistArgs a;
    if (c->transform == VSOM_CLR) {
        a.xa = c->XP;
        a.xb = c->YP;
        a.ldx = (int)c->part_pitch;
        a.ma = c->map;
        a.mb = c->map + c->part_pitch;
    } else {
        a.xa = c->Xs;
        a.xb = c->Xs;
        a.ldx = (int)c->xpitch;
        a.ma = c->map;
        a.mb = c->map;
    }
    a.ldm = (int)c->pitch;
    a.L = (int)c->part_len;
    return a;
}

// ------------------------------------------------------------------------------------------
// full search: 64 x 64 tile per workgroup, 4 x 4 pairs x 8 class accumulators per thread
// ------------------------------------------------------------------------------------------
#define TILE 64
#define LDT 36   // LDS row stride in floats: 16-B aligned, lane rows land on distinct 4-bank slots

// TI = sample rows per thread: 4 for Standard / Median (64 x 64 tile, 128 accumulators per thread).  The CLR
// residual needs two more operand arrays (y', B): with 4 x 4 pairs the kernel sat at 256 VGPRs with 46 spilled
// dwords in the hot loop and no room to prefetch (r2: VALU 54 % busy, 2.3 ms at C5).  CLR therefore takes
// TI = 2 -- a 32-sample x 64-node tile, 64 accumulators -- which leaves registers for the next K-chunk's loads
// in flight while the current one is consumed and lets three workgroups share a CU.
template <bool CLR, int TI>
__device__ __forceinline__ void bmu_tile_body(const DistArgs &a, int s0, int s1, int N,
                                              u64 *__restrict__ partial, int pstride,
                                              unsigned char *__restrict__ nan0,
                                              const int *__restrict__ slist,
                                              const u64 *__restrict__ hits, u64 min_hits, int by, int bx,
                                              const int *__restrict__ nlist, const unsigned *__restrict__ ncount)
{
    constexpr int TS = 16 * TI;                 // samples per tile
    // nlist / ncount: search the listed nodes only -- the lowest-index representative of every class of bit-identical
    // model rows (bmu_dedupe_*: equal rows give equal distances and the strict `<` keeps the lowest index, Som.cpp:299).
    // Positions past the count are not evaluated; a tile wholly past it only reports "nothing here".
    if (ncount && *ncount == 0xFFFFFFFFu) {     // the passes left without a list (short redo list): every node, no indirection
        ncount = nullptr;
        nlist = nullptr;
    }
    if (ncount) {
        const int cnt = (int)*ncount;
        N = cnt < N ? cnt : N;
        if (bx * TILE >= N) {                   // workgroup-uniform
            if ((int)threadIdx.x < TS) {
                const int s = s0 + by * TS + (int)threadIdx.x;
                if (s < s1)
                    partial[(size_t)bx * pstride + (slist ? slist[s - s0] : s)] = ~0ull;
            }
            return;
        }
    }
    constexpr int NX = TS * 8 / 256;            // float4 of a sample operand per thread and K-chunk (1 or 2)
    __shared__ __attribute__((aligned(16))) float sx[TILE * LDT];   // TS rows used (also the key scratch: 64 x 16 u64 max)
    __shared__ __attribute__((aligned(16))) float sm[TILE * LDT];
    __shared__ __attribute__((aligned(16))) float sy[CLR ? TS * LDT : 4];
    __shared__ __attribute__((aligned(16))) float sb[CLR ? TILE * LDT : 4];

    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int nbase = bx * TILE;
    const int sbase = s0 + by * TS;
    const int L = a.L, L8 = L & ~7;
    const int nchunks = (L + VSOM_TK - 1) / VSOM_TK;

    float acc[TI][4][8];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                acc[i][j][k] = 0.f;

    // global -> register staging of one K-chunk; the NEXT chunk's loads stay in flight while the current one is
    // consumed (register prefetch: the loads used to be issued and waited for between the two barriers, with
    // only two wavefronts per SIMD to cover them)
    float4 gx[NX], gm[2], gy[NX], gb[2];
    size_t mrow[2];                             // the model rows this thread stages (through the node list, if any)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = nbase + ((tid + 256 * i) >> 3);
        mrow[i] = n < N ? (size_t)(nlist ? nlist[n] : n) : 0;
    }
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            int f = tid + 256 * i;
            int row = f >> 3, c4 = (f & 7) * 4;
            int s = sbase + row;
            gx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            gy[i] = gx[i];
            if (s < s1) {
                const size_t srow = slist ? (size_t)slist[s - s0] : (size_t)s;
                gx[i] = *reinterpret_cast<const float4 *>(a.xa + srow * a.ldx + k0 + c4);
                if (CLR)
                    gy[i] = *reinterpret_cast<const float4 *>(a.xb + srow * a.ldx + k0 + c4);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int f = tid + 256 * i;
            int row = f >> 3, c4 = (f & 7) * 4;
            int n = nbase + row;
            gm[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            gb[i] = gm[i];
            if (n < N) {
                gm[i] = *reinterpret_cast<const float4 *>(a.ma + mrow[i] * a.ldm + k0 + c4);
                if (CLR)
                    gb[i] = *reinterpret_cast<const float4 *>(a.mb + mrow[i] * a.ldm + k0 + c4);
            }
        }
    };
    gload(0);
    int dk = 0;
    for (int ch = 0; ch < nchunks; ++ch, dk += VSOM_TK) {
        if (ch > 0)
            __syncthreads();
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            int f = tid + 256 * i;
            int row = f >> 3, c4 = (f & 7) * 4;
            *reinterpret_cast<float4 *>(&sx[row * LDT + c4]) = gx[i];
            if (CLR)
                *reinterpret_cast<float4 *>(&sy[row * LDT + c4]) = gy[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int f = tid + 256 * i;
            int row = f >> 3, c4 = (f & 7) * 4;
            *reinterpret_cast<float4 *>(&sm[row * LDT + c4]) = gm[i];
            if (CLR)
                *reinterpret_cast<float4 *>(&sb[row * LDT + c4]) = gb[i];
        }
        __syncthreads();
        if (ch + 1 < nchunks)
            gload(dk + VSOM_TK);
#pragma unroll
        for (int kk = 0; kk < VSOM_TK; kk += 8) {
            if (dk + kk < L8) {   // whole 8-blocks only; the remainder is handled in Eigen's order below
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float4 xv[TI], mv[4], yv[TI], bv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        mv[j] = *reinterpret_cast<const float4 *>(&sm[(tx + 16 * j) * LDT + kk + 4 * h]);
                        if (CLR)
                            bv[j] = *reinterpret_cast<const float4 *>(&sb[(tx + 16 * j) * LDT + kk + 4 * h]);
                    }
#pragma unroll
                    for (int i = 0; i < TI; ++i) {
                        xv[i] = *reinterpret_cast<const float4 *>(&sx[(ty + 16 * i) * LDT + kk + 4 * h]);
                        if (CLR)
                            yv[i] = *reinterpret_cast<const float4 *>(&sy[(ty + 16 * i) * LDT + kk + 4 * h]);
                    }
#pragma unroll
                    for (int i = 0; i < TI; ++i) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float r0 = vsom_resid<CLR>(xv[i].x, CLR ? yv[i].x : 0.f, mv[j].x, CLR ? bv[j].x : 0.f);
                            float r1 = vsom_resid<CLR>(xv[i].y, CLR ? yv[i].y : 0.f, mv[j].y, CLR ? bv[j].y : 0.f);
                            float r2 = vsom_resid<CLR>(xv[i].z, CLR ? yv[i].z : 0.f, mv[j].z, CLR ? bv[j].z : 0.f);
                            float r3 = vsom_resid<CLR>(xv[i].w, CLR ? yv[i].w : 0.f, mv[j].w, CLR ? bv[j].w : 0.f);
                            float p0 = r0 * r0, p1 = r1 * r1, p2 = r2 * r2, p3 = r3 * r3;
                            acc[i][j][4 * h + 0] = acc[i][j][4 * h + 0] + p0;
                            acc[i][j][4 * h + 1] = acc[i][j][4 * h + 1] + p1;
                            acc[i][j][4 * h + 2] = acc[i][j][4 * h + 2] + p2;
                            acc[i][j][4 * h + 3] = acc[i][j][4 * h + 3] + p3;
                        }
                    }
                }
            }
        }
    }

    // reduction tree + remainder (the last chunk is still in LDS)
    const int rem = L - L8;
    const int roff = L8 - (nchunks - 1) * VSOM_TK;   // column of element L8 inside the last chunk
    float dist[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float q0 = acc[i][j][0] + acc[i][j][4];
            float q1 = acc[i][j][1] + acc[i][j][5];
            float q2 = acc[i][j][2] + acc[i][j][6];
            float q3 = acc[i][j][3] + acc[i][j][7];
            const int xr = (ty + 16 * i) * LDT + roff, mr = (tx + 16 * j) * LDT + roff;
            int t = 0;
            if (rem >= 4) {
                float r0 = vsom_resid<CLR>(sx[xr + 0], CLR ? sy[xr + 0] : 0.f, sm[mr + 0], CLR ? sb[mr + 0] : 0.f);
                float r1 = vsom_resid<CLR>(sx[xr + 1], CLR ? sy[xr + 1] : 0.f, sm[mr + 1], CLR ? sb[mr + 1] : 0.f);
                float r2 = vsom_resid<CLR>(sx[xr + 2], CLR ? sy[xr + 2] : 0.f, sm[mr + 2], CLR ? sb[mr + 2] : 0.f);
                float r3 = vsom_resid<CLR>(sx[xr + 3], CLR ? sy[xr + 3] : 0.f, sm[mr + 3], CLR ? sb[mr + 3] : 0.f);
                float p0 = r0 * r0, p1 = r1 * r1, p2 = r2 * r2, p3 = r3 * r3;
                q0 = q0 + p0;
                q1 = q1 + p1;
                q2 = q2 + p2;
                q3 = q3 + p3;
                t = 4;
            }
            float t02 = q0 + q2, t13 = q1 + q3;
            float res = t02 + t13;
            for (; t < rem; ++t) {
                float r = vsom_resid<CLR>(sx[xr + t], CLR ? sy[xr + t] : 0.f, sm[mr + t], CLR ? sb[mr + t] : 0.f);
                float p = r * r;
                res = res + p;
            }
            dist[i][j] = res;
        }

    // node 0's NaN flag: `cur < NaN` is never true, so a NaN at node 0 pins the BMU to 0 (Som.cpp:293-299)
    if (bx == 0 && tx == 0) {
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            int s = sbase + ty + 16 * i;
            if (s < s1)
                nan0[slist ? slist[s - s0] : s] = (dist[i][0] != dist[i][0]) ? 1 : 0;
        }
    }

    __syncthreads();   // everyone is done with sx before it is reused for the keys
    u64 *keys = reinterpret_cast<u64 *>(sx);   // TS samples x 16 tx
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        u64 kmin = ~0ull;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = nbase + tx + 16 * j;
            const int node = (nlist && n < N) ? nlist[n] : n;
            // findRestrictedBmu (Som.cpp:316-322): node 0 seeds unconditionally, others need the hits
            const bool allowed = n < N && (hits == nullptr || node == 0 || hits[node] >= min_hits);
            u64 k = allowed ? vsom_key(dist[i][j], (uint32_t)node) : ~0ull;
            kmin = k < kmin ? k : kmin;
        }
        keys[(ty + 16 * i) * 16 + tx] = kmin;
    }
    __syncthreads();
    if (tid < TS) {
        int s = sbase + tid;
        if (s < s1) {
            u64 kmin = ~0ull;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                u64 k = keys[tid * 16 + t];
                kmin = k < kmin ? k : kmin;
            }
            partial[(size_t)bx * pstride + (slist ? slist[s - s0] : s)] = kmin;
        }
    }
}

// LIST = false: grid.y = sample tiles.  LIST = true: only the samples the shortlist path lists (slist / scount,
// vsom_shortlist.hip; s0 / s1 then index the list) with a SMALL grid.y, every workgroup walking on through the list's
// tiles in steps of grid.y -- the list is usually empty, and a grid sized for the whole chunk costs 7 us of workgroups
// that start only to leave (16384 of them at C3 / C4).  (Two instantiations: the loop keeps the per-thread offsets
// alive across the body, 19 spilled registers that the plain search should not pay.)
template <bool CLR, int TI, bool LIST>
__global__ __launch_bounds__(256, 2) void bmu_tile_kernel(DistArgs a, int s0, int s1, int N,
                                                          u64 *__restrict__ partial, int pstride,
                                                          unsigned char *__restrict__ nan0,
                                                          const int *__restrict__ slist,
                                                          const unsigned *__restrict__ scount,
                                                          const u64 *__restrict__ hits, u64 min_hits, SlFeedback fb,
                                                          const int *__restrict__ nlist, const unsigned *__restrict__ ncount)
{
    constexpr int TS = 16 * TI;
    if (LIST && fb.scal && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64)
        sl_feedback_write(fb);               // (vsom_digits.hpp: the search's statistics for the host)
    if (!LIST) {
        bmu_tile_body<CLR, TI>(a, s0, s1, N, partial, pstride, nan0, nullptr, hits, min_hits, (int)blockIdx.y, (int)blockIdx.x, nlist,
                               ncount);
        return;
    }
    const int cnt = (int)*scount;
    s1 = s0 + cnt < s1 ? s0 + cnt : s1;
    // The workgroups of the (node tile, walker) grid share out the (node tile, sample tile) pairs that EXIST: with a node
    // list the live node tiles are a device-side count -- were the walk tied to blockIdx.x, the workgroups of dead node tiles
    // would leave and the others keep their full share (C5's collapsed maps: 7 of 16 node tiles live, the search 9 % shorter
    // instead of 55 %).
    const int nts = (s1 - s0 + TS - 1) / TS;
    int neff = N;
    if (ncount && *ncount != 0xFFFFFFFFu) {
        const int nc = (int)*ncount;
        neff = nc < N ? nc : N;
    }
    const int ntn = (neff + TILE - 1) / TILE;
    const int G = (int)(gridDim.x * gridDim.y), f0 = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    for (int w = f0; w < ntn * nts; w += G) {
        if (w != f0)
            __syncthreads();                    // the key scratch of the previous tile has been read
        bmu_tile_body<CLR, TI>(a, s0, s1, N, partial, pstride, nan0, slist, hits, min_hits, w / ntn, w % ntn, nlist, ncount);
    }
}

__global__ void bmu_reduce_kernel(const u64 *__restrict__ partial, int pstride, int ntiles,
                                  const unsigned char *__restrict__ nan0, int s0, int s1,
                                  u64 *__restrict__ lastbmu, float *__restrict__ sqres,
                                  const int *__restrict__ slist, const unsigned *__restrict__ scount,
                                  const unsigned *__restrict__ ncount)
{
    if (ncount && *ncount != 0xFFFFFFFFu) {     // node list: only its tiles were searched (and written)
        const int live = ((int)*ncount + TILE - 1) / TILE;
        ntiles = live < ntiles ? live : ntiles;
    }
    int s = s0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (scount) {
        const int cnt = (int)*scount;
        s1 = s0 + cnt < s1 ? s0 + cnt : s1;
    }
    if (s >= s1)
        return;
    if (slist)
        s = slist[s - s0];
    u64 kmin = ~0ull;
    for (int t = 0; t < ntiles; ++t) {
        u64 k = partial[(size_t)t * pstride + s];
        kmin = k < kmin ? k : kmin;
    }
    if (nan0[s]) {
        lastbmu[s] = 0;
        sqres[s] = __uint_as_float(0x7FC00000u);
    } else {
        lastbmu[s] = kmin & 0xFFFFFFFFull;
        sqres[s] = __uint_as_float((uint32_t)(kmin >> 32));
    }
}

// ---- duplicate model rows: the exact search evaluates one representative per class -------------------------------------
// Batch training on chunks whose samples lie on (nearly) one line leaves maps with few DISTINCT rows (C5's CLR maps after
// every other chunk: 426 of 1024; an empty chunk leaves one: Som.cpp:840-875) and the exact kernel paid for every copy.
// Equal rows give equal distances and the reference's strict `<` from node 0 keeps the lowest index (Som.cpp:293-304), so
// searching the lowest-index member of every class of BIT-IDENTICAL rows returns the same index and distance.
//   bmu_row_hash_kernel   64-bit hash of every row's bits (order-independent sum of mixed (position, bits) words)
//   bmu_row_twin_kernel   rep[n] = the lowest m < n with the same hash AND the same bits, else n
//   bmu_unique_kernel     the representatives in index order + their count
// `scount` (the redo list's device-side length) below `min_list`: the passes leave at once and the list is the identity.
__device__ __forceinline__ u64 bmu_mix64(u64 z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void bmu_row_hash_kernel(const float *__restrict__ map, int ldm, int N, u64 *__restrict__ hash,
                                                           const unsigned *__restrict__ scount, unsigned min_list)
{
    __shared__ u64 sh[4];
    if (scount && *scount < min_list)
        return;
    for (int n = blockIdx.x; n < N; n += gridDim.x) {        // (a bounded grid: leaving early is then a 2 us launch)
        const unsigned *row = reinterpret_cast<const unsigned *>(map + (size_t)n * ldm);
        u64 h = 0;
        for (int d = threadIdx.x * 4; d < ldm; d += 1024) {  // pitches are multiples of 32 floats
            const uint4 v = *reinterpret_cast<const uint4 *>(row + d);
            h += bmu_mix64(((u64)(d + 0) << 32) | v.x) + bmu_mix64(((u64)(d + 1) << 32) | v.y) +
                 bmu_mix64(((u64)(d + 2) << 32) | v.z) + bmu_mix64(((u64)(d + 3) << 32) | v.w);
        }
        for (int off = 32; off > 0; off >>= 1)
            h += __shfl_xor(h, off);
        if ((threadIdx.x & 63) == 0)
            sh[threadIdx.x >> 6] = h;
        __syncthreads();
        if (threadIdx.x == 0)
            hash[n] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void bmu_row_twin_kernel(const float *__restrict__ map, int ldm, int N, const u64 *__restrict__ hash,
                                                           int *__restrict__ rep, const unsigned *__restrict__ scount,
                                                           unsigned min_list)
{
    const int lane = threadIdx.x & 63;
    if (scount && *scount < min_list)
        return;                                              // (bmu_unique_kernel then marks "no list": rep is not read)
    for (int n = blockIdx.x * 4 + ((int)threadIdx.x >> 6); n < N; n += 4 * (int)gridDim.x) {
    const u64 hn = hash[n];
    const unsigned *rn = reinterpret_cast<const unsigned *>(map + (size_t)n * ldm);
    int found = n;
    for (int m0 = 0; m0 < n && found == n; m0 += 64) {
        const int m = m0 + lane;
        u64 mask = __ballot(m < n && hash[m] == hn);
        while (mask) {                                       // (wavefront-uniform) candidates in ascending order
            const int mc = m0 + __ffsll((long long)mask) - 1;
            mask &= mask - 1ull;
            const unsigned *rm = reinterpret_cast<const unsigned *>(map + (size_t)mc * ldm);
            bool differ = false;
            for (int d = lane * 4; d < ldm && !__ballot(differ); d += 256) {
                const uint4 a = *reinterpret_cast<const uint4 *>(rn + d), b = *reinterpret_cast<const uint4 *>(rm + d);
                differ = a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w;
            }
            if (!__ballot(differ)) {
                found = mc;                                  // the first match is the class's lowest index (equality is transitive)
                break;
            }
        }
    }
    if (lane == 0)
        rep[n] = found;
    }
}

__global__ __launch_bounds__(1024) void bmu_unique_kernel(const int *__restrict__ rep, int N, int *__restrict__ ulist,
                                                          unsigned *__restrict__ ucount, const unsigned *__restrict__ scount,
                                                          unsigned min_list)
{
    __shared__ int swave[16];
    __shared__ int sbase;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (scount && *scount < min_list) {                      // short redo list: no representatives pass, no list
        if (tid == 0)
            *ucount = 0xFFFFFFFFu;
        return;
    }
    if (tid == 0)
        sbase = 0;
    __syncthreads();
    for (int n0 = 0; n0 < N; n0 += 1024) {
        const int n = n0 + tid;
        const bool uniq = n < N && rep[n] == n;
        const u64 bm = __ballot(uniq);
        if (lane == 0)
            swave[wave] = __popcll(bm);
        __syncthreads();
        int off = sbase;
        for (int w = 0; w < wave; ++w)
            off += swave[w];
        if (uniq)
            ulist[off + __popcll(bm & ((1ull << lane) - 1ull))] = n;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w)
                tot += swave[w];
            sbase += tot;
        }
        __syncthreads();
    }
    if (tid == 0)
        *ucount = (unsigned)sbase;
}

// enqueue the three passes; *nlist / *ncount = what bmu_tile_kernel takes (buffers kept with the context)
static int launch_bmu_dedupe(vsom_ctx *c, const unsigned *scount, unsigned min_list, const int **nlist, const unsigned **ncount)
{
    if (!c->dd_hash) {
        VSOM_HIP_CHECK(hipMalloc(&c->dd_hash, (size_t)c->N * sizeof(u64)));
        VSOM_HIP_CHECK(hipMalloc(&c->dd_rep, (size_t)c->N * sizeof(int)));
        VSOM_HIP_CHECK(hipMalloc(&c->dd_list, (size_t)c->N * sizeof(int) + 64));
    }
    u64 *hash = reinterpret_cast<u64 *>(c->dd_hash);
    unsigned *cnt = reinterpret_cast<unsigned *>(c->dd_list + c->N);
    hipLaunchKernelGGL(bmu_row_hash_kernel, dim3(std::min<unsigned>((unsigned)c->N, 2048u)), dim3(256), 0, c->stream, c->map,
                       (int)c->pitch, (int)c->N, hash, scount, min_list);
    hipLaunchKernelGGL(bmu_row_twin_kernel, dim3(std::min<unsigned>((unsigned)((c->N + 3) / 4), 1024u)), dim3(256), 0, c->stream,
                       c->map, (int)c->pitch, (int)c->N, hash, c->dd_rep, scount, min_list);
    hipLaunchKernelGGL(bmu_unique_kernel, dim3(1), dim3(1024), 0, c->stream, c->dd_rep, (int)c->N, c->dd_list, cnt, scount, min_list);
    *nlist = c->dd_list;
    *ncount = cnt;
    return VSOM_OK;
}

int launch_bmu_full_exact_masked(vsom_ctx *c, size_t s0, size_t s1, const int *slist, const unsigned *scount,
                                 const u64 *hits, u64 min_hits, const SlFeedback *fbp = nullptr)
{
    SlFeedback fb = {nullptr, nullptr, 0u, nullptr, nullptr};
    if (fbp)
        fb = *fbp;
    if (s1 <= s0)
        return VSOM_OK;
    const int ntn = (int)((c->N + TILE - 1) / TILE);
    const int TS = c->transform == VSOM_CLR ? 32 : TILE;      // samples per tile (bmu_tile_kernel<CLR, TI>)
    const int nts = (int)((s1 - s0 + TS - 1) / TS);
    size_t need = (size_t)ntn * c->Bcap;
    if (need > c->partial_cap) {
        if (c->partial)
            VSOM_HIP_CHECK(hipFree(c->partial));
        c->partial = nullptr;
        VSOM_HIP_CHECK(hipMalloc(&c->partial, need * sizeof(u64)));
        c->partial_cap = need;
    }
    DistArgs a = make_dist_args(c);
    // (a redo list is usually empty or short: ~512 workgroups -- the two per CU the kernel's registers allow -- each
    // walking on through the list's tiles; 768 for CLR measured slower: a second, half-empty round)
    const bool list = slist != nullptr && scount != nullptr;
    const int gy = list ? std::min(nts, std::max(1, 512 / ntn)) : nts;
    dim3 grid((unsigned)ntn, (unsigned)gy);
    // Representatives of the duplicate rows only -- where the search is large enough to repay three passes over the map
    // (C3's size: hash 15 + twins 5 + list 17 us): 2e10 (sample, node, value) triples = ~0.5 ms of this kernel
    // (vsom_set_row_dedupe moves the threshold; 0 = always).  With a redo list the passes look at its device-side length
    // first and leave when it is short.  Not for restricted searches: bit-identical rows may differ in their hit counts.
    const int *nlist = nullptr;
    const unsigned *ncount = nullptr;
    const double work = (double)(s1 - s0) * (double)c->N * (double)c->D;
    bool dd = !hits && c->N >= 256 && c->dedupe && c->dd_min_work >= 0 && work >= c->dd_min_work;
    if (dd && list && c->dd_min_work > 0 && c->transform != VSOM_CLR) {
        // A redo list is usually empty or a handful of non-finite samples (C3: always empty), and then even three launches
        // that look at its length and leave are ~8 us of a step for nothing.  Whole-chunk redo lists are what the CLR
        // shortlist produces when it recognises a collapsed map on the device (vsom_shortlist.hip, sl_clr_degenerate): the
        // passes are enqueued behind CLR searches (8 us of a 5.4 ms step when the list is short) and, for the other
        // transformations, only when vsom_set_row_dedupe(ctx, 0) asks for them always.  (A hint from the previous search's
        // feedback words was tried: the host enqueues whole steps ahead of the device, the words are stale when it matters.)
        dd = false;
    }
    if (dd) {
        if (int rc = launch_bmu_dedupe(c, list ? scount : nullptr, 256u, &nlist, &ncount))
            return rc;
    }
#define VSOM_TILE_LAUNCH(K)                                                                                             \
    hipLaunchKernelGGL(K, grid, dim3(256), 0, c->stream, a, (int)s0, (int)s1, (int)c->N, c->partial, (int)c->Bcap, c->nan0, \
                       slist, scount, hits, min_hits, fb, nlist, ncount)
    if (c->transform == VSOM_CLR) {
        if (list)
            VSOM_TILE_LAUNCH((bmu_tile_kernel<true, 2, true>));
        else
            VSOM_TILE_LAUNCH((bmu_tile_kernel<true, 2, false>));
    } else {
        if (list)
            VSOM_TILE_LAUNCH((bmu_tile_kernel<false, 4, true>));
        else
            VSOM_TILE_LAUNCH((bmu_tile_kernel<false, 4, false>));
    }
#undef VSOM_TILE_LAUNCH
    hipLaunchKernelGGL(bmu_reduce_kernel, dim3((unsigned)((s1 - s0 + 255) / 256)), dim3(256), 0,
                       c->stream, c->partial, (int)c->Bcap, ntn, c->nan0, (int)s0, (int)s1, c->lastbmu,
                       c->sqres, slist, scount, ncount);
    VSOM_HIP_CHECK(hipGetLastError());
    return VSOM_OK;
}

int launch_bmu_full_exact_list(vsom_ctx *c, size_t s0, size_t s1, const int *slist, const unsigned *scount, const SlFeedback *fb)
{
    return launch_bmu_full_exact_masked(c, s0, s1, slist, scount, nullptr, 0, fb);
}

int launch_bmu_restricted(vsom_ctx *c, u64 min_hits)
{
    TimerScope ts(c, VSOM_T_BMU);
    return launch_bmu_full_exact_masked(c, 0, c->B, nullptr, nullptr, c->hits, min_hits);
}

int launch_bmu_full_shortlist(vsom_ctx *c, size_t s0, size_t s1);

int launch_bmu_full(vsom_ctx *c, size_t s0, size_t s1)
{
    TimerScope ts(c, VSOM_T_BMU);
    // the MFMA shortlist covers the plain squared-distance comparer (Standard / Median);
    // it pays once the map is large enough to fill the matrix pipes and the vectors are long enough
    // for the contraction to outweigh writing and re-reading the B x N approximation matrix
    // (measured on 64x64 maps, B = 16384: D = 32 exact 0.22 vs 0.36 ms, D = 64 0.34 vs 0.39, D = 128
    // 0.61 vs 0.47)
    // (CLR: one contraction of length P + 3J over derived features, vsom_shortlist.hip "CLR shortlist")
    const bool can = true;
    bool want = c->bmu_mode == VSOM_BMU_SHORTLIST;
    // Rows of at most 64 values (Standard / Median): the contraction keeps tile minima only, nothing of size B x N is
    // written (sl_k64_kernel, vsom_sl_i8.hip) -- it beats the exact kernel once that has ~0.1 ms of work
    // (C4, 16384 x 4096 x 32: exact 0.21 ms)
    const bool short_rows = c->transform != VSOM_CLR && c->D <= 64 && c->D >= 16 &&
                            (double)(s1 - s0) * (double)c->N * (double)c->D >= 1.0e9;
    if (c->bmu_mode == VSOM_BMU_AUTO && c->N >= 1024 && (s1 - s0) >= 64 && (c->D > 64 || short_rows)) {
        want = true;
        // feedback of the previous shortlist call (pinned host words written by the device, read
        // without synchronising: possibly one call stale): when more than a quarter of the samples
        // had to be redone exactly the shortlist does not pay on this map -- skip it for a while
        if (c->sl_fb) {
            volatile unsigned *fb = c->sl_fb;
            unsigned redo = fb[0], rows = fb[2], seq = fb[3];
            if (seq != c->sl_seq_seen) {             // the verdict of a shortlist search not looked at yet
                c->sl_seq_seen = seq;
                if (rows > 0 && redo * 4u > rows) {
                    // A failed probe costs a fraction of the exact search it falls back to (Standard C3: 0.35 of
                    // 2.9 ms; CLR: a collapsed map is recognised on the device before the contraction runs, vsom_shortlist.hip)
                    // and a wrongly skipped one several times its own cost, and every chunk rebuilds the map from zero
                    // (Som.cpp:843,870) -- one bad map says little about the next: no pause after a first failure,
                    // 4 searches after a second in a row, doubling up to 32
                    c->sl_skip = c->sl_fail_streak == 0 ? 0 : (2 << (c->sl_fail_streak < 4 ? c->sl_fail_streak : 4));
                    ++c->sl_fail_streak;
                } else {
                    c->sl_fail_streak = 0;
                }
            }
            if (c->sl_skip > 0) {
                --c->sl_skip;
                want = false;
            }
        }
    }
    if (can && want)
        return launch_bmu_full_shortlist(c, s0, s1);
    return launch_bmu_full_exact_list(c, s0, s1, nullptr, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------
// local search: Som::findLocalBmu, unsigned (size_t) arithmetic kept literal
// ------------------------------------------------------------------------------------------
template <bool CLR>
__global__ __launch_bounds__(256) void bmu_local_kernel(DistArgs a, int s0, int s1, u64 width,
                                                        u64 height, u64 *__restrict__ lastbmu,
                                                        float *__restrict__ sqres)
{
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    const int s = s0 + wave;
    if (s >= s1)
        return;   // wave-uniform
    const float *xa = a.xa + (size_t)s * a.ldx;
    const float *xb = a.xb + (size_t)s * a.ldx;
    u64 minIndex;
    float minDist;
    vsom_local_walk<CLR>(a, xa, xb, width, height, lastbmu[s], lane, minIndex, minDist);
    if (lane == 0) {
        lastbmu[s] = minIndex;
        sqres[s] = minDist;   // == ||Comparer(x, M[minIndex])||^2 (same reduction order)
    }
}

int launch_bmu_local(vsom_ctx *c, size_t s0, size_t s1)
{
    TimerScope ts(c, VSOM_T_BMU);
    if (s1 <= s0)
        return VSOM_OK;
    DistArgs a = make_dist_args(c);
    size_t waves = s1 - s0;
    dim3 grid((unsigned)((waves + 3) / 4));
    if (c->transform == VSOM_CLR)
        hipLaunchKernelGGL(bmu_local_kernel<true>, grid, dim3(256), 0, c->stream, a, (int)s0, (int)s1,
                           (u64)c->W, (u64)c->H, c->lastbmu, c->sqres);
    else
        hipLaunchKernelGGL(bmu_local_kernel<false>, grid, dim3(256), 0, c->stream, a, (int)s0, (int)s1,
                           (u64)c->W, (u64)c->H, c->lastbmu, c->sqres);
    VSOM_HIP_CHECK(hipGetLastError());
    return VSOM_OK;
}

// ------------------------------------------------------------------------------------------
// arbitrary (node,row) pairs: Som::euclidianWeightedDist
// ------------------------------------------------------------------------------------------
template <bool CLR>
__global__ __launch_bounds__(256) void pair_dist_kernel(DistArgs a, const u64 *__restrict__ nodes,
                                                        const u64 *__restrict__ rows, int count,
                                                        float *__restrict__ out)
{
    const int gid = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 3);
    const int k = threadIdx.x & 7;
    const int p = gid < count ? gid : count - 1;
    u64 node = nodes[p], row = rows[p];
    float d = vsom_group_dist<CLR>(a.xa + (size_t)row * a.ldx, a.xb + (size_t)row * a.ldx,
                                   a.ma + (size_t)node * a.ldm, a.mb + (size_t)node * a.ldm, a.L, k);
    if (gid < count && k == 0)
        out[gid] = d;
}

int launch_pair_dist(vsom_ctx *c, const u64 *nodes_dev, const u64 *rows_dev, size_t count,
                     float *out_dev)
{
    if (count == 0)
        return VSOM_OK;
    DistArgs a = make_dist_args(c);
    dim3 grid((unsigned)((count * 8 + 255) / 256));
    if (c->transform == VSOM_CLR)
        hipLaunchKernelGGL(pair_dist_kernel<true>, grid, dim3(256), 0, c->stream, a, nodes_dev,
                           rows_dev, (int)count, out_dev);
    else
        hipLaunchKernelGGL(pair_dist_kernel<false>, grid, dim3(256), 0, c->stream, a, nodes_dev,
                           rows_dev, (int)count, out_dev);
    VSOM_HIP_CHECK(hipGetLastError());
    return VSOM_OK;
}

// every node against one chunk row (input of Som::findRestrictedBmd)
template <bool CLR>
__global__ __launch_bounds__(256) void row_dist_kernel(DistArgs a, int row, int N, float *__restrict__ out)
{
    const int gid = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 3);
    const int k = threadIdx.x & 7;
    const int n = gid < N ? gid : N - 1;
    float d = vsom_group_dist<CLR>(a.xa + (size_t)row * a.ldx, a.xb + (size_t)row * a.ldx,
                                   a.ma + (size_t)n * a.ldm, a.mb + (size_t)n * a.ldm, a.L, k);
    if (gid < N && k == 0)
        out[gid] = d;
}

int launch_row_dist(vsom_ctx *c, size_t row, float *out_dev)
{
    DistArgs a = make_dist_args(c);
    dim3 grid((unsigned)(((size_t)c->N * 8 + 255) / 256));
    if (c->transform == VSOM_CLR)
        hipLaunchKernelGGL(row_dist_kernel<true>, grid, dim3(256), 0, c->stream, a, (int)row, (int)c->N, out_dev);
    else
        hipLaunchKernelGGL(row_dist_kernel<false>, grid, dim3(256), 0, c->stream, a, (int)row, (int)c->N, out_dev);
    VSOM_HIP_CHECK(hipGetLastError());
    return VSOM_OK;
}

// Som::euclidianWeightedDistRaw (Som.cpp:143-157), valid = weights = 1:
//   a = (M - v)/sM, b = ((M - v)*1)/sM = a, sum of a*b in Eigen's reduction order;
//   sM = sigma < 1e-5 ? 1e-5 : sigma.  8 lanes per pair (one per accumulator class).
__global__ __launch_bounds__(256) void raw_dist_kernel(const float *__restrict__ map, const float *__restrict__ sigma,
                                                       int ldm, const float *__restrict__ vbase, int ldv, int D,
                                                       int P, int ppitch, int v_is_model,
                                                       const u64 *__restrict__ nodes, const u64 *__restrict__ vrows,
                                                       int count, float *__restrict__ out)
{
    const int gid = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 3);
    const int k = threadIdx.x & 7;
    const int p = gid < count ? gid : count - 1;
    const float *M = map + (size_t)nodes[p] * ldm, *sg = sigma + (size_t)nodes[p] * ldm;
    const float *v = vbase + (size_t)vrows[p] * ldv;
    auto prod = [&](int d) {
        // logical element d of a model vector: CLR rows are stored [A(P) | pad | B(P) | pad]
        const int pm = d < P ? d : ppitch + (d - P);
        const int pv = v_is_model ? pm : d;
        float s = sg[pm] < 0.00001f ? 0.00001f : sg[pm];
        float r = M[pm] - v[pv];
        float a = r / s;
        return a * a;
    };
    const int D8 = D & ~7;
    float acc = 0.f;
    for (int d = k; d < D8; d += 8)
        acc = acc + prod(d);
    float q = acc + __shfl_xor(acc, 4);
    const int rem = D - D8;
    if (rem >= 4)
        q = q + prod(D8 + (k & 3));
    float t = q + __shfl_xor(q, 2);
    float res = t + __shfl_xor(t, 1);
    for (int tt = (rem >= 4 ? 4 : 0); tt < rem; ++tt)
        res = res + prod(D8 + tt);
    if (gid < count && k == 0)
        out[gid] = res;
}

int launch_raw_dist(vsom_ctx *c, const u64 *nodes_dev, const u64 *vrows_dev, size_t count, int from_map,
                    float *out_dev)
{
    if (count == 0)
        return VSOM_OK;
    if (c->transform == VSOM_CLR && !from_map)
        return vsom_fail(VSOM_ERR_UNSUPPORTED, "euclidianWeightedDistRaw against a sample needs depth == sample length");
    dim3 grid((unsigned)((count * 8 + 255) / 256));
    hipLaunchKernelGGL(raw_dist_kernel, grid, dim3(256), 0, c->stream, c->map, c->sigma, (int)c->pitch,
                       from_map ? c->map : c->Xs, from_map ? (int)c->pitch : (int)c->xpitch, (int)c->D,
                       (int)c->part_len, (int)c->part_pitch, from_map ? 1 : 0, nodes_dev, vrows_dev, (int)count,
                       out_dev);
    VSOM_HIP_CHECK(hipGetLastError());
    return VSOM_OK;
}

// ------------------------------------------------------------------------------------------
// finish: bmuHits and MSE (Som.cpp:777-781 / 800-804), sample order fixed (Q13)
// ------------------------------------------------------------------------------------------
// bmuHits[idx]++ (:778) for 1024 samples.  BMUs cluster on few nodes, and same-address atomics serialise in L2:
// each wavefront first merges its equal indices (one atomic per distinct value).
__device__ __forceinline__ void finish_hits(const u64 *__restrict__ lastbmu, int B, u64 *__restrict__ hits, int base)
{
    const int i = base + (int)threadIdx.x;
    const bool in = i < B;
    const u64 idx = in ? lastbmu[i] : 0;
    bool todo = in;
    while (__any(todo)) {
        const u64 lead = __shfl(idx, __ffsll((long long)__ballot(todo)) - 1);
        const bool mine = todo && idx == lead;
        const u64 m = __ballot(mine);
        if (mine && (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1)
            atomicAdd(&hits[lead], (u64)__popcll(m));
        todo = todo && !mine;
    }
}

#define FIN_TILE 8192
// one launch: workgroup 0 = the MSE running sum, workgroups 1.. = bmuHits of 1024 samples each
__global__ __launch_bounds__(1024) void finish_kernel(const float *__restrict__ sqres, int B, float *__restrict__ mse,
                                                      const u64 *__restrict__ lastbmu, u64 *__restrict__ hits)
{
    if (blockIdx.x > 0) {
        finish_hits(lastbmu, B, hits, ((int)blockIdx.x - 1) * 1024);
        return;
    }
    // the running sum is serial by definition (fp32, sample order); everything around it is not:
    // the other threads fill the next tile (divisions) while thread 0 adds the current one out of
    // LDS, 16 values (4 x ds_read_b128) per dependent burst.
    __shared__ __attribute__((aligned(16))) float q[2][FIN_TILE];
    const float fB = (float)B;
    float run = 0.f;
    const int ntiles = (B + FIN_TILE - 1) / FIN_TILE;
    auto fill = [&](int t) {
        const int base = t * FIN_TILE;
        const int n = B - base < FIN_TILE ? B - base : FIN_TILE;
        float *dst = q[t & 1];
        for (int i = (int)threadIdx.x - 64; i < n; i += (int)blockDim.x - 64)   // wavefronts 1.. only
            dst[i] = sqres[base + i] / fB;         // squaredNorm()/(float)epochSize :781
    };
    if (threadIdx.x >= 64)
        fill(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        if (threadIdx.x == 0) {
            const int base = t * FIN_TILE;
            const int n = B - base < FIN_TILE ? B - base : FIN_TILE;
            const float *src = q[t & 1];
            int i = 0;
            for (; i + 16 <= n; i += 16) {
                const float4 a = *(const float4 *)(src + i), b = *(const float4 *)(src + i + 4),
                             cc = *(const float4 *)(src + i + 8), d = *(const float4 *)(src + i + 12);
                run = run + a.x; run = run + a.y; run = run + a.z; run = run + a.w;
                run = run + b.x; run = run + b.y; run = run + b.z; run = run + b.w;
                run = run + cc.x; run = run + cc.y; run = run + cc.z; run = run + cc.w;
                run = run + d.x; run = run + d.y; run = run + d.z; run = run + d.w;
            }
            for (; i < n; ++i)
                run = run + src[i];
        } else if (threadIdx.x >= 64 && t + 1 < ntiles) {
            fill(t + 1);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        *mse = run;
}

int launch_finish(vsom_ctx *c)
{
    // the serial MSE sum (~5 ns per sample) reads sqres and writes mse only, bmuHits is read by no training step: both
    // run as ONE launch on the side stream beside phase 2, joined at the end of launch_phase2 / by the next entry point
    TimerScope ts(c, VSOM_T_FINISH);
    VSOM_HIP_CHECK(hipEventRecord(c->ev_fork, c->stream));
    VSOM_HIP_CHECK(hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
    hipLaunchKernelGGL(finish_kernel, dim3(1u + (unsigned)((c->B + 1023) / 1024)), dim3(1024), 0, c->aux_stream, c->sqres,
                       (int)c->B, c->mse, c->lastbmu, c->hits);
    VSOM_HIP_CHECK(hipGetLastError());
    VSOM_HIP_CHECK(hipEventRecord(c->ev_join, c->aux_stream));
    c->aux_pending = true;
    return VSOM_OK;
}

int vsom_join_aux(vsom_ctx *c)
{
    if (c->aux_pending) {
        VSOM_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
        c->aux_pending = false;
    }
    return VSOM_OK;
}
