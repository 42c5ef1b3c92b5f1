// vsom_shortlist.hip -- Som::findBmu (Som.cpp:291-309) through an MFMA shortlist (gfx950).
//
// Standard / Median comparer (distance = sum_d (M_d - x_d)^2) and, further down, the
// CombinatorialLinearRegression comparer (Transformation.cpp:82-106).  The result is REQUIRED to
// be identical to the exact-order search of vsom_bmu.hip (and is checked against it and against
// the oracle in tests): the matrix pipe only prunes, every returned index/distance comes from an
// exact-order evaluation.
//
//   norm_kernel     nrm_i = sum_d M_id^2 (fp32), max over nodes, +inf flag
//   gemm_kernel     G[s][i] = nrm_i - 2 * <x_s, M_i> with v_mfma_f32_32x32x2_f32 (an fp32 fmaf
//                   chain, MI355X guide "FP32-input MFMA"); 128x128 output tile per workgroup,
//                   64x64 per wavefront, K streamed through LDS in chunks of 32
//   select_kernel   one wavefront per sample: m = min_i G[s][i]; every node with
//                   G[s][i] <= m + T_s is re-evaluated in the reference's fp32 order
//                   (vsom_group_dist) and the argmin is taken with the reference's rules.
//
// Threshold (u = 2^-24, K = D).  With d = true squared distance, e = exact-order fp32 value,
// G + nx = approximate value (nx = |x|^2 is constant per sample and dropped):
//   |G_i + nx - d_i| <= Ea_i = 2*g1*(nM_i + nx),  g1 = (GK + K/GK + 3)u: the dot product is
//                       accumulated per K-chunk of GK=32 by the MFMA fmaf chain (gamma_GK) and the
//                       chunk sums are added in fp32 (gamma_{K/GK}); norm + final subtraction: 3u
//   |e_i - d_i|      <= g2 * d_i,                  g2 = (K/8+10)u (products + reduction tree depth)
// If i* is the reference argmin (e_i* <= e_j for all j) then for jm = argmin G (m = G_jm):
//   G_i* <= m + Ea_jm + Ea_i* + 2.1*g2*d_jm,   d_jm <= m + nx + Ea_jm
//        <= m + 4*g1*(nMmax + nx) + 2.1*g2*(m + nx + 2*g1*(nMmax + nx)) =: m + T_s
// T_s is inflated by 1.05 to cover the fp32 rounding of nrm, nx, m + nx and of T_s itself.  Nodes whose row contains NaN have G = NaN and an exact distance of NaN: excluded,
// as in the reference (a NaN never wins; node 0 is checked explicitly, Som.cpp:293-299).
// Samples for which the bound is not applicable (non-finite nx or nrm) or whose shortlist exceeds
// SL_CMAX are appended to a device-side redo list; the exact-order tile kernel then handles
// exactly those samples (its workgroups beyond the list length exit at once).  The redo fraction
// is fed back to the host through pinned memory so that AUTO mode can stop using the shortlist on
// maps where it does not prune (very smooth maps early in training).
#include "vsom_digits.hpp"
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SL_CMAX 2048

__global__ __launch_bounds__(256) void sl_norm_kernel(const float *__restrict__ map, int ldm, int Dp, int N,
                                                      float *__restrict__ nrm, unsigned *__restrict__ scal)
{
    // 16 lanes per node, float4 per lane (rows are zero padded to Dp, a multiple of 32);
    // scal[0] = max finite nrm (as uint bits), scal[1] = non-finite flag
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int node = gid >> 4, sub = gid & 15;
    const int nc = node < N ? node : N - 1;
    const float4 *row = reinterpret_cast<const float4 *>(map + (size_t)nc * ldm);
    const int n4 = Dp >> 2;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    bool nz = false;
    // 8 independent 16-byte loads in flight per lane (the kernel is a single pass over the map and
    // was latency-bound at 4)
    int q = sub;
    for (; q + 7 * 16 < n4; q += 8 * 16) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = row[q + 16 * u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s0 = s0 + v[u].x * v[u].x;
            s1 = s1 + v[u].y * v[u].y;
            s2 = s2 + v[u].z * v[u].z;
            s3 = s3 + v[u].w * v[u].w;
            nz |= !(v[u].x == 0.f) || !(v[u].y == 0.f) || !(v[u].z == 0.f) || !(v[u].w == 0.f);
        }
    }
    for (; q < n4; q += 16) {
        float4 v = row[q];
        s0 = s0 + v.x * v.x;
        s1 = s1 + v.y * v.y;
        s2 = s2 + v.z * v.z;
        s3 = s3 + v.w * v.w;
        nz |= !(v.x == 0.f) || !(v.y == 0.f) || !(v.z == 0.f) || !(v.w == 0.f);
    }
    if (__ballot(nz && node < N) && (threadIdx.x & 63) == 0 && scal[SLI_NONZERO] == 0u)
        atomicOr(&scal[SLI_NONZERO], 1u);        // some model value is not +-0 (sl_select_kernel: an all-zero map keeps node 0)
    float s = (s0 + s1) + (s2 + s3);
    for (int off = 8; off > 0; off >>= 1)
        s = s + __shfl_xor(s, off);
    // one atomic per workgroup: 16384 same-address atomics serialise in L2 (they were most of this
    // kernel's 55 us)
    __shared__ unsigned smax[4];
    unsigned bits = 0u;
    if (sub == 0 && node < N) {
        nrm[node] = s;
        if (s == s) {   // NaN rows are legal (excluded from every search)
            if (s > 3.0e38f)
                atomicOr(&scal[1], 1u);
            else
                bits = __float_as_uint(s);   // s >= 0: the bit pattern orders like the value
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        unsigned o = (unsigned)__shfl_xor((int)bits, off);
        bits = o > bits ? o : bits;
    }
    if ((threadIdx.x & 63) == 0)
        smax[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = smax[0];
        for (int i = 1; i < 4; ++i)
            m = smax[i] > m ? smax[i] : m;
        if (m)
            atomicMax(&scal[0], m);
    }
}

// CLR: a map whose nodes (nearly) coincide -- what batch training leaves when a chunk's samples lie on one line: every
// node's parameters converge to the same regression -- makes every node a candidate of every sample; the contraction
// would be wasted, the exact-order kernel searches such a map.  Decided on the device from the spread of the nodes'
// sum B^2 and max A^2 (clr_node_feat_kernel: scal[0] / [3] the maxima, scal[5] / [7] the inverted minima): the
// feature, contraction and refinement kernels of that search exit at once and every sample goes to the redo list.
// (A heuristic about SPEED only: whichever way it decides, indices and distances come from exact-order evaluations.)
__device__ __forceinline__ bool sl_clr_degenerate(const unsigned *__restrict__ scal)
{
    const float nbmax = __uint_as_float(scal[0]), a2max = __uint_as_float(scal[3]);
    const float nbmin = __uint_as_float(~scal[5]), a2min = __uint_as_float(~scal[7]);
    return (nbmax - nbmin) <= 1.0e-3f * nbmax && (a2max - a2min) <= 1.0e-3f * a2max;
}

#define GT 128     // block tile (samples x nodes)
#define GK 32      // K chunk
#define GLD 36     // LDS row stride (floats)

__global__ __launch_bounds__(256, 2) void sl_gemm_kernel(const float *__restrict__ X, int ldx, int s0, int s1,
                                                         const float *__restrict__ M, int ldm, int N, int Kp,
                                                         const float *__restrict__ nrm,
                                                         float *__restrict__ G, int ldg,
                                                         float *__restrict__ tmin, int ntm,
                                                         const unsigned *__restrict__ kp_dev,
                                                         const unsigned *__restrict__ clr_scal)
{
    if (clr_scal && sl_clr_degenerate(clr_scal))         // CLR on a collapsed map: the exact kernel searches (wave-uniform)
        return;
    // X / M gathered onto the chunk's live columns (vsom_compact.hip): the contraction length is a device value
    if (kp_dev)
        Kp = (int)kp_dev[2];
    __shared__ __attribute__((aligned(16))) float As[GT * GLD];
    __shared__ __attribute__((aligned(16))) float Bs[GT * GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int sbase = s0 + blockIdx.y * GT, nbase = blockIdx.x * GT;
    const int lr = lane & 31, lh = lane >> 5;

    f32x16 tot[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tot[i][j][r] = 0.f;

    constexpr int F4R = GK / 4;               // float4 per tile row
    constexpr int NLD = GT * F4R / 256;       // float4 per thread per operand
    float4 pa[NLD], pb[NLD];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int f = tid + 256 * i;
            int row = f / F4R, c4 = (f % F4R) * 4;
            int s = sbase + row, n = nbase + row;
            const bool kin = k0 + c4 < Kp;    // Kp is a multiple of 32, GK may be 64
            pa[i] = (s < s1 && kin) ? *reinterpret_cast<const float4 *>(X + (size_t)s * ldx + k0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            pb[i] = (n < N && kin) ? *reinterpret_cast<const float4 *>(M + (size_t)n * ldm + k0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    gload(0);
    for (int k0 = 0; k0 < Kp; k0 += GK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int f = tid + 256 * i;
            int row = f / F4R, c4 = (f % F4R) * 4;
            *reinterpret_cast<float4 *>(&As[row * GLD + c4]) = pa[i];
            *reinterpret_cast<float4 *>(&Bs[row * GLD + c4]) = pb[i];
        }
        __syncthreads();
        if (k0 + GK < Kp)
            gload(k0 + GK);
        // one fmaf chain of length GK per chunk (short chains keep the error bound tight)
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[i][j][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < GK; kb += 8) {
            float4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const float4 *>(&As[(wm * 64 + i * 32 + lr) * GLD + kb + 4 * lh]);
                b[i] = *reinterpret_cast<const float4 *>(&Bs[(wn * 64 + i * 32 + lr) * GLD + kb + 4 * lh]);
            }
            // k order per accumulator is unchanged (x, y, z, w); consecutive MFMAs go to different
            // accumulators so none waits for the previous one's result
            const float av[2][4] = {{a[0].x, a[0].y, a[0].z, a[0].w}, {a[1].x, a[1].y, a[1].z, a[1].w}};
            const float bv[2][4] = {{b[0].x, b[0].y, b[0].z, b[0].w}, {b[1].x, b[1].y, b[1].z, b[1].w}};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][c], bv[j][c], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tot[i][j][r] = tot[i][j][r] + acc[i][j][r];
    }
    // epilogue: G = nrm - 2*dot   (C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
    // plus the minimum of every sample row over this wavefront's 64 nodes (NaN never replaces)
    const float inf = __uint_as_float(0x7F800000u);
    const int col0 = nbase + wn * 64 + lr, col1 = col0 + 32;
    const float nm0 = col0 < N ? nrm[col0] : 0.f, nm1 = col1 < N ? nrm[col1] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = sbase + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float g0 = nm0 - 2.f * tot[i][0][r];
            float g1 = nm1 - 2.f * tot[i][1][r];
            if (row < s1) {
                if (col0 < N)
                    G[(size_t)(row - s0) * ldg + col0] = g0;
                if (col1 < N)
                    G[(size_t)(row - s0) * ldg + col1] = g1;
            }
            // exact minimum of the finite entries (select_kernel skips tiles by it): NaN -> +inf
            const float m0 = (col0 < N && g0 == g0) ? g0 : inf;
            const float m1 = (col1 < N && g1 == g1) ? g1 : inf;
            float mn = m1 < m0 ? m1 : m0;
            mn = sl_min32_dpp(mn);                     // the 32 lanes that share this row; valid in lanes 31 / 63
            if (lr == 31 && row < s1)
                tmin[(size_t)(row - s0) * ntm + blockIdx.x * 2 + wn] = mn;
        }
    }
}

// one workgroup (4 wavefronts) per sample: the tile scan and the exact-order evaluations of the
// shortlist are spread over the wavefronts (the evaluations are chains of dependent row reads, so
// more wavefronts per sample is what shortens them).  stats: [0] samples sent to the exact redo
// list, [1] total candidates
// CLR candidate evaluation for sl_select_kernel<true>: the sample's x' / y' rows sit in LDS (every
// candidate of the sample reads them), the node's A / B rows come from L2 in batches of U elements per
// accumulator class, all 2U loads issued before the first use.  Same operations in the same order per
// class as vsom_group_dist<true>, hence the same bits.
template <int U>
__device__ __forceinline__ float sl_clr_dist(const float *xl, const float *yl, const float *__restrict__ ma,
                                             const float *__restrict__ mb, int L, int k)
{
    const int L8 = L & ~7;
    float acc = 0.f;
    int d = k;
    for (; d + 8 * (U - 1) < L8; d += 8 * U) {
        float av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            av[u] = ma[d + 8 * u];
            bv[u] = mb[d + 8 * u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float r = vsom_resid<true>(xl[d + 8 * u], yl[d + 8 * u], av[u], bv[u]);
            const float p = r * r;
            acc = acc + p;
        }
    }
    for (; d < L8; d += 8) {
        const float r = vsom_resid<true>(xl[d], yl[d], ma[d], mb[d]);
        const float p = r * r;
        acc = acc + p;
    }
    float q = acc + __shfl_xor(acc, 4);
    const int rem = L - L8;
    if (rem >= 4) {
        const int e = L8 + (k & 3);
        const float r = vsom_resid<true>(xl[e], yl[e], ma[e], mb[e]);
        const float p = r * r;
        q = q + p;
    }
    const float t = q + __shfl_xor(q, 2);
    float res = t + __shfl_xor(t, 1);
    for (int tt = (rem >= 4 ? 4 : 0); tt < rem; ++tt) {
        const int e = L8 + tt;
        const float r = vsom_resid<true>(xl[e], yl[e], ma[e], mb[e]);
        const float p = r * r;
        res = res + p;
    }
    return res;
}

// CLR = true: the CombinatorialLinearRegression variant (features and bound: "CLR shortlist" below);
// xraw = the staged sample rows (J values each) for the per-sample constants, c_e1 the coefficient of the
// sqrt term of the exact-order error.
template <bool CLR>
__global__ __launch_bounds__(256) void sl_select_kernel(DistArgs a, int s0, int s1, int N, int D,
                                                        const float *__restrict__ G, int ldg,
                                                        const float *__restrict__ tmin, int ntm,
                                                        const unsigned *__restrict__ scal, float c_g1, float c_g2,
                                                        u64 *__restrict__ lastbmu, float *__restrict__ sqres,
                                                        unsigned *__restrict__ redo_count, int *__restrict__ redo_list,
                                                        unsigned *__restrict__ stats,
                                                        const float *__restrict__ xraw, int ldxr, int J, float c_e1,
                                                        unsigned cmax, const float *__restrict__ l1x, unsigned lstride,
                                                        float c_l1, const unsigned *__restrict__ xflag,
                                                        const float *__restrict__ nrmn, const float *__restrict__ a2n,
                                                        const float *__restrict__ nrm0)
{
    __shared__ unsigned cand[SL_CMAX];
    __shared__ float s_um[4];
    __shared__ unsigned s_cnt;
    __shared__ u64 s_best[4];
    __shared__ int s_nan0;
    extern __shared__ __attribute__((aligned(16))) float s_xy[];   // CLR: x' row | y' row (2 * ldx floats)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = s0 + blockIdx.x;
    if (s >= s1)
        return;   // block-uniform
    if (threadIdx.x == 0)
        s_cnt = 0u;
    const float *xr = a.xa + (size_t)s * a.ldx;
    const float *yr = a.xb + (size_t)s * a.ldx;          // CLR: y' row (Standard: same as xr, unused)
    const float *g = G + (size_t)(s - s0) * ldg;
    const float *tm = tmin + (size_t)(s - s0) * ntm;
    if (CLR) {   // rows are padded to a multiple of 32 floats: 16-byte copies
        const float4 *sx = reinterpret_cast<const float4 *>(xr), *sy = reinterpret_cast<const float4 *>(yr);
        float4 *dx = reinterpret_cast<float4 *>(s_xy), *dy = reinterpret_cast<float4 *>(s_xy + a.ldx);
        for (int i = threadIdx.x; i < (a.ldx >> 2); i += 256) {
            dx[i] = sx[i];
            dy[i] = sy[i];
        }
    }
    // |x|^2 and the row minimum of the approximations: every wavefront computes both, in the same
    // order, so all of them hold identical values (a NaN never replaces the incumbent minimum)
    float nx = 0.f, cx = 0.f;
    if (CLR) {
        // cx = sum_p x'_p^2, cy = sum_p y'_p^2: column t is the first index of J-1-t pairs and the
        // second index of t pairs (pairs i<j, Transformation.cpp:94-101); nx holds cy
        const float *xw = xraw + (size_t)s * ldxr;
        for (int t = lane; t < J; t += 64) {
            const float v = xw[t];
            const float q = v * v;
            cx = cx + q * (float)(J - 1 - t);
            nx = nx + q * (float)t;
        }
        for (int off = 32; off > 0; off >>= 1) {
            cx = cx + __shfl_xor(cx, off);
            nx = nx + __shfl_xor(nx, off);
        }
    } else if (c_l1 > 0.f) {
        nx = l1x[4 * (size_t)lstride + s];               // integer contraction: |x|^2 from the quantisation pass
    } else {
        for (int d = lane; d < D; d += 64) {
            float v = xr[d];
            float p = v * v;
            nx = nx + p;
        }
        for (int off = 32; off > 0; off >>= 1)
            nx = nx + __shfl_xor(nx, off);
    }
    // integer contraction (vsom_sl_i8.hip, c_l1 > 0): the maxima of |M|^2, of the digit residual eps and of |M|_1 sit
    // in 32 line-sized slots each; the approximation error is c_l1 (e_s L1Mmax + l1eff_s eps_max) + c_g1 (nMmax + |x|^2),
    // with (e_s, l1eff_s) = (0, |x|_1) for a chunk of the uint8 kind (`xflag` clear) and the digit-grid terms otherwise
    float nmax = __uint_as_float(scal[0]), epsmax = 0.f, l1mmax = 0.f;
    if (c_l1 > 0.f) {                                    // folded by the contraction kernel (sl_fold_maxima, vsom_sl_i8.hip)
        nmax = __uint_as_float(scal[8]);                 // non-negative floats: the bit patterns order like the values
        epsmax = __uint_as_float(scal[9]);
        l1mmax = __uint_as_float(scal[10]);
    }
    bool bad = (scal[1] != 0u) || !(nx <= 3.0e38f) || !(cx <= 3.0e38f) || (CLR && sl_clr_degenerate(scal));
    // an all-zero map (what the epoch over an EMPTY chunk leaves, Som.cpp:840-875 -- every chunked MnistDataLoader pass
    // ends with one): every distance is the same sum over x in the same order, strict `<` keeps node 0 (Som.cpp:293-304)
    const bool zero_map = scal[SLI_NONZERO] == 0u;
    float m = __uint_as_float(0x7F800000u);
    for (int i = lane; i < ntm; i += 64) {
        float v = tm[i];
        m = v < m ? v : m;
    }
    for (int off = 32; off > 0; off >>= 1) {
        float o = __shfl_xor(m, off);
        m = o < m ? o : m;
    }
    __syncthreads();   // s_cnt = 0 visible
    if (!bad && !zero_map) {
        const u64 below = (1ull << lane) - 1ull;
        if (CLR) {
            // "CLR shortlist" below, with every bound PER NODE: the exact-order value of node i (minus the sample's
            // constant cy) lies in [G_i - w_i, G_i + w_i], w_i = Ea_i + Ee_i from the node's own Q_i = A2max_i cx + nB_i + cy.
            // The reference argmin i* satisfies G_i* - w_i* <= min_j (G_j + w_j): that set is the shortlist.  (A map-wide
            // Q -- round 2 -- fails on the maps CLR training really leaves: a tenth of the nodes run away to |A| ~ 1e3..1e11
            // on correlated data, and ONE such node made every node a candidate of every sample.)
            auto width = [&](float gi, int i) -> float {
                const float Q = 1.01f * (a2n[i] * cx + nrmn[i] + nx);
                const float ea = c_g1 * Q;                   // c_g1 = ga
                float dj = gi + nx;
                dj = dj + ea;
                dj = dj > 0.f ? dj : 0.f;
                const float dd = 1.02f * dj + 2.0e-5f * Q;   // covers the true squared distance (derivation below)
                const float ee = c_e1 * sqrtf(dd * Q) + c_g2 * dd;
                return 1.05f * (ea + ee);
            };
            float um = __uint_as_float(0x7F800000u);
            for (int i = threadIdx.x; i < N; i += 256) {
                const float gi = g[i];
                const float u = gi + width(gi, i);
                um = u < um ? u : um;                        // NaN (a NaN row) never replaces
            }
            for (int off = 32; off > 0; off >>= 1) {
                const float o = __shfl_xor(um, off);
                um = o < um ? o : um;
            }
            if (lane == 0)
                s_um[wave] = um;
            __syncthreads();
            um = fminf(fminf(s_um[0], s_um[1]), fminf(s_um[2], s_um[3]));
            if (!(um < 3.0e38f))
                bad = true;                                  // nothing finite to compare with
            for (int i0 = wave * 64; i0 < N && !bad; i0 += 256) {
                const int i = i0 + lane;
                bool c = false;
                if (i < N) {
                    const float gi = g[i];
                    c = gi - width(gi, i) <= um;
                }
                const u64 mask = __ballot(c);
                if (mask) {
                    unsigned base = 0;
                    if (lane == 0)
                        base = atomicAdd(&s_cnt, (unsigned)__popcll(mask));
                    base = __shfl(base, 0);
                    const unsigned slot = base + (unsigned)__popcll(mask & below);
                    if (c && slot < SL_CMAX)
                        cand[slot] = (unsigned)i;
                }
            }
        } else {
            // T_s = 4*g1*(nMmax+nx) + 2.1*g2*(m + nx + 2*g1*(nMmax+nx)), inflated by 1.05 (header)
            float ea = c_g1 * (nmax + nx);                   // c_g1 = 2*g1 (fp32 chain) / 5.5u (integer contraction)
            if (c_l1 > 0.f) {
                const bool gen = xflag[0] != 0u;
                const float l1e = l1x[(gen ? 2 * (size_t)lstride : 0) + s], es = gen ? l1x[3 * (size_t)lstride + s] : 0.f;
                // (+ the fp32 epilogues' second rounding: 2^-7 eps_n for the uint8 kind, 2^-8 t_s eps_n for the general one)
                ea = ea + 1.001f * c_l1 * (l1e * epsmax + es * l1mmax) + 0.01f * epsmax * (gen ? fmaxf(1.f, l1x[(size_t)lstride + s]) : 1.f);
            }
            float dj = m + nx;
            dj = dj + ea;
            dj = dj > 0.f ? dj : 0.f;
            const float T = 1.05f * (2.f * ea + c_g2 * dj);  // c_g2 = 2.1*g2
            const float thr = m + T;
            if (!(thr < 3.0e38f))
                bad = true;                                  // nothing finite to compare with
            // collect candidates (order irrelevant: the key min decides).  tmin[t] is the exact minimum of
            // the non-NaN entries of columns [64t, 64t+64), so only tiles with tmin <= thr can hold a
            // candidate: one coalesced 256-byte read per such tile; wavefront w takes tiles 64w + 256k + lane
            for (int t0 = wave * 64; t0 < ntm && !bad; t0 += 256) {
                const int t = t0 + lane;
                u64 hm = __ballot(t < ntm && tm[t] <= thr);
                while (hm) {
                    const int tl = __ffsll((long long)hm) - 1;
                    hm &= hm - 1ull;
                    const int i = (t0 + tl) * 64 + lane;
                    const bool c = i < N && g[i] <= thr;
                    const u64 mask = __ballot(c);
                    if (mask) {
                        unsigned base = 0;
                        if (lane == 0)
                            base = atomicAdd(&s_cnt, (unsigned)__popcll(mask));
                        base = __shfl(base, 0);
                        const unsigned slot = base + (unsigned)__popcll(mask & below);
                        if (c && slot < SL_CMAX)
                            cand[slot] = (unsigned)i;
                    }
                }
            }
        }
    }
    __syncthreads();
    const unsigned cnt = s_cnt;
    if (cnt > cmax)     // cmax <= SL_CMAX: beyond it the exact tile kernel is the cheaper way to search this sample
        bad = true;
    if (bad) {   // block-uniform
        if (threadIdx.x == 0) {
            unsigned slot = atomicAdd(redo_count, 1u);
            redo_list[slot] = s;
            atomicAdd(&stats[0], 1u);
        }
        return;
    }
    // exact-order evaluation: 8 lanes per candidate, 32 candidates per pass over the workgroup;
    // node 0 always included (wavefront 0)
    const int grp = lane >> 3, k = lane & 7;
    u64 best = ~0ull;
    // Node 0 seeds the reference's search (Som.cpp:293-299): a NaN there pins the BMU to 0.  Standard / Median: with a
    // finite sample and no inf in the map (both checked above) its distance is NaN exactly when its row holds a NaN, i.e.
    // when |M_0|^2 is NaN; otherwise it is a node like any other, and the argmin always is among the candidates (its G is
    // the row minimum or within the bound of it) -- so its 784-element evaluation, which used to double wavefront 0's
    // critical path, is skipped.
    const bool need0 = CLR || nrm0 == nullptr || zero_map || nrm0[0] != nrm0[0];
    if (threadIdx.x == 0)
        s_nan0 = 0;
    if (wave == 0 && need0) {
        float d0 = CLR ? sl_clr_dist<8>(s_xy, s_xy + a.ldx, a.ma, a.mb, a.L, k) : vsom_group_dist<false>(xr, xr, a.ma, a.ma, a.L, k);
        d0 = __shfl(d0, 0);
        best = vsom_key(d0, 0u);
        if (lane == 0)
            s_nan0 = d0 != d0;
    }
    for (unsigned c0 = 0; c0 < cnt; c0 += 32) {
        const unsigned ci = c0 + (unsigned)(wave * 8 + grp);
        const unsigned node = cand[ci < cnt ? ci : cnt - 1];
        float d = CLR ? sl_clr_dist<8>(s_xy, s_xy + a.ldx, a.ma + (size_t)node * a.ldm, a.mb + (size_t)node * a.ldm, a.L, k)
                      : vsom_group_dist<false>(xr, xr, a.ma + (size_t)node * a.ldm, a.ma, a.L, k);
        u64 key = ci < cnt ? vsom_key(d, node) : ~0ull;
        best = key < best ? key : best;
    }
    for (int off = 32; off >= 8; off >>= 1) {
        u64 o = __shfl_xor(best, off);
        best = o < best ? o : best;
    }
    if (lane == 0)
        s_best[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        // statistics only, but 4096 same-address atomics per chunk queue up at the memory side: the
        // count goes to one of 32 line-sized slots (scal[16 + 32 s]); sl_feedback_write adds them up
        atomicAdd(&stats[12 + 32 * (blockIdx.x & 31)], cnt);
        u64 b = s_best[0];
        for (int i = 1; i < 4; ++i)
            b = s_best[i] < b ? s_best[i] : b;
        if (s_nan0) {
            lastbmu[s] = 0;
            sqres[s] = __uint_as_float(0x7FC00000u);
        } else {
            lastbmu[s] = b & 0xFFFFFFFFull;
            sqres[s] = __uint_as_float((uint32_t)(b >> 32));
        }
    }
}

// The refinement of the G-less integer contraction (sl_k64_kernel, vsom_sl_i8.hip: at most 64 contracted columns).
// One WAVEFRONT per sample: tmin[sample][t] is the minimum of the approximations over tile t = 16 nodes
// (32 (t >> 1) + 4 (t & 1) + {0..3, 8..11, 16..19, 24..27}); every node of every tile with tmin <= row minimum + T_s
// (the bound of sl_select_kernel, same terms) is evaluated in the reference's order, 8 lanes per node, node 0 always.
// More than `tmax` such tiles, or a sample / map the bound does not cover: the redo list.
// WPS = wavefronts per sample: 1 for short rows (four samples per workgroup); 2 or 4 for long ones (the (tile, half) items dealt
// round robin to the sample's wavefronts: a 784-element row is seven dependent load batches per evaluation, and the ~21
// candidate nodes of a C3 sample are three passes of eight for one wavefront.  Measured at C3: 69 us with one wavefront
// per sample, 58 with four -- a workgroup per sample is 2.7 rounds of resident workgroups -- 53 with two).
template <int WPS>
__global__ __launch_bounds__(256) void sl_pick_kernel(DistArgs a, int s0, int s1, int N, int D,
                                                      const float *__restrict__ tmin, int ntl,
                                                      const unsigned *__restrict__ scal, float c_g1, float c_g2,
                                                      u64 *__restrict__ lastbmu, float *__restrict__ sqres,
                                                      unsigned *__restrict__ redo_count, int *__restrict__ redo_list,
                                                      unsigned *__restrict__ stats, unsigned tmax,
                                                      const float *__restrict__ l1x, unsigned lstride, float c_l1,
                                                      const unsigned *__restrict__ xflag, const float *__restrict__ nrm0)
{
    // candidate statistics: one global atomic per workgroup (WPS = 1: the last of its four wavefronts to arrive sends the sum)
    __shared__ unsigned s_tot, s_arr;
    __shared__ u64 s_best[4];
    // long rows (WPS > 1, at most 1024 values): the sample's row in LDS -- the evaluations then fetch model rows only, 28
    // elements per class at a time instead of 14 + 14
    constexpr int XL = WPS == 1 ? 4 : 1024;
    __shared__ __attribute__((aligned(16))) float s_x[WPS == 1 ? 1 : 4 / WPS][XL];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s_own = s0 + (int)blockIdx.x * (4 / WPS) + wave / WPS;   // WPS wavefronts per sample, 4 / WPS samples per workgroup
    const int sub = wave % WPS;
    // WPS > 1: the wavefronts of a workgroup meet at a barrier at the end, so none may leave before it.  The wavefronts of a
    // sample past the range (the last workgroup of an odd range) work on the range's last sample with every side effect
    // switched off (`active`) instead of returning early.
    const bool active = s_own < s1;
    const int s = (WPS == 1 || active) ? s_own : s1 - 1;
    const bool xlds = WPS != 1 && a.L <= XL;             // kernel-uniform
    if (xlds) {
        const float4 *src = reinterpret_cast<const float4 *>(a.xa + (size_t)s * a.ldx);
        for (int i = sub * 64 + lane; i < ((a.L + 3) >> 2); i += 64 * WPS)     // rows are zero padded to a multiple of 32
            reinterpret_cast<float4 *>(s_x[wave / WPS])[i] = src[i];
    }
    if (threadIdx.x == 0) {
        s_tot = 0u;
        s_arr = 0u;
    }
    __syncthreads();      // WPS = 1: the only barrier: every wavefront passes it before any leaves
    auto leave = [&](unsigned cands) {
        if (WPS != 1) {
            if (lane == 0 && sub == 0 && cands)
                atomicAdd(&stats[12 + 32 * (blockIdx.x & 31)], cands);
            return;
        }
        if (lane == 0) {
            atomicAdd(&s_tot, cands);
            __threadfence_block();
            if (atomicAdd(&s_arr, 1u) == 3u) {
                const unsigned tot = atomicAdd(&s_tot, 0u);
                if (tot)
                    atomicAdd(&stats[12 + 32 * (blockIdx.x & 31)], tot);
            }
        }
    };
    if (WPS == 1 && s >= s1) {   // wavefront-uniform (the barrier above was this form's only one)
        leave(0u);
        return;
    }
    const float *xr = xlds ? s_x[wave / WPS] : a.xa + (size_t)s * a.ldx;
    const float *tm = tmin + (size_t)(s - s0) * ntl;
    // |x|^2 from the quantisation pass (plane 4 of l1x), the map-wide maxima from sl_k64_kernel (scal[8..10])
    const float nx = l1x[4 * (size_t)lstride + s];
    const float nmax = __uint_as_float(scal[8]), epsmax = __uint_as_float(scal[9]), l1mmax = __uint_as_float(scal[10]);
    bool bad = (scal[1] != 0u) || !(nx <= 3.0e38f);
    const bool zero_map = scal[SLI_NONZERO] == 0u;       // sl_select_kernel: every node ties, node 0 wins
    // the sample's tile minima: up to 1024 of them (a 128 x 128 map) stay in registers for the candidate pass below -- 16
    // loads in flight once instead of one dependent load per 64 tiles twice over
    constexpr int TV = WPS == 1 ? 4 : 16;                // (short rows: small maps' 256 tiles; fewer registers, more wavefronts)
    const float inf = __uint_as_float(0x7F800000u);
    float tv[TV];
#pragma unroll
    for (int i = 0; i < TV; ++i)
        tv[i] = i * 64 + lane < ntl ? tm[i * 64 + lane] : inf;
    float m = inf;
#pragma unroll
    for (int i = 0; i < TV; ++i)
        m = tv[i] < m ? tv[i] : m;
    for (int i = TV * 64 + lane; i < ntl; i += 64) {
        const float v = tm[i];
        m = v < m ? v : m;
    }
    m = sl_min32_dpp(m);                                 // (entries are never NaN: the contraction kernels' fminf)
    m = fminf(__shfl(m, 31), __shfl(m, 63));
    // T_s as in sl_select_kernel (integer contraction, fp32 epilogue)
    float ea = c_g1 * (nmax + nx);
    {
        const bool gen = xflag[0] != 0u;
        const float l1e = l1x[(gen ? 2 * (size_t)lstride : 0) + s], es = gen ? l1x[3 * (size_t)lstride + s] : 0.f;
        ea = ea + 1.001f * c_l1 * (l1e * epsmax + es * l1mmax) + 0.01f * epsmax;
        if (gen)     // sl_k64_kernel converts u = 128 a1 + a2 (< 2^27) to fp32: off by at most 4, times t_s s_n 2^-13
            ea = ea + 16.1f * l1x[(size_t)lstride + s] * epsmax;
    }
    float dj = m + nx;
    dj = dj + ea;
    dj = dj > 0.f ? dj : 0.f;
    const float T = 1.05f * (2.f * ea + c_g2 * dj);
    const float thr = m + T;
    if (!(thr < 3.0e38f))
        bad = true;
    const int grp = lane >> 3, k = lane & 7;
    // Node 0 seeds the reference's search (Som.cpp:293-299): a NaN there pins the BMU to 0.  With a finite sample and no
    // inf in the map (both checked above) its distance is NaN exactly when its row holds a NaN, i.e. when |M_0|^2 is NaN;
    // otherwise node 0 is a node like any other and the argmin's tile is always among the candidates (tmin = m <= thr).
    const float n0 = nrm0[0];
    const bool nan0 = n0 != n0;
    u64 best = ~0ull;
    if (zero_map || nan0) {
        float d0 = vsom_group_dist_lat<false>(xr, xr, a.ma, a.ma, a.L, k);
        d0 = __shfl(d0, 0);
        best = vsom_key(d0, 0u);
    }
    unsigned ntiles = 0, item = 0;
    if (!bad && !zero_map && !nan0) {
#pragma unroll 1
        for (int t0 = 0; t0 < ntl; t0 += 64) {
            const int t = t0 + lane, ti = t0 >> 6;
            float tval = inf;
            if (ti < TV) {
#pragma unroll
                for (int i = 0; i < TV; ++i)             // (a register array indexed by a loop counter: a select chain)
                    tval = ti == i ? tv[i] : tval;
            } else if (t < ntl) {
                tval = tm[t];
            }
            u64 hm = __ballot(tval <= thr);
            ntiles += (unsigned)__popcll(hm);
            if (ntiles > tmax) {
                bad = true;
                break;
            }
            while (hm) {
                const int tl = t0 + __ffsll((long long)hm) - 1;
                hm &= hm - 1ull;
                const int nbase = 32 * (tl >> 1) + 4 * (tl & 1);
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    if (WPS != 1 && (int)(item++ % WPS) != sub)
                        continue;                        // another wavefront's item
                    const int j = pass * 8 + grp;
                    const int node = nbase + (j & 3) + 8 * (j >> 2);
                    const bool ok = node < N;
                    const float *mrow = a.ma + (size_t)(ok ? node : 0) * a.ldm;
                    const float d = xlds ? vsom_group_dist_lat<false, 28>(s_x[wave / WPS], s_x[wave / WPS], mrow, a.ma, a.L, k)
                                         : vsom_group_dist_lat<false>(xr, xr, mrow, a.ma, a.L, k);
                    const u64 key = ok ? vsom_key(d, (unsigned)node) : ~0ull;
                    best = key < best ? key : best;
                }
            }
        }
    }
    auto redo = [&]() {   // `bad` is wavefront-uniform, and every wavefront of a sample reaches the same verdict
        if (lane == 0 && sub == 0) {
            const unsigned slot = atomicAdd(redo_count, 1u);
            redo_list[slot] = s;
            atomicAdd(&stats[0], 1u);
        }
        leave(0u);
    };
    if (WPS == 1 && bad) {
        redo();
        return;
    }
    for (int off = 32; off >= 8; off >>= 1) {
        const u64 o = __shfl_xor(best, off);
        best = o < best ? o : best;
    }
    if (WPS != 1) {
        if (lane == 0)
            s_best[wave] = best;
        __syncthreads();                                 // every wavefront of the workgroup arrives here
        if (sub != 0 || !active)
            return;
        if (bad) {
            redo();
            return;
        }
        for (int w = 1; w < WPS; ++w)
            best = s_best[wave + w] < best ? s_best[wave + w] : best;
    }
    leave(16u * ntiles);
    if (lane == 0) {
        if (nan0) {
            lastbmu[s] = 0;
            sqres[s] = __uint_as_float(0x7FC00000u);
        } else {
            lastbmu[s] = best & 0xFFFFFFFFull;
            sqres[s] = __uint_as_float((uint32_t)(best >> 32));
        }
    }
}

// host side ------------------------------------------------------------------------------------
int launch_bmu_full_exact_list(vsom_ctx *c, size_t s0, size_t s1, const int *slist, const unsigned *scount, const SlFeedback *fb);
int launch_sl_i8(vsom_ctx *c, size_t s0, size_t s1, size_t ldg, size_t ntm, unsigned *scal, unsigned *xflag, int plan);   // vsom_sl_i8.hip
int sl_i8_plan(vsom_ctx *c, size_t s0, size_t s1);

static int launch_bmu_full_shortlist_clr(vsom_ctx *c, size_t s0, size_t s1);

int launch_bmu_full_shortlist(vsom_ctx *c, size_t s0, size_t s1)
{
    if (s1 <= s0)
        return VSOM_OK;
    if (c->transform == VSOM_CLR)
        return launch_bmu_full_shortlist_clr(c, s0, s1);
    const size_t nrows = s1 - s0;
    // the contraction in exact integer arithmetic on the int8 matrix pipe (vsom_sl_i8.hip): one digit per sample value
    // for chunks of small non-negative integers (MNIST pixels), three for any other data -- which of the two is a
    // device-side fact of the staged chunk (`xflag`) that the kernels read; nothing is decided here
    const bool i8 = c->xpitch <= 4096;
    // no B x N matrix where the contraction kernels keep 16-node tile minima only (rows of at most 64 values; big
    // problems of at most 960 contracted columns): sl_pick_kernel refines
    const int plan = i8 ? sl_i8_plan(c, s0, s1) : 0;
    const bool gless = plan != 0;
    const size_t ldg = ((size_t)c->N + 127) / 128 * 128;
    const size_t need = gless ? 0 : nrows * ldg;
    if (need > c->sl_cap) {
        if (c->sl_G)
            VSOM_HIP_CHECK(hipFree(c->sl_G));
        c->sl_G = nullptr;
        VSOM_HIP_CHECK(hipMalloc(&c->sl_G, need * sizeof(float)));
        c->sl_cap = need;
    }
    const size_t ntm = gless ? (((size_t)c->N + 31) / 32) * 2 : (((size_t)c->N + GT - 1) / GT) * 2;
    if (nrows * ntm > c->sl_tmin_cap) {
        if (c->sl_tmin)
            VSOM_HIP_CHECK(hipFree(c->sl_tmin));
        c->sl_tmin = nullptr;
        VSOM_HIP_CHECK(hipMalloc(&c->sl_tmin, nrows * ntm * sizeof(float)));
        c->sl_tmin_cap = nrows * ntm;
    }
    if (nrows > c->sl_list_cap) {
        if (c->sl_list)
            VSOM_HIP_CHECK(hipFree(c->sl_list));
        c->sl_list = nullptr;
        VSOM_HIP_CHECK(hipMalloc(&c->sl_list, nrows * sizeof(int)));
        c->sl_list_cap = nrows;
    }
    if (!c->sl_nrm) {
        VSOM_HIP_CHECK(hipMalloc(&c->sl_nrm, (size_t)c->N * sizeof(float)));
        // two alternating sets of {16 words + 3 x 32 line-sized slots} (4096 words each) + the chunk's data-kind flag
        VSOM_HIP_CHECK(hipMalloc(&c->sl_scal, 3 * 16384));
        VSOM_HIP_CHECK(hipMemsetAsync(c->sl_scal, 0, 3 * 16384, c->stream));
        VSOM_HIP_CHECK(hipHostMalloc(&c->sl_fb, 64));
        std::memset(c->sl_fb, 0, 64);
    }
    // scal: [0] max nrm bits, [1] non-finite flag, [2] redo count, [4] redo samples, [5] candidates
    unsigned *scal = c->sl_scal + 4096 * c->sl_par, *scal_next = c->sl_scal + 4096 * (c->sl_par ^ 1);
    c->sl_par ^= 1;
    unsigned *xflag = c->sl_scal + 8192;
    dim3 grid((unsigned)((c->N + GT - 1) / GT), (unsigned)((nrows + GT - 1) / GT));
    const double u = 5.9604644775390625e-08;   // 2^-24
    const double g2 = ((double)c->D / 8.0 + 10.0) * u;
    if (i8) {
        int rc = launch_sl_i8(c, s0, s1, ldg, ntm, scal, xflag, plan);
        if (rc)
            return rc;
    } else {
    hipLaunchKernelGGL(sl_norm_kernel, dim3((unsigned)(((size_t)c->N * 16 + 255) / 256)), dim3(256), 0, c->stream,
                       c->map, (int)c->pitch, (int)c->part_pitch, (int)c->N, c->sl_nrm, scal);
    if (c->cc_valid) {
        // columns that are zero in every row of the chunk add exactly 0 to every <x, M>: contract over the live
        // ones (norms, bound and refinement keep the whole rows)
        int rc = vsom_cc_gather_map(c);
        if (rc)
            return rc;
        hipLaunchKernelGGL(sl_gemm_kernel, grid, dim3(256), 0, c->stream, c->Xc, (int)c->cpitch, (int)s0, (int)s1,
                           c->Mc, (int)c->cpitch, (int)c->N, (int)c->cpitch, c->sl_nrm, c->sl_G, (int)ldg, c->sl_tmin, (int)ntm,
                           (const unsigned *)c->cc_meta, (const unsigned *)nullptr);
    } else
    hipLaunchKernelGGL(sl_gemm_kernel, grid, dim3(256), 0, c->stream, c->Xs, (int)c->xpitch, (int)s0, (int)s1,
                       c->map, (int)c->pitch, (int)c->N, (int)c->xpitch, c->sl_nrm, c->sl_G, (int)ldg, c->sl_tmin, (int)ntm,
                       (const unsigned *)nullptr, (const unsigned *)nullptr);
    }
    DistArgs a;
    a.xa = c->Xs;
    a.xb = c->Xs;
    a.ldx = (int)c->xpitch;
    a.ma = c->map;
    a.mb = c->map;
    a.ldm = (int)c->pitch;
    a.L = (int)c->part_len;
    const double K = (double)c->xpitch;
    const double g1 = ((double)GK + K / GK + 3.0) * u;
    // integer contraction: |G - (|M|^2 - 2<x,M>)| <= 2 (e_s L1Mmax + l1eff_s eps_max) + 5.1u (nMmax + |x|^2) + 2^-7 eps_max
    // (vsom_sl_i8.hip; 3.1u with the fp64 epilogue, 5.1u with the two-rounding fp32 one of the uint8 kind)
    if (gless && c->D <= 64)
        hipLaunchKernelGGL(sl_pick_kernel<1>, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, c->stream, a, (int)s0, (int)s1, (int)c->N,
                           (int)c->D, c->sl_tmin, (int)ntm, scal, (float)(5.5 * u), (float)(2.1 * g2), c->lastbmu, c->sqres, scal + 2,
                           c->sl_list, scal + 4, 64u, (const float *)c->sl_l1, (unsigned)c->Bcap, 2.0f, (const unsigned *)xflag,
                           (const float *)c->sl_nrm);
    else if (gless && nrows < 4096)     // few samples (a rank's share of a chunk): a workgroup per sample (0.109 -> 0.089 ms at 512)
        hipLaunchKernelGGL(sl_pick_kernel<4>, dim3((unsigned)nrows), dim3(256), 0, c->stream, a, (int)s0, (int)s1, (int)c->N,
                           (int)c->D, c->sl_tmin, (int)ntm, scal, (float)(5.5 * u), (float)(2.1 * g2), c->lastbmu, c->sqres, scal + 2,
                           c->sl_list, scal + 4, 64u, (const float *)c->sl_l1, (unsigned)c->Bcap, 2.0f, (const unsigned *)xflag,
                           (const float *)c->sl_nrm);
    else if (gless)
        hipLaunchKernelGGL(sl_pick_kernel<2>, dim3((unsigned)((nrows + 1) / 2)), dim3(256), 0, c->stream, a, (int)s0, (int)s1, (int)c->N,
                           (int)c->D, c->sl_tmin, (int)ntm, scal, (float)(5.5 * u), (float)(2.1 * g2), c->lastbmu, c->sqres, scal + 2,
                           c->sl_list, scal + 4, 64u, (const float *)c->sl_l1, (unsigned)c->Bcap, 2.0f, (const unsigned *)xflag,
                           (const float *)c->sl_nrm);
    else
    hipLaunchKernelGGL(sl_select_kernel<false>, dim3((unsigned)nrows), dim3(256), 0, c->stream, a, (int)s0,
                       (int)s1, (int)c->N, (int)c->D, c->sl_G, (int)ldg, c->sl_tmin, (int)ntm, scal, (float)(i8 ? 5.5 * u : 2.0 * g1),
                       (float)(2.1 * g2), c->lastbmu, c->sqres, scal + 2, c->sl_list, scal + 4, (const float *)nullptr, 0, 0, 0.f,
                       (unsigned)SL_CMAX, (const float *)c->sl_l1, (unsigned)c->Bcap, i8 ? 2.0f : 0.f, (const unsigned *)xflag,
                       (const float *)nullptr, (const float *)nullptr, (const float *)c->sl_nrm);
    VSOM_HIP_CHECK(hipGetLastError());
    // exact-order redo of the listed samples (device-side count; its workgroups walk the list) + the feedback words
    const SlFeedback fb = {scal, c->sl_fb, (unsigned)nrows, i8 ? (const unsigned *)xflag : (const unsigned *)nullptr, scal_next};
    return launch_bmu_full_exact_list(c, s0, s1, c->sl_list, scal + 2, &fb);
}

// ==============================================================================================
// CLR shortlist: Som::findBmu for the CombinatorialLinearRegression comparer (Transformation.cpp:82-106)
// ==============================================================================================
// Exact distance of sample s and node n:  d = sum_p r_p^2,  r_p = A_p x'_p + B_p - y'_p,  x'_p = x[i(p)],
// y'_p = x[j(p)], pairs i<j lexicographic, P = J(J-1)/2.  Expanding the square and grouping by input column,
//   d - sum_p y'_p^2 = nB_n - 2 * < phi(s), psi(n) >,         nB_n = sum_p B_p^2,
//   phi = [ x_i x_j (P) | x_t^2 (J) | x_t (J) | x_t (J) ],
//   psi = [ A_p     (P) | -1/2 sum_{p:i(p)=t} A_p^2 | -sum_{p:i(p)=t} A_p B_p | +sum_{p:j(p)=t} B_p ],
// i.e. ONE contraction of length K = P + 3J (not 4P) that sl_gemm_kernel computes as it stands
// (G = nrm - 2 X.M^T with X = phi rows, M = psi rows, nrm = nB).  sum_p y'^2 =: cy_s is constant per sample.
//
// Bound (u = 2^-24).  Let Q_ns = sum_p (A_p x'_p)^2 + B_p^2 + y'_p^2  <=  amax2 * cx_s + nBmax + cy_s =: Q_s
// (amax2 = max A_p^2 and nBmax = max nB over the map, cx_s = sum_p x'_p^2).
//  (a) approximation:  sum_k |phi_k psi_k| <= sum_p [A^2x'^2 + 2|A B x'| + 2|A x' y'| + 2|B y'|] <= 3 Q_ns
//      (2|ab| <= a^2 + b^2).  Features carry relative errors <= gf = (J+3)u (one product + a sequential sum of
//      <= J terms), the contraction g1 = (GK + K/GK + 3)u as for the Standard search, nB (P/256+12)u:
//          |G_ns + cy_s - d_ns| <= Ea := ga * Q_s,     ga = 3.01*(g1 + gf) + (P/256 + 19)u
//      (the last 7u: rounding of nB - 2*dot itself, |G| <= nB + 6 Q_ns).
//  (b) exact-order value e_ns (three roundings per residual, a square, Eigen's sum tree):
//      |fl(r_p) - r_p| <= 3.01u * rho_p,  rho_p = |A x'| + |B| + |y'|,  sum rho_p^2 <= 3 Q_ns, so
//          |e_ns - d_ns| <= 6.1u * sqrt(3 d_ns Q_s) + g2 * d_ns =: Ee(d_ns),     g2 = (P/8 + 10)u.
//  (c) with i* the reference argmin and jm = argmin G (m = G_jm):  d_jm <= dj := m + cy + Ea, and
//      d_i* <= e_i* + Ee <= e_jm + Ee <= dj + Ee(dj) + Ee(d_i*), which for coefficients of this size gives
//      d_i* <= 1.0002 dj + 1e-6 Q_s; both are below dd := 1.02 dj + 2e-5 Q_s.  Then
//          G_i* <= m + 2 Ea + 2 Ee(dd) =: m + T_s            (inflated by 1.05 for the fp32 evaluation).
// Everything within T_s of the row minimum is re-evaluated in the reference's order; rows with NaN drop
// out (G = NaN), node 0 is always evaluated, an inf anywhere in the map sends the chunk to the exact kernel.
// A candidate costs ~1 ns here (16 KB of model rows gathered from L2) against ~280 ns per sample in the
// exact tile kernel at C5's size, so a sample with more than 128 candidates -- maps with many (near-)
// duplicate nodes: after a batch epoch on strongly correlated data whole rows of the map can coincide --
// goes to the redo list instead (measured at C5: 1 candidate per sample on a random map 0.65 ms, 32 per
// sample 0.69 ms, every node a candidate 8-9 ms before the cap; the exact kernel alone 2.3-2.5 ms).

// phi rows.  One workgroup per sample; Kp = P32 + roundup(3J, 32), P32 = roundup(P, 32).
__global__ __launch_bounds__(256) void clr_sample_feat_kernel(const float *__restrict__ XP, const float *__restrict__ YP, int ldp,
                                                              int P, const float *__restrict__ Xs, int ldx, int J,
                                                              float *__restrict__ Fs, int Kp, int P32, int s0, int s1,
                                                              const unsigned *__restrict__ scal)
{
    const int s = s0 + blockIdx.x;
    if (s >= s1 || sl_clr_degenerate(scal))
        return;
    const float *xp = XP + (size_t)s * ldp, *yp = YP + (size_t)s * ldp, *x = Xs + (size_t)s * ldx;
    float *f = Fs + (size_t)s * Kp;
    for (int p = threadIdx.x; p < P32; p += 256)
        f[p] = p < P ? xp[p] * yp[p] : 0.f;
    for (int t = threadIdx.x; t < Kp - P32; t += 256) {
        float v = 0.f;
        if (t < J)
            v = x[t] * x[t];
        else if (t < 3 * J)
            v = x[t < 2 * J ? t - J : t - 2 * J];
        f[P32 + t] = v;
    }
}

// psi rows, nB, and the map-wide maxima.  One workgroup per node.
__global__ __launch_bounds__(256) void clr_node_feat_kernel(const float *__restrict__ map, int ldm, int ppitch, int P, int J,
                                                            float *__restrict__ Fm, int Kp, int P32, int N,
                                                            float *__restrict__ nB, float *__restrict__ a2n,
                                                            unsigned *__restrict__ scal)
{
    const int n = blockIdx.x;
    if (n >= N)
        return;
    const float *A = map + (size_t)n * ldm, *Bv = A + ppitch;
    float *f = Fm + (size_t)n * Kp;
    float sb = 0.f, amax = 0.f;
    bool inf = false, nan = false, nzero = false;
    for (int p = threadIdx.x; p < P32; p += 256) {
        const float a = p < P ? A[p] : 0.f, b = p < P ? Bv[p] : 0.f;
        f[p] = a;
        if (!(a == 0.f) || !(b == 0.f))
            nzero = true;
        const float a2 = a * a, b2 = b * b;
        sb = sb + b2;
        nan |= (a2 != a2) || (b2 != b2);
        inf |= (a2 > 3.0e38f) || (b2 > 3.0e38f);
        amax = a2 > amax ? a2 : amax;
    }
    for (int t = threadIdx.x; t < Kp - P32; t += 256) {
        float v = 0.f;
        if (t < J) {                                 // -1/2 sum_{i(p)=t} A_p^2 : pairs off(t) .. off(t)+J-2-t
            const int off = t * (2 * J - t - 1) / 2;
            float acc = 0.f;
            for (int q = 0; q < J - 1 - t; ++q) {
                const float a = A[off + q];
                acc = acc + a * a;
            }
            v = -0.5f * acc;
        } else if (t < 2 * J) {                      // -sum_{i(p)=t} A_p B_p
            const int tt = t - J, off = tt * (2 * J - tt - 1) / 2;
            float acc = 0.f;
            for (int q = 0; q < J - 1 - tt; ++q) {
                const float ab = A[off + q] * Bv[off + q];
                acc = acc + ab;
            }
            v = -acc;
        } else if (t < 3 * J) {                      // +sum_{j(p)=t} B_p : pairs (i, t), i < t, at off(i) + t-i-1
            const int tt = t - 2 * J;
            float acc = 0.f;
            for (int i = 0; i < tt; ++i)
                acc = acc + Bv[i * (2 * J - i - 1) / 2 + (tt - i - 1)];
            v = acc;
        }
        f[P32 + t] = v;
    }
    // block reductions: nB (sum), amax2 (max), flags
    __shared__ float ssum[4], smax[4];
    __shared__ int sflag[2];
    if (threadIdx.x < 2)
        sflag[threadIdx.x] = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) {
        sb = sb + __shfl_xor(sb, off);
        const float o = __shfl_xor(amax, off);
        amax = o > amax ? o : amax;
    }
    if (inf)
        atomicOr(&sflag[0], 1);
    if (nan)
        atomicOr(&sflag[1], 1);
    if (__ballot(nzero) && (threadIdx.x & 63) == 0 && scal[SLI_NONZERO] == 0u)
        atomicOr(&scal[SLI_NONZERO], 1u);        // (an all-zero A / B map: every residual is -y', node 0 stays)
    if ((threadIdx.x & 63) == 0) {
        ssum[threadIdx.x >> 6] = sb;
        smax[threadIdx.x >> 6] = amax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
        float am = smax[0];
        for (int i = 1; i < 4; ++i)
            am = smax[i] > am ? smax[i] : am;
        nB[n] = tot;
        a2n[n] = am;                                 // the node's own max A^2: the select kernel's bounds are per node
        if (sflag[0] || tot > 3.0e38f)
            atomicOr(&scal[1], 1u);                  // an inf somewhere: the bound does not apply
        else if (!sflag[1]) {                        // NaN rows are legal and excluded from every search
            atomicMax(&scal[0], __float_as_uint(tot));
            atomicMax(&scal[3], __float_as_uint(am));
            atomicMax(&scal[5], ~__float_as_uint(tot));      // the minima, inverted (the sets are reset to 0)
            atomicMax(&scal[7], ~__float_as_uint(am));
        }
    }
}

static int launch_bmu_full_shortlist_clr(vsom_ctx *c, size_t s0, size_t s1)
{
    // the select kernel keeps the sample's x' / y' rows in LDS (2 * part_pitch floats beside 8.3 KB of
    // static storage): beyond 48 KB (J > 110) the exact-order kernel searches instead
    if ((size_t)2 * c->part_pitch * sizeof(float) > 48 * 1024)
        return launch_bmu_full_exact_list(c, s0, s1, nullptr, nullptr, nullptr);
    const size_t nrows = s1 - s0;
    const uint32_t P = c->part_len, J = c->J;
    const uint32_t P32 = (P + 31) / 32 * 32, Kp = P32 + (3 * J + 31) / 32 * 32;
    const size_t ldg = ((size_t)c->N + 127) / 128 * 128;
    const size_t ntm = (((size_t)c->N + GT - 1) / GT) * 2;
    auto grow = [](void **buf, size_t *cap, size_t need_bytes) -> int {
        if (need_bytes <= *cap)
            return VSOM_OK;
        if (*buf)
            VSOM_HIP_CHECK(hipFree(*buf));
        *buf = nullptr;
        *cap = 0;
        VSOM_HIP_CHECK(hipMalloc(buf, need_bytes));
        *cap = need_bytes;
        return VSOM_OK;
    };
    int rc;
    size_t capG = c->sl_cap * sizeof(float), capT = c->sl_tmin_cap * sizeof(float), capL = c->sl_list_cap * sizeof(int);
    if ((rc = grow((void **)&c->sl_G, &capG, nrows * ldg * sizeof(float))) ||
        (rc = grow((void **)&c->sl_tmin, &capT, nrows * ntm * sizeof(float))) ||
        (rc = grow((void **)&c->sl_list, &capL, nrows * sizeof(int))) ||
        (rc = grow((void **)&c->sl_fs, &c->sl_fs_cap, c->B * (size_t)Kp * sizeof(float))) ||
        (rc = grow((void **)&c->sl_fm, &c->sl_fm_cap, (size_t)c->N * Kp * sizeof(float))))
        return rc;
    c->sl_cap = capG / sizeof(float);
    c->sl_tmin_cap = capT / sizeof(float);
    c->sl_list_cap = capL / sizeof(int);
    if (!c->sl_nrm) {
        VSOM_HIP_CHECK(hipMalloc(&c->sl_nrm, (size_t)c->N * sizeof(float)));
        VSOM_HIP_CHECK(hipMalloc(&c->sl_scal, 3 * 16384));
        VSOM_HIP_CHECK(hipMemsetAsync(c->sl_scal, 0, 3 * 16384, c->stream));
        VSOM_HIP_CHECK(hipHostMalloc(&c->sl_fb, 64));
        std::memset(c->sl_fb, 0, 64);
    }
    if (!c->sl_a2)
        VSOM_HIP_CHECK(hipMalloc(&c->sl_a2, (size_t)c->N * sizeof(float)));
    // scal: [0] max nB bits, [1] non-finite flag, [2] redo count, [3] max A^2 bits, [4] redo samples
    unsigned *scal = c->sl_scal + 4096 * c->sl_par, *scal_next = c->sl_scal + 4096 * (c->sl_par ^ 1);
    c->sl_par ^= 1;
    hipLaunchKernelGGL(clr_node_feat_kernel, dim3((unsigned)c->N), dim3(256), 0, c->stream, c->map, (int)c->pitch,
                       (int)c->part_pitch, (int)P, (int)J, c->sl_fm, (int)Kp, (int)P32, (int)c->N, c->sl_nrm, c->sl_a2, scal);
    hipLaunchKernelGGL(clr_sample_feat_kernel, dim3((unsigned)nrows), dim3(256), 0, c->stream, c->XP, c->YP, (int)c->part_pitch,
                       (int)P, c->Xs, (int)c->xpitch, (int)J, c->sl_fs, (int)Kp, (int)P32, (int)s0, (int)s1, (const unsigned *)scal);
    dim3 grid((unsigned)((c->N + GT - 1) / GT), (unsigned)((nrows + GT - 1) / GT));
    hipLaunchKernelGGL(sl_gemm_kernel, grid, dim3(256), 0, c->stream, c->sl_fs, (int)Kp, (int)s0, (int)s1, c->sl_fm, (int)Kp,
                       (int)c->N, (int)Kp, c->sl_nrm, c->sl_G, (int)ldg, c->sl_tmin, (int)ntm, (const unsigned *)nullptr,
                       (const unsigned *)scal);
    DistArgs a;
    a.xa = c->XP;
    a.xb = c->YP;
    a.ldx = (int)c->part_pitch;
    a.ma = c->map;
    a.mb = c->map + c->part_pitch;
    a.ldm = (int)c->pitch;
    a.L = (int)P;
    const double u = 5.9604644775390625e-08;   // 2^-24
    const double g1 = ((double)GK + (double)Kp / GK + 3.0) * u, gf = ((double)J + 3.0) * u;
    const double ga = 3.01 * (g1 + gf) + ((double)P / 256.0 + 12.0 + 7.0) * u;   // + the final nB - 2*dot rounding
    const double g2 = ((double)P / 8.0 + 10.0) * u, e1 = 6.1 * 1.7320508075688773 * u;
    const size_t xy_bytes = (size_t)2 * c->part_pitch * sizeof(float);   // + 8.3 KB static: fits the default 64 KB up to J = 120
    hipLaunchKernelGGL(sl_select_kernel<true>, dim3((unsigned)nrows), dim3(256), xy_bytes, c->stream, a, (int)s0, (int)s1, (int)c->N,
                       (int)c->D, c->sl_G, (int)ldg, c->sl_tmin, (int)ntm, scal, (float)(1.0001 * ga), (float)(1.0001 * g2),
                       c->lastbmu, c->sqres, scal + 2, c->sl_list, scal + 4, c->Xs, (int)c->xpitch, (int)J, (float)(1.0001 * e1), 128u,
                       (const float *)nullptr, 0u, 0.f, (const unsigned *)nullptr, (const float *)c->sl_nrm, (const float *)c->sl_a2,
                       (const float *)nullptr);
    VSOM_HIP_CHECK(hipGetLastError());
    const SlFeedback fb = {scal, c->sl_fb, (unsigned)nrows, (const unsigned *)nullptr, scal_next};
    return launch_bmu_full_exact_list(c, s0, s1, c->sl_list, scal + 2, &fb);
}
