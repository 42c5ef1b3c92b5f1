// vsom_sl_i8.hip -- the shortlist contraction of Som::findBmu (Som.cpp:291-309; vsom_shortlist.hip) on the INTEGER
// matrix pipe, for ANY finite fp32 chunk (round 5; round 4 covered uint8-valued chunks only).
//
// The matrix pipe only PRUNES: G[s][n] ~ |M_n|^2 - 2 <x_s, M_n> with a proven error bound, every returned index and
// distance comes from the exact-order evaluation of sl_select_kernel.  The approximation is computed in exact integer
// arithmetic (v_mfma_i32_32x32x32_i8 runs 32x the multiply-adds per cycle of v_mfma_f32_32x32x2_f32):
//   * model rows: M_nk = s_n (q1 + q2/128 + q3/16384)_nk + r_nk, s_n a power of two with |M_nk| / s_n < 64, digits in
//     [-64, 64], |r_nk| <= s_n 2^-15 =: eps_n (vsom_digits.hpp; sl_prepare_i8_kernel)
//   * samples, one of two kinds -- a device-side fact of the staged chunk (`xflag`, written by sl_quant_rows_kernel), so
//     nothing is decided on the host and a stream may change kind with every chunk:
//       uint8 kind   every value an integer in [0, 255] (MnistDataLoader.cpp:73-75 yields raw pixels): ONE int8 plane
//                    x - 128, exact; the offset is put back with the row sums of q.  3 integer products per element.
//       general kind any other finite data (normalised pixels, REAL columns of SqliteDataLoader.cpp:481-548): per sample
//                    the same digit grid, x_sk = t_s (p1 + p2/128 + p3/16384)_sk + rho_sk, |rho_sk| <= t_s 2^-15 =: e_s.
//                    The six digit products of weight >= 2^-14 go to three accumulator sets (weights 1, 2^-7, 2^-14);
//                    the three lighter ones (p2 q3, p3 q2, p3 q3) are DROPPED and bounded.
//   * <x^, M^> = t_s s_n 2^-14 [ 16384 A0 + 128 A1 + A2 ],  A_w exact in int32; the combination, the offset term, the
//     scales and |M|^2 - 2 <x^, M^> are evaluated in fp64 (exact up to the final rounding to fp32)
// Bound (u = 2^-24, a_s = sum_k |p2| + |p3| of the sample, K = contracted columns):
//   |<x,M> - approx| <= |<rho, M>| + |<x^, r>| + dropped
//                    <= e_s |M_n|_1 + (|x_s|_1 + K e_s) eps_n + t_s s_n 64 (2^-21 + 2^-28) a_s
//                    <= e_s L1Mmax + l1eff_s eps_max,      l1eff_s := |x_s|_1 + K e_s + 1.01 t_s a_s   (s_n <= 2^15 eps_n)
//   |G - (|M_n|^2 - 2 <x, M_n>)| <= 2 (e_s L1Mmax + l1eff_s eps_max) + 3.1 u (nMmax + |x|^2)     [fp64 epilogue; the fp32
//                                   epilogue of the uint8 kind, sl_i8_value_fast: 5.1 u (...) + 2^-7 eps_max]
// (uint8 kind: e_s = 0, l1eff_s = |x_s|_1 -- round 4's bound) which sl_select_kernel uses in place of the fp32 chain's
// 2 g1 (nMmax + |x|^2), g1 = (32 + K/32 + 3) u.
// Samples holding NaN / inf / overflowing values are redone exactly through the |x|^2 test of the select kernel.
#include "vsom_digits.hpp"
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#ifdef VSOM_DEVELOPMENT
__device__ int vsom_sl_dbg = 0;      // timing experiments (WRONG results): 1 = one K chunk only, 2 = no epilogue, 4 = values only (no stores)
#endif

// The chunk's kind word xflag[0] (0: every value an integer in [0, 255]) is written through 32 line-sized slots
// xflag[32 (1 + k)]: the thousands of wavefronts of a quantisation pass that start together all find the word clear, and
// that many writes to ONE address queue up at the memory side (4096 atomics 30 us, 4096 plain stores 80 us of a 5 us
// kernel).  The next search's prepare kernel folds the slots into word 0 and clears them (sl_kind_fold);
// stage_rows_kernel clears word 0 and the slots for a new chunk.
__device__ __forceinline__ void sl_kind_mark(unsigned *xflag, unsigned who)
{
    unsigned *slot = xflag + 32 * (1 + (who & 31u));
    if (__atomic_load_n(slot, __ATOMIC_RELAXED) == 0u)
        __atomic_store_n(slot, 1u, __ATOMIC_RELAXED);
}
// first wavefront of ONE workgroup of a kernel that runs after the quantisation and before the word's first reader
__device__ __forceinline__ void sl_kind_fold(unsigned *xflag)
{
    const int lane = threadIdx.x & 63;
    const unsigned v = lane < 32 ? xflag[32 * (1 + lane)] : 0u;
    if (__ballot(v != 0u)) {
        if (lane < 32 && v != 0u)
            xflag[32 * (1 + lane)] = 0u;
        if (lane == 0)
            xflag[0] = 1u;
    }
}

// ---- samples: both int8 images, the per-sample bound terms, the chunk's kind -----------------------------------------
// One workgroup per row.  src = the staged rows; idx = the compaction's live-column list (then dst = the row gathered onto
// them, vsom_compact.hip) or null (identity: the padded row itself).  Planes of xi: [0] x - 128 (uint8 kind), [1..3] the
// three digits (general kind), row pitch kp8 bytes, columns past the row's length hold x = 0.  lx: [0] |x|_1 of the
// uint8 image (exact), [1] t_s, [2] l1eff_s, [3] e_s (header), each `lstride` floats.  xflag |= 1 when some value of
// the chunk is not an integer in [0, 255].
__global__ __launch_bounds__(256) void sl_quant_rows_kernel(const float *__restrict__ src, int lds_, float *__restrict__ dst,
                                                            int ldd, const int *__restrict__ idx, int nrows,
                                                            signed char *__restrict__ xi, size_t xplane, int kp8,
                                                            float *__restrict__ lx, size_t lstride, unsigned *__restrict__ xflag)
{
    __shared__ float ssum[4], smax[4], sl1[4], snx[4];
    __shared__ int sa[4];
    const int row = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (row >= nrows)
        return;
    const float *s = src + (size_t)row * lds_;
    auto value = [&](int k) -> float {                   // column k of the (gathered) row; ldd is a multiple of 32
        if (k >= ldd)
            return 0.f;
        const int c = idx ? idx[k] : k;
        return c >= 0 ? s[c] : 0.f;
    };
    const int kend = xi ? kp8 : ldd;
    float sum = 0.f, mx = 0.f, l1f = 0.f, nx2 = 0.f;
    bool bad = false;
    float keep[4] = {0.f, 0.f, 0.f, 0.f};                // rows of <= 1024 values: the thread's four, kept for the digits
    for (int k4 = threadIdx.x * 4; k4 < kend; k4 += 1024) {
        const float vv[4] = {value(k4), value(k4 + 1), value(k4 + 2), value(k4 + 3)};
#pragma unroll
        for (int u = 0; u < 4; ++u)
            keep[u] = vv[u];
        if (dst && k4 < ldd)
            *reinterpret_cast<float4 *>(dst + (size_t)row * ldd + k4) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        if (xi) {
            char4 q;
            signed char *qq = reinterpret_cast<signed char *>(&q);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float f = vv[u];
                const bool ok = f >= 0.f && f <= 255.f && f == rintf(f);      // NaN fails the comparisons
                bad |= !ok;
                const int iv = ok ? (int)f : 0;
                sum += (float)iv;                        // exact: integers, total < 2^24 for rows up to 65536 values
                qq[u] = (signed char)(iv - 128);
                const float af = fabsf(f);
                const bool fin = af <= 3.0e38f;
                mx = (fin && af > mx) ? af : mx;
                l1f += fin ? af : 0.f;
                nx2 += f * f;
            }
            *reinterpret_cast<char4 *>(xi + (size_t)row * kp8 + k4) = q;
        }
    }
    if (!xi)
        return;
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off);
        l1f += __shfl_xor(l1f, off);
        nx2 += __shfl_xor(nx2, off);
        const float o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
    }
    // (one atomic per chunk, not per wavefront: 16384 same-address atomics take 0.13 ms; a stale read only repeats it)
    if (__ballot(bad) && lane == 0)
        sl_kind_mark(xflag, blockIdx.x);
    if (lane == 0) {
        ssum[wave] = sum;
        smax[wave] = mx;
        sl1[wave] = l1f;
        snx[wave] = nx2;
    }
    __syncthreads();
    mx = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    float t1, it1, es;
    sl_row_scale(mx, t1, it1, es);
    int asum = 0;
    for (int k4 = threadIdx.x * 4; k4 < kp8; k4 += 1024) {
        char4 o1, o2, o3;
        signed char *p1 = reinterpret_cast<signed char *>(&o1), *p2 = reinterpret_cast<signed char *>(&o2),
                    *p3 = reinterpret_cast<signed char *>(&o3);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int a, b, c3;
            sl_digits3(kp8 <= 1024 ? keep[u] : value(k4 + u), t1, it1, a, b, c3);
            p1[u] = (signed char)a;
            p2[u] = (signed char)b;
            p3[u] = (signed char)c3;
            asum += (b < 0 ? -b : b) + (c3 < 0 ? -c3 : c3);
        }
        signed char *o = xi + xplane + (size_t)row * kp8 + k4;
        *reinterpret_cast<char4 *>(o) = o1;
        *reinterpret_cast<char4 *>(o + xplane) = o2;
        *reinterpret_cast<char4 *>(o + 2 * xplane) = o3;
    }
    for (int off = 32; off > 0; off >>= 1)
        asum += __shfl_xor(asum, off);
    if (lane == 0)
        sa[wave] = asum;
    __syncthreads();
    if (threadIdx.x == 0) {
        lx[row] = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);                 // exact: integers below 2^24
        lx[lstride + row] = t1;
        const float l1 = ((sl1[0] + sl1[1]) + (sl1[2] + sl1[3])) * 1.001f;   // fp32 sum of <= 4096 magnitudes, rounded up
        const float at = (float)(sa[0] + sa[1] + sa[2] + sa[3]);             // exact: < 2^24
        lx[2 * lstride + row] = (l1 + (float)kp8 * es + 1.01f * t1 * at) * 1.001f;
        lx[3 * lstride + row] = es;
        lx[4 * lstride + row] = (snx[0] + snx[1]) + (snx[2] + snx[3]);      // |x|^2 (any order: it only enters bounds)
    }
}

// The same for rows of at most 64 columns (kp8 == 64): 16 lanes per row, four columns each, no LDS, no barrier (a
// workgroup per row spends its time in the two barriers: 50 us for C4's 16384 rows of 32 values against 5 here).
__global__ __launch_bounds__(256) void sl_quant_rows64_kernel(const float *__restrict__ src, int lds_, float *__restrict__ dst,
                                                              int ldd, const int *__restrict__ idx, int nrows,
                                                              signed char *__restrict__ xi, size_t xplane,
                                                              float *__restrict__ lx, size_t lstride, unsigned *__restrict__ xflag)
{
    const int row = blockIdx.x * 16 + ((int)threadIdx.x >> 4), k4 = ((int)threadIdx.x & 15) * 4;
    if (row >= nrows)
        return;   // whole 16-lane groups leave; the exchanges below stay inside a group
    const float *s = src + (size_t)row * lds_;
    float vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = k4 + u;
        const int c = k < ldd ? (idx ? idx[k] : k) : -1;
        vv[u] = c >= 0 ? s[c] : 0.f;
    }
    if (dst && k4 < ldd)
        *reinterpret_cast<float4 *>(dst + (size_t)row * ldd + k4) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    if (!xi)
        return;
    float sum = 0.f, mx = 0.f, l1f = 0.f;
    bool bad = false;
    char4 q;
    signed char *qq = reinterpret_cast<signed char *>(&q);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float f = vv[u];
        const bool ok = f >= 0.f && f <= 255.f && f == rintf(f);
        bad |= !ok;
        const int iv = ok ? (int)f : 0;
        sum += (float)iv;
        qq[u] = (signed char)(iv - 128);
        const float af = fabsf(f);
        const bool fin = af <= 3.0e38f;
        mx = (fin && af > mx) ? af : mx;
        l1f += fin ? af : 0.f;
    }
    *reinterpret_cast<char4 *>(xi + (size_t)row * 64 + k4) = q;
    float nx2 = (vv[0] * vv[0] + vv[1] * vv[1]) + (vv[2] * vv[2] + vv[3] * vv[3]);
    for (int off = 8; off > 0; off >>= 1) {
        nx2 += __shfl_xor(nx2, off);
        sum += __shfl_xor(sum, off);
        l1f += __shfl_xor(l1f, off);
        const float o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0)
        sl_kind_mark(xflag, blockIdx.x);
    float t1, it1, es;
    sl_row_scale(mx, t1, it1, es);
    char4 o1, o2, o3;
    signed char *p1 = reinterpret_cast<signed char *>(&o1), *p2 = reinterpret_cast<signed char *>(&o2),
                *p3 = reinterpret_cast<signed char *>(&o3);
    int asum = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int a, b, c3;
        sl_digits3(vv[u], t1, it1, a, b, c3);
        p1[u] = (signed char)a;
        p2[u] = (signed char)b;
        p3[u] = (signed char)c3;
        asum += (b < 0 ? -b : b) + (c3 < 0 ? -c3 : c3);
    }
    signed char *o = xi + xplane + (size_t)row * 64 + k4;
    *reinterpret_cast<char4 *>(o) = o1;
    *reinterpret_cast<char4 *>(o + xplane) = o2;
    *reinterpret_cast<char4 *>(o + 2 * xplane) = o3;
    for (int off = 8; off > 0; off >>= 1)
        asum += __shfl_xor(asum, off);
    if ((threadIdx.x & 15) == 0) {
        lx[row] = sum;
        lx[lstride + row] = t1;
        lx[2 * lstride + row] = (l1f * 1.001f + 64.f * es + 1.01f * t1 * (float)asum) * 1.001f;
        lx[3 * lstride + row] = es;
        lx[4 * lstride + row] = nx2;                     // |x|^2 (any order: it only enters bounds), NaN / inf if x has one
    }
}

// ---- model rows: |M|^2 (fp64 sum), live columns gathered, three 7-bit digits, row sums -------------------------------
// one WAVEFRONT per node (4 per workgroup), no LDS, no barriers.  idx = live-column list of the compaction (null:
// identity), kp = contraction length (device value kp_dev[2] when compacted).  q planes: [3][N][kp8].
__global__ __launch_bounds__(256) void sl_prepare_i8_kernel(const float *__restrict__ map, int ldm, int Dp, int N,
                                                            const int *__restrict__ idx, int kp, const unsigned *__restrict__ kp_dev,
                                                            int kp8, signed char *__restrict__ q, float *__restrict__ nrm,
                                                            double *__restrict__ qscale, double *__restrict__ qcorr,
                                                            int4 *__restrict__ qfast, unsigned *__restrict__ scal,
                                                            unsigned *__restrict__ xflag)
{
    __shared__ __attribute__((aligned(16))) float s_row[4][1024];      // the rows of the four wavefronts (rows of <= 1024 values)
    const int n_own = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x < 64)
        sl_kind_fold(xflag);
    const bool small = kp8 <= 1024 && Dp <= 1024;        // workgroup-uniform
    // wavefronts past the last node: the `small` path has a workgroup barrier, which every wavefront must reach -- they
    // load the last node's row and leave right behind the barrier, before anything is written
    const bool live = n_own < N;
    const int n = live ? n_own : N - 1;
    if (!small && !live)
        return;
    const int kalloc = kp;                               // entries of idx (those past the live columns hold -1)
    if (kp_dev)
        kp = (int)kp_dev[2];
    const float *src = map + (size_t)n * ldm;
    double ss = 0.0;
    bool nz = false;
    float mx = 0.f, l1 = 0.f, s1, is1, eps, nf;
    int r1 = 0, r2 = 0, r3 = 0;
    const size_t plane = (size_t)N * kp8;
    signed char *q1 = q + (size_t)n * kp8, *q2 = q1 + plane, *q3 = q2 + plane;
    if (small) {
        // Rows of up to 1024 values: ONE round trip to memory per row -- the row itself (4 x 16 bytes per lane, coalesced)
        // and the live-column list (the same) -- then the live values are gathered from the row's copy in LDS, 16 per
        // lane, and kept for both the scale and the digits.  (Rolled loops over the row, the list and the list again took
        // one or two round trips per iteration: 30 per row, 50 us for C3's 16384 rows.  Gathering the 16 values per lane
        // from global memory instead: 120-158 us -- each instruction touches 64 separate 4-byte pieces.)
        float *row = s_row[threadIdx.x >> 6];
        float4 rv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = lane * 4 + 256 * j;
            rv[j] = d < Dp ? *reinterpret_cast<const float4 *>(src + d) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        int ci[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k4 = lane * 4 + 256 * j;
            int4 t = make_int4(k4, k4 + 1, k4 + 2, k4 + 3);
            if (idx)
                t = k4 < kalloc ? *reinterpret_cast<const int4 *>(idx + k4) : make_int4(-1, -1, -1, -1);
            const int tt[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int u = 0; u < 4; ++u)
                ci[j][u] = (k4 + u < kp && tt[u] >= 0 && tt[u] < Dp) ? tt[u] : -1;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<float4 *>(row + lane * 4 + 256 * j) = rv[j];
            ss += (double)rv[j].x * (double)rv[j].x + (double)rv[j].y * (double)rv[j].y;
            ss += (double)rv[j].z * (double)rv[j].z + (double)rv[j].w * (double)rv[j].w;
            nz |= !(rv[j].x == 0.f) || !(rv[j].y == 0.f) || !(rv[j].z == 0.f) || !(rv[j].w == 0.f);
        }
        __syncthreads();
        if (!live)
            return;                                      // (no barrier follows)
        float gv[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                gv[j][u] = ci[j][u] >= 0 ? row[ci[j][u]] : 0.f;
        for (int off = 32; off > 0; off >>= 1)
            ss += __shfl_xor(ss, off);
        nf = (float)ss;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v = fabsf(gv[j][u]);
                const bool fin = v <= 3.0e38f;
                mx = (fin && v > mx) ? v : mx;
                l1 += fin ? v : 0.f;
            }
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(mx, off);
            mx = o > mx ? o : mx;
            l1 += __shfl_xor(l1, off);
        }
        sl_row_scale(mx, s1, is1, eps);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k4 = lane * 4 + 256 * j;
            if (k4 < kp8) {
                char4 o1, o2, o3;
                signed char *p1 = reinterpret_cast<signed char *>(&o1), *p2 = reinterpret_cast<signed char *>(&o2),
                            *p3 = reinterpret_cast<signed char *>(&o3);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int a, b, c3;
                    sl_digits3(gv[j][u], s1, is1, a, b, c3);     // columns past the live ones: 0 -> digits 0
                    p1[u] = (signed char)a;
                    p2[u] = (signed char)b;
                    p3[u] = (signed char)c3;
                    r1 += a;
                    r2 += b;
                    r3 += c3;
                }
                *reinterpret_cast<char4 *>(q1 + k4) = o1;
                *reinterpret_cast<char4 *>(q2 + k4) = o2;
                *reinterpret_cast<char4 *>(q3 + k4) = o3;
            }
        }
    } else {
    for (int d = lane * 4; d < Dp; d += 256) {          // rows are zero padded to Dp, a multiple of 32
        const float4 v = *reinterpret_cast<const float4 *>(src + d);
        ss += (double)v.x * (double)v.x + (double)v.y * (double)v.y;
        ss += (double)v.z * (double)v.z + (double)v.w * (double)v.w;
        nz |= !(v.x == 0.f) || !(v.y == 0.f) || !(v.z == 0.f) || !(v.w == 0.f);
    }
    for (int off = 32; off > 0; off >>= 1)
        ss += __shfl_xor(ss, off);
    nf = (float)ss;                                      // NaN rows stay NaN, overflow -> inf
    // the row's largest live magnitude and |M_n|_1 over the live columns (non-finite values count as 0: such a row is
    // excluded / redone anyway)
    for (int k = lane; k < kp; k += 64) {
        const int c = idx ? idx[k] : k;
        float v = c >= 0 && c < Dp ? src[c] : 0.f;
        v = fabsf(v);
        const bool fin = v <= 3.0e38f;
        mx = (fin && v > mx) ? v : mx;
        l1 += fin ? v : 0.f;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
        l1 += __shfl_xor(l1, off);
    }
    sl_row_scale(mx, s1, is1, eps);
    for (int k4 = lane * 4; k4 < kp8; k4 += 256) {      // four columns per lane and step: 4-byte stores
        char4 o1, o2, o3;
        signed char *p1 = reinterpret_cast<signed char *>(&o1), *p2 = reinterpret_cast<signed char *>(&o2),
                    *p3 = reinterpret_cast<signed char *>(&o3);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k4 + u;
            int a = 0, b = 0, c3 = 0;
            if (k < kp) {
                const int c = idx ? idx[k] : k;
                sl_digits3(c >= 0 && c < Dp ? src[c] : 0.f, s1, is1, a, b, c3);
            }
            p1[u] = (signed char)a;
            p2[u] = (signed char)b;
            p3[u] = (signed char)c3;
            r1 += a;
            r2 += b;
            r3 += c3;
        }
        *reinterpret_cast<char4 *>(q1 + k4) = o1;
        *reinterpret_cast<char4 *>(q2 + k4) = o2;
        *reinterpret_cast<char4 *>(q3 + k4) = o3;
    }
    }
    for (int off = 32; off > 0; off >>= 1) {
        r1 += __shfl_xor(r1, off);
        r2 += __shfl_xor(r2, off);
        r3 += __shfl_xor(r3, off);
    }
    if (__ballot(nz) && lane == 0 && scal[SLI_NONZERO] == 0u)
        atomicOr(&scal[SLI_NONZERO], 1u);                // (an all-zero map -- what an empty chunk leaves -- skips the refinement)
    if (lane == 0) {
        nrm[n] = nf;
        qscale[n] = (double)s1 * 0x1.0p-14;              // s 2^-14: multiplies 16384 A0 + 128 A1 + A2
        const long long rr = (long long)r1 * 16384 + (long long)r2 * 128 + (long long)r3;
        qcorr[n] = 128.0 * (double)rr;                   // the (x - 128) offset put back
        // sl_i8_value_fast: cr = 128 rr = 16384 (rr >> 7) + ((rr & 127) << 7)
        qfast[n] = make_int4((int)__float_as_uint(2.f * s1), (int)__float_as_uint(s1 * 0x1.0p-13f), (int)(rr >> 7), (int)((rr & 127) << 7));
        const int slot = n & 31;
        if (nf == nf) {
            if (nf > 3.0e38f)
                atomicOr(&scal[1], 1u);                  // an inf somewhere: the bound does not apply
            else
                atomicMax(&scal[SLI_NMAX(slot)], __float_as_uint(nf));
        }
        atomicMax(&scal[SLI_EMAX(slot)], __float_as_uint(eps));
        atomicMax(&scal[SLI_L1MAX(slot)], __float_as_uint(l1 * 1.001f));     // fp32 sum of <= 4096 magnitudes, rounded up
    }
}

// The same for rows of at most 64 columns (kp8 == 64, Dp <= 64): 16 lanes per node, one atomic set per wavefront
__global__ __launch_bounds__(256) void sl_prepare64_kernel(const float *__restrict__ map, int ldm, int Dp, int N,
                                                           const int *__restrict__ idx, int kp, const unsigned *__restrict__ kp_dev,
                                                           signed char *__restrict__ q, float *__restrict__ nrm,
                                                           double *__restrict__ qscale, double *__restrict__ qcorr,
                                                           int4 *__restrict__ qfast, unsigned *__restrict__ scal,
                                                           unsigned *__restrict__ xflag)
{
    const int n0 = blockIdx.x * 16 + ((int)threadIdx.x >> 4), k4 = ((int)threadIdx.x & 15) * 4;
    if (blockIdx.x == 0 && threadIdx.x < 64)
        sl_kind_fold(xflag);
    const bool ok = n0 < N;
    const int n = ok ? n0 : N - 1;                       // (no early exit: the wavefront's maxima are exchanged below)
    if (kp_dev)
        kp = (int)kp_dev[2];
    const float *src = map + (size_t)n * ldm;
    double ss = 0.0;
    bool nz = false;
    if (k4 < Dp) {
        const float4 v = *reinterpret_cast<const float4 *>(src + k4);
        ss = (double)v.x * (double)v.x + (double)v.y * (double)v.y;
        ss += (double)v.z * (double)v.z + (double)v.w * (double)v.w;
        nz = !(v.x == 0.f) || !(v.y == 0.f) || !(v.z == 0.f) || !(v.w == 0.f);
    }
    float vv[4], mx = 0.f, l1 = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = k4 + u;
        const int c = k < kp ? (idx ? idx[k] : k) : -1;
        vv[u] = c >= 0 && c < Dp ? src[c] : 0.f;
        const float v = fabsf(vv[u]);
        const bool fin = v <= 3.0e38f;
        mx = (fin && v > mx) ? v : mx;
        l1 += fin ? v : 0.f;
    }
    for (int off = 8; off > 0; off >>= 1) {
        ss += __shfl_xor(ss, off);
        const float o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
        l1 += __shfl_xor(l1, off);
    }
    const float nf = (float)ss;
    float s1, is1, eps;
    sl_row_scale(mx, s1, is1, eps);
    int r1 = 0, r2 = 0, r3 = 0;
    char4 o1, o2, o3;
    signed char *p1 = reinterpret_cast<signed char *>(&o1), *p2 = reinterpret_cast<signed char *>(&o2),
                *p3 = reinterpret_cast<signed char *>(&o3);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int a, b, c3;
        sl_digits3(vv[u], s1, is1, a, b, c3);            // columns past kp: 0 -> digits 0
        p1[u] = (signed char)a;
        p2[u] = (signed char)b;
        p3[u] = (signed char)c3;
        r1 += a;
        r2 += b;
        r3 += c3;
    }
    const size_t plane = (size_t)N * 64;
    if (ok) {
        signed char *q1 = q + (size_t)n * 64 + k4;
        *reinterpret_cast<char4 *>(q1) = o1;
        *reinterpret_cast<char4 *>(q1 + plane) = o2;
        *reinterpret_cast<char4 *>(q1 + 2 * plane) = o3;
    }
    for (int off = 8; off > 0; off >>= 1) {
        r1 += __shfl_xor(r1, off);
        r2 += __shfl_xor(r2, off);
        r3 += __shfl_xor(r3, off);
    }
    if (ok && (threadIdx.x & 15) == 0) {
        nrm[n] = nf;
        qscale[n] = (double)s1 * 0x1.0p-14;
        const long long rr = (long long)r1 * 16384 + (long long)r2 * 128 + (long long)r3;
        qcorr[n] = 128.0 * (double)rr;
        qfast[n] = make_int4((int)__float_as_uint(2.f * s1), (int)__float_as_uint(s1 * 0x1.0p-13f), (int)(rr >> 7), (int)((rr & 127) << 7));
    }
    // the wavefront's four nodes: one atomic per quantity (rows past N repeat node N - 1)
    unsigned nb = (nf == nf && nf <= 3.0e38f) ? __float_as_uint(nf) : 0u, eb = __float_as_uint(eps), lb = __float_as_uint(l1 * 1.001f);
    const bool isinf = nf == nf && nf > 3.0e38f;
    for (int off = 16; off <= 32; off <<= 1) {
        const unsigned a1 = (unsigned)__shfl_xor((int)nb, off), a2 = (unsigned)__shfl_xor((int)eb, off), a3 = (unsigned)__shfl_xor((int)lb, off);
        nb = a1 > nb ? a1 : nb;
        eb = a2 > eb ? a2 : eb;
        lb = a3 > lb ? a3 : lb;
    }
    const bool anynz = __ballot(nz) != 0ull, anyinf = __ballot(isinf) != 0ull;
    if ((threadIdx.x & 63) == 0) {
        const int slot = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)) & 31;
        if (anynz && scal[SLI_NONZERO] == 0u)
            atomicOr(&scal[SLI_NONZERO], 1u);
        if (anyinf)
            atomicOr(&scal[1], 1u);
        atomicMax(&scal[SLI_NMAX(slot)], nb);
        atomicMax(&scal[SLI_EMAX(slot)], eb);
        atomicMax(&scal[SLI_L1MAX(slot)], lb);
    }
}

// The maxima of |M|^2, eps and |M|_1 over the 32 slots the prepare kernel filled -> words 8..10 of the counter set: the
// first wavefront of ONE workgroup of the contraction kernel does it, the refinement kernels (one workgroup or wavefront
// per sample) then read three words instead of reducing 96 each.
__device__ __forceinline__ void sl_fold_maxima(unsigned *__restrict__ scal)
{
    const int lane = threadIdx.x & 63;
    unsigned nb = scal[SLI_NMAX(lane & 31)], eb = scal[SLI_EMAX(lane & 31)], lb = scal[SLI_L1MAX(lane & 31)];
    for (int off = 16; off > 0; off >>= 1) {
        const unsigned o1 = (unsigned)__shfl_xor((int)nb, off), o2 = (unsigned)__shfl_xor((int)eb, off),
                       o3 = (unsigned)__shfl_xor((int)lb, off);
        nb = o1 > nb ? o1 : nb;
        eb = o2 > eb ? o2 : eb;
        lb = o3 > lb ? o3 : lb;
    }
    if (lane == 0) {
        scal[8] = nb;
        scal[9] = eb;
        scal[10] = lb;
    }
}

// ---- G = |M|^2 - 2 <x^, M^>, tile minima ----------------------------------------------------------------------------
// XD = sample planes (1: uint8 kind, 3: general kind), chosen at run time from the chunk's flag: the kernels below hold
// both bodies.  Accumulator set w collects the digit products of weight 128^-w: (sample plane pl) x (model plane l),
// pl + l = w <= 2.

// epilogue value (C/D layout of the 32 x 32 MFMA tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)),
// in fp64: exact up to the final rounding to fp32
__device__ __forceinline__ float sl_i8_value(int a0, int a1, int a2, double cr, double sc, double nm)
{
    const double t = (double)a0 * 16384.0 + (double)a1 * 128.0 + (double)a2 + cr;
    return (float)(nm - 2.0 * (sc * t));
}

// The same value for the uint8 kind WITHOUT fp64 (the epilogue was half of the contraction kernel's time: 67 M elements x
// 3 int -> fp64 conversions, 5 fp64 operations and a conversion back, at a half to a sixteenth of the fp32 rate).
// T = 16384 a0 + 128 a1 + a2 + cr is an integer below 2^47; with u = 128 a1 + a2 (int32, exact) it is split as
// T = 16384 A + Bq,  A = a0 + (u >> 14) + (cr >> 14),  Bq = (u & 16383) + (cr & 16383)  -- |A| < 2^24 for K <= 960 contracted
// columns, 0 <= Bq < 2^15: both exact as fp32 -- and the scale 2 sc = s 2^-13 is a power of two, so
//   g = fma(-s 2^-13, Bq, fma(-2 s, A, |M|^2))
// has TWO roundings where the fp64 form has one: |g - exact| <= 2u |G| + 2^-7 eps_n, which the select kernel's bound
// carries as c_g1 = 5.5u (instead of 3.3u) and a 0.01 eps_max term.  f = {bits of 2 s, bits of s 2^-13, cr >> 14, cr & 16383}.
__device__ __forceinline__ float sl_i8_value_fast(int a0, int a1, int a2, int4 f, float nm)
{
    const int u = (a1 << 7) + a2;
    const int A = a0 + (u >> 14) + f.z;                  // arithmetic shift: floor
    const int Bq = (u & 16383) + f.w;
    const float g1 = fmaf(-__int_as_float(f.x), (float)A, nm);
    return fmaf(-__int_as_float(f.y), (float)Bq, g1);
}

// The general kind without fp64: no offset, and the scale t_s s_n 2^-14 is a product of two powers of two >= 2^-50 each
// (vsom_digits.hpp) -- exact in fp32.  Same split of T as above (|u| = |128 a1 + a2| < 2^31 and |A| < 2^23 for K <= 960), two
// roundings; the second one's extra term u |t s 2^-13 Bq| <= 2^-8 t_s eps_n is carried by the select kernel's bound.
__device__ __forceinline__ float sl_i8_value_fast3(int a0, int a1, int a2, int4 f, float ts, float nm)
{
    const int u = (a1 << 7) + a2;
    const int A = a0 + (u >> 14);
    const int Bq = u & 16383;
    const float g1 = fmaf(-(ts * __int_as_float(f.x)), (float)A, nm);
    return fmaf(-(ts * __int_as_float(f.y)), (float)Bq, g1);
}

// workgroup tile (64 MI) samples x 64 nodes, wavefront tile (32 MI) x 32 (MI MFMA tiles of 32 x 32), K streamed through
// LDS in chunks of 64 bytes with the next chunk's global loads in flight.  For problems too small to fill the chip with
// the ring kernel's 256 x 128 tiles.
#define IT_N 64
#define IK 64
#define ILD 80      // LDS row stride in bytes (64 + 16: conflict-free 16-byte fragment reads)
template <int MI, int XD>
__device__ __forceinline__ void sl_gemm_i8_body(const signed char *__restrict__ xi, size_t xplane, int s0, int s1,
                                                const signed char *__restrict__ q, int N, int kp, int kp8,
                                                const float *__restrict__ nrm, const double *__restrict__ qscale,
                                                const double *__restrict__ qcorr, const float *__restrict__ xscale,
                                                float *__restrict__ G, int ldg, float *__restrict__ tmin, int ntm,
                                                signed char *As, signed char *Bs, float *smin, const int4 *__restrict__ qfast)
{
    const int k64 = (kp + IK - 1) / IK * IK;             // <= kp8; columns past kp hold q = 0
    constexpr int IT_S = 64 * MI;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int sbase = s0 + blockIdx.y * IT_S, nbase = blockIdx.x * IT_N;
    const size_t plane = (size_t)N * kp8;
    const signed char *xbase = XD == 1 ? xi : xi + xplane;        // plane 0: uint8 image; planes 1..3: digits

    v16i acc[3][MI];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[l][i][r] = 0;

    // staging: A planes XD x (64 MI) rows x 64 B (XD MI 16-byte pieces per thread); B tiles 3 x 64 rows x 64 B (3)
    v4i pa[XD][MI], pb[3];
    auto gload = [&](int k0) {
#pragma unroll
        for (int pl = 0; pl < XD; ++pl)
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int f = tid + 256 * i, r = f >> 2, c = (f & 3) * 16;
                const int s = sbase + r;
                pa[pl][i] = s < s1 ? *reinterpret_cast<const v4i *>(xbase + pl * xplane + (size_t)s * kp8 + k0 + c) : v4i{0, 0, 0, 0};
            }
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const int r = tid >> 2, c = (tid & 3) * 16;
            const int n = nbase + r;
            pb[l] = n < N ? *reinterpret_cast<const v4i *>(q + l * plane + (size_t)n * kp8 + k0 + c) : v4i{0, 0, 0, 0};
        }
    };
    gload(0);
    for (int k0 = 0; k0 < k64; k0 += IK) {
        __syncthreads();
#pragma unroll
        for (int pl = 0; pl < XD; ++pl)
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int f = tid + 256 * i, r = f >> 2, c = (f & 3) * 16;
                *reinterpret_cast<v4i *>(&As[(pl * IT_S + r) * ILD + c]) = pa[pl][i];
            }
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const int r = tid >> 2, c = (tid & 3) * 16;
            *reinterpret_cast<v4i *>(&Bs[(l * IT_N + r) * ILD + c]) = pb[l];
        }
        __syncthreads();
        if (k0 + IK < k64)
            gload(k0 + IK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v4i a[XD][MI], b[3];
#pragma unroll
            for (int pl = 0; pl < XD; ++pl)
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    a[pl][i] = *reinterpret_cast<const v4i *>(&As[(pl * IT_S + wm * 32 * MI + i * 32 + lr) * ILD + ks * 32 + 16 * lh]);
#pragma unroll
            for (int l = 0; l < 3; ++l)
                b[l] = *reinterpret_cast<const v4i *>(&Bs[(l * IT_N + wn * 32 + lr) * ILD + ks * 32 + 16 * lh]);
#pragma unroll
            for (int pl = 0; pl < XD; ++pl)
#pragma unroll
                for (int l = 0; l + pl < 3; ++l)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        acc[pl + l][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[pl][i], b[l], acc[pl + l][i], 0, 0, 0);
        }
    }
    // epilogue in three phases (constants, values, stores: see sl_gemm_i8_ring_body)
    const float inf = __uint_as_float(0x7F800000u);
    const int col = nbase + wn * 32 + lr;
    const bool cok = col < N;
    const int cc = cok ? col : N - 1;
    const bool fast = qfast != nullptr;                  // workgroup-uniform (K <= 960 contracted columns)
    const float nmf = nrm[cc];
    const double nm = (double)nmf, sc = qscale[cc], cr = XD == 1 ? qcorr[cc] : 0.0;
    const int4 ff = fast ? qfast[cc] : make_int4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        float rsf[16];
        if (XD != 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = sbase + wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                rsf[r] = xscale[row < s1 ? row : s1 - 1];
            }
        }
        if (fast) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[0][i][r] = __float_as_int(XD == 1 ? sl_i8_value_fast(acc[0][i][r], acc[1][i][r], acc[2][i][r], ff, nmf)
                                                      : sl_i8_value_fast3(acc[0][i][r], acc[1][i][r], acc[2][i][r], ff, rsf[r], nmf));
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[0][i][r] = __float_as_int(sl_i8_value(acc[0][i][r], acc[1][i][r], acc[2][i][r], cr,
                                                          XD == 1 ? sc : sc * (double)rsf[r], nm));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lrow = wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float g = __int_as_float(acc[0][i][r]);
            // exact minimum of the finite entries of this row over the workgroup's 64 nodes: NaN -> +inf
            float mn = (cok && g == g) ? g : inf;
            mn = sl_min32_dpp(mn);                       // the 32 lanes that share this row; valid in lanes 31 / 63
            if (lr == 31)
                smin[wn * IT_S + lrow] = mn;
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = sbase + wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < s1 && cok)
                G[(size_t)(row - s0) * ldg + col] = __int_as_float(acc[0][i][r]);
        }
    }
    __syncthreads();
    if (tid < IT_S) {
        const int row = sbase + tid;
        if (row < s1) {
            const float a0 = smin[tid], a1 = smin[IT_S + tid];
            tmin[(size_t)(row - s0) * ntm + blockIdx.x] = a1 < a0 ? a1 : a0;
        }
    }
}

// one kernel per kind (the general body needs the registers of two workgroups per CU, the uint8 body runs three): both
// are launched, the one whose kind the chunk is not exits at once
template <int MI, int XD>
__global__ __launch_bounds__(256, XD == 1 ? 3 : 2) void sl_gemm_i8_kernel(const signed char *__restrict__ xi, size_t xplane, int s0, int s1,
                                                            const signed char *__restrict__ q, int N, int kp,
                                                            const unsigned *__restrict__ kp_dev, int kp8,
                                                            const float *__restrict__ nrm, const double *__restrict__ qscale,
                                                            const double *__restrict__ qcorr, const float *__restrict__ xscale,
                                                            float *__restrict__ G, int ldg, float *__restrict__ tmin, int ntm,
                                                            const unsigned *__restrict__ xflag, const int4 *__restrict__ qfast,
                                                            unsigned *__restrict__ scal)
{
    if ((xflag[0] != 0u) != (XD == 3))   // wavefront-uniform (a scalar load)
        return;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64)
        sl_fold_maxima(scal);
    if (kp_dev)
        kp = (int)kp_dev[2];
    __shared__ __attribute__((aligned(16))) signed char As[XD * 64 * MI * ILD];
    __shared__ __attribute__((aligned(16))) signed char Bs[3 * IT_N * ILD];
    __shared__ float smin[2 * 64 * MI];
    sl_gemm_i8_body<MI, XD>(xi, xplane, s0, s1, q, N, kp, kp8, nrm, qscale, qcorr, xscale, G, ldg, tmin, ntm, As, Bs, smin, qfast);
}

// ---- the same contraction with big tiles and a ring of LDS stages filled by LDS-DMA ---------------------------------------
// What the register-staged kernel above waits for is its staging loads.  Here: workgroup tile 256 samples x 128 nodes,
// 8 wavefronts of 64 x 64 (2 x 2 MFMA tiles x three accumulator sets = 192 accumulator registers), K in chunks of 64
// bytes through a ring of LDS stages that global_load_lds_dwordx4 fills without passing through registers, one barrier
// per chunk.  uint8 kind: stage = A 16 KB | q1 | q2 | q3 8 KB each = 40 KB, ring of THREE (two chunks in flight while one
// is consumed), 12 MFMAs per wavefront and half chunk.  General kind: stage = three sample planes + three model planes =
// 72 KB, ring of TWO (24 MFMAs per half chunk cover the one chunk in flight).
// A wave-instruction deposits 1 KB contiguously (16 rows x 64 B); the 16-byte pieces of a row are stored XOR-swizzled
// (slot = piece ^ ((row >> 2) & 3), applied on the GLOBAL side: each lane picks the piece that belongs into its slot),
// which makes the 16-byte fragment reads of 32 consecutive rows conflict-free without row padding.
#define RT_S 256
#define RT_N 128
#define RING_BYTES 147456      // max(3 x 40960, 2 x 73728)
template <int XD, int WM, int NS, bool FAST, bool GLESS = false>
__device__ __forceinline__ void sl_gemm_i8_ring_body(const signed char *__restrict__ xi, size_t xplane, int s0, int s1,
                                                     const signed char *__restrict__ q, int N, int kp, int kp8,
                                                     const float *__restrict__ nrm, const double *__restrict__ qscale,
                                                     const double *__restrict__ qcorr, const float *__restrict__ xscale,
                                                     float *__restrict__ G, int ldg, float *__restrict__ tmin, int ntm,
                                                     signed char *ring, const int4 *__restrict__ qfast)
{
    // WM = wavefront rows of the tile (64 samples each; 2 wavefront columns of 64 nodes), NS = ring depth
    constexpr int AUP = 4 * WM;                          // 1 KB units (16 rows each) of ONE sample plane per stage
    constexpr int AU = AUP * XD;                         // ... of all sample planes
    constexpr int UNITS = AU + 24;                       // + 3 model planes x 8 units
    constexpr int UPW = UNITS / (2 * WM);                // per wavefront: 5 / 9 (WM = 4), 8 (WM = 2, one sample plane)
    constexpr int STAGE = UNITS * 1024, QOFF = AU * 1024, RTS = 64 * WM;
    static_assert(UNITS % (2 * WM) == 0 && NS * STAGE <= RING_BYTES, "ring layout");
    const int nchunks = (kp + IK - 1) / IK;             // columns past kp hold q = 0
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int sbase = s0 + blockIdx.y * RTS, nbase = blockIdx.x * RT_N;
    const size_t plane = (size_t)N * kp8;
    const signed char *xbase = XD == 1 ? xi : xi + xplane;

    // per lane: 32-bit offsets from a wavefront-uniform base (unit ids are per wavefront), so that the loads take the
    // scalar-base + vector-offset form and the addresses cost one register each
    unsigned soff[UPW];
    int dst[UPW];
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        const int id = wave * UPW + i;
        const int rin = lane >> 2, slot = lane & 3;
        if (id < AU) {
            const int pl = id / AUP, row = (id % AUP) * 16 + rin;
            int sr = sbase + row;
            sr = sr < s1 ? sr : s1 - 1;                  // rows past the chunk: re-read the last one (never stored)
            soff[i] = (unsigned)(pl * xplane + (size_t)sr * kp8 + ((slot ^ ((row >> 2) & 3)) << 4));
            dst[i] = id * 1024;
        } else {
            const int pl = (id - AU) >> 3, u = (id - AU) & 7;
            const int row = u * 16 + rin;
            int n = nbase + row;
            n = n < N ? n : N - 1;
            soff[i] = (unsigned)(pl * plane + (size_t)n * kp8 + ((slot ^ ((row >> 2) & 3)) << 4));
            dst[i] = QOFF + pl * 8192 + u * 1024;
        }
    }
    auto issue = [&](int c) {                            // chunk c -> stage c % NS
        const int sb = (c % NS) * STAGE;
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const signed char *base = (wave * UPW + i < AU ? xbase : q) + (size_t)c * IK;      // uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + soff[i]),
                                             (__attribute__((address_space(3))) void *)(ring + sb + dst[i]), 16, 0, 0);
        }
    };

    v16i acc[3][2][2];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[l][i][j][r] = 0;

#pragma unroll
    for (int c = 0; c < NS - 1; ++c)
        if (c < nchunks)
            issue(c);
#ifdef VSOM_DEVELOPMENT
    const int dbg = vsom_sl_dbg;
#else
    constexpr int dbg = 0;
#endif
    for (int c = 0; c < ((dbg & 1) ? 1 : nchunks); ++c) {
        // my pieces of chunk c have landed (the NS - 2 chunks behind it may still be in flight)
        if (NS == 3 && c + 1 < nchunks)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UPW) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                 // everybody's pieces of chunk c; everybody is done with chunk c-1
        if (c + NS - 1 < nchunks)
            issue(c + NS - 1);                           // into the stage chunk c-1 was read from
        const signed char *st = ring + (c % NS) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v4i a[XD][2], b[3][2];
            const int P = ks * 2 + lh;
#pragma unroll
            for (int pl = 0; pl < XD; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = wm * 64 + i * 32 + lr;
                    a[pl][i] = *reinterpret_cast<const v4i *>(st + pl * (AUP * 1024) + r * 64 + ((P ^ ((r >> 2) & 3)) << 4));
                }
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int r = wn * 64 + j * 32 + lr;
                    b[l][j] = *reinterpret_cast<const v4i *>(st + QOFF + l * 8192 + r * 64 + ((P ^ ((r >> 2) & 3)) << 4));
                }
#pragma unroll
            for (int pl = 0; pl < XD; ++pl)
#pragma unroll
                for (int l = 0; l + pl < 3; ++l)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[pl + l][i][j] = GLESS      // (G-less: nodes along the accumulator registers, samples along the lanes)
                                                    ? __builtin_amdgcn_mfma_i32_32x32x32_i8(b[l][j], a[pl][i], acc[pl + l][i][j], 0, 0, 0)
                                                    : __builtin_amdgcn_mfma_i32_32x32x32_i8(a[pl][i], b[l][j], acc[pl + l][i][j], 0, 0, 0);
        }
    }
    if (dbg & 2)
        return;
    if (GLESS) {
        // No G (FAST only: K <= 960).  As sl_k64_kernel below: with the tile transposed a lane's minimum over the 16
        // registers of a 32 x 32 block is ONE sample's minimum over 16 nodes -- no cross-lane step, no B x N matrix (268 MB
        // at C3, a quarter of this kernel's time to write and most of the refinement's to read): tmin[sample][tile], tile
        // t = 2 (node block of 32) + (lane >> 5) = the nodes 32 (t >> 1) + 4 (t & 1) + {0..3, 8..11, 16..19, 24..27}, and
        // sl_pick_kernel evaluates every node of the tiles within the bound in the reference's order.  The 128 nodes'
        // constants go through the ring's memory (free once every wavefront has left the K loop).
        float *s_nrm = reinterpret_cast<float *>(ring);
        int4 *s_f = reinterpret_cast<int4 *>(ring + 512);
        __syncthreads();
        if (tid < RT_N) {
            const int n = nbase + tid;
            s_nrm[tid] = n < N ? nrm[n] : __uint_as_float(0x7F800000u);     // past N: +inf (f = 0): never a minimum
            s_f[tid] = n < N ? qfast[n] : make_int4(0, 0, 0, 0);
        }
        float ts[2];
        int srow[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            srow[i] = sbase + wm * 64 + i * 32 + lr;
            ts[i] = XD == 3 ? xscale[srow[i] < s1 ? srow[i] : s1 - 1] : 1.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nblk = nbase + wn * 64 + j * 32;   // first node of this block (wavefront-uniform)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float mn = __uint_as_float(0x7F800000u);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nl = wn * 64 + j * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
                    const float g = XD == 1 ? sl_i8_value_fast(acc[0][i][j][r], acc[1][i][j][r], acc[2][i][j][r], s_f[nl], s_nrm[nl])
                                            : sl_i8_value_fast3(acc[0][i][j][r], acc[1][i][j][r], acc[2][i][j][r], s_f[nl], ts[i], s_nrm[nl]);
                    mn = fminf(mn, g);                   // a NaN (a NaN row of the map) never replaces the minimum
                }
                if (srow[i] < s1 && nblk < N)
                    tmin[(size_t)(srow[i] - s0) * ntm + (nblk >> 4) + lh] = mn;
            }
        }
        return;
    }
    // epilogue; a wavefront's 64 columns are exactly one 64-node tile of `tmin`.  Three phases, so that no load is pending
    // while values are formed and stored: with the constants loaded under `if (col < N)` and a conditional store behind
    // every value, hipcc put `s_waitcnt vmcnt(0)` in front of EVERY value -- 64 times the latency of the previous store per
    // wavefront, half of the kernel's time (round 5: 257 -> see profiles/EXPERIMENTS.md).
    // phase 0: the per-column constants (columns past N: those of the last column, never stored)
    const float inf = __uint_as_float(0x7F800000u);
    int col[2];
    double nm[2], sc[2], cr[2];
    int4 ff[2];
    float nmf[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        col[j] = nbase + wn * 64 + j * 32 + lr;
        const int cc = col[j] < N ? col[j] : N - 1;
        nmf[j] = nrm[cc];
        if (FAST) {
            ff[j] = qfast[cc];
        } else {
            nm[j] = (double)nmf[j];
            sc[j] = qscale[cc];
            cr[j] = XD == 1 ? qcorr[cc] : 0.0;
        }
    }
    // phase 1: the values (into the first accumulator set) and the row minima over this wavefront's 64 nodes (into the
    // second, whose sums have been consumed by then); general kind: the 16 row scales of each half tile loaded together
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float rsf[16];
        if (XD != 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = sbase + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                rsf[r] = xscale[row < s1 ? row : s1 - 1];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float mn = inf;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float g;
                if (FAST)
                    g = XD == 1 ? sl_i8_value_fast(acc[0][i][j][r], acc[1][i][j][r], acc[2][i][j][r], ff[j], nmf[j])
                                : sl_i8_value_fast3(acc[0][i][j][r], acc[1][i][j][r], acc[2][i][j][r], ff[j], rsf[r], nmf[j]);
                else
                    g = sl_i8_value(acc[0][i][j][r], acc[1][i][j][r], acc[2][i][j][r], cr[j],
                                    XD == 1 ? sc[j] : sc[j] * (double)rsf[r], nm[j]);
                acc[0][i][j][r] = __float_as_int(g);
                const float m = (col[j] < N && g == g) ? g : inf;       // exact minimum of the finite entries: NaN -> +inf
                mn = m < mn ? m : mn;
            }
            acc[1][i][0][r] = __float_as_int(sl_min32_dpp(mn));   // the 32 lanes that share this row; valid in lanes 31 / 63
        }
    }
    if (dbg & 4)
        return;
    // phase 2: stores only
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = sbase + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < s1) {
                float *grow = G + (size_t)(row - s0) * ldg;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (col[j] < N)
                        grow[col[j]] = __int_as_float(acc[0][i][j][r]);
                if (lr == 31)
                    tmin[(size_t)(row - s0) * ntm + blockIdx.x * 2 + wn] = __int_as_float(acc[1][i][0][r]);
            }
        }
    }
}

__global__ __launch_bounds__(512, 1) void sl_gemm_i8_ring_kernel(const signed char *__restrict__ xi, size_t xplane, int s0, int s1,
                                                                 const signed char *__restrict__ q, int N, int kp,
                                                                 const unsigned *__restrict__ kp_dev, int kp8,
                                                                 const float *__restrict__ nrm, const double *__restrict__ qscale,
                                                                 const double *__restrict__ qcorr, const float *__restrict__ xscale,
                                                                 float *__restrict__ G, int ldg, float *__restrict__ tmin, int ntm,
                                                                 const unsigned *__restrict__ xflag,
                                                                 const int4 *__restrict__ qfast, unsigned *__restrict__ scal)
{
    if (kp_dev)
        kp = (int)kp_dev[2];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64)
        sl_fold_maxima(scal);
    extern __shared__ __attribute__((aligned(1024))) signed char ring[];
    if (xflag[0] != 0u) {        // wavefront-uniform (a scalar load)
        if (qfast)
            sl_gemm_i8_ring_body<3, 4, 2, true>(xi, xplane, s0, s1, q, N, kp, kp8, nrm, qscale, qcorr, xscale, G, ldg, tmin, ntm, ring, qfast);
        else
            sl_gemm_i8_ring_body<3, 4, 2, false>(xi, xplane, s0, s1, q, N, kp, kp8, nrm, qscale, qcorr, xscale, G, ldg, tmin, ntm, ring, qfast);
    } else if (qfast)
        sl_gemm_i8_ring_body<1, 4, 3, true>(xi, xplane, s0, s1, q, N, kp, kp8, nrm, qscale, qcorr, xscale, G, ldg, tmin, ntm, ring, qfast);
    else
        sl_gemm_i8_ring_body<1, 4, 3, false>(xi, xplane, s0, s1, q, N, kp, kp8, nrm, qscale, qcorr, xscale, G, ldg, tmin, ntm, ring, qfast);
}

// the same without G (sl_pick_kernel refines): K <= 960 contracted columns
__global__ __launch_bounds__(512, 1) void sl_gemm_i8_ring_gless_kernel(const signed char *__restrict__ xi, size_t xplane, int s0, int s1,
                                                                       const signed char *__restrict__ q, int N, int kp,
                                                                       const unsigned *__restrict__ kp_dev, int kp8,
                                                                       const float *__restrict__ nrm, const float *__restrict__ xscale,
                                                                       float *__restrict__ tmin, int ntl,
                                                                       const unsigned *__restrict__ xflag,
                                                                       const int4 *__restrict__ qfast, unsigned *__restrict__ scal)
{
    if (kp_dev)
        kp = (int)kp_dev[2];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64)
        sl_fold_maxima(scal);
    extern __shared__ __attribute__((aligned(1024))) signed char ring[];
    if (xflag[0] != 0u)          // wavefront-uniform (a scalar load)
        sl_gemm_i8_ring_body<3, 4, 2, true, true>(xi, xplane, s0, s1, q, N, kp, kp8, nrm, nullptr, nullptr, xscale, nullptr, 0, tmin, ntl, ring, qfast);
    else
        sl_gemm_i8_ring_body<1, 4, 3, true, true>(xi, xplane, s0, s1, q, N, kp, kp8, nrm, nullptr, nullptr, xscale, nullptr, 0, tmin, ntl, ring, qfast);
}

// ---- at most 64 contracted columns: no G at all ----------------------------------------------------------------------------
// With K <= 64 a (sample, node) value costs two MFMAs per digit product and NO K loop: writing the B x N matrix G and
// reading it back in the refinement would be all of the time (C4, 16384 x 4096: 268 MB each way against 0.6 MB of
// operands).  Here the operands go from global memory straight into the MFMA registers (rows of the int8 planes are 64
// bytes: the 16-byte fragment of lane (row lr, k half lh) is one load), the tile is evaluated TRANSPOSED -- nodes along
// the accumulator registers, samples along the lanes -- so that the minimum over a lane's 16 registers is the minimum of
// one sample over 16 nodes without any cross-lane step, and only those minima are written: tmin[sample][tile], tile
// t = 2 (node block of 32) + (lane >> 5) = the nodes 32 (t >> 1) + 4 (t & 1) + {0..3, 8..11, 16..19, 24..27}.  The
// refinement (sl_pick_kernel, vsom_shortlist.hip) evaluates every node of every tile whose minimum is within the bound of the
// row minimum in the reference's order.
// Both kinds evaluate the value in fp32: the uint8 kind as sl_i8_value_fast, the general kind as |M|^2 - t_s (2 s_n w) with
// w = (16384 a0 + 128 a1 + a2) / 16384 -- the scales are powers of two >= 2^-50 each (vsom_digits.hpp), so the products
// with them are exact; the conversion of u = 128 a1 + a2 (< 2^27) is off by at most 4, which the refinement's bound
// carries as 16 t_s eps_n.  The kernel is bound by the vector instructions of its epilogue (7 per value), not by the
// matrix pipe: accumulators start from the MFMA's zero operand instead of 48 moves.  (Two digits per value instead of three -- half the MFMAs, 6
// instructions per value, bound terms x 128 -- measured at C4: kernel 30 instead of 34 us, but 4.7 instead of 1.0
// candidate tiles per sample on the trained map: refinement 45 instead of 22 us.  Dropped.)
#define K64_NB 16         // node blocks of 32 per workgroup (its 256 threads stage the constants of these 512 nodes)
#define K64_SB 2          // sample blocks of 32 per wavefront: every model fragment feeds two MFMAs (see below)
template <int XD>         // sample planes: 1 (uint8 image) or 3 (digits)
__device__ __forceinline__ void sl_k64_body(const signed char *__restrict__ xi, size_t xplane, int s0, int s1,
                                            const signed char *__restrict__ q, int N, const float *__restrict__ xscale,
                                            float *__restrict__ tmin, int ntl, const float *s_nrm, const int4 *s_f)
{
    // What the first version (one sample block per wavefront) waited for was the texture-address unit, not the matrix
    // pipe: a 16-byte load of 64 lanes occupies it for 16 cycles, six model fragments per node block = 96 cycles per
    // wavefront against 384 cycles of MFMAs on ITS SIMD -- but one unit serves the CU's four SIMDs.  With two sample
    // blocks per wavefront the same fragments feed twice the MFMAs.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int nbase = blockIdx.x * (K64_NB * 32);
    const size_t plane = (size_t)N * 64;
    v4i a[K64_SB][XD][2];
    float ts[K64_SB];
    bool sok[K64_SB];
    float *out[K64_SB];
#pragma unroll
    for (int sb = 0; sb < K64_SB; ++sb) {
        const int srow = s0 + (blockIdx.y * 4 + wave) * (32 * K64_SB) + sb * 32 + lr;
        sok[sb] = srow < s1;
        const size_t sr = (size_t)(sok[sb] ? srow : s1 - 1);
        const signed char *xbase = (XD == 1 ? xi : xi + xplane) + sr * 64 + 16 * lh;
#pragma unroll
        for (int pl = 0; pl < XD; ++pl)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                a[sb][pl][ks] = *reinterpret_cast<const v4i *>(xbase + pl * xplane + ks * 32);
        ts[sb] = XD == 3 ? xscale[sr] : 1.f;
        out[sb] = tmin + (size_t)(srow - s0) * ntl + (nbase >> 4) + lh;
    }
    const int nbn = (N - nbase + 31) / 32 < K64_NB ? (N - nbase + 31) / 32 : K64_NB;      // node blocks of this workgroup
    for (int nb = 0; nb < nbn; ++nb) {
        int n = nbase + nb * 32 + lr;
        n = n < N ? n : N - 1;
        const signed char *qb = q + (size_t)n * 64 + 16 * lh;
        v4i b[3][2];
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                b[l][ks] = *reinterpret_cast<const v4i *>(qb + l * plane + ks * 32);
        // the per-node constants of this block (16 per lane), shared by the sample blocks
        float nm[16], fx[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nl = nb * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
            nm[r] = s_nrm[nl];                           // nodes past N: +inf (f = 0): never a minimum
            fx[r] = __int_as_float(s_f[nl].x);
        }
#pragma unroll
        for (int sb = 0; sb < K64_SB; ++sb) {
            const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            // first operand = rows of the result = nodes.  acc[w]: products (sample plane pl) x (model plane l), pl + l = w
            v16i acc[3];
#pragma unroll
            for (int l = 0; l < 3; ++l) {                // sample plane 0 opens every accumulator set
                acc[l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b[l][0], a[sb][0][0], zero, 0, 0, 0);
                acc[l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b[l][1], a[sb][0][1], acc[l], 0, 0, 0);
            }
#pragma unroll
            for (int pl = 1; pl < XD; ++pl)
#pragma unroll
                for (int l = 0; l + pl < 3; ++l)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        acc[pl + l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(b[l][ks], a[sb][pl][ks], acc[pl + l], 0, 0, 0);
            float mn = __uint_as_float(0x7F800000u);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float g;
                if (XD == 1) {
                    g = sl_i8_value_fast(acc[0][r], acc[1][r], acc[2][r], s_f[nb * 32 + 4 * lh + (r & 3) + 8 * (r >> 2)], nm[r]);
                } else {
                    const int u = (acc[1][r] << 7) + acc[2][r];
                    const float w = fmaf((float)u, 0x1.0p-14f, (float)acc[0][r]);
                    g = fmaf(-ts[sb], fx[r] * w, nm[r]);
                }
                mn = fminf(mn, g);                       // a NaN (a NaN row of the map) never replaces the minimum
            }
            if (sok[sb])
                out[sb][2 * nb] = mn;
        }
    }
}

// scal: the counter set of this search (sl_fold_maxima)
__global__ __launch_bounds__(256, 2) void sl_k64_kernel(const signed char *__restrict__ xi, size_t xplane, int s0, int s1,
                                                        const signed char *__restrict__ q, int N,
                                                        const float *__restrict__ nrm, const int4 *__restrict__ qfast,
                                                        const float *__restrict__ xscale, float *__restrict__ tmin, int ntl,
                                                        const unsigned *__restrict__ xflag, unsigned *__restrict__ scal)
{
    __shared__ float s_nrm[K64_NB * 32];
    __shared__ int4 s_f[K64_NB * 32];
    for (int i = threadIdx.x; i < K64_NB * 32; i += 256) {
        const int n = blockIdx.x * (K64_NB * 32) + i;
        s_nrm[i] = n < N ? nrm[n] : __uint_as_float(0x7F800000u);
        s_f[i] = n < N ? qfast[n] : make_int4(0, 0, 0, 0);
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64)
        sl_fold_maxima(scal);
    __syncthreads();
    if (xflag[0] != 0u)                                  // workgroup-uniform (a scalar load)
        sl_k64_body<3>(xi, xplane, s0, s1, q, N, xscale, tmin, ntl, s_nrm, s_f);
    else
        sl_k64_body<1>(xi, xplane, s0, s1, q, N, xscale, tmin, ntl, s_nrm, s_f);
}

// ---- host --------------------------------------------------------------------------------------------------------------
// buffers of the int8 images: model planes for N rows, sample planes for Bcap rows of kp8 bytes
static int sl_i8_ensure(vsom_ctx *c, uint32_t kp8)
{
    if (!c->sl_q || c->sl_kp8 != kp8) {
        if (c->sl_q) {
            VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
            VSOM_HIP_CHECK(hipFree(c->sl_q));
            VSOM_HIP_CHECK(hipFree(c->sl_qscale));
            VSOM_HIP_CHECK(hipFree(c->sl_qcorr));
            VSOM_HIP_CHECK(hipFree(c->sl_qfast));
        }
        c->sl_q = nullptr;
        VSOM_HIP_CHECK(hipMalloc(&c->sl_q, (size_t)3 * c->N * kp8));
        VSOM_HIP_CHECK(hipMalloc(&c->sl_qscale, (size_t)c->N * sizeof(double)));
        VSOM_HIP_CHECK(hipMalloc(&c->sl_qcorr, (size_t)c->N * sizeof(double)));
        VSOM_HIP_CHECK(hipMalloc(&c->sl_qfast, (size_t)c->N * sizeof(int4)));
        c->sl_kp8 = kp8;
        c->xi_valid = false;
        c->sl_xi_cap = 0;
    }
    if ((size_t)c->Bcap * kp8 > c->sl_xi_cap) {
        if (c->sl_xi) {
            VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
            VSOM_HIP_CHECK(hipFree(c->sl_xi));
            VSOM_HIP_CHECK(hipFree(c->sl_l1));
        }
        c->sl_xi = nullptr;
        VSOM_HIP_CHECK(hipMalloc(&c->sl_xi, (size_t)4 * c->Bcap * kp8));             // [0] x - 128, [1..3] digits
        VSOM_HIP_CHECK(hipMalloc(&c->sl_l1, (size_t)5 * c->Bcap * sizeof(float)));   // |x|_1, t_s, l1eff_s, e_s, |x|^2 (rows of <= 64 values)
        c->sl_xi_cap = (size_t)c->Bcap * kp8;
        c->xi_valid = false;
    }
    return VSOM_OK;
}

// the compaction's gather pass (vsom_compact.hip): the staged rows onto the live columns and -- once the integer
// shortlist's buffers exist (the first search of a context allocates them) -- their int8 images in the same pass
int launch_sl_gather_quant(vsom_ctx *c, size_t B, hipStream_t stream, const int *idx, bool *xi_out)
{
    const uint32_t kp8 = (c->cpitch + 63) / 64 * 64;
    const bool xi = c->sl_xi && c->sl_kp8 == kp8 && (size_t)c->Bcap * kp8 <= c->sl_xi_cap && c->sl_scal;
    if (kp8 == 64)
        hipLaunchKernelGGL(sl_quant_rows64_kernel, dim3((unsigned)((B + 15) / 16)), dim3(256), 0, stream, c->Xs, (int)c->xpitch, c->Xc,
                           (int)c->cpitch, idx, (int)B, xi ? c->sl_xi : (signed char *)nullptr, (size_t)c->Bcap * kp8,
                           xi ? c->sl_l1 : (float *)nullptr, (size_t)c->Bcap, xi ? c->sl_scal + 8192 : (unsigned *)nullptr);
    else
    hipLaunchKernelGGL(sl_quant_rows_kernel, dim3((unsigned)B), dim3(256), 0, stream, c->Xs, (int)c->xpitch, c->Xc,
                       (int)c->cpitch, idx, (int)B, xi ? c->sl_xi : (signed char *)nullptr, (size_t)c->Bcap * kp8, (int)kp8,
                       xi ? c->sl_l1 : (float *)nullptr, (size_t)c->Bcap, xi ? c->sl_scal + 8192 : (unsigned *)nullptr);
    VSOM_HIP_CHECK(hipGetLastError());
    *xi_out = xi;
    return VSOM_OK;
}

// prepares the int8 images and computes G / tmin for samples [s0, s1) (ldg, ntm as the fp32 path lays them out: 64-node
// tile minima); scal = the counter set THIS search's select / feedback kernels read (the two sets alternate,
// vsom_shortlist.hip), already reset
// Which form a search of samples [s0, s1) takes: 0 = G and 64-node tile minima (sl_select_kernel refines), 1 = no G, 16-node
// tile minima from sl_k64_kernel (at most 64 contracted columns), 2 = no G, from the ring kernel (big problems, K <= 960);
// 1 and 2 are refined by sl_pick_kernel.  (The attribute is per DEVICE -- a group drives several from one process -- so it
// is raised on every call, as launch_phase2 does.)
int sl_i8_plan(vsom_ctx *c, size_t s0, size_t s1)
{
    const uint32_t kp8 = ((c->cc_valid ? c->cpitch : c->xpitch) + 63) / 64 * 64;
    if (kp8 == 64)
        return 1;
    const size_t big_tiles = ((size_t)c->N + RT_N - 1) / RT_N * ((s1 - s0 + RT_S - 1) / RT_S);
    if (kp8 <= 960 && big_tiles >= 256) {
        if (hipFuncSetAttribute((const void *)sl_gemm_i8_ring_gless_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                RING_BYTES) == hipSuccess)
            return 2;
        (void)hipGetLastError();
    }
    return 0;
}

int launch_sl_i8(vsom_ctx *c, size_t s0, size_t s1, size_t ldg, size_t ntm, unsigned *scal, unsigned *xflag, int plan)
{
    const bool gless = plan == 1;
    const bool compact = c->cc_valid;
    const uint32_t kmax = compact ? c->cpitch : c->xpitch;
    const uint32_t kp8 = (kmax + 63) / 64 * 64;
    if (int rc = sl_i8_ensure(c, kp8))
        return rc;
#ifdef VSOM_DEVELOPMENT
    {
        static int last = -1;
        const char *e = std::getenv("VSOM_SL_DBG");
        const int v = e ? std::atoi(e) : 0;
        if (v != last) {
            (void)hipMemcpyToSymbol(HIP_SYMBOL(vsom_sl_dbg), &v, sizeof(int));
            last = v;
        }
    }
#endif
    const size_t xplane = (size_t)c->Bcap * kp8;
    if (!c->xi_valid) {      // once per staged chunk (the compaction's gather pass does it when the buffers exist)
        if (kp8 == 64)
            hipLaunchKernelGGL(sl_quant_rows64_kernel, dim3((unsigned)((c->B + 15) / 16)), dim3(256), 0, c->stream,
                               compact ? c->Xc : c->Xs, (int)kmax, (float *)nullptr, (int)kmax, (const int *)nullptr, (int)c->B,
                               c->sl_xi, xplane, c->sl_l1, (size_t)c->Bcap, xflag);
        else
        hipLaunchKernelGGL(sl_quant_rows_kernel, dim3((unsigned)c->B), dim3(256), 0, c->stream, compact ? c->Xc : c->Xs, (int)kmax,
                           (float *)nullptr, (int)kmax, (const int *)nullptr, (int)c->B, c->sl_xi, xplane, (int)kp8, c->sl_l1,
                           (size_t)c->Bcap, xflag);
        c->xi_valid = true;
    }
    const float *xscale = c->sl_l1 + c->Bcap;
    const unsigned *kp_dev = compact ? (const unsigned *)c->cc_meta : nullptr;
    if (kp8 == 64 && c->part_pitch <= 64)
        hipLaunchKernelGGL(sl_prepare64_kernel, dim3((unsigned)((c->N + 15) / 16)), dim3(256), 0, c->stream, c->map, (int)c->pitch,
                           (int)c->part_pitch, (int)c->N, compact ? (const int *)c->cc_idx : (const int *)nullptr, (int)kmax, kp_dev,
                           c->sl_q, c->sl_nrm, c->sl_qscale, c->sl_qcorr, (int4 *)c->sl_qfast, scal, xflag);
    else
    hipLaunchKernelGGL(sl_prepare_i8_kernel, dim3((unsigned)((c->N + 3) / 4)), dim3(256), 0, c->stream, c->map, (int)c->pitch,
                       (int)c->part_pitch, (int)c->N, compact ? (const int *)c->cc_idx : (const int *)nullptr, (int)kmax, kp_dev,
                       (int)kp8, c->sl_q, c->sl_nrm, c->sl_qscale, c->sl_qcorr, (int4 *)c->sl_qfast, scal, xflag);
    // the fp32 two-fma epilogue of the uint8 kind needs |A| < 2^24: K <= 960 contracted columns (sl_i8_value_fast)
    const int4 *qfast = kp8 <= 960 ? (const int4 *)c->sl_qfast : (const int4 *)nullptr;
    if (gless) {             // K <= 64: tile minima only (ntm = 16-node tiles), the refinement is sl_pick_kernel
        if (kp8 != 64)
            return VSOM_ERR_INVALID;
        dim3 grid((unsigned)((c->N + K64_NB * 32 - 1) / (K64_NB * 32)), (unsigned)((s1 - s0 + 128 * K64_SB - 1) / (128 * K64_SB)));
        hipLaunchKernelGGL(sl_k64_kernel, grid, dim3(256), 0, c->stream, c->sl_xi, xplane, (int)s0, (int)s1, c->sl_q, (int)c->N,
                           c->sl_nrm, (const int4 *)c->sl_qfast, xscale, c->sl_tmin, (int)ntm, (const unsigned *)xflag, scal);
        VSOM_HIP_CHECK(hipGetLastError());
        return VSOM_OK;
    }
    if (plan == 2) {         // the ring kernel without G (ntm = 16-node tiles)
        dim3 grid((unsigned)((c->N + RT_N - 1) / RT_N), (unsigned)((s1 - s0 + RT_S - 1) / RT_S));
        hipLaunchKernelGGL(sl_gemm_i8_ring_gless_kernel, grid, dim3(512), RING_BYTES, c->stream, c->sl_xi, xplane, (int)s0, (int)s1,
                           c->sl_q, (int)c->N, (int)kmax, kp_dev, (int)kp8, c->sl_nrm, xscale, c->sl_tmin, (int)ntm,
                           (const unsigned *)xflag, (const int4 *)c->sl_qfast, scal);
        VSOM_HIP_CHECK(hipGetLastError());
        return VSOM_OK;
    }
    // big maps and chunks: 256 x 128 tiles through the LDS-DMA ring; otherwise (few tiles: they would not fill the
    // chip) 128 x 64 tiles staged through registers
    const size_t big_tiles = ((size_t)c->N + RT_N - 1) / RT_N * ((s1 - s0 + RT_S - 1) / RT_S);
    // the attribute is per DEVICE (a group drives several from one process): raised on every launch, as launch_phase2 does
    bool ring_ok = false;
    if (big_tiles >= 1024) {
        ring_ok = hipFuncSetAttribute((const void *)sl_gemm_i8_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      RING_BYTES) == hipSuccess;
        if (!ring_ok)
            (void)hipGetLastError();
    }
    if (ring_ok) {
        dim3 grid((unsigned)(ntm / 2), (unsigned)((s1 - s0 + RT_S - 1) / RT_S));   // ntm = 2 ceil(N / 128) 64-node tiles
        hipLaunchKernelGGL(sl_gemm_i8_ring_kernel, grid, dim3(512), RING_BYTES, c->stream, c->sl_xi, xplane, (int)s0, (int)s1,
                           c->sl_q, (int)c->N, (int)kmax, kp_dev, (int)kp8, c->sl_nrm, c->sl_qscale, c->sl_qcorr, xscale, c->sl_G,
                           (int)ldg, c->sl_tmin, (int)ntm, (const unsigned *)xflag, qfast, scal);
    } else {
        constexpr int MI = 2;
        dim3 grid((unsigned)ntm, (unsigned)((s1 - s0 + 64 * MI - 1) / (64 * MI)));   // ntm 64-node tiles (the last may lie past N: minima +inf)
        hipLaunchKernelGGL((sl_gemm_i8_kernel<MI, 1>), grid, dim3(256), 0, c->stream, c->sl_xi, xplane, (int)s0, (int)s1, c->sl_q,
                           (int)c->N, (int)kmax, kp_dev, (int)kp8, c->sl_nrm, c->sl_qscale, c->sl_qcorr, xscale, c->sl_G, (int)ldg,
                           c->sl_tmin, (int)ntm, (const unsigned *)xflag, qfast, scal);
        hipLaunchKernelGGL((sl_gemm_i8_kernel<MI, 3>), grid, dim3(256), 0, c->stream, c->sl_xi, xplane, (int)s0, (int)s1, c->sl_q,
                           (int)c->N, (int)kmax, kp_dev, (int)kp8, c->sl_nrm, c->sl_qscale, c->sl_qcorr, xscale, c->sl_G, (int)ldg,
                           c->sl_tmin, (int)ntm, (const unsigned *)xflag, qfast, scal);
    }
    VSOM_HIP_CHECK(hipGetLastError());
    return VSOM_OK;
}
