// vsom_device.hpp -- device helpers shared by the gfx950 kernels.
//
// Strict-arithmetic rules (the kernels must reproduce the reference's SSE2, non-FMA fp32
// results bit for bit): this translation unit is compiled with -ffp-contract=off and
// -fhip-fp32-correctly-rounded-divide-sqrt; f32 subnormals are kept (hipcc default mode).
#pragma once
#include "vsom_internal.hpp"

// operands of one distance evaluation (Som::euclidianWeightedDist, Som.cpp:124-141)
struct DistArgs {
    const float *xa;   // sample rows: Xs (Standard/Median) or XP (CLR)
    const float *xb;   // CLR: YP rows
    int ldx;
    const float *ma;   // model rows: map (A part for CLR)
    const float *mb;   // CLR: B part (= map + part_pitch)
    int ldm;
    int L;             // comparer length: D (Standard/Median) or P = D/2 (CLR)
};

// residual element of Transformation::Comparer
//   Standard/Median: model - value                     (Transformation.cpp:8,46)
//   CLR: (A*x' + B) - y', one rounding per operation   (Transformation.cpp:104)
template <bool CLR>
__device__ __forceinline__ float vsom_resid(float x, float y, float m, float b)
{
    if (CLR) {
        float t = m * x;
        t = t + b;
        t = t - y;
        return t;
    }
    return m - x;
}

// order-preserving key for argmin with lowest-index tie break; NaN never wins (Som.cpp:299)
__device__ __forceinline__ u64 vsom_key(float d, uint32_t node)
{
    uint32_t bits = (d != d) ? 0xFFFFFFFFu : __float_as_uint(d);
    return ((u64)bits << 32) | (u64)node;
}

// Distance of one (sample,node) pair computed by a group of 8 consecutive lanes, lane k
// owning Eigen's accumulator class k (elements d = k mod 8), followed by the reduction tree
// of Eigen's SSE linear-vectorised redux:
//   q_k = p0_k + p1_k ; [+ one more packet] ; (q0+q2)+(q1+q3) ; + scalar tail.
// All 8 lanes return the same value.  xa/xb/ma/mb are the row pointers of the pair.
template <bool CLR, int UNR = 14>
__device__ __forceinline__ float vsom_group_dist(const float *xa, const float *xb,
                                                 const float *ma, const float *mb, int L, int k)
{
    const int L8 = L & ~7;
    float acc = 0.f;
#pragma unroll UNR
    for (int d = k; d < L8; d += 8) {
        float r = vsom_resid<CLR>(xa[d], CLR ? xb[d] : 0.f, ma[d], CLR ? mb[d] : 0.f);
        float p = r * r;
        acc = acc + p;
    }
    float q = acc + __shfl_xor(acc, 4);          // p0_k + p1_k (commutative: same bits in k, k^4)
    const int rem = L - L8;
    if (rem >= 4) {
        int d = L8 + (k & 3);
        float r = vsom_resid<CLR>(xa[d], CLR ? xb[d] : 0.f, ma[d], CLR ? mb[d] : 0.f);
        float p = r * r;
        q = q + p;
    }
    float t = q + __shfl_xor(q, 2);              // (q0+q2) in lanes 0,2 ; (q1+q3) in lanes 1,3
    float res = t + __shfl_xor(t, 1);            // (q0+q2)+(q1+q3)
    for (int tt = (rem >= 4 ? 4 : 0); tt < rem; ++tt) {
        int d = L8 + tt;
        float r = vsom_resid<CLR>(xa[d], CLR ? xb[d] : 0.f, ma[d], CLR ? mb[d] : 0.f);
        float p = r * r;
        res = res + p;
    }
    return res;
}


// The same distance for callers that wait on it alone (the online path's local search and post step:
// one wavefront, nothing else to hide the memory latency).  vsom_group_dist's unrolled loop leaves a
// rolled remainder loop with one load in flight -- a whole round trip per element for short rows
// (D = 32: 4 trips) and for the last < 14 elements of long ones; here the remainder is fetched in
// blocks of up to 16 / 4 elements before the first is used.  Same operations in the same order per
// accumulator class as vsom_group_dist, hence the same bits.
template <bool CLR, int BLK>
__device__ __forceinline__ float vsom_group_acc_block(float acc, const float *xa, const float *xb, const float *ma,
                                                      const float *mb, int k, int base, int nb)
{
    float xv[BLK], yv[BLK], mv[BLK], bv[BLK];
#pragma unroll
    for (int u = 0; u < BLK; ++u) {
        const int d = k + 8 * (base + (u < nb ? u : nb - 1));   // clamped: always a valid element
        xv[u] = xa[d];
        mv[u] = ma[d];
        yv[u] = CLR ? xb[d] : 0.f;
        bv[u] = CLR ? mb[d] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < BLK; ++u) {
        if (u < nb) {                                            // wavefront-uniform
            const float r = vsom_resid<CLR>(xv[u], yv[u], mv[u], bv[u]);
            const float p = r * r;
            acc = acc + p;
        }
    }
    return acc;
}

// BIG = elements per class fetched together on long rows (14; 28 where the sample row sits in LDS and only the model row's
// loads need registers: sl_pick_kernel).
template <bool CLR, int BIG = 14>
__device__ __forceinline__ float vsom_group_dist_lat(const float *xa, const float *xb,
                                                     const float *ma, const float *mb, int L, int k)
{
    const int L8 = L & ~7;
    const int ni = L8 >> 3;                      // elements per class
    float acc = 0.f;
    for (int base = 0; base < ni;) {
        const int left = ni - base;
        if (left > BIG + 2) {     // long rows: 14 at a time, like vsom_group_dist (49 at a time measured slower)
            acc = vsom_group_acc_block<CLR, BIG>(acc, xa, xb, ma, mb, k, base, BIG);
            base += BIG;
        } else if (BIG > 16 && left > 16) {
            acc = vsom_group_acc_block<CLR, 14>(acc, xa, xb, ma, mb, k, base, 14);
            base += 14;
        } else if (left > 4) {
            acc = vsom_group_acc_block<CLR, 16>(acc, xa, xb, ma, mb, k, base, left);
            base += left;
        } else {
            acc = vsom_group_acc_block<CLR, 4>(acc, xa, xb, ma, mb, k, base, left);
            base += left;
        }
    }
    float q = acc + __shfl_xor(acc, 4);
    const int rem = L - L8;
    if (rem >= 4) {
        int d = L8 + (k & 3);
        float r = vsom_resid<CLR>(xa[d], CLR ? xb[d] : 0.f, ma[d], CLR ? mb[d] : 0.f);
        float p = r * r;
        q = q + p;
    }
    float t = q + __shfl_xor(q, 2);
    float res = t + __shfl_xor(t, 1);
    for (int tt = (rem >= 4 ? 4 : 0); tt < rem; ++tt) {
        int d = L8 + tt;
        float r = vsom_resid<CLR>(xa[d], CLR ? xb[d] : 0.f, ma[d], CLR ? mb[d] : 0.f);
        float p = r * r;
        res = res + p;
    }
    return res;
}

// Som::findLocalBmu (Som.cpp:335-454) for one sample by one wavefront: 8 candidates x 8 accumulator
// classes across the 64 lanes, unsigned (size_t) arithmetic kept literal.  Every lane
// returns the same (minIndex, minDist); minDist == ||Comparer(x, M[minIndex])||^2.
template <bool CLR>
__device__ __forceinline__ void vsom_local_walk(const DistArgs &a, const float *xa, const float *xb, u64 width,
                                                u64 height, u64 start, int lane, u64 &minIndexOut, float &minDistOut)
{
    const int g = lane >> 3, k = lane & 7;
    const u64 m1 = ~0ull;   // -1uz
    // firstSearchX / firstSearchY (Som.cpp:341-342)
    const u64 fsx = (g == 0 || g >= 6) ? m1 : ((g == 1 || g == 5) ? 0ull : 1ull);
    const u64 fsy = (g <= 2) ? 1ull : ((g == 3 || g == 7) ? 0ull : m1);

    u64 lastBMU = start;
    float minDist = vsom_group_dist<CLR>(xa, xb, a.ma + (size_t)lastBMU * a.ldm,
                                         a.mb + (size_t)lastBMU * a.ldm, a.L, k);
    minDist = __shfl(minDist, 0);
    u64 minIndex = lastBMU;
    u64 lastMeasured = lastBMU;

    for (;;) {
        const u64 lmX = lastMeasured % width, lmY = lastMeasured / width;
        const u64 lbX = lastBMU % width, lbY = lastBMU / width;
        if (lastMeasured == lastBMU) {   // first try: 8 neighbours, wrap-then-clamp (Som.cpp:362-385)
            u64 cx = lmX + fsx;
            cx = cx < width - 1 ? cx : width - 1;
            u64 cy = lmY + fsy;
            cy = cy < height - 1 ? cy : height - 1;
            u64 node = cy * width + cx;
            float d = vsom_group_dist<CLR>(xa, xb, a.ma + (size_t)node * a.ldm,
                                           a.mb + (size_t)node * a.ldm, a.L, k);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float di = __shfl(d, i * 8);
                u64 ni = __shfl(node, i * 8);
                if (di < minDist) {
                    minDist = di;
                    minIndex = ni;
                }
            }
            if (minIndex == lastBMU)
                break;
            lastMeasured = minIndex;
        } else {
            if (lmX - lbX) {   // moving in X: 3 nodes ahead (Som.cpp:390-403)
                u64 cx = lmX + lmX - lbX;
                cx = cx < width - 1 ? cx : width - 1;
                u64 off = (u64)(long long)((g < 3 ? g : 0) - 1);   // i = -1,0,1
                u64 cy = lmY + off;
                cy = cy < height - 1 ? cy : height - 1;
                u64 node = cy * width + cx;
                float d = vsom_group_dist<CLR>(xa, xb, a.ma + (size_t)node * a.ldm,
                                               a.mb + (size_t)node * a.ldm, a.L, k);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    float di = __shfl(d, i * 8);
                    u64 ni = __shfl(node, i * 8);
                    if (di < minDist) {
                        minDist = di;
                        minIndex = ni;
                    }
                }
            }
            // moving in Y (Som.cpp:406-437): the reference's loop starts at SIZE_MAX and its
            // condition `i < endX + 1` is false at once, so no node is evaluated.
            (void)lbY;
            if (minIndex == lastMeasured)
                break;
            lastBMU = lastMeasured;
            lastMeasured = minIndex;
        }
    }
    minIndexOut = minIndex;
    minDistOut = minDist;
}

// SomIndex(const Som&, size_t) (SomIndex.cpp:13-18): y divides by HEIGHT (Q10)
__device__ __forceinline__ void vsom_somindex(u64 idx, u64 W, u64 H, int &x, int &y)
{
    u64 xm = idx % W;
    x = (int)xm;
    y = (int)((idx - xm) / H);
}
