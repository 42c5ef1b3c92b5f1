// vsom_online.hip -- online path: Som::trainSingle (Som.cpp:885-947) and the per-chunk loop of
// Som::trainBasicSom (Som.cpp:1159-1171) on gfx950.
//
// The path is strictly sequential in samples (sample j's BMU search reads the map sample j-1
// wrote), so one sample = two launches (sigma > 1) or one (sigma <= 1) enqueued back to back on
// the context stream, without host synchronisation:
//   online_scan_kernel    sigma > 1: Som::findBmu, 8 lanes per node (one per Eigen accumulator
//                         class), atomicMin on an order-preserving (distance, index) key.  Key and node-0-NaN flag are double-buffered by
//                         sample parity, so no launch is needed to resolve / re-arm them.
//   online_window_kernel  the +-2.5 sigma window (Som.cpp:899-944): one workgroup per window
//                         node, elementwise over the model vector; weightMap / map / SMap /
//                         sigmaMap updated with the reference's double->float narrowing (Q11).
//                         The workgroup that owns the BMU node then does the post step: residual
//                         + distance of the BMU after the update (:946), addBmu (:1189-1192), MSE
//                         running sum (:1167), lastBMU (:895), re-arming the other parity's key.
//   online_small_kernel   sigma <= 1: Som::findLocalBmu + <=6x6 window + post in one launch of one
//                         1024-thread workgroup (16 wavefronts share the window's nodes: 15.0 vs 17.9 us
//                         with 4; staging the walk's candidate rows through LDS was measured slower, 20 us)
// The bandwidth roofline of one sample is 4*N*D (scan) + 20*k*D (k window nodes) bytes.
#include "vsom_device.hpp"
#include <cmath>
#include <algorithm>
#include <cstring>

// onl_state (u64): the argmin key of a sample lives in ONL_SLOTS slots, one 128-byte line each (slot s
// of parity p at [(p * ONL_SLOTS + s) * 16]); a workgroup of the scan folds its minimum into slot
// blockIdx % ONL_SLOTS and the reader takes the minimum over the slots.  With ONE slot the 512
// same-address device-scope atomics of a 128x128 scan serialise and cost 4.5 of its 13.5 us
// (tools/exp/scan_bw_bench.hip: 8.9 us without the atomic, 13.5 with one slot, 9.1-9.4 with 8-64).
// The node-0-NaN flags of the two parities sit at [ONL_FLAG], [ONL_FLAG + 1].
// onl_f: [0] dist, [1] mse sum
constexpr int ONL_SLOTS = 16;
constexpr int ONL_FLAG = 2 * ONL_SLOTS * 16;
constexpr size_t ONL_STATE_BYTES = (ONL_FLAG + 16) * sizeof(u64);
static_assert(ONL_STATE_BYTES <= VSOM_ONL_STATE_BYTES, "vsom_create allocates VSOM_ONL_STATE_BYTES");
__device__ __forceinline__ u64 *online_slot(u64 *state, int par, int s) { return state + (par * ONL_SLOTS + s) * 16; }
struct OnlineArgs {
    DistArgs d;          // xa/xb point at the sample's row(s)
    u64 *state;
    float *fstate;
    int N, W, H;
    int par;             // sample parity: which key / flag this sample uses
    // chunk loop: the search launch of sample j also finishes sample j-1 (its extra workgroup, online_scan_kernel)
    const float *pxa, *pxb;   // rows of the sample being finished
    int do_scan, do_post;
};

// BMU of the sample from its scan results; a NaN distance at node 0 pins it to 0 (Som.cpp:293-299)
// (whole wavefronts call this: the slots are read by 16 lanes and folded with shuffles)
__device__ __forceinline__ u64 online_resolve(const u64 *state, int par)
{
    u64 key = state[(par * ONL_SLOTS + ((int)threadIdx.x & (ONL_SLOTS - 1))) * 16];
#pragma unroll
    for (int off = ONL_SLOTS / 2; off >= 1; off >>= 1) {
        const u64 o = __shfl_xor(key, off);
        key = o < key ? o : key;
    }
    return state[ONL_FLAG + par] ? 0ull : (key & 0xFFFFFFFFull);
}

#ifndef VSOM_SCAN_UNR
#define VSOM_SCAN_UNR 14
#endif
template <bool CLR>
__device__ __forceinline__ void online_post(const OnlineArgs &a, const float *xa, const float *xb, u64 bmu, int lane, u64 *hits,
                                            u64 *lastbmu_out, float *residual, float fB, int add_hit);

// POSTWG: the chunk loop's form.  The LAST workgroup does not search: it finishes the PREVIOUS sample (residual and
// distance of its BMU after the window update :946, addBmu, MSE, lastBMU) and re-arms that sample's key set -- 2-3 us of
// dependent row reads that used to sit at the end of the window launch's critical path, now hidden beside the 9.6 us
// search (the window launch of the previous sample has completed: kernel boundary; the next window launch has not
// started).  After the last sample of a chunk the kernel is launched once more with do_scan = 0.
template <bool CLR, bool POSTWG>
__global__ __launch_bounds__(256) void online_scan_kernel(OnlineArgs a, u64 *hits, u64 *lastbmu_out, float fB, int add_hit)
{
    __shared__ u64 skey[4];
    if (POSTWG && blockIdx.x == gridDim.x - 1) {
        if (a.do_post && threadIdx.x < 64) {
            const u64 bmu = online_resolve(a.state, a.par ^ 1);
            online_post<CLR>(a, a.pxa, a.pxb, bmu, (int)threadIdx.x, hits, lastbmu_out, nullptr, fB, add_hit);
            if (threadIdx.x < ONL_SLOTS)
                *online_slot(a.state, a.par ^ 1, (int)threadIdx.x) = ~0ull;   // the key set of the sample after this one
        }
        return;
    }
    if (POSTWG && !a.do_scan)
        return;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int node = gid >> 3, k = threadIdx.x & 7;
    const int nc = node < a.N ? node : a.N - 1;
    float d = vsom_group_dist<CLR, VSOM_SCAN_UNR>(a.d.xa, a.d.xb, a.d.ma + (size_t)nc * a.d.ldm,
                                                  a.d.mb + (size_t)nc * a.d.ldm, a.d.L, k);
    u64 key = (node < a.N) ? vsom_key(d, (uint32_t)node) : ~0ull;
    if (node == 0 && k == 0)
        a.state[ONL_FLAG + a.par] = (d != d) ? 1ull : 0ull;
    // wave min, then block min, then one atomic per block
    for (int off = 32; off >= 8; off >>= 1) {
        u64 o = __shfl_xor(key, off);
        key = o < key ? o : key;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        skey[wave] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 m = skey[0];
        for (int i = 1; i < 4; ++i)
            m = skey[i] < m ? skey[i] : m;
        atomicMin(online_slot(a.state, a.par, (int)(blockIdx.x % ONL_SLOTS)), m);
    }
}

// g-th candidate of the walk's first step: the 8 neighbours of `from`, wrap-then-clamp (Som.cpp:362-385)
__device__ __forceinline__ u64 online_first_try_node(u64 from, int g, u64 width, u64 height)
{
    const u64 m1 = ~0ull;
    const u64 fsx = (g == 0 || g >= 6) ? m1 : ((g == 1 || g == 5) ? 0ull : 1ull);
    const u64 fsy = (g <= 2) ? 1ull : ((g == 3 || g == 7) ? 0ull : m1);
    u64 cx = from % width + fsx;
    cx = cx < width - 1 ? cx : width - 1;
    u64 cy = from / width + fsy;
    cy = cy < height - 1 ? cy : height - 1;
    return cy * width + cx;
}

// single-sample Som::findLocalBmu (same walk as bmu_local_kernel in vsom_bmu.hip).  The distance of
// the starting node (own) and of the first step's 8 neighbours (dfirst, one per lane group) do not
// depend on each other; the caller may have them evaluated side by side by two wavefronts and pass
// them in (own != nullptr), which takes one of the walk's latency-bound distance passes off its path.
template <bool CLR>
__device__ __forceinline__ u64 online_local_search(const OnlineArgs &a, const u64 *lastbmu, int lane,
                                                   const float *own = nullptr, float dfirst = 0.f)
{
    const int g = lane >> 3, k = lane & 7;
    const u64 width = (u64)a.W, height = (u64)a.H;
    const float *xa = a.d.xa, *xb = a.d.xb;
    u64 lastBMU = *lastbmu;
    float minDist;
    if (own) {
        minDist = *own;
    } else {
        minDist = vsom_group_dist_lat<CLR>(xa, xb, a.d.ma + (size_t)lastBMU * a.d.ldm,
                                           a.d.mb + (size_t)lastBMU * a.d.ldm, a.d.L, k);
        minDist = __shfl(minDist, 0);
    }
    bool precomputed = own != nullptr;
    u64 minIndex = lastBMU, lastMeasured = lastBMU;
    for (;;) {
        const u64 lmX = lastMeasured % width, lmY = lastMeasured / width;
        const u64 lbX = lastBMU % width;
        if (lastMeasured == lastBMU) {
            const u64 node = online_first_try_node(lastMeasured, g, width, height);
            float d;
            if (precomputed) {
                d = dfirst;
                precomputed = false;
            } else {
                d = vsom_group_dist_lat<CLR>(xa, xb, a.d.ma + (size_t)node * a.d.ldm,
                                             a.d.mb + (size_t)node * a.d.ldm, a.d.L, k);
            }
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
            if (lmX - lbX) {
                u64 cx = lmX + lmX - lbX;
                cx = cx < width - 1 ? cx : width - 1;
                u64 off = (u64)(long long)((g < 3 ? g : 0) - 1);
                u64 cy = lmY + off;
                cy = cy < height - 1 ? cy : height - 1;
                u64 node = cy * width + cx;
                float d = vsom_group_dist_lat<CLR>(xa, xb, a.d.ma + (size_t)node * a.d.ldm,
                                               a.d.mb + (size_t)node * a.d.ldm, a.d.L, k);
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
            if (minIndex == lastMeasured)
                break;
            lastBMU = lastMeasured;
            lastMeasured = minIndex;
        }
    }
    return minIndex;   // identical in every lane
}

__device__ __forceinline__ float onl_sign(float a)
{
    return a > 0.f ? 1.f : (a < 0.f ? -1.f : (a != a ? a : 0.f));
}

// update of ONE window node (Som.cpp:911-943) by the `nthr` threads tid = 0..nthr-1 of a group
// (a workgroup with BLOCK_SYNC, or a single wavefront running in lockstep without)
// store flavours of the window update's rows: 0 plain; 1 nontemporal; 2 write-through (sc1: the line does not stay dirty in
// the XCD's L2 until the end-of-kernel write-back)
template <int ST>
__device__ __forceinline__ void onl_store(float *p, float v)
{
    if (ST == 1)
        __builtin_nontemporal_store(v, p);
    else if (ST == 2)
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}

template <int KIND, bool BLOCK_SYNC, int ST = 0, bool SIG = true>   // SIG = false: sigmaMap is written later (onl_sigma_kernel)
__device__ __forceinline__ void online_node_update(size_t n, double h, int tid, int nthr,
                                                   const float *__restrict__ xs, const float *__restrict__ xp,
                                                   const float *__restrict__ yp, int D, int P, int ppitch, int pitch,
                                                   double eta, int decay_fn, float *map, float *Smap, float *sigmap,
                                                   float *weight, float *mkeep = nullptr)   // mkeep[4]: the new M of elements tid + u * nthr
{
    float *M = map + n * pitch, *S = Smap + n * pitch, *sg = sigmap + n * pitch;
    // workgroup form, Standard / Median: this thread's first four elements are requested BEFORE the barrier below, so that
    // weightMap[n], the table entry h and the rows travel in ONE memory round trip (the barrier waits for all of them)
    constexpr bool PRE = BLOCK_SYNC && KIND != VSOM_CLR;
    float px[4], pm[4], ps[4];
    if (PRE) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int d = tid + u * nthr;
            const bool ok = d < D;
            px[u] = ok ? xs[d] : 0.f;
            pm[u] = ok ? M[d] : 0.f;
            ps[u] = ok ? S[d] : 0.f;
        }
    }
    const float wold = weight[n];
    float wnew, scM;
    if (decay_fn == VSOM_EXPONENTIAL) {
        wnew = wold + (float)(h * eta);              // :924
        scM = (float)(h * eta);                      // double scalar narrowed before the fp32 product :925
    } else {
        wnew = wold + (float)h;                      // :930
        double tw = wnew == 0 ? 1.0 : h / (double)wnew;   // :933
        scM = (float)tw;
    }
    const double tw2 = wnew == 0 ? 0.000001 : (double)wnew;   // :939
    const float twf = (float)tw2, hf = (float)h;
    if (BLOCK_SYNC)
        __syncthreads();   // every thread has read weight[n]
    if (tid == 0)
        weight[n] = wnew;

    if (KIND == VSOM_CLR) {
        for (int p = tid; p < P; p += nthr) {
            const float x1 = xp[p], y1 = yp[p];
            float A = M[p], Bv = M[ppitch + p];
            float inner = A * x1;
            inner = inner + Bv;
            inner = inner - y1;
            float m2 = -2.f * inner;
            float dA = m2 * x1, dB = m2;                     // Stepper :912
            float tA = scM * dA, tB = scM * dB;
            A = A + tA;                                      // :925 / :935
            Bv = Bv + tB;
            float in2 = A * x1;
            in2 = in2 + Bv;
            in2 = in2 - y1;
            float n2 = -2.f * in2;
            float dA2 = n2 * x1, dB2 = n2;                   // Stepper(v, map_new) :941
            float pa = dA * dA2, pb = dB * dB2;
            float ua = hf * pa, ub = hf * pb;
            float SA = S[p] + ua, SB = S[ppitch + p] + ub;   // :941
            M[p] = A;
            M[ppitch + p] = Bv;
            S[p] = SA;
            S[ppitch + p] = SB;
            sg[p] = sqrtf(fabsf(SA / twf));                  // :942
            sg[ppitch + p] = sqrtf(fabsf(SB / twf));
        }
    } else {
        for (int d = tid, u = 0; d < D; d += nthr, ++u) {
            const float x = PRE && u < 4 ? px[u < 4 ? u : 0] : xs[d];
            float m = PRE && u < 4 ? pm[u < 4 ? u : 0] : M[d];
            const float s_old = PRE && u < 4 ? ps[u < 4 ? u : 0] : S[d];
            float dl = x - m;                                // Stepper :912
            if (KIND == VSOM_MEDIAN)
                dl = onl_sign(dl);
            float t = scM * dl;
            m = m + t;                                       // :925 / :935
            float dl2 = x - m;                               // Stepper(v, map_new) :941
            if (KIND == VSOM_MEDIAN)
                dl2 = onl_sign(dl2);
            float pr = dl * dl2;
            float uu = hf * pr;
            float s = s_old + uu;                            // :941
            onl_store<ST>(M + d, m);
            onl_store<ST>(S + d, s);
            if (SIG)
                onl_store<ST>(sg + d, sqrtf(fabsf(s / twf)));    // :942
            if (mkeep && u < 4)
                mkeep[u < 4 ? u : 0] = m;
        }
    }
}

// window bounds of Som.cpp:899-903 (truncating, asymmetric, Q6)
__device__ __forceinline__ void online_window(u64 bmu, int W, int H, double sigma, int &bx, int &by,
                                              u64 &startX, u64 &startY, u64 &endX, u64 &endY)
{
    bx = (int)(bmu % (u64)W);
    by = (int)(bmu / (u64)W);   // bmu.getX()/getY() (:306)
    const double sxd = fmax((double)bx - 2.5 * sigma, 0.), syd = fmax((double)by - 2.5 * sigma, 0.);
    const double exd = fmin((double)bx + 2.5 * sigma, (double)W), eyd = fmin((double)by + 2.5 * sigma, (double)H);
    startX = (u64)sxd;
    startY = (u64)syd;
    endX = (u64)exd;
    endY = (u64)eyd;
}

// residual / distance of the BMU after the update (:946), addBmu, MSE, lastBMU
template <bool CLR>
__device__ __forceinline__ void online_post(const OnlineArgs &a, const float *xa, const float *xb, u64 bmu, int lane, u64 *hits,
                                            u64 *lastbmu_out, float *residual, float fB, int add_hit)
{
    const int k = lane & 7;
    const float *ma = a.d.ma + (size_t)bmu * a.d.ldm, *mb = a.d.mb + (size_t)bmu * a.d.ldm;
    if (residual) {
        for (int d = lane; d < a.d.L; d += 64)
            residual[d] = vsom_resid<CLR>(xa[d], CLR ? xb[d] : 0.f, ma[d], CLR ? mb[d] : 0.f);
    }
    float dist = vsom_group_dist_lat<CLR>(xa, xb, ma, mb, a.d.L, k);
    if (lane == 0) {
        a.fstate[0] = dist;
        float q = dist / fB;                 // residual.squaredNorm() / epochSize  (:1167)
        a.fstate[1] = a.fstate[1] + q;
        if (add_hit)
            hits[bmu] += 1ull;               // addBmu (:1165, :1189-1192)
        *lastbmu_out = bmu;                  // lastBMU = by*W + bx (:895)
    }
}

// one workgroup per node of the (maximal) window; nodes outside the actual window exit.  The
// workgroup of the BMU node finishes the sample (post step) once its own update is visible.
template <int KIND, bool POST>
__global__ __launch_bounds__(256) void online_window_kernel(
    OnlineArgs a, const float *__restrict__ xs, const float *__restrict__ xp, const float *__restrict__ yp,
    const double *__restrict__ lutd, int lutw, int D, int P, int ppitch, int pitch, double eta, double sigma,
    int decay_fn, float *map, float *Smap, float *sigmap, float *weight, u64 *hits, u64 *lastbmu_out,
    float *residual, float fB, int add_hit)   // no __restrict__: a.d.ma aliases map
{
    constexpr bool CLR = KIND == VSOM_CLR;
    const u64 bmu = online_resolve(a.state, a.par);
    int bx, by;
    u64 startX, startY, endX, endY;
    online_window(bmu, a.W, a.H, sigma, bx, by, startX, startY, endX, endY);
    const u64 i = startX + blockIdx.x, j = startY + blockIdx.y;
    if (i >= endX || j >= endY)
        return;
    const size_t n = (size_t)(j * (u64)a.W + i);
    int dx = (int)i - bx, dy = (int)j - by;
    dx = dx < 0 ? -dx : dx;
    dy = dy < 0 ? -dy : dy;
    const double h = lutd[(size_t)dy * lutw + dx];   // calculateNeighbourhoodWeight(i,j,bx,by,sigma) :915
    online_node_update<KIND, true>(n, h, threadIdx.x, blockDim.x, xs, xp, yp, D, P, ppitch, pitch, eta, decay_fn,
                                   map, Smap, sigmap, weight);
    if (!POST || n != (size_t)bmu)   // (chunk loop: the next search launch finishes the sample; the BMU always lies inside
        return;                      //  its own window, sigma > 1)
    __syncthreads();             // this workgroup's writes of the BMU row are visible to its wave 0
    if (threadIdx.x < 64) {
        online_post<CLR>(a, a.d.xa, a.d.xb, bmu, threadIdx.x, hits, lastbmu_out, residual, fB, add_hit);
        if (threadIdx.x < ONL_SLOTS)
            *online_slot(a.state, a.par ^ 1, (int)threadIdx.x) = ~0ull;   // arm the next sample's key (nobody reads it during this launch)
    }
}

// sigma <= 1 (Som.cpp:891: findLocalBmu, indicator neighbourhood, window of at most 6x6 nodes):
// the whole trainSingle step in ONE small launch -- wave 0 walks the local search, each wave then
// updates window nodes round-robin, wave 0 finishes with the residual / bookkeeping.
template <int KIND>
__global__ __launch_bounds__(1024) void online_small_kernel(
    OnlineArgs a, const float *__restrict__ xs, const float *__restrict__ xp, const float *__restrict__ yp,
    const double *__restrict__ lutd, int lutw, int D, int P, int ppitch, int pitch, double eta, double sigma,
    int decay_fn, float *map, float *Smap, float *sigmap, float *weight, u64 *hits, u64 *lastbmu_io,
    float *residual, float fB, int add_hit)   // no __restrict__: a.d.ma aliases map, lastbmu_io is read and written
{
    constexpr bool CLR = KIND == VSOM_CLR;
    __shared__ u64 sbmu;
    __shared__ float sown;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // the walk's starting distance (wavefront 1) beside its first step's 8 neighbours (wavefront 0)
    float dfirst = 0.f;
    if (wave <= 1) {
        const u64 from = *lastbmu_io;
        const u64 node = wave == 0 ? online_first_try_node(from, lane >> 3, (u64)a.W, (u64)a.H) : from;
        const float d = vsom_group_dist_lat<CLR>(a.d.xa, a.d.xb, a.d.ma + (size_t)node * a.d.ldm,
                                                 a.d.mb + (size_t)node * a.d.ldm, a.d.L, lane & 7);
        if (wave == 0)
            dfirst = d;
        else if (lane == 0)
            sown = d;
    }
    __syncthreads();
    if (wave == 0) {
        const u64 b = online_local_search<CLR>(a, lastbmu_io, lane, &sown, dfirst);
        if (lane == 0)
            sbmu = b;
    }
    __syncthreads();
    const u64 bmu = sbmu;
    int bx, by;
    u64 startX, startY, endX, endY;
    online_window(bmu, a.W, a.H, sigma, bx, by, startX, startY, endX, endY);
    const u64 nx = endX > startX ? endX - startX : 0, ny = endY > startY ? endY - startY : 0;
    for (u64 idx = (u64)wave; idx < nx * ny; idx += (u64)(blockDim.x >> 6)) {
        const u64 i = startX + idx % nx, j = startY + idx / nx;
        int dx = (int)i - bx, dy = (int)j - by;
        dx = dx < 0 ? -dx : dx;
        dy = dy < 0 ? -dy : dy;
        const double h = lutd[(size_t)dy * lutw + dx];
        const size_t n = (size_t)(j * (u64)a.W + i);
        online_node_update<KIND, false>(n, h, lane, 64, xs, xp, yp, D, P, ppitch, pitch,
                                        eta, decay_fn, map, Smap, sigmap, weight);
        if (n == (size_t)bmu) {        // the wavefront that rewrote the BMU's row finishes the sample while
            __threadfence_block();     // the others update their nodes
            online_post<CLR>(a, a.d.xa, a.d.xb, bmu, lane, hits, lastbmu_io, residual, fB, add_hit);
        }
    }
    // sigma < 0.4 truncates the window to nothing (:899-903): the BMU is not updated, the sample still counts
    const bool inside = (u64)bx >= startX && (u64)bx < endX && (u64)by >= startY && (u64)by < endY;
    if (!inside && wave == 0)
        online_post<CLR>(a, a.d.xa, a.d.xb, bmu, lane, hits, lastbmu_io, residual, fB, add_hit);
}

__global__ void online_init_kernel(u64 *state, float *fstate, int keep_mse)
{
    for (int s = 0; s < 2 * ONL_SLOTS; ++s)
        state[s * 16] = ~0ull;
    state[ONL_FLAG] = 0ull;
    state[ONL_FLAG + 1] = 0ull;
    fstate[0] = 0.f;
    if (!keep_mse)
        fstate[1] = 0.f;   // else: the epoch's running MSE continues across chunks (Som.cpp:1153,1167)
}

// =====================================================================================================================
// Image-bounded search of the chunk loop (sigma > 1, Standard / Median, rows of at most 1024 values).
//
// The exact scan streams the fp32 map once per SAMPLE (4 N D bytes: 51 MB at 128 x 128 x 784).  Here the search reads a
// one-digit image of it instead -- one byte per model value, u = q + 128 with M_nk = s_n q_nk + r_nk, s_n a power of two,
// q in [-127, 127], plus four scalars per node -- and only PRUNES with it: every index and distance still comes from an
// exact-order evaluation (onl_refine_kernel: vsom_group_dist's arithmetic on the fp32 rows of the nodes that survive),
// so results are bit-identical to the exact scan's.  Two launches per sample, as before, but differently cut:
//
//   onl_fused_kernel    window update of sample j  ||  image scan for sample j + 1.  The window workgroups rewrite their
//                       node's fp32 rows exactly as online_window_kernel does, then re-digit the row they hold in
//                       registers and score it against sample j + 1; the scan workgroups score the nodes OUTSIDE the window
//                       from the image (8 lanes per node, 16 bytes per load, every load of a lane in flight at once).  A
//                       score is an interval [L_n, U_n] that provably contains the exact-order fp32 distance e_n (minus
//                       the sample's |x|^2): L_n goes to onl_lb[n], min_n U_n into 64 line-sized slots.
//   onl_refine_kernel   candidates {n : L_n <= min U} (the argmin of the exact-order distances is always one of them: its
//                       L is below its own e, which is below every e_m <= U_m), evaluated in the reference's order, argmin
//                       by the same (distance, index) key and node-0-NaN flag as online_scan_kernel; its extra workgroup
//                       finishes sample j - 1 (residual / distance after the update, addBmu, MSE, lastBMU).
//
// Bound.  u = 2^-24, K = columns.  Per node (stored by the digit passes): s_n, nM_n ~ |M_n|^2, eps_n = max_k |r_nk|,
// rho_n >= |r_n|_2; per sample (onl_prep_kernel): l1 >= |x|_1, nx ~ |x|^2, sx = sum x.  With T' the fp32 value of
// <x, q_n> (scan: fma chain of x_k * float(u_nk) over <= 128 elements per lane + 3 adds, then - 128 sx; |T' - <x,q>| <=
// 35000 u l1) and A_n = fma(-2 s_n, T', nM_n):
//     |A_n - (d_n - |x|^2)| <= w_n := 2 min(l1 eps_n, |x|_2 rho_n) + s_n l1 (2 * 35000 + 256) u + 25 u nM_n
//     |e_n - d_n| <= g2 d_n,  g2 = (K / 8 + 16) u        (exact-order evaluation: products + Eigen's sum tree)
//     e_n - |x|^2 in [A_n - slack_n, A_n + slack_n],  slack_n = 1.01 w_n + 1.05 g2 (max(A_n + nx, 0) + w_n) + 4 u |A_n| + 1e-30
// Anything non-finite on the way (a NaN / inf in the row or in the sample, an overflow) gives L = -inf, U = +inf: the
// node is always a candidate and never lowers the threshold -- a sample with a NaN then costs a full exact evaluation,
// spread over the refinement's workgroups.  Node 0 is always evaluated (Som.cpp:293-299: a NaN there pins the BMU).
#ifdef VSOM_DEVELOPMENT
// phase stamps of the two kernels (100 MHz wall clock), one writer per role: tools/exp/onl_stamps.py
__device__ unsigned long long vsom_onl_stamps[32];
#define ONL_STAMP(i) (vsom_onl_stamps[i] = wall_clock64())
extern "C" int vsom_dev_onl_stamps(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(vsom_onl_stamps), sizeof(vsom_onl_stamps)) == hipSuccess ? 0 : -2;
}
// entry / exit time of every workgroup of the last fused launch that had both roles
__device__ unsigned long long vsom_onl_trace[2 * 4096];
#define ONL_TRACE(slot) do { if (f.do_scan && f.do_window && threadIdx.x == 0 && blockIdx.x < 4096) vsom_onl_trace[2 * blockIdx.x + (slot)] = wall_clock64(); } while (0)
extern "C" int vsom_dev_onl_trace(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(vsom_onl_trace), sizeof(vsom_onl_trace)) == hipSuccess ? 0 : -2;
}
#else
#define ONL_STAMP(i) ((void)0)
#define ONL_TRACE(slot) ((void)0)
#endif
constexpr int ONL_USLOTS = 32;                           // min U: 32 line-sized slots per parity
constexpr size_t ONL_U_BYTES = 2 * ONL_USLOTS * 128;
constexpr int ONL_KSLOTS = 2;                            // exact keys of a refinement: ~40 workgroups have one, two slots do
constexpr int ONL_REF_NODES = 32;                        // nodes per refinement workgroup (one wavefront per candidate)

struct OnlI8 {
    unsigned char *img;      // [N][ipitch] u = q + 128 (pad bytes 128)
    float4 *nsc;             // [N] {s, nM, eps, rho}; s = NaN: row holds a non-finite value
    float *lb;               // [N] L_n of the sample being searched
    unsigned *uslots;        // [2][ONL_USLOTS][32]
    const float4 *xsc;       // [B] {l1, nx, sx, -}
    int ipitch, ni;          // ni = ipitch / 128: 16-byte pieces per lane
    int nref;                // refinement workgroups = ceil(N / 32): the layout of lb
    float cT, g2c;           // (2 * 35000 + 256) u ; 1.05 (K / 8 + 16) u
    u64 *stats;              // [0] samples searched, [1] nodes evaluated exactly, [2] refinement workgroups with work
    unsigned char *dirty;    // [N] the node's sigmaMap row is owed
};

// order-preserving key of a (signed, finite) float for the unsigned atomicMin; 0xFFFFFFFF = "no finite U yet"
__device__ __forceinline__ unsigned onl_fkey(float v)
{
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float onl_funkey(unsigned k)
{
    if (k == 0xFFFFFFFFu)
        return __uint_as_float(0x7F800000u);
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// [L, U] of a node from its approximate value A and the bound terms; non-finite -> (-inf, +inf)
__device__ __forceinline__ void onl_interval(float A, float s, float nM, float eps, float rho, float4 xs, float cT, float g2c,
                                             float &L, float &U)
{
    const float l1 = xs.x, nx = xs.y;
    const float x2 = sqrtf(nx * 1.0001f) * 1.0001f;
    const float w = (2.f * fminf(l1 * eps, x2 * rho) + s * l1 * cT + 1.5e-6f * nM) * 1.01f;     // 25 u = 1.49e-6
    const float dpos = fmaxf(A + nx * 1.0001f, 0.f);
    const float slack = w + g2c * (dpos + w) + 2.4e-7f * fabsf(A) + 1e-30f;
    const bool ok = slack < 3.0e38f && fabsf(A) < 3.0e38f;                                       // NaN fails both
    const float inf = __uint_as_float(0x7F800000u);
    L = ok ? A - slack : -inf;
    U = ok ? A + slack : inf;
}

// scale 2^E of a row whose largest finite magnitude is mx: |M| / s < 128 (rows below 2^-44 keep E = -50)
__device__ __forceinline__ void onl_row_scale(float mx, float &s, float &is)
{
    int e = mx > 0.f ? (int)((__float_as_uint(mx) >> 23) & 0xFF) - 127 : -100;
    e = mx > 0.f && ((__float_as_uint(mx) >> 23) & 0xFF) == 0 ? -126 : e;
    int E = e - 6;
    E = E < -50 ? -50 : E;
    s = __uint_as_float((unsigned)(E + 127) << 23);
    is = __uint_as_float((unsigned)(127 - E) << 23);
}

// digit of one value on the grid of scale s: q in [-127, 127]; r = m - q s is exact in fp32 (Sterbenz: q s and m are
// within a factor of two of each other whenever q != 0)
__device__ __forceinline__ void onl_digit(float m, float s, float is, float &q, float &r)
{
    const float mm = fabsf(m) <= 3.0e38f ? m : 0.f;
    float t = rintf(mm * is);
    t = fminf(fmaxf(t, -127.f), 127.f);
    q = t;
    r = mm - t * s;
}

// per chunk: the whole map -> image + node scalars.  One wavefront per node (rows of at most 1024 values: 16 per lane).
__global__ __launch_bounds__(256) void onl_digit_kernel(const float *__restrict__ map, int ldm, int D, int N, OnlI8 o)
{
    const int n = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N)
        return;
    const float *row = map + (size_t)n * ldm;
    float v[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int d = lane * 16 + 4 * j;                 // rows are zero padded to a multiple of 32 floats
        const float4 t = d < D ? *reinterpret_cast<const float4 *>(row + d) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[4 * j] = t.x;
        v[4 * j + 1] = t.y;
        v[4 * j + 2] = t.z;
        v[4 * j + 3] = t.w;
    }
    float mx = 0.f, nm = 0.f;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float a = fabsf(v[i]);
        const bool fin = a <= 3.0e38f;
        bad |= !fin;
        mx = (fin && a > mx) ? a : mx;
        nm = nm + v[i] * v[i];
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_xor(mx, off);
        mx = om > mx ? om : mx;
        nm += __shfl_xor(nm, off);
    }
    bad = __ballot(bad) != 0ull;
    float s, is;
    onl_row_scale(mx, s, is);
    float eps = 0.f, r2 = 0.f;
    unsigned w4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned pk = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float q, r;
            onl_digit(v[4 * j + i], s, is, q, r);
            const int d = lane * 16 + 4 * j + i;
            const unsigned ub = d < D ? (unsigned)((int)q + 128) : 128u;
            pk |= ub << (8 * i);
            const float ar = fabsf(r);
            eps = ar > eps ? ar : eps;
            r2 = r2 + r * r;
        }
        w4[j] = pk;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float oe = __shfl_xor(eps, off);
        eps = oe > eps ? oe : eps;
        r2 += __shfl_xor(r2, off);
    }
    if (lane * 16 < o.ipitch)
        *reinterpret_cast<uint4 *>(o.img + (size_t)n * o.ipitch + lane * 16) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    if (lane == 0)
        o.nsc[n] = make_float4(bad ? __uint_as_float(0x7FC00000u) : s, nm, eps, sqrtf(r2 * 1.0001f) * 1.0001f);
}

// per chunk: the bound terms of every sample.  One wavefront per sample.
__global__ __launch_bounds__(256) void onl_prep_kernel(const float *__restrict__ X, int ldx, int D, int B, float4 *__restrict__ xsc)
{
    const int s = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= B)
        return;
    const float *row = X + (size_t)s * ldx;
    float l1 = 0.f, nx = 0.f;
    double sx = 0.0;
    for (int d = lane; d < D; d += 64) {
        const float v = row[d];
        l1 += fabsf(v);
        nx += v * v;
        sx += (double)v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        l1 += __shfl_xor(l1, off);
        nx += __shfl_xor(nx, off);
        sx += __shfl_xor(sx, off);
    }
    if (lane == 0)
        xsc[s] = make_float4(l1 * 1.0001f, nx, (float)sx, 0.f);   // (a NaN / inf in the row makes l1 / nx non-finite: every interval opens)
}

// BMU of the sample of parity `par` from the refinement's key slots (whole wavefronts call this)
__device__ __forceinline__ u64 onl_resolve_i8(const u64 *state, int par)
{
    u64 key = state[(par * ONL_SLOTS + ((int)threadIdx.x & (ONL_KSLOTS - 1))) * 16];
    for (int off = ONL_KSLOTS / 2; off >= 1; off >>= 1) {
        const u64 o = __shfl_xor(key, off);
        key = o < key ? o : key;
    }
    return state[ONL_FLAG + par] ? 0ull : (key & 0xFFFFFFFFull);
}

// onl_lb is stored in the REFINEMENT's order: workgroup b of nref owns the nodes b + t nref (t < 32), and its 32 bounds sit
// side by side (one 128-byte line per workgroup instead of 32 lines 4 nref bytes apart)
__device__ __forceinline__ int onl_lb_index(int n, int nref) { return (n % nref) * ONL_REF_NODES + n / nref; }

// min U of the sample of parity `par` (whole wavefronts call this: one slot per lane)
__device__ __forceinline__ float onl_umin(const unsigned *uslots, int par)
{
    unsigned k = uslots[(par * ONL_USLOTS + ((int)threadIdx.x & (ONL_USLOTS - 1))) * 32];
    for (int off = ONL_USLOTS / 2; off > 0; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)k, off);
        k = o < k ? o : k;
    }
    return onl_funkey(k);
}

struct OnlFusedArgs {
    OnlineArgs a;            // a.d.xa = row of sample j (the window's sample), a.par = parity of sample j
    const float *xnext;      // row of sample j + 1 (the one being scored)
    int jnext;               // its index into xsc
    int do_window, do_scan;
    int nwin_x, nwin_y;      // maximal window extents = window workgroups (do_window)
    int nscan, passes;       // scan workgroups (do_scan), groups of 32 nodes each of them scores
};

template <int KIND, int ST, int PASSES>   // PASSES: groups of 32 nodes per scan workgroup (compile-time: the one-group form keeps 64 VGPRs)
__global__ __launch_bounds__(256, PASSES == 1 ? 8 : 1) void onl_fused_kernel(
    OnlFusedArgs f, OnlI8 o, const double *__restrict__ lutd, int lutw, int D, int pitch, double eta, double sigma, int decay_fn,
    float *map, float *Smap, float *sigmap, float *weight)   // no __restrict__: f.a.d.ma aliases map
{
    __shared__ __attribute__((aligned(16))) float s_x[1024];
    __shared__ float s_red[4][4];
    __shared__ unsigned s_u[4];
    const OnlineArgs &a = f.a;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    ONL_TRACE(0);
    // The scan workgroups come FIRST in dispatch order: their loads are in flight from the start, and a window larger than
    // one round of resident workgroups does not keep them waiting.  (Window first, 512 scan workgroups behind 1764 window
    // workgroups at sigma = 8: the chip holds 2048, the last 228 scan workgroups started when window workgroups ended and
    // the launch took the SUM of both roles, 11.3 us under the profiler against 5.1 + 5.7 alone.)
    const int nscan = f.do_scan ? f.nscan : 0;
    const int parn = a.par ^ 1;                          // parity of the sample being scored
    const float inf = __uint_as_float(0x7F800000u);
    if ((int)blockIdx.x >= nscan) {
        const int wb = (int)blockIdx.x - nscan;
        const bool st_ = tid == 0 && wb == (f.nwin_x * f.nwin_y) / 2 + f.nwin_x / 2;
        if (st_)
            ONL_STAMP(8);
        if (tid == 0 && blockIdx.x == gridDim.x - 1 && f.do_scan)
            ONL_STAMP(21);
        const float4 xs = o.xsc[f.do_scan ? f.jnext : 0];   // (uniform address, requested before anything depends on it)
        // ---- window role: one workgroup per node of the (maximal) window of sample j
        float xn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                    // (independent of the BMU: in flight beside the resolve)
            const int d = tid + u * 256;
            xn[u] = (f.do_scan && d < D) ? f.xnext[d] : 0.f;
        }
        const u64 bmu = onl_resolve_i8(a.state, a.par);
        if (st_)
            ONL_STAMP(9);
        int bx, by;
        u64 startX, startY, endX, endY;
        online_window(bmu, a.W, a.H, sigma, bx, by, startX, startY, endX, endY);
        const u64 i = startX + (u64)(wb % f.nwin_x), j = startY + (u64)(wb / f.nwin_x);
        if (i >= endX || j >= endY) {
            ONL_TRACE(1);
            return;
        }
        const size_t n = (size_t)(j * (u64)a.W + i);
        int dx = (int)i - bx, dy = (int)j - by;
        dx = dx < 0 ? -dx : dx;
        dy = dy < 0 ? -dy : dy;
        const double h = lutd[(size_t)dy * lutw + dx];   // calculateNeighbourhoodWeight(i,j,bx,by,sigma) :915
        float mk[4] = {0.f, 0.f, 0.f, 0.f};
        online_node_update<KIND, true, ST, false>(n, h, tid, 256, a.d.xa, nullptr, nullptr, D, 0, 0, pitch, eta, decay_fn,
                                                  map, Smap, sigmap, weight, mk);
        if (tid == 0)
            o.dirty[n] = 1;                              // sigmaMap[n] = sqrt(|S / w|) is owed (onl_sigma_kernel, end of the chunk)
        if (st_)
            ONL_STAMP(10);
        if (!f.do_scan)
            return;
        // re-digit the row this workgroup holds and score it against sample j + 1
        float mx = 0.f;
        bool bad = false;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float av = fabsf(mk[u]);
            const bool fin = av <= 3.0e38f;
            bad |= !fin;
            mx = (fin && av > mx) ? av : mx;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float om = __shfl_xor(mx, off);
            mx = om > mx ? om : mx;
        }
        bad = __ballot(bad) != 0ull;
        if (lane == 0) {
            s_red[wave][0] = mx;
            s_u[wave] = bad ? 1u : 0u;
        }
        __syncthreads();
        mx = fmaxf(fmaxf(s_red[0][0], s_red[1][0]), fmaxf(s_red[2][0], s_red[3][0]));
        bad = (s_u[0] | s_u[1] | s_u[2] | s_u[3]) != 0u;
        __syncthreads();
        float s, is;
        onl_row_scale(mx, s, is);
        float dot = 0.f, nm = 0.f, r2 = 0.f, eps = 0.f;
        unsigned char *irow = o.img + n * (size_t)o.ipitch;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int d = tid + u * 256;
            float q, r;
            onl_digit(mk[u], s, is, q, r);
            if (d < D) {
                irow[d] = (unsigned char)((int)q + 128);
                dot = fmaf(xn[u], q, dot);
                nm = nm + mk[u] * mk[u];
                r2 = r2 + r * r;
                const float ar = fabsf(r);
                eps = ar > eps ? ar : eps;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            dot += __shfl_xor(dot, off);
            nm += __shfl_xor(nm, off);
            r2 += __shfl_xor(r2, off);
            const float oe = __shfl_xor(eps, off);
            eps = oe > eps ? oe : eps;
        }
        if (lane == 0) {
            s_red[wave][0] = dot;
            s_red[wave][1] = nm;
            s_red[wave][2] = r2;
            s_red[wave][3] = eps;
        }
        __syncthreads();
        if (st_)
            ONL_STAMP(11);
        if (tid == 0) {
            dot = (s_red[0][0] + s_red[1][0]) + (s_red[2][0] + s_red[3][0]);
            nm = (s_red[0][1] + s_red[1][1]) + (s_red[2][1] + s_red[3][1]);
            r2 = (s_red[0][2] + s_red[1][2]) + (s_red[2][2] + s_red[3][2]);
            eps = fmaxf(fmaxf(s_red[0][3], s_red[1][3]), fmaxf(s_red[2][3], s_red[3][3]));
            const float rho = sqrtf(r2 * 1.0001f) * 1.0001f;
            const float sn = bad ? __uint_as_float(0x7FC00000u) : s;
            o.nsc[n] = make_float4(sn, nm, eps, rho);
            const float A = fmaf(-2.f * sn, dot, nm);
            float L, U;
            onl_interval(A, sn, nm, eps, rho, xs, o.cT, o.g2c, L, U);
            o.lb[onl_lb_index((int)n, o.nref)] = L;
            if (U < inf)
                atomicMin(&o.uslots[(parn * ONL_USLOTS + (wb & (ONL_USLOTS - 1))) * 32], onl_fkey(U));
            if (st_)
                ONL_STAMP(12);
            if (blockIdx.x == gridDim.x - 1)
                ONL_STAMP(22);
        }
        ONL_TRACE(1);
        return;
    }
    // ---- scan role: f.passes groups of 32 nodes per workgroup, one after the other; 8 lanes per node, from the image
    const int sb = (int)blockIdx.x;
    const bool st_ = tid == 0 && sb == nscan / 2;
    if (st_)
        ONL_STAMP(16);
    if (tid == 0 && sb == 0)
        ONL_STAMP(20);
    if (sb == 0 && tid < ONL_USLOTS)                     // nobody reads or writes the other parity's slots during this launch
        o.uslots[((a.par) * ONL_USLOTS + tid) * 32] = 0xFFFFFFFFu;
    const int k = tid & 7;
    {
        const int d = tid * 4;
        const float4 t = d < D ? *reinterpret_cast<const float4 *>(f.xnext + d) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(s_x + d) = t;        // rows are zero padded to a multiple of 32 floats
    }
    const float4 xs = o.xsc[f.jnext];
    u64 startX = 0, startY = 0, endX = 0, endY = 0;       // (no window: empty)
    uint4 pc[8];
    // (nodes of the window are scored by their own workgroups: their image rows are not even loaded.  Requesting the rows
    // BEFORE the window is known -- a tenth of them read for nothing at sigma = 8, the workgroup through one round trip
    // earlier -- measured no faster: 14.68-14.73 against 14.54-14.66 us per sample.)
    if (f.do_window) {
        const u64 bmu = onl_resolve_i8(a.state, a.par);
        int bx, by;
        online_window(bmu, a.W, a.H, sigma, bx, by, startX, startY, endX, endY);
    }
    const unsigned wx0 = (unsigned)startX, wx1 = (unsigned)endX, wy0 = (unsigned)startY, wy1 = (unsigned)endY, mapw = (unsigned)a.W;
    auto in_window = [&](int nc) {                       // (32-bit: the bounds are at most the map's sides)
        const unsigned ny_ = (unsigned)nc / mapw, nx_ = (unsigned)nc - ny_ * mapw;
        return nx_ >= wx0 && nx_ < wx1 && ny_ >= wy0 && ny_ < wy1;
    };
    auto issue = [&](int n_, bool live) {
        const unsigned char *irow = o.img + (size_t)(n_ < a.N ? n_ : a.N - 1) * o.ipitch + 16 * k;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (i < o.ni)                                // (the last round of a row is ragged: 16-byte pieces past it are not read)
                pc[i] = (live && 128 * i + 16 * k < o.ipitch) ? *reinterpret_cast<const uint4 *>(irow + 128 * i) : make_uint4(0u, 0u, 0u, 0u);
    };
    int node = sb * 32 * PASSES + (tid >> 3);
    issue(node, node < a.N && !in_window(node));
    __syncthreads();
    if (st_)
        ONL_STAMP(17);
    unsigned ukmin = 0xFFFFFFFFu;
#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass, node += 32) {
        const int nc = node < a.N ? node : a.N - 1;
        const float4 nsc = o.nsc[nc];
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < o.ni) {
                const float4 *xp = reinterpret_cast<const float4 *>(s_x + (i * 8 + k) * 16);
                const unsigned wv[4] = {pc[i].x, pc[i].y, pc[i].z, pc[i].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 xv = xp[c];
                    const unsigned w = wv[c];
                    t0 = fmaf(xv.x, (float)(w & 0xFFu), t0);
                    t1 = fmaf(xv.y, (float)((w >> 8) & 0xFFu), t1);
                    t0 = fmaf(xv.z, (float)((w >> 16) & 0xFFu), t0);
                    t1 = fmaf(xv.w, (float)(w >> 24), t1);
                }
            }
        }
        if (pass + 1 < PASSES)                           // the next group's rows travel while this one's interval is formed
            issue(node + 32, node + 32 < a.N && !in_window(node + 32));
        float T = t0 + t1;
        T += __shfl_xor(T, 1);
        T += __shfl_xor(T, 2);
        T += __shfl_xor(T, 4);
        const float Tq = fmaf(-128.f, xs.z, T);          // <x, q> = <x, u> - 128 sum x
        const float A = fmaf(-2.f * nsc.x, Tq, nsc.y);
        float L, U;
        onl_interval(A, nsc.x, nsc.y, nsc.z, nsc.w, xs, o.cT, o.g2c, L, U);
        const bool mine = node < a.N && !in_window(nc);
        if (mine && k == 0)
            o.lb[onl_lb_index(node, o.nref)] = L;
        const unsigned uk = (mine && U < inf) ? onl_fkey(U) : 0xFFFFFFFFu;
        ukmin = uk < ukmin ? uk : ukmin;
    }
    if (st_)
        ONL_STAMP(18);
    for (int off = 32; off >= 8; off >>= 1) {
        const unsigned ou = (unsigned)__shfl_xor((int)ukmin, off);
        ukmin = ou < ukmin ? ou : ukmin;
    }
    if (lane == 0)
        s_u[wave] = ukmin;
    __syncthreads();
    if (tid == 0) {
        const unsigned uk = min(min(s_u[0], s_u[1]), min(s_u[2], s_u[3]));
        if (uk != 0xFFFFFFFFu)
            atomicMin(&o.uslots[(parn * ONL_USLOTS + (sb & (ONL_USLOTS - 1))) * 32], uk);
    }
    ONL_TRACE(1);
}

// Exact-order distance of ONE (sample, node) pair by ONE wavefront.  The reference's sum runs eight sequential accumulator
// chains (Eigen's packet classes, vsom_group_dist) -- that part cannot be spread over more lanes -- but the loads and the
// squares can: the 64 lanes fetch the model row as float4 (at most four each: ONE memory round trip, coalesced), form
// p = fl(fl(m - x)^2) exactly as vsom_resid / vsom_group_dist do, and park the squares in LDS; eight lanes' worth of chains
// (every group of eight lanes runs the same ones) then only add LDS values in the reference's order.  8 lanes per
// candidate with 98 strided loads each took 7.9 us per pass through clamped 64-bit addresses and 3.2 us through a buffer
// descriptor (profiles/EXPERIMENTS.md); this form is bound by one round trip + 98 dependent additions.
//   mrow: the node's row (pitch a multiple of 32 floats, pad columns zero); x: the sample's row in LDS (1024 floats, zero
//   beyond L); sp: this wavefront's 1024-float LDS scratch.
__device__ __forceinline__ float onl_wave_dist(const float *__restrict__ mrow, const float *__restrict__ x, float *__restrict__ sp, int L,
                                               int lane)
{
    float4 mv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int d = (lane + 64 * t) * 4;
        mv[t] = d < L ? *reinterpret_cast<const float4 *>(mrow + d) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int d = (lane + 64 * t) * 4;
        if (d < L) {
            const float4 xv = *reinterpret_cast<const float4 *>(x + d);
            float4 r, p;
            r.x = mv[t].x - xv.x;
            r.y = mv[t].y - xv.y;
            r.z = mv[t].z - xv.z;
            r.w = mv[t].w - xv.w;
            p.x = r.x * r.x;
            p.y = r.y * r.y;
            p.z = r.z * r.z;
            p.w = r.w * r.w;
            *reinterpret_cast<float4 *>(sp + d) = p;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): this wavefront's LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    const int k = lane & 7, L8 = L & ~7, ni = L8 >> 3;
    const float *pk = sp + k;
    float acc = 0.f;
#pragma unroll 14
    for (int u = 0; u < ni; ++u)
        acc = acc + pk[8 * u];
    float q = acc + __shfl_xor(acc, 4);                  // p0_k + p1_k
    const int rem = L - L8;
    if (rem >= 4)
        q = q + sp[L8 + (k & 3)];
    const float t2 = q + __shfl_xor(q, 2);
    float res = t2 + __shfl_xor(t2, 1);
    for (int tt = (rem >= 4 ? 4 : 0); tt < rem; ++tt)
        res = res + sp[L8 + tt];
    __builtin_amdgcn_wave_barrier();                     // (the next evaluation overwrites sp)
    return res;
}

// refinement of sample j (a.par, row a.d.xa) + the post step of sample j - 1 (rows a.pxa) in the last workgroup
__global__ __launch_bounds__(256) void onl_refine_kernel(OnlineArgs a, OnlI8 o, int do_refine, u64 *hits, u64 *lastbmu_out, float fB,
                                                         int add_hit)
{
    __shared__ __attribute__((aligned(16))) float s_x[1024];
    __shared__ __attribute__((aligned(16))) float s_p[4][1024];
    __shared__ int s_list[ONL_REF_NODES];
    __shared__ int s_cnt;
    __shared__ u64 skey[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (blockIdx.x == gridDim.x - 1) {
        if (a.do_post && tid < 64) {
            if (tid == 0)
                ONL_STAMP(5);
            const u64 bmu = onl_resolve_i8(a.state, a.par ^ 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {                // the row of the sample being finished (this wavefront only uses it)
                const int d = (i * 64 + tid) * 4;
                const float4 t = d < a.d.L ? *reinterpret_cast<const float4 *>(a.pxa + d) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(s_x + d) = t;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            // residual / distance of the BMU after the update (:946), addBmu, MSE, lastBMU -- online_post's steps
            const float dist = onl_wave_dist(a.d.ma + (size_t)bmu * a.d.ldm, s_x, s_p[0], a.d.L, tid);
            if (tid == 0) {
                a.fstate[0] = dist;
                const float q = dist / fB;               // residual.squaredNorm() / epochSize  (:1167)
                a.fstate[1] = a.fstate[1] + q;
                if (add_hit)
                    hits[bmu] += 1ull;                   // addBmu (:1165, :1189-1192)
                *lastbmu_out = bmu;                      // lastBMU = by*W + bx (:895)
                ONL_STAMP(6);
            }
            if (tid < ONL_SLOTS)
                *online_slot(a.state, a.par ^ 1, tid) = ~0ull;    // the key set of the sample after this one
        }
        return;
    }
    if (!do_refine)
        return;
    // Workgroup b owns the nodes b + t * nref (t < 32), NOT 32 consecutive ones: the candidates of a sample are its BMU's
    // neighbours on the map -- consecutive indices -- and one wavefront evaluates one candidate at a time; with consecutive
    // ownership the BMU's own row segment put 20-30 candidates into one workgroup (5-8 rounds of 1.4 us while 500
    // workgroups sat idle: the next launch started 6 us after workgroup 0 was done).
    const int nref = (int)gridDim.x - 1;
    if (tid == 0 && blockIdx.x == 0)
        ONL_STAMP(0);
    {
        const int d = tid * 4;
        const float4 t = d < a.d.L ? *reinterpret_cast<const float4 *>(a.d.xa + d) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(s_x + d) = t;
    }
    if (wave == 0) {
        const float umin = onl_umin(o.uslots, a.par);
        bool cand = false;
        const int n = (int)blockIdx.x + lane * nref;
        if (lane < ONL_REF_NODES && n < a.N) {
            const float L = o.lb[(int)blockIdx.x * ONL_REF_NODES + lane];
            cand = !(L > umin) || n == 0;                // (node 0 seeds the reference's search, Som.cpp:293-299)
        }
        const u64 bm = __ballot(cand);
        if (cand)
            s_list[__popcll(bm & ((1ull << lane) - 1ull))] = n;
        if (lane == 0)
            s_cnt = __popcll(bm);
    }
    __syncthreads();
    const int cnt = s_cnt;
    if (cnt == 0)
        return;
    if (tid == 0) {          // diagnostics (vsom_get_online_search_stats): a few workgroups per sample get here
        atomicAdd(&o.stats[1], (u64)cnt);
        atomicAdd(&o.stats[2], 1ull);
        if (blockIdx.x == 0) {
            atomicAdd(&o.stats[0], 1ull);                // (node 0 is always a candidate of workgroup 0)
            ONL_STAMP(2);
        }
    }
    u64 best = ~0ull;
    for (int c = wave; c < cnt; c += 4) {                // one wavefront per candidate
        const int node = s_list[c];
        const float d = onl_wave_dist(a.d.ma + (size_t)node * a.d.ldm, s_x, s_p[wave], a.d.L, lane);
        const u64 key = vsom_key(d, (uint32_t)node);
        best = key < best ? key : best;
        if (node == 0 && lane == 0)
            a.state[ONL_FLAG + a.par] = (d != d) ? 1ull : 0ull;
    }
    if (tid == 0 && blockIdx.x == 0)
        ONL_STAMP(3);
    if (lane == 0)
        skey[wave] = best;
    __syncthreads();
    if (tid == 0) {
        u64 m = skey[0];
        for (int i = 1; i < 4; ++i)
            m = skey[i] < m ? skey[i] : m;
        if (m != ~0ull)
            atomicMin(online_slot(a.state, a.par, (int)(blockIdx.x % ONL_KSLOTS)), m);
        if (blockIdx.x == 0)
            ONL_STAMP(4);
    }
}

// End of a chunk: sigmaMap rows of the nodes its windows rewrote.  Som.cpp:942 writes sqrt(|S / w|) at every update of a
// node; nothing of the training reads sigmaMap, and S and weightMap change at the node's updates only, so the value after the
// node's LAST update -- computed here from the final S and weight with :939's and :942's operations -- is the same bits,
// for a fifth fewer bytes per window (5 of the 25 N_w D per sample).  Nodes no window touched keep what they had.
__global__ __launch_bounds__(256) void onl_sigma_kernel(const float *__restrict__ Smap, float *__restrict__ sigmap,
                                                        const float *__restrict__ weight, unsigned char *__restrict__ dirty, int D,
                                                        int pitch, int N)
{
    const int n = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N || !dirty[n])
        return;
    const float wnew = weight[n];
    const double tw2 = wnew == 0 ? 0.000001 : (double)wnew;   // :939
    const float twf = (float)tw2;
    const float *S = Smap + (size_t)n * pitch;
    float *sg = sigmap + (size_t)n * pitch;
    for (int d = lane; d < D; d += 64)
        sg[d] = sqrtf(fabsf(S[d] / twf));                // :942
    __builtin_amdgcn_wave_barrier();
    if (lane == 0)
        dirty[n] = 0;
}

__global__ void onl_uslots_init_kernel(unsigned *uslots)
{
    const int t = threadIdx.x;
    if (t < 2 * ONL_USLOTS)
        uslots[t * 32] = 0xFFFFFFFFu;
}

// double-precision table of calculateNeighbourhoodWeight for the online path (the batch path uses its float cast), cached per
// sigma.  The schedule changes sigma every epoch (Som.cpp:1146), so a training run builds one table per epoch: the host image
// goes into one of two pinned slots and from there to the device by a copy ENQUEUED on the stream -- ordered behind the
// kernels that still read the previous table, no stream wait, no pageable staging (a synchronous copy was ~20 us of the
// 10 x 10 x 9 scenario's epoch).
static int ensure_lutd_host(vsom_ctx *c, double sigma, const double **out, int *slot_out)
{
    const uint32_t lw = c->W, lh = c->H;
    const size_t need = (size_t)lw * lh;
    if (need > c->lutd_host_cap) {
        VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->lutd_host)
            VSOM_HIP_CHECK(hipHostFree(c->lutd_host));
        c->lutd_host = nullptr;
        c->lutd_host_cap = 0;
        VSOM_HIP_CHECK(hipHostMalloc(&c->lutd_host, 2 * need * sizeof(double)));
        c->lutd_host_cap = need;
        c->lutd_host_sigma[0] = c->lutd_host_sigma[1] = -1.0;
        c->lutd_ev_valid[0] = c->lutd_ev_valid[1] = false;
    }
    for (int k = 0; k < 2; ++k)
        if (c->lutd_host_sigma[k] == sigma) {
            *out = c->lutd_host + (size_t)k * c->lutd_host_cap;
            *slot_out = k;
            return VSOM_OK;
        }
    const int k = c->lutd_slot ^= 1;
    if (!c->lutd_ev[k])
        VSOM_HIP_CHECK(hipEventCreateWithFlags(&c->lutd_ev[k], hipEventDisableTiming));
    if (c->lutd_ev_valid[k])
        VSOM_HIP_CHECK(hipEventSynchronize(c->lutd_ev[k]));      // its last reader (two tables ago): long done
    double *host = c->lutd_host + (size_t)k * c->lutd_host_cap;
    c->lutd_host_sigma[k] = -1.0;
    for (uint32_t dy = 0; dy < lh; ++dy)
        for (uint32_t dx = 0; dx < lw; ++dx)
            host[(size_t)dy * lw + dx] = vsom_neighbourhood_weight(dx, dy, 0, 0, sigma);
    c->lutd_host_sigma[k] = sigma;
    *out = host;
    *slot_out = k;
    return VSOM_OK;
}

// (behind the last enqueued reader of host slot k)
static int lutd_host_used(vsom_ctx *c, int k)
{
    VSOM_HIP_CHECK(hipEventRecord(c->lutd_ev[k], c->stream));
    c->lutd_ev_valid[k] = true;
    return VSOM_OK;
}

static int ensure_lutd(vsom_ctx *c, double sigma, const double **out, int *lutw)
{
    const uint32_t lw = c->W, lh = c->H;
    const size_t need = (size_t)lw * lh;
    *lutw = (int)lw;
    if (c->lutd && c->lutd_sigma == sigma) {
        *out = c->lutd;
        return VSOM_OK;
    }
    if (need > c->lutd_cap) {
        VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->lutd)
            VSOM_HIP_CHECK(hipFree(c->lutd));
        c->lutd = nullptr;
        c->lutd_cap = 0;
        VSOM_HIP_CHECK(hipMalloc(&c->lutd, need * sizeof(double)));
        c->lutd_cap = need;
    }
    const double *host = nullptr;
    int k = 0;
    if (int rc = ensure_lutd_host(c, sigma, &host, &k))
        return rc;
    c->lutd_sigma = -1.0;
    VSOM_HIP_CHECK(hipMemcpyAsync(c->lutd, host, need * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (int rc = lutd_host_used(c, k))
        return rc;
    c->lutd_sigma = sigma;
    *out = c->lutd;
    return VSOM_OK;
}

// enqueue one trainSingle on sample rows (xs / xp / yp), lastBMU in/out at `lastbmu_dev`
// chunk = true: the chunk loop's pipelined form (sigma > 1) -- this search launch also finishes the PREVIOUS sample
// (rows pxs / pxp / pyp, lastBMU out at lastbmu_dev - 1; j = 0: nothing to finish) and the window launch leaves its own
// sample unfinished; the caller ends the chunk with enqueue_chunk_tail
static int enqueue_single(vsom_ctx *c, const float *xs, const float *xp, const float *yp,
                          double eta, double sigma, int decay_fn, u64 *lastbmu_dev,
                          float *residual_dev, float fB, int add_hit, const double *lutd, int lutw, int par,
                          float *fstate = nullptr, bool chunk = false, const float *pxs = nullptr, const float *pxp = nullptr,
                          const float *pyp = nullptr)
{
    OnlineArgs a;
    a.par = par & 1;
    const bool clr = c->transform == VSOM_CLR;
    a.pxa = clr ? pxp : pxs;
    a.pxb = clr ? pyp : pxs;
    a.do_scan = 1;
    a.do_post = chunk && pxs != nullptr;
    a.d.xa = clr ? xp : xs;
    a.d.xb = clr ? yp : xs;
    a.d.ldx = 0;
    a.d.ma = c->map;
    a.d.mb = clr ? c->map + c->part_pitch : c->map;
    a.d.ldm = (int)c->pitch;
    a.d.L = (int)c->part_len;
    a.state = c->onl_state;
    a.fstate = fstate ? fstate : c->onl_f;   // {distance of the BMU, MSE running sum}
    a.N = (int)c->N;
    a.W = (int)c->W;
    a.H = (int)c->H;

    if (sigma > 1) {   // SIGMA_SWITCH_TO_LOCAL (SOM.hpp:37, Som.cpp:891)
        dim3 grid((unsigned)(((size_t)c->N * 8 + 255) / 256) + (chunk ? 1u : 0u));   // + the workgroup that finishes sample j-1
        u64 *lb_prev = chunk && lastbmu_dev ? lastbmu_dev - 1 : nullptr;
        if (chunk) {
            if (clr)
                hipLaunchKernelGGL((online_scan_kernel<true, true>), grid, dim3(256), 0, c->stream, a, c->hits, lb_prev, fB, add_hit);
            else
                hipLaunchKernelGGL((online_scan_kernel<false, true>), grid, dim3(256), 0, c->stream, a, c->hits, lb_prev, fB, add_hit);
        } else {
            if (clr)
                hipLaunchKernelGGL((online_scan_kernel<true, false>), grid, dim3(256), 0, c->stream, a, c->hits, lb_prev, fB, add_hit);
            else
                hipLaunchKernelGGL((online_scan_kernel<false, false>), grid, dim3(256), 0, c->stream, a, c->hits, lb_prev, fB, add_hit);
        }
    } else {
        // sigma <= 1: one fused launch (local search + <=6x6 window + post)
#define LAUNCH_SMALL(KIND)                                                                                  \
    hipLaunchKernelGGL(online_small_kernel<KIND>, dim3(1), dim3(1024), 0, c->stream, a, xs, xp, yp, lutd, lutw, \
                       (int)c->D, (int)c->part_len, (int)c->part_pitch, (int)c->pitch, eta, sigma, decay_fn, \
                       c->map, c->S, c->sigma, c->weight, c->hits, lastbmu_dev, residual_dev, fB, add_hit)
        if (c->transform == VSOM_CLR)
            LAUNCH_SMALL(VSOM_CLR);
        else if (c->transform == VSOM_MEDIAN)
            LAUNCH_SMALL(VSOM_MEDIAN);
        else
            LAUNCH_SMALL(VSOM_STANDARD);
#undef LAUNCH_SMALL
        return VSOM_OK;
    }
    // maximal window extents: trunc(b+2.5s) - trunc(b-2.5s) <= floor(5s)+1, clipped to the map
    double ext = std::floor(5.0 * sigma) + 2.0;
    unsigned gx = (unsigned)std::min<double>((double)c->W, ext), gy = (unsigned)std::min<double>((double)c->H, ext);
    dim3 wgrid(gx ? gx : 1, gy ? gy : 1);
    const int L = (int)c->part_len;
    int bs = L >= 256 ? 256 : ((L + 63) / 64) * 64;
    if (bs < 64)
        bs = 64;
#define LAUNCH_WIN(KIND, POST)                                                                              \
    hipLaunchKernelGGL((online_window_kernel<KIND, POST>), wgrid, dim3(bs), 0, c->stream, a, xs, xp, yp, lutd, lutw, \
                       (int)c->D, (int)c->part_len, (int)c->part_pitch, (int)c->pitch, eta, sigma, decay_fn, \
                       c->map, c->S, c->sigma, c->weight, c->hits, lastbmu_dev, residual_dev, fB, add_hit)
    if (chunk) {
        if (c->transform == VSOM_CLR)
            LAUNCH_WIN(VSOM_CLR, false);
        else if (c->transform == VSOM_MEDIAN)
            LAUNCH_WIN(VSOM_MEDIAN, false);
        else
            LAUNCH_WIN(VSOM_STANDARD, false);
    } else {
        if (c->transform == VSOM_CLR)
            LAUNCH_WIN(VSOM_CLR, true);
        else if (c->transform == VSOM_MEDIAN)
            LAUNCH_WIN(VSOM_MEDIAN, true);
        else
            LAUNCH_WIN(VSOM_STANDARD, true);
    }
#undef LAUNCH_WIN
    return VSOM_OK;
}

// end of a pipelined chunk (sigma > 1): finish its last sample (rows pxs / pxp / pyp, parity par_last); no search
static int enqueue_chunk_tail(vsom_ctx *c, const float *pxs, const float *pxp, const float *pyp, u64 *lastbmu_last, float fB,
                              int par_last)
{
    OnlineArgs a;
    const bool clr = c->transform == VSOM_CLR;
    a.par = (par_last & 1) ^ 1;      // the post workgroup works on parity par ^ 1
    a.d.xa = a.pxa = clr ? pxp : pxs;
    a.d.xb = a.pxb = clr ? pyp : pxs;
    a.d.ldx = 0;
    a.d.ma = c->map;
    a.d.mb = clr ? c->map + c->part_pitch : c->map;
    a.d.ldm = (int)c->pitch;
    a.d.L = (int)c->part_len;
    a.state = c->onl_state;
    a.fstate = c->onl_f;
    a.N = (int)c->N;
    a.W = (int)c->W;
    a.H = (int)c->H;
    a.do_scan = 0;
    a.do_post = 1;
    if (clr)
        hipLaunchKernelGGL((online_scan_kernel<true, true>), dim3(1), dim3(256), 0, c->stream, a, c->hits, lastbmu_last, fB, 1);
    else
        hipLaunchKernelGGL((online_scan_kernel<false, true>), dim3(1), dim3(256), 0, c->stream, a, c->hits, lastbmu_last, fB, 1);
    return VSOM_OK;
}

// one host vector -> the single-sample device rows [xs | xp | yp | residual] (v_dev), through a pinned
// host buffer; CLR expands x'/y' (Transformation.cpp:94-101).  The copy is enqueued on the stream.
// copy = false: the rows stay in the pinned buffer only -- single-wavefront kernels read them from there (below)
static int stage_single(vsom_ctx *c, const float *v_host, bool copy = true)
{
    const size_t xs_n = c->xpitch, pp = c->part_pitch;
    const size_t nstage = xs_n + 2 * pp;
    if (!c->v_dev) {
        // device: [xs | xp | yp | residual(pp) | tail(16)]; tail = {u64 lastBMU/bmu, float dist, float mse}
        VSOM_HIP_CHECK(hipMalloc(&c->v_dev, (xs_n + 3 * pp + 16) * sizeof(float)));
        VSOM_HIP_CHECK(hipMalloc(&c->res_dev, 64));
        // pinned: the same rows, 32 floats of tails, and an image of onl_state for vsom_find_bmu
        VSOM_HIP_CHECK(hipHostMalloc(&c->v_pinned, (xs_n + 3 * pp + 32) * sizeof(float) + ONL_STATE_BYTES));
    } else {
        // the previous call's copy out of the pinned buffer has been waited for (every caller
        // synchronises the stream before returning)
    }
    float *host = c->v_pinned;
    std::fill(host, host + nstage, 0.f);
    for (uint32_t d = 0; d < c->J; ++d)
        host[d] = v_host[d];
    if (c->transform == VSOM_CLR) {
        size_t p = 0;
        for (uint32_t i = 0; i < c->J; ++i)
            for (uint32_t j = i + 1; j < c->J; ++j) {
                host[xs_n + p] = v_host[i];
                host[xs_n + pp + p] = v_host[j];
                ++p;
            }
    }
    if (copy)
        VSOM_HIP_CHECK(hipMemcpyAsync(c->v_dev, host, nstage * sizeof(float), hipMemcpyHostToDevice, c->stream));
    return VSOM_OK;
}

// =====================================================================================================================
// Tiny maps: the whole chunk loop of Som::trainBasicSom (Som.cpp:1159-1171) in ONE launch of ONE workgroup.
//
// The reference's own performance scenario trains a 10 x 10 map on 20 nine-dimensional rows (tests/performance/
// perf_tests.cpp:74-112): two dependent launches per sample were 220-260 us per epoch there against 50 us of one CPU thread.
// For maps of at most 4096 values (N D) and 1024 nodes, Standard / Median: every thread owns U = 1, 2 or 4 model values
// (M, S) IN REGISTERS for the whole chunk, plus a copy of its node's weight -- the window update of Som.cpp:911-943 is
// elementwise, so nothing of it crosses threads -- and per sample the workgroup meets at two barriers:
//   B  one thread per node adds its row's squares in Eigen's order (the eight accumulator classes, the tree, the tail:
//      vsom_group_dist's arithmetic); per wavefront a DPP minimum of the distances' bit patterns, the lowest lane that holds
//      it (strict <: the lowest index wins), one LDS atomic minimum of the (distance, index) key.  Meanwhile the LAST thread
//      finishes the PREVIOUS sample: the distance after the update (:946), the MSE running sum (:1167), addBmu, lastBMU.
//      sigma <= 1 (:891): the distances go to LDS and, behind one more barrier, one thread walks findLocalBmu over them.
//   -- barrier --
//   C  every thread reads the BMU, applies online_window's bounds (a per-BMU table) and the neighbourhood table and updates
//      its values with online_node_update's operations (same expressions, same order: same bits)
//   D  the BMU's threads publish their new squares (double-buffered: read by the post step beside the next sample's B)
//   A  squares p = fl(fl(m - x)^2) of the NEXT sample against every value, into LDS
//   -- barrier --
// sigmaMap is written once at the end, from the final S and weight, for the nodes some window touched (the value of the
// node's last update: onl_sigma_kernel's argument).  Results are bit-identical to the per-sample kernels and the oracle.
// The kernel is bound by instruction issue (16 wavefronts on 4 SIMDs) and LDS latency: 0.95 us per sample at 10 x 10 x 9.
struct OnlTinyArgs {
    const float *X;          // staged rows
    int ldx, B, N, W, H, D, pitch;
    float *map, *Smap, *sigmap, *weight;
    u64 *hits, *lastbmu;
    float *fstate;           // [0] distance of the last sample's BMU, [1] MSE running sum (in/out)
    const double *lutd;
    int lutw;
    double eta, sigma;
    int decay_fn;
    float fB;
    int keep_mse;            // 0: first chunk of an epoch, the running MSE starts at 0 (online_init_kernel's argument)
    float *mse_out;          // vsom_get_mse's value
    u64 *lastbmu_host;       // pinned host copy of lastBMU (vsom_train_online_chunk_fetch), or NULL
};

#ifdef VSOM_DEVELOPMENT
// cycle stamps of one sample of the one-launch chunk (sample 5 of the chunk; s_memtime): [2 w + p] = wavefront w (0: first,
// 1: last) at phase boundary p (tools/exp/tiny_stamps.py)
__device__ unsigned long long vsom_tiny_stamps[64];
#define TINY_STAMP(p) do { if (j == 5 && (tid == 0 || tid == 1023)) vsom_tiny_stamps[(tid ? 16 : 0) + (p)] = __builtin_readcyclecounter(); } while (0)
extern "C" int vsom_dev_tiny_stamps(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(vsom_tiny_stamps), sizeof(vsom_tiny_stamps)) == hipSuccess ? 0 : -2;
}
#else
#define TINY_STAMP(p) ((void)0)
#endif
constexpr int TINY_XBLOCK = 2048;      // values of staged rows held in LDS at a time (TINY_XBLOCK / D samples)

// minimum of a u64 over the wavefront, in lane 63: four row shifts, two row broadcasts (DPP moves of both halves; the
// cross-lane LDS permutes of a shuffle ladder were a third of a tiny-map sample)
__device__ __forceinline__ u64 onl_wave_min_u64(u64 v)
{
#define ONL_DPP_STEP(CTRL, ROWS)                                                                                         \
    {                                                                                                                    \
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)(unsigned)v, CTRL, ROWS, 0xF, false);         \
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)(unsigned)(v >> 32), CTRL, ROWS, 0xF, false); \
        const u64 o = (u64)hi << 32 | lo;                                                                                \
        v = o < v ? o : v;                                                                                               \
    }
    ONL_DPP_STEP(0x111, 0xF)   // row_shr:1
    ONL_DPP_STEP(0x112, 0xF)   // row_shr:2
    ONL_DPP_STEP(0x114, 0xF)   // row_shr:4
    ONL_DPP_STEP(0x118, 0xF)   // row_shr:8   (lane 15 of every row: the row's minimum)
    ONL_DPP_STEP(0x142, 0xA)   // row_bcast:15 into rows 1 and 3
    ONL_DPP_STEP(0x143, 0xC)   // row_bcast:31 into rows 2 and 3
#undef ONL_DPP_STEP
    return v;
}

// minimum of a u32 over the wavefront, in lane 63 (same ladder, one v_min_u32 with a DPP operand per step)
__device__ __forceinline__ unsigned onl_wave_min_u32(unsigned v)
{
#define ONL_DPP_STEP32(CTRL, ROWS)                                                                  \
    {                                                                                               \
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROWS, 0xF, false); \
        v = o < v ? o : v;                                                                          \
    }
    ONL_DPP_STEP32(0x111, 0xF)
    ONL_DPP_STEP32(0x112, 0xF)
    ONL_DPP_STEP32(0x114, 0xF)
    ONL_DPP_STEP32(0x118, 0xF)
    ONL_DPP_STEP32(0x142, 0xA)
    ONL_DPP_STEP32(0x143, 0xC)
#undef ONL_DPP_STEP32
    return v;
}

struct OnlTinyNode {           // what a thread needs to know about the sample's BMU: its coordinates and window (Som.cpp:899-907)
    unsigned short bx, by, startX, endX, startY, endY, pad0, pad1;
};

// Som::findLocalBmu (Som.cpp:335-454) over a table of the sample's distances to every node, by one thread: vsom_local_walk's
// steps and comparisons in its order (the 8 neighbours of the first try, then 3 nodes ahead while moving in X; the size_t
// wrap-then-clamp of :362-385 in 32 bits -- a wrapped coordinate is far above width - 1 either way)
__device__ __forceinline__ int onl_tiny_walk(const float *s_d, const OnlTinyNode *s_win, unsigned W, unsigned H, unsigned start)
{
    // (the coordinates of lastMeasured / lastBMU / the running minimum travel in registers: every candidate is built from its
    //  coordinates, so a step is ONE LDS round trip -- the candidates' distances, requested together, compared in order)
    unsigned lastBMU = start, minIndex = start, lastMeasured = start;
    unsigned lmX = s_win[start].bx, lmY = s_win[start].by, lbX = lmX, minX = lmX, minY = lmY;
    float minDist = s_d[start];
    for (;;) {
        if (lastMeasured == lastBMU) {
            unsigned node[8], cxs[8], cys[8];
            float di[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const unsigned fsx = (g == 0 || g >= 6) ? ~0u : ((g == 1 || g == 5) ? 0u : 1u);   // firstSearchX / Y (:341-342)
                const unsigned fsy = g <= 2 ? 1u : ((g == 3 || g == 7) ? 0u : ~0u);
                unsigned cx = lmX + fsx, cy = lmY + fsy;
                cx = cx < W - 1 ? cx : W - 1;
                cy = cy < H - 1 ? cy : H - 1;
                cxs[g] = cx;
                cys[g] = cy;
                node[g] = cy * W + cx;
                di[g] = s_d[node[g]];
            }
#pragma unroll
            for (int g = 0; g < 8; ++g)
                if (di[g] < minDist) {
                    minDist = di[g];
                    minIndex = node[g];
                    minX = cxs[g];
                    minY = cys[g];
                }
            if (minIndex == lastBMU)
                break;
            lastMeasured = minIndex;
            lmX = minX;
            lmY = minY;
        } else {
            if (lmX - lbX) {                                     // moving in X: 3 nodes ahead (:390-403)
                unsigned cx = lmX + lmX - lbX;
                cx = cx < W - 1 ? cx : W - 1;
                unsigned node[3], cys[3];
                float di[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    unsigned cy = lmY + (unsigned)(i - 1);
                    cy = cy < H - 1 ? cy : H - 1;
                    cys[i] = cy;
                    node[i] = cy * W + cx;
                    di[i] = s_d[node[i]];
                }
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    if (di[i] < minDist) {
                        minDist = di[i];
                        minIndex = node[i];
                        minX = cx;
                        minY = cys[i];
                    }
            }
            // (moving in Y evaluates nothing: :406-437, vsom_local_walk)
            if (minIndex == lastMeasured)
                break;
            lastBMU = lastMeasured;
            lbX = lmX;
            lastMeasured = minIndex;
            lmX = minX;
            lmY = minY;
        }
    }
    return (int)minIndex;
}

// a row of squares added in Eigen's order: the eight accumulator classes, the packet tree, the scalar tail (vsom_group_dist's
// arithmetic by one thread)
__device__ __forceinline__ float onl_tiny_row_sum(const float *p, int L8, int rem)
{
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < L8; d += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            acc[k] = acc[k] + p[d + k];
    }
    float q0 = acc[0] + acc[4], q1 = acc[1] + acc[5], q2 = acc[2] + acc[6], q3 = acc[3] + acc[7];
    int t = 0;
    if (rem >= 4) {
        q0 = q0 + p[L8];
        q1 = q1 + p[L8 + 1];
        q2 = q2 + p[L8 + 2];
        q3 = q3 + p[L8 + 3];
        t = 4;
    }
    const float t02 = q0 + q2, t13 = q1 + q3;
    float res = t02 + t13;
    for (; t < rem; ++t)
        res = res + p[L8 + t];
    return res;
}

template <int KIND, bool LOCAL, int U>     // U = values per thread (1, 2, 4): N D <= 1024 U
__global__ __launch_bounds__(1024) void online_tiny_chunk_kernel(OnlTinyArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char tiny_onl_smem[];
    const int N = a.N, D = a.D, ND = N * D, tid = threadIdx.x, Dp = D | 1;      // (odd row pitch: phase B's rows hit distinct banks)
    // LDS: per-BMU window table, the neighbourhood table (double for :933's division; its two float images for :924-925),
    // the squares, a block of staged rows, the chunk's addBmu counts and lastBMU
    OnlTinyNode *s_win = reinterpret_cast<OnlTinyNode *>(tiny_onl_smem);        // [N]
    double *s_lut = reinterpret_cast<double *>(s_win + N);                      // [W * H]
    float2 *s_lf = reinterpret_cast<float2 *>(s_lut + N);                       // [W * H]  {(float)(h eta), (float)h}
    float *s_p = reinterpret_cast<float *>(s_lf + N);                           // [N Dp] squares of the sample against every node
    float *s_p2 = s_p + N * Dp;                                                 // [2][D]  squares of the BMU's row after its update
    float *s_x = s_p2 + 2 * ((D + 3) & ~3);                                     // [TINY_XBLOCK] staged rows of this block
    unsigned *s_hits = reinterpret_cast<unsigned *>(s_x + TINY_XBLOCK);         // [N]  addBmu counts of this chunk
    float *s_d = reinterpret_cast<float *>(s_hits + N);                         // [N]  LOCAL: the sample's distances
    unsigned short *s_last = reinterpret_cast<unsigned short *>(s_d + N);       // [B]  lastBMU of every sample (a global store
                                                                                //      per sample made its wavefront wait for it)
    __shared__ u64 s_key[2];
    if (LOCAL)                                               // findLocalBmu starts from the sample's BMU of the last epoch (:891)
        for (int i = tid; i < a.B; i += 1024)
            s_last[i] = (unsigned short)a.lastbmu[i];
    for (int i = tid; i < N; i += 1024) {
        const double h = a.lutd[(size_t)(i / a.W) * a.lutw + (i % a.W)];
        s_lut[i] = h;
        s_lf[i] = make_float2((float)(h * a.eta), (float)h);
        s_hits[i] = 0u;
        int bx, by;
        u64 startX, startY, endX, endY;
        online_window((u64)i, a.W, a.H, a.sigma, bx, by, startX, startY, endX, endY);   // (ends <= W, H <= 1024)
        OnlTinyNode t;
        t.bx = (unsigned short)bx;
        t.by = (unsigned short)by;
        t.startX = (unsigned short)startX;
        t.endX = (unsigned short)endX;
        t.startY = (unsigned short)startY;
        t.endY = (unsigned short)endY;
        t.pad0 = t.pad1 = 0;
        s_win[i] = t;
    }
    // this thread's values: flattened (node, dim) indices tid + 1024 u
    float m[U], sv[U], w[U];
    int node[U], dim[U], nx[U], ny[U];
    bool own[U], touched[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = tid + 1024 * u;
        own[u] = e < ND;
        node[u] = own[u] ? e / D : 0;
        dim[u] = own[u] ? e - node[u] * D : 0;
        nx[u] = node[u] % a.W;
        ny[u] = node[u] / a.W;
        m[u] = own[u] ? a.map[(size_t)node[u] * a.pitch + dim[u]] : 0.f;
        sv[u] = own[u] ? a.Smap[(size_t)node[u] * a.pitch + dim[u]] : 0.f;
        w[u] = own[u] ? a.weight[node[u]] : 0.f;
        touched[u] = false;
    }
    float mse = a.keep_mse ? a.fstate[1] : 0.f, lastdist = 0.f;
    if (tid < 2)
        s_key[tid] = ~0ull;
    const int L8 = D & ~7, rem = D - L8, KB = TINY_XBLOCK / D;
    int pj = -1, pbmu = 0;                                   // the sample whose post step is owed, and its BMU
    // post step of a sample (one thread): the distance of its BMU after the update (:946) from the squares the BMU's threads
    // left in that sample's parity, the MSE running sum (:1167), addBmu (:1165), lastBMU (:895); re-arms the parity's key
    auto post = [&](int sj, int sbmu) {
        const int ppar = sj & 1;
        const float res = onl_tiny_row_sum(s_p2 + ppar * ((D + 3) & ~3), L8, rem);
        lastdist = res;
        const float q = res / a.fB;                          // residual.squaredNorm() / epochSize  (:1167)
        mse = mse + q;
        atomicAdd(&s_hits[sbmu], 1u);
        s_last[sj] = (unsigned short)sbmu;
        s_key[ppar] = ~0ull;
    };
    const bool exp_decay = a.decay_fn == VSOM_EXPONENTIAL;
    for (int j0 = 0; j0 < a.B; j0 += KB) {
        const int kb = min(KB, a.B - j0);
        if (j0 == 0)
            __syncthreads();                                 // the tables (later blocks: the sample loop's last barrier -- nobody reads the old rows)
        for (int i = tid; i < kb * D; i += 1024)
            s_x[i] = a.X[(size_t)(j0 + i / D) * a.ldx + (i % D)];
        __syncthreads();
        // A (first sample of the block): squares of the sample against every node
        float x[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (own[u]) {
                x[u] = s_x[dim[u]];
                const float r = m[u] - x[u];
                s_p[node[u] * Dp + dim[u]] = r * r;
            }
        __syncthreads();
        for (int jj = 0; jj < kb; ++jj) {
            const int j = j0 + jj, par = j & 1;
            TINY_STAMP(2);
            // post step of the PREVIOUS sample (distance after the update :946, MSE :1167, addBmu :1165, lastBMU :895) by the last
            // thread, in the shadow of phase B -- its wavefront has nothing to do there on maps of at most 960 nodes.  It reads
            // the previous parity's squares and re-arms that parity's key, which nobody touches before sample j + 1's phase B.
            if (tid == 1023 && pj >= 0)
                post(pj, pbmu);
            // B: distances in Eigen's order, argmin with the reference's rules (strict <, lowest index, NaN never wins,
            //    a NaN at node 0 pins the BMU: Som.cpp:293-304 -- key 0 is below every other key and names node 0)
            if (tid < ((N + 63) & ~63)) {                        // whole wavefronts
                unsigned mybits = 0xFFFFFFFFu;
                if (tid < N) {
                    const float res = onl_tiny_row_sum(s_p + tid * Dp, L8, rem);
                    // (distances are sums of squares: their bit patterns order like their values; NaN -> all ones, never a
                    //  minimum; a NaN at node 0 -> 0, below everything)
                    mybits = (res != res) ? (tid == 0 ? 0u : 0xFFFFFFFFu) : __float_as_uint(res);
                    if (LOCAL)
                        s_d[tid] = res;
                }
                if (!LOCAL) {
                    // the wavefront's minimum, then the LOWEST lane that holds it (strict <: the lowest index wins), one LDS
                    // atomic per wavefront (N same-address atomics serialise: 100 of them were 2 us of a sample)
                    const unsigned wmin = (unsigned)__builtin_amdgcn_readlane((int)onl_wave_min_u32(mybits), 63);
                    const u64 holders = __ballot(mybits == wmin);
                    if ((tid & 63) == 0)
                        atomicMin(&s_key[par], (u64)wmin << 32 | (u64)((tid & ~63) + (__ffsll((long long)holders) - 1)));
                }
            }
            TINY_STAMP(3);
            __syncthreads();
            TINY_STAMP(4);
            if (LOCAL) {                                         // sigma <= 1: the walk from the sample's last BMU (Som.cpp:891)
                if (tid == 1023)
                    s_key[par] = (u64)onl_tiny_walk(s_d, s_win, (unsigned)a.W, (unsigned)a.H, (unsigned)s_last[j]);
                __syncthreads();
            }
            const int bmu = (int)(s_key[par] & 0xFFFFFFFFull);
            // C: the window of Som.cpp:899-944 around the BMU (online_window's bounds, from the table)
            const OnlTinyNode bn = s_win[bmu];
            TINY_STAMP(5);
            const int bx = bn.bx, by = bn.by;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!own[u])
                    continue;
                if (nx[u] < (int)bn.startX || nx[u] >= (int)bn.endX || ny[u] < (int)bn.startY || ny[u] >= (int)bn.endY)
                    continue;
                int dx = nx[u] - bx, dy = ny[u] - by;
                dx = dx < 0 ? -dx : dx;
                dy = dy < 0 ? -dy : dy;
                const int at = dy * a.W + dx;                    // calculateNeighbourhoodWeight(i,j,bx,by,sigma) :915
                const float wold = w[u];
                const float2 lf = s_lf[at];
                const float hf = lf.y;
                float wnew, scM;
                if (exp_decay) {
                    scM = lf.x;                                  // (float)(h eta) :925
                    wnew = wold + scM;                           // :924
                } else {
                    wnew = wold + hf;                            // :930
                    const double tw = wnew == 0 ? 1.0 : s_lut[at] / (double)wnew;   // :933
                    scM = (float)tw;
                }
                float dl = x[u] - m[u];                          // Stepper :912
                if (KIND == VSOM_MEDIAN)
                    dl = onl_sign(dl);
                const float tt = scM * dl;
                const float mn = m[u] + tt;                      // :925 / :935
                float dl2 = x[u] - mn;                           // Stepper(v, map_new) :941
                if (KIND == VSOM_MEDIAN)
                    dl2 = onl_sign(dl2);
                const float pr = dl * dl2;
                const float uu = hf * pr;
                sv[u] = sv[u] + uu;                              // :941
                m[u] = mn;
                w[u] = wnew;
                touched[u] = true;
            }
            TINY_STAMP(6);
            // D: distance of the BMU after the update (:946), MSE (:1167), addBmu (:1165), lastBMU (:895)
            float *p2 = s_p2 + par * ((D + 3) & ~3);
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (own[u] && node[u] == bmu) {
                    const float r = m[u] - x[u];
                    p2[dim[u]] = r * r;
                }
            TINY_STAMP(7);
            // A of the NEXT sample before the same barrier (its squares go where phase B of this sample, long done, read)
            if (jj + 1 < kb) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (own[u]) {
                        x[u] = s_x[(jj + 1) * D + dim[u]];
                        const float r = m[u] - x[u];
                        s_p[node[u] * Dp + dim[u]] = r * r;
                    }
            }
            TINY_STAMP(8);
            __syncthreads();
            TINY_STAMP(9);
            pj = j;
            pbmu = bmu;
        }
    }
    __syncthreads();
    if (tid == 1023 && pj >= 0)                               // the chunk's last sample
        post(pj, pbmu);
    __syncthreads();
    // state back: M and S of every value, weight by the node's first value, sigmaMap = sqrt(|S / w|) (:939-942) where a
    // window touched the node during this chunk
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!own[u])
            continue;
        const size_t at = (size_t)node[u] * a.pitch + dim[u];
        a.map[at] = m[u];
        a.Smap[at] = sv[u];
        if (touched[u]) {
            const double tw2 = w[u] == 0 ? 0.000001 : (double)w[u];   // :939
            const float twf = (float)tw2;
            a.sigmap[at] = sqrtf(fabsf(sv[u] / twf));                 // :942
        }
        if (dim[u] == 0)
            a.weight[node[u]] = w[u];
    }
    for (int i = tid; i < N; i += 1024)
        if (s_hits[i])
            a.hits[i] += (u64)s_hits[i];
    for (int i = tid; i < a.B; i += 1024)
        a.lastbmu[i] = (u64)s_last[i];
    if (a.lastbmu_host)
        for (int i = tid; i < a.B; i += 1024)
            a.lastbmu_host[i] = (u64)s_last[i];
    if (tid == 1023) {
        a.fstate[0] = lastdist;
        a.fstate[1] = mse;
        *a.mse_out = mse;
    }
}

static size_t online_tiny_lds_bytes(const vsom_ctx *c)
{
    const size_t N = c->N, D = c->part_len;
    return N * (16 + 8 + 8) + (N * (D | 1) + 2 * ((D + 3) & ~(size_t)3) + TINY_XBLOCK) * sizeof(float) + N * 8 +
           (c->B + 8) * sizeof(unsigned short);
}

static bool online_tiny_applies(const vsom_ctx *c, double sigma)
{
    return c->use_tiny && c->transform != VSOM_CLR && sigma == sigma && c->B > 0 && c->N <= 1024 &&
           (size_t)c->N * c->part_len <= 4096 && c->part_len <= 512 && c->B <= 4096 && c->bmu_mode == VSOM_BMU_AUTO;   // (EXACT / SHORTLIST name the per-sample forms)
}

static int enqueue_chunk_tiny(vsom_ctx *c, double eta, double sigma, int decay_fn, const double *lutd, int lutw, int first_chunk,
                              u64 *lastbmu_host)
{
    OnlTinyArgs a;
    a.X = c->Xs;
    a.ldx = (int)c->xpitch;
    a.B = (int)c->B;
    a.N = (int)c->N;
    a.W = (int)c->W;
    a.H = (int)c->H;
    a.D = (int)c->part_len;
    a.pitch = (int)c->pitch;
    a.map = c->map;
    a.Smap = c->S;
    a.sigmap = c->sigma;
    a.weight = c->weight;
    a.hits = c->hits;
    a.lastbmu = c->lastbmu;
    a.fstate = c->onl_f;
    a.lutd = lutd;
    a.lutw = lutw;
    a.eta = eta;
    a.sigma = sigma;
    a.decay_fn = decay_fn;
    a.fB = (float)c->B;
    a.keep_mse = first_chunk ? 0 : 1;
    a.mse_out = c->mse;
    a.lastbmu_host = lastbmu_host;
    const size_t smem = online_tiny_lds_bytes(c);
    const bool local = !(sigma > 1);                  // SIGMA_SWITCH_TO_LOCAL (SOM.hpp:37, Som.cpp:891)
    const size_t nd = (size_t)c->N * c->part_len;
    const int upt = nd <= 1024 ? 1 : (nd <= 2048 ? 2 : 4);     // values per thread (the kernel is bound by instruction issue)
#define VSOM_TINY_LAUNCH(KIND, LOC, UU)                                                                                             \
    do {                                                                                                                            \
        if (!c->tiny_lds_attr[(LOC ? 3 : 0) + (UU == 1 ? 0 : (UU == 2 ? 1 : 2))]) {                                                 \
            /* (more than the 64 KiB a launch may ask for by default; 160 KiB per CU) */                                            \
            VSOM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(online_tiny_chunk_kernel<KIND, LOC, UU>),             \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10));                              \
            c->tiny_lds_attr[(LOC ? 3 : 0) + (UU == 1 ? 0 : (UU == 2 ? 1 : 2))] = true;                                            \
        }                                                                                                                           \
        hipLaunchKernelGGL((online_tiny_chunk_kernel<KIND, LOC, UU>), dim3(1), dim3(1024), smem, c->stream, a);                     \
    } while (0)
#define VSOM_TINY_LAUNCH_U(KIND, LOC)                                                                                               \
    do {                                                                                                                            \
        if (upt == 1)                                                                                                               \
            VSOM_TINY_LAUNCH(KIND, LOC, 1);                                                                                         \
        else if (upt == 2)                                                                                                          \
            VSOM_TINY_LAUNCH(KIND, LOC, 2);                                                                                         \
        else                                                                                                                        \
            VSOM_TINY_LAUNCH(KIND, LOC, 4);                                                                                         \
    } while (0)
    if (c->transform == VSOM_MEDIAN) {
        if (local)
            VSOM_TINY_LAUNCH_U(VSOM_MEDIAN, true);
        else
            VSOM_TINY_LAUNCH_U(VSOM_MEDIAN, false);
    } else {
        if (local)
            VSOM_TINY_LAUNCH_U(VSOM_STANDARD, true);
        else
            VSOM_TINY_LAUNCH_U(VSOM_STANDARD, false);
    }
#undef VSOM_TINY_LAUNCH_U
#undef VSOM_TINY_LAUNCH
    return VSOM_OK;
}

// ---- single-vector queries (Som::euclidianWeightedDist / Som::findLocalBmu of ONE host vector) ----------------------
// one wavefront; results into the 16-byte tail behind the single-sample rows: {u64 index, float distance}
template <bool CLR>
__global__ __launch_bounds__(64) void single_dist_kernel(DistArgs d, u64 node, u64 *out_idx, float *out_dist)
{
    const int lane = threadIdx.x;
    const float v = vsom_group_dist_lat<CLR>(d.xa, d.xb, d.ma + (size_t)node * d.ldm, d.mb + (size_t)node * d.ldm, d.L, lane & 7);
    if (lane == 0) {
        *out_idx = node;
        *out_dist = v;
    }
}

template <bool CLR>
__global__ __launch_bounds__(64) void single_local_kernel(DistArgs d, u64 W, u64 H, u64 start, u64 *out_idx, float *out_dist)
{
    u64 idx;
    float dist;
    vsom_local_walk<CLR>(d, d.xa, d.xb, W, H, start, (int)threadIdx.x, idx, dist);
    if (threadIdx.x == 0) {
        *out_idx = idx;
        *out_dist = dist;
    }
}

// Som::findRestrictedBmu / the distances of Som::findRestrictedBmd for ONE vector: online_scan_kernel's evaluation (8 lanes
// per node, the (distance, index) key, the node-0-NaN flag) with the hit-count filter of Som.cpp:316-322 -- node 0 seeds
// the search whatever its hits, any other node needs bmuHits >= min_hits -- and, optionally, every node's distance kept.
template <bool CLR>
__global__ __launch_bounds__(256) void single_scan_kernel(OnlineArgs a, const u64 *__restrict__ hits, u64 min_hits, int use_hits,
                                                          float *__restrict__ all_dist)
{
    __shared__ u64 skey[4];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int node = gid >> 3, k = threadIdx.x & 7;
    const int nc = node < a.N ? node : a.N - 1;
    const float d = vsom_group_dist<CLR, VSOM_SCAN_UNR>(a.d.xa, a.d.xb, a.d.ma + (size_t)nc * a.d.ldm, a.d.mb + (size_t)nc * a.d.ldm,
                                                        a.d.L, k);
    if (all_dist && node < a.N && k == 0)
        all_dist[node] = d;
    const bool allowed = node < a.N && (!use_hits || node == 0 || hits[nc] >= min_hits);
    u64 key = allowed ? vsom_key(d, (uint32_t)node) : ~0ull;
    if (node == 0 && k == 0)
        a.state[ONL_FLAG + a.par] = (d != d) ? 1ull : 0ull;
    for (int off = 32; off >= 8; off >>= 1) {
        const u64 o = __shfl_xor(key, off);
        key = o < key ? o : key;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        skey[wave] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 m = skey[0];
        for (int i = 1; i < 4; ++i)
            m = skey[i] < m ? skey[i] : m;
        atomicMin(online_slot(a.state, a.par, (int)(blockIdx.x % ONL_SLOTS)), m);
    }
}

// ---- host side of the image-bounded search ------------------------------------------------------------------------
// Does the chunk loop of this context search through the image?  VSOM_BMU_EXACT: never; VSOM_BMU_SHORTLIST: whenever the
// kernels apply (Standard / Median, rows of at most 1024 values, sigma > 1); VSOM_BMU_AUTO: where it pays.  Measured on
// MI355X (tools/online_sweep.py, profiles/r6_online_sweep.jsonl; microseconds per sample, image minus exact):
//     +3.3 (the refinement launch and its boundary)  - 0.43e-6 N D (three of the scan's four bytes per value saved)
//     + 0.55e-3 k (re-digiting and scoring each of the k window nodes)
// e.g. 128 x 128 x 784: -2.2 at sigma 2, -1.0 at sigma 8, +1.3 at sigma 16, +6.6 at sigma 32; 192 x 192 x 784: -9.0 ... -1.1;
// 64 x 64 x 784 and below: 0 ... +5; rows of 256 values or fewer lose 2-5 us everywhere (the per-node costs of the refinement
// and of the slots outweigh the bytes).  AUTO takes the image where that estimate gains at least 0.3 us, for rows of at
// least 512 values and chunks long enough to repay the per-chunk passes over the map (digit pass in, sigmaMap pass out).
static bool onl_i8_applies(const vsom_ctx *c, double sigma)
{
    if (c->transform == VSOM_CLR || c->part_len > 1024 || !(sigma > 1) || c->bmu_mode == VSOM_BMU_EXACT || c->B == 0)
        return false;
    if (c->bmu_mode == VSOM_BMU_SHORTLIST)
        return true;
    if (c->part_len < 512 || c->B < 64)
        return false;
    const double ext = std::floor(5.0 * sigma) + 2.0;
    const double k = std::min<double>((double)c->W, ext) * std::min<double>((double)c->H, ext);
    return 0.43e-6 * (double)c->N * (double)c->part_len - 0.55e-3 * k > 3.6;
}

static int onl_i8_ensure(vsom_ctx *c, OnlI8 *o)
{
    const uint32_t ipitch = (c->part_len + 15) / 16 * 16;    // rows of 16-byte pieces
    if (!c->onl_img) {
        VSOM_HIP_CHECK(hipMalloc(&c->onl_img, (size_t)c->N * ipitch));
        VSOM_HIP_CHECK(hipMalloc(&c->onl_nsc, (size_t)c->N * sizeof(float4)));
        VSOM_HIP_CHECK(hipMalloc(&c->onl_lb, (((size_t)c->N + ONL_REF_NODES - 1) / ONL_REF_NODES) * ONL_REF_NODES * sizeof(float)));
        VSOM_HIP_CHECK(hipMalloc(&c->onl_u, ONL_U_BYTES + 64));
        VSOM_HIP_CHECK(hipMemsetAsync((char *)c->onl_u + ONL_U_BYTES, 0, 64, c->stream));
        VSOM_HIP_CHECK(hipMalloc(&c->onl_dirty, (size_t)c->N));
        VSOM_HIP_CHECK(hipMemsetAsync(c->onl_dirty, 0, (size_t)c->N, c->stream));
    }
    if (c->onl_xsc_cap < c->B) {
        VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->onl_xsc)
            (void)hipFree(c->onl_xsc);
        c->onl_xsc = nullptr;
        c->onl_xsc_cap = 0;
        VSOM_HIP_CHECK(hipMalloc(&c->onl_xsc, c->Bcap * sizeof(float4)));
        c->onl_xsc_cap = c->Bcap;
    }
    const double u = 5.9604644775390625e-8;
    o->img = c->onl_img;
    o->nsc = (float4 *)c->onl_nsc;
    o->lb = c->onl_lb;
    o->uslots = c->onl_u;
    o->stats = reinterpret_cast<u64 *>((char *)c->onl_u + ONL_U_BYTES);
    o->xsc = (const float4 *)c->onl_xsc;
    o->dirty = c->onl_dirty;
    o->ipitch = (int)ipitch;
    o->ni = (int)((ipitch + 127) / 128);
    o->nref = (int)((c->N + ONL_REF_NODES - 1) / ONL_REF_NODES);
    o->cT = (float)((2.0 * 35000.0 + 256.0) * u);
    o->g2c = (float)(1.05 * ((double)c->part_len / 8.0 + 16.0) * u);
    return VSOM_OK;
}

// the chunk loop through the image: per chunk a digit pass over the map and the samples' bound terms, then per sample
// onl_refine_kernel (sample j; finishes sample j - 1) and onl_fused_kernel (window of sample j || image scan for j + 1)
static int enqueue_chunk_i8(vsom_ctx *c, double eta, double sigma, int decay_fn, const double *lutd, int lutw)
{
    OnlI8 o;
    if (int rc = onl_i8_ensure(c, &o))
        return rc;
    const int N = (int)c->N, D = (int)c->part_len;
    const size_t B = c->B;
    hipLaunchKernelGGL(onl_digit_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, c->stream, c->map, (int)c->pitch, D, N, o);
    hipLaunchKernelGGL(onl_prep_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, c->stream, c->Xs, (int)c->xpitch, D, (int)B,
                       (float4 *)c->onl_xsc);
    hipLaunchKernelGGL(onl_uslots_init_kernel, dim3(1), dim3(128), 0, c->stream, c->onl_u);
    OnlineArgs a;
    a.d.ldx = 0;
    a.d.ma = a.d.mb = c->map;
    a.d.ldm = (int)c->pitch;
    a.d.L = D;
    a.state = c->onl_state;
    a.fstate = c->onl_f;
    a.N = N;
    a.W = (int)c->W;
    a.H = (int)c->H;
    a.do_scan = 1;
    const float fB = (float)B;
    double ext = std::floor(5.0 * sigma) + 2.0;          // maximal window extents (enqueue_single)
    const int gx = (int)std::min<double>((double)c->W, ext), gy = (int)std::min<double>((double)c->H, ext);
    // scan workgroups: the window launch of sigma = 8 on a 128 x 128 map is 1764 workgroups and the chip holds 2048 of
    // them; two groups of 32 nodes per scan workgroup keep the launch one round (256 + 1764)
    // One group of 32 nodes per scan workgroup (64 VGPRs: eight workgroups per CU; at sigma = 8 on a 128 x 128 map 512 +
    // 1764 workgroups, 2048 resident: the last window workgroups take the scan workgroups' places -- 14.75 us per sample;
    // two groups per workgroup, one round but 86 VGPRs = five per CU: 15.2) and write-through (sc1) stores of the window's
    // rows (nothing re-reads them from this XCD's L2 before the next launch, and dirty lines are written back at the
    // kernel's end: plain 16.05 us per sample, nontemporal 15.7, write-through 15.3 -- profiles/EXPERIMENTS.md).
#ifdef VSOM_DEVELOPMENT
    static const int passes_env = std::getenv("VSOM_ONL_PASSES") ? std::atoi(std::getenv("VSOM_ONL_PASSES")) : 0;
    static const int store_env = std::getenv("VSOM_ONL_STORE") ? std::atoi(std::getenv("VSOM_ONL_STORE")) : 2;
    const int passes = passes_env == 2 ? 2 : 1, store_kind = store_env;
#else
    const int passes = 1, store_kind = 2;
#endif
    const unsigned nscan = (unsigned)((N + 32 * passes - 1) / (32 * passes)),
                   nref = (unsigned)((N + ONL_REF_NODES - 1) / ONL_REF_NODES);
    const bool med = c->transform == VSOM_MEDIAN;
    auto fused = [&](const float *xs_j, int par_j, const float *xs_next, size_t jnext, bool win, bool scan) {
        OnlFusedArgs f;
        f.a = a;
        f.a.d.xa = f.a.d.xb = xs_j;
        f.a.par = par_j;
        f.xnext = xs_next;
        f.jnext = (int)jnext;
        f.do_window = win ? 1 : 0;
        f.do_scan = scan ? 1 : 0;
        f.nwin_x = gx > 0 ? gx : 1;
        f.nwin_y = gy > 0 ? gy : 1;
        f.nscan = (int)nscan;
        f.passes = passes;
        const unsigned grid = (win ? (unsigned)(f.nwin_x * f.nwin_y) : 0u) + (scan ? nscan : 0u);
#define ONL_FUSED2(KIND, ST, P)                                                                                            \
    hipLaunchKernelGGL((onl_fused_kernel<KIND, ST, P>), dim3(grid), dim3(256), 0, c->stream, f, o, lutd, lutw, D, (int)c->pitch, eta, \
                       sigma, decay_fn, c->map, c->S, c->sigma, c->weight)
#define ONL_FUSED(KIND, ST) do { if (passes == 2) ONL_FUSED2(KIND, ST, 2); else ONL_FUSED2(KIND, ST, 1); } while (0)
        if (med) {
            if (store_kind == 2) ONL_FUSED(VSOM_MEDIAN, 2); else if (store_kind == 1) ONL_FUSED(VSOM_MEDIAN, 1); else ONL_FUSED(VSOM_MEDIAN, 0);
        } else {
            if (store_kind == 2) ONL_FUSED(VSOM_STANDARD, 2); else if (store_kind == 1) ONL_FUSED(VSOM_STANDARD, 1); else ONL_FUSED(VSOM_STANDARD, 0);
        }
#undef ONL_FUSED2
#undef ONL_FUSED
    };
    auto refine = [&](const float *xs_j, int par_j, const float *xs_prev, u64 *lb_prev, bool do_refine) {
        OnlineArgs r = a;
        r.d.xa = r.d.xb = xs_j;
        r.par = par_j;
        r.pxa = r.pxb = xs_prev;
        r.do_post = xs_prev != nullptr;
        const dim3 grid((do_refine ? nref : 0u) + 1u);
        hipLaunchKernelGGL(onl_refine_kernel, grid, dim3(256), 0, c->stream, r, o, do_refine ? 1 : 0, c->hits, lb_prev, fB, 1);
    };
    // scores of sample 0 (no window yet): the launch works on "sample -1" of parity 1
    fused(c->Xs, 1, c->Xs, 0, false, true);
    const float *prev = nullptr;
    for (size_t j = 0; j < B; ++j) {
        const float *xs = c->Xs + j * c->xpitch;
        refine(xs, (int)(j & 1), prev, j ? c->lastbmu + (j - 1) : nullptr, true);
        const bool more = j + 1 < B;
        fused(xs, (int)(j & 1), more ? xs + c->xpitch : xs, more ? j + 1 : j, true, more);
        prev = xs;
    }
    // finish the last sample: its post step runs as the "previous sample" of a launch of the other parity
    refine(prev, (int)(B & 1), prev, c->lastbmu + (B - 1), false);
    hipLaunchKernelGGL(onl_sigma_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, c->stream, c->S, c->sigma, c->weight,
                       c->onl_dirty, D, (int)c->pitch, N);
    return VSOM_OK;
}

extern "C" {

// Som::findBmu(v) for ONE host vector (Som.cpp:283-309) without touching the staged chunk: copy,
// one scan launch (8 lanes per node, atomicMin on the (distance, index) key), 32 bytes back.
int vsom_find_bmu(vsom_ctx *c, const float *v_host, uint64_t *bmu_out, float *dist_out)
{
    if (!c)
        return vsom_fail(VSOM_ERR_INVALID, "null context");
    VSOM_HIP_CHECK(hipSetDevice(c->device));
    if (int jrc = vsom_join_aux(c))
        return jrc;
    if (!v_host)
        return vsom_fail(VSOM_ERR_INVALID, "null argument");
    int rc = stage_single(c, v_host);
    if (rc)
        return rc;
    const bool clr = c->transform == VSOM_CLR;
    const size_t xs_n = c->xpitch, pp = c->part_pitch;
    float *xs = c->v_dev, *xp = c->v_dev + xs_n, *yp = xp + pp;
    OnlineArgs a;
    a.par = 0;
    a.d.xa = clr ? xp : xs;
    a.d.xb = clr ? yp : xs;
    a.d.ldx = 0;
    a.d.ma = c->map;
    a.d.mb = clr ? c->map + c->part_pitch : c->map;
    a.d.ldm = (int)c->pitch;
    a.d.L = (int)c->part_len;
    a.state = c->onl_state;
    a.fstate = c->onl_f;
    a.N = (int)c->N;
    a.W = (int)c->W;
    a.H = (int)c->H;
    VSOM_HIP_CHECK(hipMemsetAsync(c->onl_state, 0xFF, ONL_SLOTS * 16 * sizeof(u64), c->stream));   // arm the keys of parity 0
    dim3 grid((unsigned)(((size_t)c->N * 8 + 255) / 256));
    a.pxa = a.pxb = nullptr;
    a.do_scan = 1;
    a.do_post = 0;
    if (clr)
        hipLaunchKernelGGL((online_scan_kernel<true, false>), grid, dim3(256), 0, c->stream, a, (u64 *)nullptr, (u64 *)nullptr, 1.f, 0);
    else
        hipLaunchKernelGGL((online_scan_kernel<false, false>), grid, dim3(256), 0, c->stream, a, (u64 *)nullptr, (u64 *)nullptr, 1.f, 0);
    VSOM_HIP_CHECK(hipGetLastError());
    u64 *st = reinterpret_cast<u64 *>(c->v_pinned + xs_n + 3 * pp + 32);   // image of onl_state (8-byte aligned: pitches are multiples of 32 floats)
    VSOM_HIP_CHECK(hipMemcpyAsync(st, c->onl_state, ONL_STATE_BYTES, hipMemcpyDeviceToHost, c->stream));
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    uint64_t key = ~0ull;             // minimum over the key slots of parity 0
    for (int sl = 0; sl < ONL_SLOTS; ++sl)
        key = std::min<uint64_t>(key, st[sl * 16]);
    const bool nan0 = st[ONL_FLAG] != 0;     // a NaN distance at node 0 pins the BMU to 0 (Som.cpp:293-299)
    const uint64_t bmu = nan0 ? 0ull : (key & 0xFFFFFFFFull);
    if (bmu_out)
        *bmu_out = bmu;
    if (dist_out) {
        const uint32_t bits = nan0 ? 0x7FC00000u : (uint32_t)(key >> 32);
        std::memcpy(dist_out, &bits, 4);
    }
    return VSOM_OK;
}

// shared by the two single-vector queries below: stage v, launch `which` (0: distance to `node`; 1: local search from
// `node`), 16 bytes back -- one synchronisation, no allocation, the staged chunk untouched
static int single_query(vsom_ctx *c, const float *v_host, uint64_t node, int which, uint64_t *idx_out, float *dist_out)
{
    if (!c)
        return vsom_fail(VSOM_ERR_INVALID, "null context");
    VSOM_HIP_CHECK(hipSetDevice(c->device));
    if (int jrc = vsom_join_aux(c))
        return jrc;
    if (!v_host)
        return vsom_fail(VSOM_ERR_INVALID, "null argument");
    if (node >= c->N)
        return vsom_fail(VSOM_ERR_INVALID, "node index out of range");
    // The distance: ONE wavefront reads the vector once and writes 16 bytes -- both go straight through the pinned buffer
    // (host memory the device addresses): no copy in, no copy out, the call is a launch and a synchronisation (18.2 -> 14.9 us).
    // The local walk evaluates up to a few dozen distances and would read the vector across the bus each time (no gain
    // measured): its vector is copied into HBM first, only its 16 bytes of results go through pinned memory.  (Kernels of
    // thousands of wavefronts that all read the vector -- vsom_find_bmu -- keep their copy too.)
    const bool zero_copy_in = which == 0;
    int rc = stage_single(c, v_host, !zero_copy_in);
    if (rc)
        return rc;
    const bool clr = c->transform == VSOM_CLR;
    const size_t xs_n = c->xpitch, pp = c->part_pitch;
    float *rows = zero_copy_in ? c->v_pinned : c->v_dev;
    float *xs = rows, *xp = rows + xs_n, *yp = xp + pp, *tail = c->v_pinned + xs_n + 3 * pp;
    DistArgs d;
    d.xa = clr ? xp : xs;
    d.xb = clr ? yp : xs;
    d.ldx = 0;
    d.ma = c->map;
    d.mb = clr ? c->map + c->part_pitch : c->map;
    d.ldm = (int)c->pitch;
    d.L = (int)c->part_len;
    u64 *oi = reinterpret_cast<u64 *>(tail);
    float *od = tail + 2;
    if (which == 0) {
        if (clr)
            hipLaunchKernelGGL(single_dist_kernel<true>, dim3(1), dim3(64), 0, c->stream, d, (u64)node, oi, od);
        else
            hipLaunchKernelGGL(single_dist_kernel<false>, dim3(1), dim3(64), 0, c->stream, d, (u64)node, oi, od);
    } else {
        if (clr)
            hipLaunchKernelGGL(single_local_kernel<true>, dim3(1), dim3(64), 0, c->stream, d, (u64)c->W, (u64)c->H, (u64)node, oi, od);
        else
            hipLaunchKernelGGL(single_local_kernel<false>, dim3(1), dim3(64), 0, c->stream, d, (u64)c->W, (u64)c->H, (u64)node, oi, od);
    }
    VSOM_HIP_CHECK(hipGetLastError());
    float *pout = tail;
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (idx_out)
        std::memcpy(idx_out, pout, 8);
    if (dist_out)
        *dist_out = pout[2];
    return VSOM_OK;
}

// one host vector against every node: restricted (or plain) BMU through the key slots, optionally all N distances
static int single_scan(vsom_ctx *c, const float *v_host, int use_hits, uint64_t min_hits, uint64_t *bmu_out, float *dist_out,
                       float *all_out_host)
{
    if (!c)
        return vsom_fail(VSOM_ERR_INVALID, "null context");
    VSOM_HIP_CHECK(hipSetDevice(c->device));
    if (int jrc = vsom_join_aux(c))
        return jrc;
    if (!v_host)
        return vsom_fail(VSOM_ERR_INVALID, "null argument");
    int rc = stage_single(c, v_host);
    if (rc)
        return rc;
    float *all_dev = nullptr;
    if (all_out_host) {
        if ((size_t)c->N * 4 > c->q_scratch_cap) {
            VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
            if (c->q_scratch)
                (void)hipFree(c->q_scratch);
            c->q_scratch = nullptr;
            c->q_scratch_cap = 0;
            const size_t cap = ((size_t)c->N * 4 + 4095) / 4096 * 4096;
            VSOM_HIP_CHECK(hipMalloc(&c->q_scratch, cap));
            c->q_scratch_cap = cap;
        }
        all_dev = reinterpret_cast<float *>(c->q_scratch);
    }
    const bool clr = c->transform == VSOM_CLR;
    const size_t xs_n = c->xpitch, pp = c->part_pitch;
    float *xs = c->v_dev, *xp = c->v_dev + xs_n, *yp = xp + pp;
    OnlineArgs a;
    a.par = 0;
    a.d.xa = clr ? xp : xs;
    a.d.xb = clr ? yp : xs;
    a.d.ldx = 0;
    a.d.ma = c->map;
    a.d.mb = clr ? c->map + c->part_pitch : c->map;
    a.d.ldm = (int)c->pitch;
    a.d.L = (int)c->part_len;
    a.state = c->onl_state;
    a.fstate = c->onl_f;
    a.N = (int)c->N;
    a.W = (int)c->W;
    a.H = (int)c->H;
    a.pxa = a.pxb = nullptr;
    a.do_scan = 1;
    a.do_post = 0;
    VSOM_HIP_CHECK(hipMemsetAsync(c->onl_state, 0xFF, ONL_SLOTS * 16 * sizeof(u64), c->stream));   // arm the keys of parity 0
    dim3 grid((unsigned)(((size_t)c->N * 8 + 255) / 256));
    if (clr)
        hipLaunchKernelGGL(single_scan_kernel<true>, grid, dim3(256), 0, c->stream, a, c->hits, (u64)min_hits, use_hits, all_dev);
    else
        hipLaunchKernelGGL(single_scan_kernel<false>, grid, dim3(256), 0, c->stream, a, c->hits, (u64)min_hits, use_hits, all_dev);
    VSOM_HIP_CHECK(hipGetLastError());
    u64 *st = reinterpret_cast<u64 *>(c->v_pinned + xs_n + 3 * pp + 32);   // image of onl_state (vsom_find_bmu)
    VSOM_HIP_CHECK(hipMemcpyAsync(st, c->onl_state, ONL_STATE_BYTES, hipMemcpyDeviceToHost, c->stream));
    if (all_out_host)
        VSOM_HIP_CHECK(hipMemcpyAsync(all_out_host, all_dev, (size_t)c->N * 4, hipMemcpyDeviceToHost, c->stream));
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    uint64_t key = ~0ull;
    for (int sl = 0; sl < ONL_SLOTS; ++sl)
        key = std::min<uint64_t>(key, st[sl * 16]);
    const bool nan0 = st[ONL_FLAG] != 0;          // a NaN distance at node 0 pins the BMU to 0 (Som.cpp:316-322: `cur < NaN` never holds)
    if (bmu_out)
        *bmu_out = nan0 ? 0ull : (key & 0xFFFFFFFFull);
    if (dist_out) {
        const uint32_t bits = nan0 ? 0x7FC00000u : (uint32_t)(key >> 32);
        std::memcpy(dist_out, &bits, 4);
    }
    return VSOM_OK;
}

// Som::findRestrictedBmu(v, ..., minBmuHits, ...) for ONE host vector (Som.cpp:313-332)
int vsom_find_restricted_bmu(vsom_ctx *c, const float *v_host, uint64_t min_hits, uint64_t *bmu_out, float *dist_out)
{
    return single_scan(c, v_host, 1, min_hits, bmu_out, dist_out, nullptr);
}

// euclidianWeightedDist(i, v) of ONE host vector to every node i (what Som::findRestrictedBmd walks, Som.cpp:457-487)
int vsom_distances_single(vsom_ctx *c, const float *v_host, float *dist_out_host)
{
    if (!dist_out_host)
        return vsom_fail(VSOM_ERR_INVALID, "null output");
    return single_scan(c, v_host, 0, 0, nullptr, nullptr, dist_out_host);
}

// Som::euclidianWeightedDist(pos, v, ...) for ONE host vector (Som.cpp:124-141)
int vsom_dist_single(vsom_ctx *c, const float *v_host, uint64_t node, float *dist_out)
{
    return single_query(c, v_host, node, 0, nullptr, dist_out);
}

// Som::findLocalBmu(v, ..., lastBMU, ...) for ONE host vector (Som.cpp:335-454)
int vsom_find_local_bmu(vsom_ctx *c, const float *v_host, uint64_t last_bmu, uint64_t *bmu_out, float *dist_out)
{
    return single_query(c, v_host, last_bmu, 1, bmu_out, dist_out);
}

// lb_host != NULL: the caller wants lastBMU back in this call -- *lb_in_pinned = the one-launch kernel has stored it into
// c->out_pinned itself (else the caller fetches it with vsom_get_last_bmu)
static int train_online_chunk_impl(vsom_ctx *c, double eta, double sigma, int decay_fn, int first_chunk, float *mse_out,
                                   bool want_lb, bool *lb_in_pinned)
{
    if (lb_in_pinned)
        *lb_in_pinned = false;
    if (!c)
        return vsom_fail(VSOM_ERR_INVALID, "null context");
    VSOM_HIP_CHECK(hipSetDevice(c->device));
    if (int jrc = vsom_join_aux(c))
        return jrc;
    if (decay_fn != VSOM_EXPONENTIAL && decay_fn != VSOM_INVERSE_PROPORTIONAL)
        return vsom_fail(VSOM_ERR_INVALID, "online training needs Exponential or InverseProportional");
    if (!c->chunk_loaded)   // an empty chunk is a no-op for the sample loop (Som.cpp:1161)
        return vsom_fail(VSOM_ERR_INVALID, "no chunk loaded");
    if (c->ahead_rows)      // (CHECK_ROWS of vsom_capi.hip: the staged rows already belong to the next chunk)
        return vsom_fail(VSOM_ERR_INVALID,
                         "the next chunk is staged ahead over the current chunk's rows: vsom_commit_chunk first");
    const double *lutd = nullptr;
    int lutw = (int)c->W, lslot = 0, rc;
    bool tiny = online_tiny_applies(c, sigma);
    if (tiny)                                 // one launch reads the table once, into LDS: straight from the pinned slot
        rc = ensure_lutd_host(c, sigma, &lutd, &lslot);
    else
        rc = ensure_lutd(c, sigma, &lutd, &lutw);
    if (rc)
        return rc;
    {
        TimerScope ts(c, VSOM_T_ONLINE);
        if (!tiny)                            // (the one-launch chunk starts the running MSE itself and uses no key slots)
            hipLaunchKernelGGL(online_init_kernel, dim3(1), dim3(1), 0, c->stream, c->onl_state, c->onl_f,
                               first_chunk ? 0 : 1);
        if (tiny) {
            u64 *lbh = nullptr;
            if (want_lb && c->B <= 8192) {
                if (!c->out_pinned)
                    VSOM_HIP_CHECK(hipHostMalloc(&c->out_pinned, 8192 * sizeof(uint64_t)));
                lbh = static_cast<u64 *>(c->out_pinned);
            }
            if ((rc = enqueue_chunk_tiny(c, eta, sigma, decay_fn, lutd, lutw, first_chunk, lbh)) || (rc = lutd_host_used(c, lslot)))
                return rc;
            if (lbh && lb_in_pinned)
                *lb_in_pinned = true;
        } else if (onl_i8_applies(c, sigma)) {
            if ((rc = enqueue_chunk_i8(c, eta, sigma, decay_fn, lutd, lutw)))
                return rc;
        } else {
        const float fB = (float)c->B;
        const bool pipelined = sigma > 1;     // the search launch of sample j finishes sample j-1 (online_scan_kernel)
        const float *pxs = nullptr, *pxp = nullptr, *pyp = nullptr;
        for (size_t j = 0; j < c->B; ++j) {
            const float *xs = c->Xs + j * c->xpitch;
            const float *xp = c->XP ? c->XP + j * c->part_pitch : nullptr;
            const float *yp = c->YP ? c->YP + j * c->part_pitch : nullptr;
            rc = enqueue_single(c, xs, xp, yp, eta, sigma, decay_fn, c->lastbmu + j, nullptr, fB, 1,
                                lutd, lutw, (int)(j & 1), nullptr, pipelined, pxs, pxp, pyp);
            if (rc)
                return rc;
            pxs = xs;
            pxp = xp;
            pyp = yp;
        }
        if (pipelined && c->B > 0 &&
            (rc = enqueue_chunk_tail(c, pxs, pxp, pyp, c->lastbmu + (c->B - 1), fB, (int)((c->B - 1) & 1))))
            return rc;
        }
        VSOM_HIP_CHECK(hipGetLastError());
    }
    // vsom_get_mse reports the chunk's MSE for callers that passed mse_out = NULL (asynchronous use)
    if (!tiny)
        VSOM_HIP_CHECK(hipMemcpyAsync(c->mse, c->onl_f + 1, 4, hipMemcpyDefault, c->stream));   // (c->mse: pinned host memory)
    if (mse_out) {
        VSOM_HIP_CHECK(hipMemcpyAsync(mse_out, c->onl_f + 1, 4, hipMemcpyDeviceToHost, c->stream));
        VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    return VSOM_OK;
}

int vsom_train_online_chunk_acc(vsom_ctx *c, double eta, double sigma, int decay_fn, int first_chunk,
                                float *mse_out)
{
    return train_online_chunk_impl(c, eta, sigma, decay_fn, first_chunk, mse_out, false, nullptr);
}

int vsom_train_online_chunk_fetch(vsom_ctx *c, double eta, double sigma, int decay_fn, int first_chunk,
                                  uint64_t *lastbmu_out, float *mse_out)
{
    bool in_pinned = false;
    int rc = train_online_chunk_impl(c, eta, sigma, decay_fn, first_chunk, nullptr, lastbmu_out != nullptr, &in_pinned);
    if (rc)
        return rc;
    if (lastbmu_out && !in_pinned) {
        if ((rc = vsom_get_last_bmu(c, lastbmu_out)))      // (synchronises)
            return rc;
    } else {
        VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (lastbmu_out)
            std::memcpy(lastbmu_out, c->out_pinned, c->B * sizeof(uint64_t));
    }
    if (mse_out)
        *mse_out = *static_cast<volatile float *>(c->mse);  // pinned host memory; written in stream order before the wait above
    return VSOM_OK;
}

int vsom_get_online_search_stats(vsom_ctx *c, uint64_t *out, int reset)
{
    if (!c || !out)
        return vsom_fail(VSOM_ERR_INVALID, "null argument");
    VSOM_HIP_CHECK(hipSetDevice(c->device));
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!c->onl_u)
        return VSOM_OK;
    u64 *st = reinterpret_cast<u64 *>((char *)c->onl_u + ONL_U_BYTES);
    VSOM_HIP_CHECK(hipMemcpyAsync(out, st, 32, hipMemcpyDeviceToHost, c->stream));
    if (reset)
        VSOM_HIP_CHECK(hipMemsetAsync(st, 0, 32, c->stream));
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return VSOM_OK;
}

int vsom_train_online_chunk(vsom_ctx *c, double eta, double sigma, int decay_fn, float *mse_out)
{
    return vsom_train_online_chunk_acc(c, eta, sigma, decay_fn, 1, mse_out);
}

int vsom_train_single(vsom_ctx *c, const float *v_host, double eta, double sigma, uint64_t *last_bmu,
                      int decay_fn, float *residual_out, float *dist_out, uint64_t *bmu_out)
{
    if (!c)
        return vsom_fail(VSOM_ERR_INVALID, "null context");
    VSOM_HIP_CHECK(hipSetDevice(c->device));
    if (int jrc = vsom_join_aux(c))
        return jrc;
    if (!v_host || !last_bmu)
        return vsom_fail(VSOM_ERR_INVALID, "null argument");
    if (decay_fn != VSOM_EXPONENTIAL && decay_fn != VSOM_INVERSE_PROPORTIONAL)
        return vsom_fail(VSOM_ERR_INVALID, "online training needs Exponential or InverseProportional");
    if (*last_bmu >= c->N)
        return vsom_fail(VSOM_ERR_INVALID, "lastBMU out of range");
    int rc = stage_single(c, v_host);
    if (rc)
        return rc;
    const size_t xs_n = c->xpitch, pp = c->part_pitch;
    const double *lutd = nullptr;
    int lutw = 0;
    rc = ensure_lutd(c, sigma, &lutd, &lutw);
    if (rc)
        return rc;
    // The sample's rows are copied into HBM (thousands of wavefronts read them); lastBMU in, residual + {bmu, distance, mse}
    // out go straight through the pinned buffer, which the device addresses: written / read by single wavefronts of the
    // step's kernels -- one copy and one synchronisation per call (three copies before: 36 us per call)
    float *xs = c->v_dev, *xp = c->v_dev + xs_n, *yp = xp + pp;
    float *res = c->v_pinned + xs_n + 2 * pp, *tail = res + pp;
    u64 *lb = reinterpret_cast<u64 *>(tail);
    float *pout = res;
    std::memcpy(tail, last_bmu, 8);
    tail[2] = 0.f;
    tail[3] = 0.f;
    {
        TimerScope ts(c, VSOM_T_ONLINE);
        hipLaunchKernelGGL(online_init_kernel, dim3(1), dim3(1), 0, c->stream, c->onl_state, c->onl_f, 1);
        rc = enqueue_single(c, xs, xp, yp, eta, sigma, decay_fn, lb, res, 1.0f, 0, lutd, lutw, 0, tail + 2);
        if (rc)
            return rc;
        VSOM_HIP_CHECK(hipGetLastError());
    }
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    uint64_t bmu = 0;
    std::memcpy(&bmu, pout + pp, 8);
    if (residual_out)
        std::memcpy(residual_out, pout, (size_t)c->part_len * 4);
    if (dist_out)
        *dist_out = pout[pp + 2];
    *last_bmu = bmu;
    if (bmu_out)
        *bmu_out = bmu;
    return VSOM_OK;
}

}   // extern "C"
