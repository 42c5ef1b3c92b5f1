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
template <int KIND, bool BLOCK_SYNC>
__device__ __forceinline__ void online_node_update(size_t n, double h, int tid, int nthr,
                                                   const float *__restrict__ xs, const float *__restrict__ xp,
                                                   const float *__restrict__ yp, int D, int P, int ppitch, int pitch,
                                                   double eta, int decay_fn, float *map, float *Smap, float *sigmap,
                                                   float *weight)
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
            M[d] = m;
            S[d] = s;
            sg[d] = sqrtf(fabsf(s / twf));                   // :942
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

// double-precision table of calculateNeighbourhoodWeight for the online path (the batch path
// uses its float cast); cached per sigma
static int ensure_lutd(vsom_ctx *c, double sigma, const double **out, int *lutw)
{
    const uint32_t lw = c->W, lh = c->H;
    const size_t need = (size_t)lw * lh;
    if (c->lutd && c->lutd_sigma == sigma) {
        *out = c->lutd;
        *lutw = (int)lw;
        return VSOM_OK;
    }
    if (need > c->lutd_cap) {
        VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->lutd)
            VSOM_HIP_CHECK(hipFree(c->lutd));
        c->lutd = nullptr;
        VSOM_HIP_CHECK(hipMalloc(&c->lutd, need * sizeof(double)));
        c->lutd_cap = need;
    }
    std::vector<double> host(need);
    for (uint32_t dy = 0; dy < lh; ++dy)
        for (uint32_t dx = 0; dx < lw; ++dx)
            host[(size_t)dy * lw + dx] = vsom_neighbourhood_weight(dx, dy, 0, 0, sigma);
    // the previous table may still be in use by enqueued kernels
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    VSOM_HIP_CHECK(hipMemcpy(c->lutd, host.data(), need * sizeof(double), hipMemcpyHostToDevice));
    c->lutd_sigma = sigma;
    *out = c->lutd;
    *lutw = (int)lw;
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
static int stage_single(vsom_ctx *c, const float *v_host)
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
    VSOM_HIP_CHECK(hipMemcpyAsync(c->v_dev, host, nstage * sizeof(float), hipMemcpyHostToDevice, c->stream));
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

int vsom_train_online_chunk_acc(vsom_ctx *c, double eta, double sigma, int decay_fn, int first_chunk,
                                float *mse_out)
{
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
    int lutw = 0;
    int rc = ensure_lutd(c, sigma, &lutd, &lutw);
    if (rc)
        return rc;
    {
        TimerScope ts(c, VSOM_T_ONLINE);
        hipLaunchKernelGGL(online_init_kernel, dim3(1), dim3(1), 0, c->stream, c->onl_state, c->onl_f,
                           first_chunk ? 0 : 1);
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
        VSOM_HIP_CHECK(hipGetLastError());
    }
    // vsom_get_mse reports the chunk's MSE for callers that passed mse_out = NULL (asynchronous use)
    VSOM_HIP_CHECK(hipMemcpyAsync(c->mse, c->onl_f + 1, 4, hipMemcpyDeviceToDevice, c->stream));
    if (mse_out) {
        VSOM_HIP_CHECK(hipMemcpyAsync(mse_out, c->onl_f + 1, 4, hipMemcpyDeviceToHost, c->stream));
        VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
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
    float *xs = c->v_dev, *xp = c->v_dev + xs_n, *yp = xp + pp, *res = yp + pp, *tail = res + pp;
    u64 *lb = reinterpret_cast<u64 *>(tail);
    // inputs and outputs travel through the pinned buffer: rows (stage_single), a 16-byte tail in,
    // residual + tail out -- three small asynchronous copies and one synchronisation per call
    float *ptail_in = c->v_pinned + xs_n + 3 * pp + 16;      // host image of the tail going in
    float *pout = c->v_pinned + xs_n + 2 * pp;               // host image of [residual | tail] coming out
    std::memcpy(ptail_in, last_bmu, 8);
    ptail_in[2] = 0.f;
    ptail_in[3] = 0.f;
    VSOM_HIP_CHECK(hipMemcpyAsync(tail, ptail_in, 16, hipMemcpyHostToDevice, c->stream));
    {
        TimerScope ts(c, VSOM_T_ONLINE);
        hipLaunchKernelGGL(online_init_kernel, dim3(1), dim3(1), 0, c->stream, c->onl_state, c->onl_f, 1);
        rc = enqueue_single(c, xs, xp, yp, eta, sigma, decay_fn, lb, res, 1.0f, 0, lutd, lutw, 0, tail + 2);
        if (rc)
            return rc;
        VSOM_HIP_CHECK(hipGetLastError());
    }
    VSOM_HIP_CHECK(hipMemcpyAsync(pout, res, (pp + 4) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
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
