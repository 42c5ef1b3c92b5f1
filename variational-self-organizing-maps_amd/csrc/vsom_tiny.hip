// vsom_tiny.hip -- Som::trainBatchSomEpoch (Som.cpp:756-879) for maps so small that the dozen launches
// of the general path cost more than the work (the reference's own perf-harness scenario is a 10x10
// map with 20 nine-dimensional rows, tests/performance/perf_tests.cpp:74-112): ONE workgroup does the
// whole epoch in one launch --
//   phase 1  every (sample, node) distance by 8-lane groups in Eigen's order (vsom_group_dist), argmin
//            through an LDS atomicMin on the (distance, index) key -- or the findLocalBmu walk, one
//            wavefront per sample (vsom_local_walk);  bmuHits;  MSE summed in sample order
//   phase 2  one thread per (node, dim) chain [CLR: per (node, pair)]: the neighbourhood weight from
//            the host table, the fp32 prefix W, c = w/W, and the mean / sigma^2 recurrences, every
//            operation rounded separately like the general kernels
// so the results are bit-identical to those kernels and to the oracle.
#include "vsom_device.hpp"

struct TinyArgs {
    DistArgs d;                  // search operands: Xs (Standard/Median) or XP,YP (CLR); map parts
    float *map, *sigma, *weight;
    u64 *hits, *lastbmu;
    float *sqres, *mse;
    const float *lut;
    int lutw;
    int N, W, H, Dc, pitch, ppitch, B, is_first;   // Dc: chains per node (D, or P for CLR)
    int stage_x;                                    // samples fit into LDS
    int luth;                                       // table rows (the table is copied into LDS)
};

__device__ __forceinline__ float tiny_sign(float a)   // as vsom_update.hip's vsom_sign (see there)
{
    const float one = __builtin_copysignf(1.f, a);
    return (a < 0.f || a > 0.f) ? one : a;
}

template <int KIND>
__global__ __launch_bounds__(256) void tiny_batch_epoch_kernel(TinyArgs a)
{
    constexpr bool CLR = KIND == VSOM_CLR;
    extern __shared__ __attribute__((aligned(16))) unsigned char tiny_smem[];
    u64 *keys = reinterpret_cast<u64 *>(tiny_smem);          // [B]
    int2 *bxy = reinterpret_cast<int2 *>(keys + a.B);        // [B]
    float *sq = reinterpret_cast<float *>(bxy + a.B);        // [B]
    int *nan0 = reinterpret_cast<int *>(sq + a.B);           // [B]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int B = a.B, N = a.N;

    // ---- phase 1 (Som.cpp:762-806) ---------------------------------------------------------------
    if (a.is_first) {
        for (int s = tid; s < B; s += 256) {
            keys[s] = ~0ull;
            nan0[s] = 0;
        }
        __syncthreads();
        const int grp = tid >> 3, k = tid & 7;
        for (int p = grp; p < B * N; p += 32) {
            const int s = p / N, n = p - s * N;
            const float dd = vsom_group_dist<CLR>(a.d.xa + (size_t)s * a.d.ldx, a.d.xb + (size_t)s * a.d.ldx,
                                                  a.d.ma + (size_t)n * a.d.ldm, a.d.mb + (size_t)n * a.d.ldm, a.d.L, k);
            if (k == 0) {
                if (n == 0 && dd != dd)
                    nan0[s] = 1;                 // a NaN at node 0 pins the BMU to 0 (Som.cpp:293-299)
                atomicMin(&keys[s], vsom_key(dd, (uint32_t)n));
            }
        }
        __syncthreads();
        for (int s = tid; s < B; s += 256) {
            const u64 key = keys[s];
            a.lastbmu[s] = nan0[s] ? 0ull : (key & 0xFFFFFFFFull);
            sq[s] = nan0[s] ? __uint_as_float(0x7FC00000u) : __uint_as_float((uint32_t)(key >> 32));
        }
    } else {
        for (int s = wave; s < B; s += 4) {
            u64 idx;
            float dist;
            vsom_local_walk<CLR>(a.d, a.d.xa + (size_t)s * a.d.ldx, a.d.xb + (size_t)s * a.d.ldx, (u64)a.W, (u64)a.H,
                                 a.lastbmu[s], lane, idx, dist);
            if (lane == 0) {
                a.lastbmu[s] = idx;
                sq[s] = dist;
            }
        }
    }
    __syncthreads();
    for (int s = tid; s < B; s += 256) {
        const u64 idx = a.lastbmu[s];
        int bx, by;
        vsom_somindex(idx, (u64)a.W, (u64)a.H, bx, by);      // SomIndex(*this, lastBMU) :847-849
        bxy[s] = make_int2(bx, by);
        a.sqres[s] = sq[s];
        atomicAdd(&a.hits[idx], 1ull);                       // bmuHits[index] += 1 :778,801
    }
    if (tid == 0) {                                          // MSE in sample order :781,804
        const float fB = (float)B;
        float run = 0.f;
        for (int s = 0; s < B; ++s)
            run = run + sq[s] / fB;
        *a.mse = run;
    }
    __syncthreads();

    // ---- phase 2 (Som.cpp:809-876): one thread per chain, the old map is not read ------------------
    const int Dc = a.Dc;
    // the samples (and y' for CLR) move into LDS when they fit: the chains below are serial in the
    // samples, and an LDS read per step instead of an L2 round trip is most of this kernel's time
    float *xl = reinterpret_cast<float *>(nan0 + B);
    const float *xsrc = a.d.xa, *ysrc = a.d.xb;
    int xld = a.d.ldx;
    float *lutl = xl + (a.stage_x ? B * Dc * (CLR ? 2 : 1) : 0);
    for (int i = tid; i < a.lutw * a.luth; i += 256)
        lutl[i] = a.lut[i];
    if (a.stage_x) {
        for (int i = tid; i < B * Dc; i += 256) {
            const int s = i / Dc, e = i - s * Dc;
            xl[i] = a.d.xa[(size_t)s * a.d.ldx + e];
            if (CLR)
                xl[B * Dc + i] = a.d.xb[(size_t)s * a.d.ldx + e];
        }
        xsrc = xl;
        ysrc = xl + B * Dc;
        xld = Dc;
    }
    __syncthreads();   // table (and samples) in LDS
    for (int c = tid; c < N * Dc; c += 256) {
        const int node = c / Dc, e = c - node * Dc;
        int cx, cy;
        vsom_somindex((u64)node, (u64)a.W, (u64)a.H, cx, cy);
        float Wsum = 0.f;                                    // sumOfWeights :840
        float M = 0.f, S = 0.f, Bv = 0.f, SB = 0.f;          // CLR: (M,S) = A chain, (Bv,SB) = B chain
        for (int s = 0; s < B; ++s) {
            const int2 b = bxy[s];
            int dx = cx - b.x, dy = cy - b.y;
            dx = dx < 0 ? -dx : dx;
            dy = dy < 0 ? -dy : dy;
            const float w = lutl[dy * a.lutw + dx];           // (float)calculateNeighbourhoodWeight :851
            Wsum = Wsum + w;                                 // :857
            const float cc = w / Wsum;                       // :864 (0/0 -> NaN)
            if (CLR) {
                const float xp = xsrc[(size_t)s * xld + e], yp = ysrc[(size_t)s * xld + e];
                float inner = M * xp;                        // Transformation.cpp:129
                inner = inner + Bv;
                inner = inner - yp;
                const float m2 = -2.f * inner;
                const float aD = m2 * xp;
                const float tA = cc * aD, tB = cc * m2;
                float uA = w * aD;
                uA = uA * aD;
                float uB = w * m2;
                uB = uB * m2;
                M = M + tA;
                Bv = Bv + tB;
                S = S + uA;
                SB = SB + uB;
            } else {
                float dl = xsrc[(size_t)s * xld + e] - M;         // Stepper (Transformation.cpp:12 / :50)
                if (KIND == VSOM_MEDIAN)
                    dl = tiny_sign(dl);
                const float t = cc * dl;
                M = M + t;                                   // :864
                float u;
                if (KIND == VSOM_MEDIAN) {
                    u = w * __builtin_fabsf(dl);             // = (w * s) * s exactly for s in {-1, +-0, 1, NaN}
                } else {
                    u = w * dl;
                    u = u * dl;
                }
                S = S + u;                                   // :867
            }
        }
        const size_t row = (size_t)node * a.pitch;
        a.map[row + e] = M;                                  // :870
        a.sigma[row + e] = sqrtf(S / Wsum);                  // :873
        if (CLR) {
            a.map[row + a.ppitch + e] = Bv;
            a.sigma[row + a.ppitch + e] = sqrtf(SB / Wsum);
        }
        if (e == 0)
            a.weight[node] = Wsum;                           // :875
    }
}

// the fused path applies when the whole epoch is a handful of microseconds of work for one workgroup
bool vsom_tiny_applies(const vsom_ctx *c)
{
    const size_t chains = (size_t)c->N * c->part_len;
    return c->use_tiny && c->B > 0 && c->B <= 256 && chains <= 4096 && (size_t)c->B * c->N <= 16384 &&
           chains * c->B <= 262144;
}

int launch_tiny_epoch(vsom_ctx *c, double sigma, int is_first)
{
    int rc = ensure_lut(c, sigma);
    if (rc)
        return rc;
    TinyArgs a;
    const bool clr = c->transform == VSOM_CLR;
    a.d.xa = clr ? c->XP : c->Xs;
    a.d.xb = clr ? c->YP : c->Xs;
    a.d.ldx = (int)(clr ? c->part_pitch : c->xpitch);
    a.d.ma = c->map;
    a.d.mb = clr ? c->map + c->part_pitch : c->map;
    a.d.ldm = (int)c->pitch;
    a.d.L = (int)c->part_len;
    a.map = c->map;
    a.sigma = c->sigma;
    a.weight = c->weight;
    a.hits = c->hits;
    a.lastbmu = c->lastbmu;
    a.sqres = c->sqres;
    a.mse = c->mse;
    a.lut = c->lut;
    a.lutw = (int)c->lut_w;
    a.N = (int)c->N;
    a.W = (int)c->W;
    a.H = (int)c->H;
    a.Dc = (int)c->part_len;
    a.pitch = (int)c->pitch;
    a.ppitch = (int)c->part_pitch;
    a.B = (int)c->B;
    a.is_first = is_first;
    size_t smem = c->B * (sizeof(u64) + sizeof(int2) + sizeof(float) + sizeof(int));
    const size_t xfloats = c->B * (size_t)c->part_len * (clr ? 2 : 1);
    a.stage_x = xfloats <= 10240 ? 1 : 0;            // 40 KB on top of the per-sample arrays (< 64 KB in all)
    if (a.stage_x)
        smem += xfloats * sizeof(float);
    a.luth = (int)c->lut_h;
    smem += (size_t)c->lut_w * c->lut_h * sizeof(float);    // N <= 4096 here: at most 16 KB
    TimerScope ts(c, VSOM_T_UPDATE);
    if (c->transform == VSOM_CLR)
        hipLaunchKernelGGL(tiny_batch_epoch_kernel<VSOM_CLR>, dim3(1), dim3(256), smem, c->stream, a);
    else if (c->transform == VSOM_MEDIAN)
        hipLaunchKernelGGL(tiny_batch_epoch_kernel<VSOM_MEDIAN>, dim3(1), dim3(256), smem, c->stream, a);
    else
        hipLaunchKernelGGL(tiny_batch_epoch_kernel<VSOM_STANDARD>, dim3(1), dim3(256), smem, c->stream, a);
    VSOM_HIP_CHECK(hipGetLastError());
    return VSOM_OK;
}
