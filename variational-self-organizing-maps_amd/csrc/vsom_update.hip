// vsom_update.hip -- phase 2 of Som::trainBatchSomEpoch (Som.cpp:809-876) on gfx950.
//
// For every node i (independent) and the samples j of the chunk in load order:
//     w = (float)calculateNeighbourhoodWeight(node, bmu_j, sigma)       Som.cpp:851, 949-975
//     W += w ; c = w / W                                                  :857, :864
//     delta = Stepper(x_j, M) ; M = M + c*delta ; S = S + (w*delta)*delta  :861-867
//   map[i] = M ; sigmaMap[i] = sqrt(S / W) ; weightMap[i] = W             :870-875
//
// Split into
//   bxy_kernel   : SomIndex(*this, lastBMU[j]) once per sample               (:847-849)
//   cwp_kernel   : the neighbourhood chain per node: LUT lookup of w (the double exp is tabulated on
//                  the host over (|dx|,|dy|), bit-identical), the serial fp32 prefix sum W and
//                  c = w/W -> cw[j][i] = (c, w).  Role-split workgroups: one wavefront only adds,
//                  eight look up / divide / store (cw_kernel / cw16_kernel are the earlier
//                  redundant-chain versions, kept behind VSOM_CW_MODE for comparison).
//   update_*     : the N*D chains.
//                  - vsom_update_{std,fma,med}_rd{14,16}_gfx950, vsom_update_clr_rp8_gfx950: hand-scheduled
//                    code object (gen_update_asm.py): lane = node, RD dims (8 CLR pairs) per lane in
//                    VGPRs, x rows through scalar loads (SGPR operands of v_pk_* ops), a ring of
//                    (c,w) loads always in flight, x rows prefetched into L2.  A ragged depth is
//                    covered by a 16/14 column split or by a last slice that runs into the rows'
//                    zero padding (vsom_update_split; the padding is re-zeroed afterwards).  `med`:
//                    the Median sign from packed clamped multiplications, bit-identical (compute_median).
//                  - update_kernel, update_clr_kernel (VSOM_NO_ASM, and depths the assembly kernels do not
//                    cover): the same decomposition in HIP, sample pairs software-pipelined.
//                  - update_chain_kernel: one lane per (node, dim) chain for maps too small to fill
//                    the chip with lane = node.
//                  Every fp32 operation is rounded separately (-ffp-contract=off), so the result is
//                  bit-identical to the reference's SSE2 build (VSOM_UPDATE_FMA opts out, 1e-5).
//   sigma_finalize_kernel : sigmaMap = sqrt(S / W) for the columns the assembly kernels left as S
//                  (+ zeroes of the padding columns a ragged last slice wrote).
#include "vsom_device.hpp"
#include <cmath>
#include <cstdlib>
#include <mutex>

// (c,w) layout: pair-interleaved, float2 at ((j>>1)*ldn + node)*2 + (j&1), i.e. one float4
// {c_j, w_j, c_j+1, w_j+1} per node and sample pair -- the assembly kernel (gen_update_asm.py)
// fetches it with one global_load_dwordx4 per two samples.
__device__ __forceinline__ size_t cw2_index(int j, int ldn, int nl)
{
    return (((size_t)(j >> 1) * ldn + nl) << 1) + (j & 1);
}

typedef const __attribute__((address_space(4))) float *vsom_cfp;   // forces s_load_* (scalar cache)

__global__ void bxy_kernel(const u64 *__restrict__ lastbmu, int B, int W, int H,
                           int2 *__restrict__ bxy)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= B)
        return;
    int x, y;
    vsom_somindex(lastbmu[j], (u64)W, (u64)H, x, y);
    bxy[j] = make_int2(x, y);
}

// Neighbourhood chain.  Only the fp32 prefix sum W_j = W_{j-1} + w_j is inherently serial; the
// table lookup of w and the division c = w/W are not.  A quad of 4 lanes serves one node: lane q
// of the quad looks up / divides / stores the samples j = 4r+q, while all four lanes run the
// same serial chain redundantly (w of the other lanes arrives through DPP quad broadcasts), so
// no lane ever waits for a cross-lane hand-off.  16 nodes per wavefront, N/16 wavefronts.
#define CWR 4   // rounds (of 4 samples) in flight per loop iteration
template <bool LUT_LDS>
__global__ __launch_bounds__(256) void cw_kernel(const int2 *__restrict__ bxy, int B, int n0, int n1,
                                                 int W, int H, const float *__restrict__ lut,
                                                 int lutw, int luth, float2 *__restrict__ cw, int ldn,
                                                 float *__restrict__ weight)
{
    extern __shared__ float slut[];
    if (LUT_LDS) {
        for (int i = threadIdx.x; i < lutw * luth; i += blockDim.x)
            slut[i] = lut[i];
        __syncthreads();
    }
    const float *tab = LUT_LDS ? slut : lut;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = gid & 3;
    const int nl = gid >> 2;
    const int node = n0 + nl;
    const bool valid = node < n1;
    int cx = 0, cy = 0;
    if (valid)
        vsom_somindex((u64)node, (u64)W, (u64)H, cx, cy);   // SomIndex(*this, index) (Som.cpp:816)
    float run = 0.f;                                         // sumOfWeights :840
    const int nlo = valid ? nl : 0;
    const int Bq = B & ~3;
    int j = 0;
    for (; j + 4 * CWR <= Bq; j += 4 * CWR) {
        float w[CWR], Wm[CWR];
#pragma unroll
        for (int r = 0; r < CWR; ++r) {
            int2 b = bxy[j + 4 * r + q];
            int dx = cx - b.x, dy = cy - b.y;
            dx = dx < 0 ? -dx : dx;
            dy = dy < 0 ? -dy : dy;
            w[r] = tab[dy * lutw + dx];      // (float)calculateNeighbourhoodWeight(...)  :851
        }
#pragma unroll
        for (int r = 0; r < CWR; ++r) {
            // samples 4r..4r+3 in order; every lane of the quad runs the same additions
            float w0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, w[r]), 0x00, 0xF, 0xF, true));
            float w1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, w[r]), 0x55, 0xF, 0xF, true));
            float w2 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, w[r]), 0xAA, 0xF, 0xF, true));
            float w3 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, w[r]), 0xFF, 0xF, 0xF, true));
            float W0 = run + w0;             // :857
            float W1 = W0 + w1;
            float W2 = W1 + w2;
            float W3 = W2 + w3;
            run = W3;
            Wm[r] = q == 0 ? W0 : (q == 1 ? W1 : (q == 2 ? W2 : W3));
        }
        if (valid) {
#pragma unroll
            for (int r = 0; r < CWR; ++r)
                cw[cw2_index(j + 4 * r + q, ldn, nlo)] = make_float2(w[r] / Wm[r], w[r]);   // c = w/W :864 (0/0 -> NaN, Q7)
        }
    }
    for (; j < B; ++j) {   // tail, every lane of the quad redundantly; lane 0 stores
        int2 b = bxy[j];
        int dx = cx - b.x, dy = cy - b.y;
        dx = dx < 0 ? -dx : dx;
        dy = dy < 0 ? -dy : dy;
        float w = tab[dy * lutw + dx];
        run = run + w;
        if (valid && q == 0)
            cw[cw2_index(j, ldn, nlo)] = make_float2(w / run, w);
    }
    if (valid && q == 0)
        weight[node] = run;   // :875
}

// Same chain with 16 lanes per node (4 nodes per wavefront, N/4 wavefronts) for maps too small to
// occupy the chip with quads: lane q of a group looks up / divides / stores sample j+q of each
// round of 16; the serial prefix runs redundantly in all 16 lanes, fed by ds_bpermute broadcasts.
template <bool LUT_LDS>
__global__ __launch_bounds__(256) void cw16_kernel(const int2 *__restrict__ bxy, int B, int n0, int n1,
                                                   int W, int H, const float *__restrict__ lut,
                                                   int lutw, int luth, float2 *__restrict__ cw, int ldn,
                                                   float *__restrict__ weight)
{
    extern __shared__ float slut[];
    if (LUT_LDS) {
        for (int i = threadIdx.x; i < lutw * luth; i += blockDim.x)
            slut[i] = lut[i];
        __syncthreads();
    }
    const float *tab = LUT_LDS ? slut : lut;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = gid & 15;
    const int nl = gid >> 4;
    const int node = n0 + nl;
    const bool valid = node < n1;
    int cx = 0, cy = 0;
    if (valid)
        vsom_somindex((u64)node, (u64)W, (u64)H, cx, cy);   // SomIndex(*this, index) (Som.cpp:816)
    float run = 0.f;                                         // sumOfWeights :840
    const int nlo = valid ? nl : 0;
    const int B16 = B & ~15;
    int j = 0;
    int2 b = B16 > 0 ? bxy[q] : make_int2(0, 0);
    for (; j < B16; j += 16) {
        int dx = cx - b.x, dy = cy - b.y;
        dx = dx < 0 ? -dx : dx;
        dy = dy < 0 ? -dy : dy;
        const float w = tab[dy * lutw + dx];     // (float)calculateNeighbourhoodWeight(...)  :851
        if (j + 16 < B16)
            b = bxy[j + 16 + q];
        float wk[16];
#pragma unroll
        for (int k = 0; k < 16; ++k)
            wk[k] = __shfl(w, k, 16);
        float Wm = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            run = run + wk[k];                   // :857, samples j..j+15 in order
            Wm = q == k ? run : Wm;
        }
        if (valid)
            cw[cw2_index(j + q, ldn, nlo)] = make_float2(w / Wm, w);   // c = w/W :864 (0/0 -> NaN, Q7)
    }
    for (; j < B; ++j) {   // tail, every lane of the group redundantly; lane 0 stores
        int2 bb = bxy[j];
        int dx = cx - bb.x, dy = cy - bb.y;
        dx = dx < 0 ? -dx : dx;
        dy = dy < 0 ? -dy : dy;
        float w = tab[dy * lutw + dx];
        run = run + w;
        if (valid && q == 0)
            cw[cw2_index(j, ldn, nlo)] = make_float2(w / run, w);
    }
    if (valid && q == 0)
        weight[node] = run;   // :875
}

// Role-split neighbourhood chain: one workgroup (9 wavefronts) serves NW nodes and walks the chunk
// in tiles of T samples (NW*T = 2048) through LDS.  Wavefront 0 does nothing but the serial part --
// the fp32 prefix W_j = W_{j-1} + w_j of tile t (one LDS read, one add, one LDS write per sample);
// wavefronts 1..8 meanwhile look up w for tile t+1 and turn tile t-1 into (c = w/W, w) pairs,
// stored as one float4 per node and sample pair.  No redundant chain work, so the time per
// workgroup is the chain's own ~B dependent adds instead of B x (lookup + division + chain).
#define CWP_WT 512                               // worker threads
#define CWP_THREADS (64 + CWP_WT)
template <int NW, int T, bool LUT_LDS>
__global__ __launch_bounds__(CWP_THREADS) void cwp_kernel(const int2 *__restrict__ bxy, int B, int n0, int n1,
                                                          int W, int H, const float *__restrict__ lut,
                                                          int lutw, int luth, float2 *__restrict__ cw, int ldn,
                                                          float *__restrict__ weight)
{
    static_assert(NW * T == 2048 && CWP_WT % NW == 0 && T % 2 == 0 && T <= CWP_WT, "worker mapping");
    constexpr int LK = NW * T / CWP_WT;          // lookups per worker thread and tile (4)
    constexpr int PK = LK / 2;                   // sample pairs per worker thread and tile (2)
    constexpr int SS = CWP_WT / NW;              // sample stride between a thread's elements
    extern __shared__ __attribute__((aligned(16))) unsigned char cwp_smem[];
    float *wL = (float *)cwp_smem;               // [3][T][NW]  w of tiles t-1, t, t+1
    float *WL = wL + 3 * T * NW;                 // [2][T][NW]  prefix sums of tiles t-1, t
    int2 *bL = (int2 *)(WL + 2 * T * NW);        // [3][T]      BMU coordinates of tiles t .. t+2
    float *slut = (float *)(bL + 3 * T);
    const int tid = threadIdx.x;
    if (LUT_LDS) {
        for (int i = tid; i < lutw * luth; i += CWP_THREADS)
            slut[i] = lut[i];
    }
    const float *tab = LUT_LDS ? slut : lut;
    const int ntiles = (B + T - 1) / T;
    const int nloc = n1 - n0;
    const int wt = tid - 64;                     // worker thread 0..511 (wavefronts 1..8)
    const int lnode = (wt < 0 ? tid : wt) & (NW - 1);
    const int s0 = wt < 0 ? 0 : wt / NW;         // first sample (lookup) / pair (emit) of this thread
    const int nl = blockIdx.x * NW + lnode;
    const bool valid = nl < nloc;
    int cx = 0, cy = 0;
    if (valid)
        vsom_somindex((u64)(n0 + nl), (u64)W, (u64)H, cx, cy);   // SomIndex(*this, index) (Som.cpp:816)

    auto load_bxy = [&](int t) {                 // workers: coordinates of tile t -> ring slot t%3
        if (t < ntiles && wt < T) {
            const int j = t * T + wt;
            bL[(t % 3) * T + wt] = j < B ? bxy[j] : make_int2(0, 0);
        }
    };
    // Both worker stages read everything first and write afterwards: the compiler does not move LDS
    // reads across LDS writes, so this is what keeps several reads in flight per thread.
    auto lookup = [&](int t) {                   // workers: w of tile t
        float *dst = wL + (t % 3) * T * NW;
        const int2 *bsrc = bL + (t % 3) * T;
        const int nt = B - t * T < T ? B - t * T : T;
        int2 b[LK];
        float w[LK];
#pragma unroll
        for (int k = 0; k < LK; ++k)
            b[k] = bsrc[s0 + k * SS];            // rows beyond nt hold (0,0) or stale coordinates: harmless
#pragma unroll
        for (int k = 0; k < LK; ++k) {
            int dx = cx - b[k].x, dy = cy - b[k].y;
            dx = dx < 0 ? -dx : dx;
            dy = dy < 0 ? -dy : dy;
            dx = dx < lutw ? dx : lutw - 1;
            dy = dy < luth ? dy : luth - 1;
            w[k] = tab[dy * lutw + dx];          // (float)calculateNeighbourhoodWeight(...) :851
        }
#pragma unroll
        for (int k = 0; k < LK; ++k)
            if (s0 + k * SS < nt)
                dst[(s0 + k * SS) * NW + lnode] = w[k];
    };
    auto emit = [&](int t) {                     // workers: (c,w) pairs of tile t
        const float *ws = wL + (t % 3) * T * NW;
        const float *Ws = WL + (t & 1) * T * NW;
        const int nt = B - t * T < T ? B - t * T : T;
        float w0[PK], w1[PK], W0[PK], W1[PK];
#pragma unroll
        for (int k = 0; k < PK; ++k) {
            const int s = (s0 + k * SS) * 2;
            w0[k] = ws[s * NW + lnode];
            W0[k] = Ws[s * NW + lnode];
            w1[k] = ws[(s + 1) * NW + lnode];
            W1[k] = Ws[(s + 1) * NW + lnode];
        }
#pragma unroll
        for (int k = 0; k < PK; ++k) {
            const int s = (s0 + k * SS) * 2;
            float4 o;
            o.x = w0[k] / W0[k];                 // c = w/W :864 (0/0 -> NaN, Q7)
            o.y = w0[k];
            const bool two = s + 1 < nt;         // odd tail: the partner slot is never read
            o.z = two ? w1[k] / W1[k] : 0.f;
            o.w = two ? w1[k] : 0.f;
            if (valid && s < nt)
                ((float4 *)cw)[(size_t)((t * T + s) >> 1) * ldn + nl] = o;
        }
    };

    float run = 0.f;                             // sumOfWeights :840
    if (wt >= 0) {
        load_bxy(0);
        load_bxy(1);
    }
    __syncthreads();
    if (wt >= 0)
        lookup(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        if (wt >= 0) {
            load_bxy(t + 2);
            if (t + 1 < ntiles)
                lookup(t + 1);
            if (t > 0)
                emit(t - 1);
        } else {
            const float *ws = wL + (t % 3) * T * NW;
            float *Ws = WL + (t & 1) * T * NW;
            const int nt = B - t * T < T ? B - t * T : T;
            int s = 0;
            // 16 reads issued together, then the 16 dependent additions, then the 16 writes
            for (; s + 16 <= nt; s += 16) {
                float r[16], o[16];
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    r[k] = ws[(s + k) * NW + lnode];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    run = run + r[k];                        // :857
                    o[k] = run;
                }
                if (tid < NW) {
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        Ws[(s + k) * NW + lnode] = o[k];
                }
            }
            for (; s < nt; ++s) {
                run = run + ws[s * NW + lnode];
                if (tid < NW)
                    Ws[s * NW + lnode] = run;
            }
        }
        __syncthreads();
    }
    if (wt >= 0)
        emit(ntiles - 1);
    else if (tid < NW && valid)
        weight[n0 + nl] = run;                   // :875
}

// Eigen scalar_sign_op<float> (Transformation.cpp:50) as the chains need it: +-1 for a nonzero
// number, NaN for NaN, and the argument itself for +-0.  The reference yields +0 for -0; inside the
// chains the difference cannot surface: t = c*s is -0 instead of +0 (or NaN either way when c is
// NaN/inf), M + (-0) == M + (+0) because M is never -0 (it starts at +0 and x + y is -0 only for
// -0 + -0), and (w*s)*s is +0 either way.  Three VALU ops (v_bfi, v_cmp_lg, v_cndmask) instead of
// five -- the sign is what the Median kernels spend their time on.
__device__ __forceinline__ float vsom_sign(float a)
{
    const float one = __builtin_copysignf(1.f, a);
    return (a < 0.f || a > 0.f) ? one : a;
}

// (w * s) * s of the sigma^2 accumulation (Som.cpp:867) when s = sign(..) is -1, +-0, +1 or NaN and w a
// finite weight >= 0: both products are exact, and equal w * |s| -- w for +-1, +0 for +-0 (w*(-0) = -0,
// (-0)*(-0) = +0), NaN for NaN.  One multiplication with a source modifier instead of two.
__device__ __forceinline__ float vsom_median_sq(float w, float s)
{
    return w * __builtin_fabsf(s);
}

// Standard / Median: lane = node, RD dims per lane, 4 waves per workgroup = 4 dim slices
template <int RD, bool MEDIAN>
__global__ __launch_bounds__(256) void update_kernel(const float *__restrict__ Xs, int ldx,
                                                     const float2 *__restrict__ cw, int ldn, int B,
                                                     int n0, int nloc, int D, int nslices, int dbase,
                                                     float *__restrict__ map,
                                                     float *__restrict__ sigma, int pitch,
                                                     const float *__restrict__ weight)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int slice = blockIdx.y * 4 + wave;
    if (slice >= nslices)
        return;
    const int d0 = dbase + slice * RD;   // dbase: first dim not covered by the assembly kernel
    const int nl = blockIdx.x * 64 + lane;
    const bool valid = nl < nloc;
    const int nlc = valid ? nl : nloc - 1;

    float M[RD], S[RD];
#pragma unroll
    for (int k = 0; k < RD; ++k) {
        M[k] = 0.f;   // currentModel.setZero()       :843
        S[k] = 0.f;   // currentModelSigma.setZero()  :844
    }
    vsom_cfp xr = (vsom_cfp)(Xs + d0);
    const float4 *cp = (const float4 *)cw + nlc;     // {c_j, w_j, c_j+1, w_j+1} of pair row r at cp[r * ldn]
    // Sample pairs, software-pipelined: the scalar loads of x and the (c,w) load of pair r+1 are
    // issued before pair r is consumed (two register sets used alternately).  Reads one pair past
    // the chunk at most: Xs has B + VSOM_ROW_PAD rows, cw ceil(B/2)+8 pair rows.
    auto load = [&](float (&xa)[RD], float (&xb)[RD], float4 &cv, int r) {
        vsom_cfp p0 = xr + (size_t)(2 * r) * ldx;
        vsom_cfp p1 = p0 + ldx;
#pragma unroll
        for (int k = 0; k < RD; ++k) {
            xa[k] = p0[k];
            xb[k] = p1[k];
        }
        cv = cp[(size_t)r * ldn];
    };
    auto step = [&](const float (&x)[RD], float c, float w) {
#pragma unroll
        for (int k = 0; k < RD; ++k) {
            float dl = x[k] - M[k];         // Stepper: value - model        (Transformation.cpp:12)
            if (MEDIAN)
                dl = vsom_sign(dl);         //          sign(value - model)  (Transformation.cpp:50)
            float t = c * dl;
            M[k] = M[k] + t;                // :864
            float u = w * dl;           // (packed: v_pk_mul_f32 has no |x| modifier, so no vsom_median_sq here)
            u = u * dl;
            S[k] = S[k] + u;                // :867
        }
    };
    const int npair = B >> 1;
    float xa0[RD], xa1[RD], xb0[RD], xb1[RD];
    float4 ca, cb;
    load(xa0, xa1, ca, 0);
    int r = 0;
    for (; r + 2 <= npair; r += 2) {
        load(xb0, xb1, cb, r + 1);
        step(xa0, ca.x, ca.y);
        step(xa1, ca.z, ca.w);
        load(xa0, xa1, ca, r + 2);
        step(xb0, cb.x, cb.y);
        step(xb1, cb.z, cb.w);
    }
    if (r < npair) {                       // one more full pair (set a holds it)
        load(xb0, xb1, cb, r + 1);
        step(xa0, ca.x, ca.y);
        step(xa1, ca.z, ca.w);
        if (B & 1)
            step(xb0, cb.x, cb.y);         // odd tail sample = first half of the next pair row
    } else if (B & 1) {
        step(xa0, ca.x, ca.y);
    }
    if (valid) {
        const size_t node = (size_t)(n0 + nl);
        const float Wf = weight[node];
#pragma unroll
        for (int k = 0; k < RD; ++k) {
            if (d0 + k < D) {
                map[node * pitch + d0 + k] = M[k];                   // :870
                sigma[node * pitch + d0 + k] = sqrtf(S[k] / Wf);     // :873
            }
        }
    }
}

// Standard / Median on maps whose node count cannot fill the chip with lane = node (fewer than a
// few hundred wavefronts, e.g. C4: 64x64x32): lane = one (node, dim) chain.  A group of DL =
// 2^dl_log2 consecutive lanes covers DL consecutive dims of one node (x is a coalesced row
// segment, (c,w) a same-address broadcast), 256/DL nodes per workgroup, blockIdx.y = dim slice.
// Loads of the next U samples are issued before the current U are consumed.  Same fp32 operation
// sequence per chain as update_kernel, so the results are bit-identical.
template <bool MEDIAN, int U, int FMA>
__global__ __launch_bounds__(256) void update_chain_kernel(const float *__restrict__ Xs, int ldx,
                                                           const float2 *__restrict__ cw, int ldn, int B,
                                                           int n0, int nloc, int D, int dl_log2,
                                                           float *__restrict__ map,
                                                           float *__restrict__ sigma, int pitch,
                                                           const float *__restrict__ weight)
{
    static_assert(U % 2 == 0, "pairs of samples share one float4 of (c,w)");
    const int DL = 1 << dl_log2;
    const int nl = blockIdx.x * (256 >> dl_log2) + ((int)threadIdx.x >> dl_log2);
    const int d = blockIdx.y * DL + ((int)threadIdx.x & (DL - 1));
    const bool valid = nl < nloc && d < D;
    const int nlc = nl < nloc ? nl : nloc - 1;
    const int dc = d < D ? d : D - 1;

    const float *xp = Xs + dc;
    const float4 *cp = (const float4 *)cw + nlc;      // pair row r at cp[r * ldn]
    float M = 0.f, S = 0.f;                           // :843-844

    // two register sets used alternately (no copies): while one is consumed the loads of the
    // following group are already in flight into the other
    float xa[U], xb[U];
    float4 ca[U / 2], cb[U / 2];
    const int nfull = B / U;
    auto load = [&](float (&x)[U], float4 (&cv)[U / 2], int g) {
        // clamped to the last full group: re-reads it, never past the buffers
        g = g < nfull ? g : nfull - 1;
        const float *xq = xp + (size_t)g * U * ldx;
        const float4 *cq = cp + (size_t)g * (U / 2) * ldn;
#pragma unroll
        for (int u = 0; u < U; ++u)
            x[u] = xq[(size_t)u * ldx];
#pragma unroll
        for (int u = 0; u < U / 2; ++u)
            cv[u] = cq[(size_t)u * ldn];
    };
    auto steps = [&](const float (&x)[U], const float4 (&cv)[U / 2]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float c = (u & 1) ? cv[u >> 1].z : cv[u >> 1].x;
            const float w = (u & 1) ? cv[u >> 1].w : cv[u >> 1].y;
            float dl = x[u] - M;            // Stepper (Transformation.cpp:12 / :50)
            if (MEDIAN)
                dl = vsom_sign(dl);
            if (FMA == 1) {                 // opt-in contracted arithmetic (VSOM_UPDATE_FMA)
                M = __builtin_fmaf(c, dl, M);
                S = __builtin_fmaf(w * dl, dl, S);
            } else if (FMA == 2 && !MEDIAN) {   // VSOM_UPDATE_FMA_SIGMA: the mean chain strict, only S contracted
                float t = c * dl;
                M = M + t;
                S = __builtin_fmaf(w * dl, dl, S);
            } else {
                float t = c * dl;
                M = M + t;                  // :864
                float s;
                if (MEDIAN) {
                    s = vsom_median_sq(w, dl);
                } else {
                    s = w * dl;
                    s = s * dl;
                }
                S = S + s;                  // :867
            }
        }
    };
    if (nfull > 0)
        load(xa, ca, 0);
    int g = 0;
    for (; g + 2 <= nfull; g += 2) {
        load(xb, cb, g + 1);
        steps(xa, ca);
        load(xa, ca, g + 2);
        steps(xb, cb);
    }
    if (g < nfull)
        steps(xa, ca);
    for (int j = nfull * U; j < B; ++j) {
        const float2 v = cw[cw2_index(j, ldn, nlc)];
        float dl = xp[(size_t)j * ldx] - M;
        if (MEDIAN)
            dl = vsom_sign(dl);
        if (FMA == 1) {
            M = __builtin_fmaf(v.x, dl, M);
            S = __builtin_fmaf(v.y * dl, dl, S);
        } else if (FMA == 2 && !MEDIAN) {
            float t = v.x * dl;
            M = M + t;
            S = __builtin_fmaf(v.y * dl, dl, S);
        } else {
            float t = v.x * dl;
            M = M + t;
            float s;
            if (MEDIAN) {
                s = vsom_median_sq(v.y, dl);
            } else {
                s = v.y * dl;
                s = s * dl;
            }
            S = S + s;
        }
    }
    if (valid) {
        const size_t node = (size_t)(n0 + nl);
        const float Wf = weight[node];
        map[node * pitch + d] = M;                   // :870
        sigma[node * pitch + d] = sqrtf(S / Wf);     // :873
    }
}

// The same chains with TWO dims per lane (packed fp32 arithmetic) for the shapes where even lane = (node, dim)
// yields no more than two wavefronts per SIMD (C4: 64x64x32): half the wavefronts, each with a ring of
// NG groups of U samples of loads in flight -- a single wavefront per SIMD has the registers for it (up to
// 512) and nothing else to cover the L2 / HBM latency with.  Median: the sign through clamped
// multiplications (see gen_update_asm.py, compute_median: p = [delta > 0], n = [delta < 0], exact fused
// accumulation, bit-identical); the clamp must pass NaN, so the kernel switches DX10_CLAMP off for itself.
typedef float vsom_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ vsom_f2 vsom_pk_mul_clamp(vsom_f2 a, vsom_f2 b)
{
    vsom_f2 r;
    asm volatile("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ vsom_f2 vsom_pk_mul_negclamp(vsom_f2 a, vsom_f2 b)
{
    vsom_f2 r;
    asm volatile("v_pk_mul_f32 %0, %1, %2 neg_lo:[1,0] neg_hi:[1,0] clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <bool MEDIAN, int U, int NG, int FMA>
__global__ __launch_bounds__(256, 1) void update_chain2_kernel(const float *__restrict__ Xs, int ldx,
                                                               const float2 *__restrict__ cw, int ldn, int B,
                                                               int n0, int nloc, int D, int dl_log2,
                                                               float *__restrict__ map,
                                                               float *__restrict__ sigma, int pitch,
                                                               const float *__restrict__ weight)
{
    static_assert(U % 2 == 0, "pairs of samples share one float4 of (c,w)");
    if (MEDIAN)
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 8, 1), 0");   // DX10_CLAMP off: clamp(NaN) = NaN
    const int DL = 1 << dl_log2;                       // dim PAIRS per node row of the workgroup
    const int nl = blockIdx.x * (256 >> dl_log2) + ((int)threadIdx.x >> dl_log2);
    const int dp = blockIdx.y * DL + ((int)threadIdx.x & (DL - 1));
    const int d = 2 * dp;                              // rows are zero padded to a multiple of 32: d+1 is readable
    const bool valid = nl < nloc && d < D;
    const int nlc = nl < nloc ? nl : nloc - 1;
    const int dc = d < D ? d : (D - 1) & ~1;

    const float *xp = Xs + dc;
    const float4 *cp = (const float4 *)cw + nlc;      // pair row r at cp[r * ldn]
    vsom_f2 M = {0.f, 0.f}, S = {0.f, 0.f};            // :843-844
    const vsom_f2 big = {0x1.0p100f, 0x1.0p100f};

    vsom_f2 x[NG][U];
    float4 cv[NG][U / 2];
    const int nfull = B / U;
    auto load = [&](int slot, int g) {
        g = g < nfull ? g : nfull - 1;                 // clamped: re-reads the last full group, never past the buffers
        const float *xq = xp + (size_t)g * U * ldx;
        const float4 *cq = cp + (size_t)g * (U / 2) * ldn;
#pragma unroll
        for (int u = 0; u < U; ++u)
            x[slot][u] = *reinterpret_cast<const vsom_f2 *>(xq + (size_t)u * ldx);
#pragma unroll
        for (int u = 0; u < U / 2; ++u)
            cv[slot][u] = cq[(size_t)u * ldn];
    };
    auto one = [&](vsom_f2 xv, float c, float w) {
        const vsom_f2 cc = {c, c}, ww = {w, w};
        vsom_f2 dl = xv - M;                           // Stepper (Transformation.cpp:12 / :50)
        if (MEDIAN) {
            const vsom_f2 t = dl * big;
            const vsom_f2 pp = vsom_pk_mul_clamp(t, big), nn = vsom_pk_mul_negclamp(t, big);
            M = __builtin_elementwise_fma(cc, pp, M);  // exact products: rounds like mul + add (:864)
            M = __builtin_elementwise_fma(-cc, nn, M);
            S = __builtin_elementwise_fma(ww, pp, S);  // (:867)
            S = __builtin_elementwise_fma(ww, nn, S);
        } else if (FMA == 1) {
            M = __builtin_elementwise_fma(cc, dl, M);
            S = __builtin_elementwise_fma(ww * dl, dl, S);
        } else if (FMA == 2) {                         // VSOM_UPDATE_FMA_SIGMA: mean chain strict
            const vsom_f2 t = cc * dl;
            M = M + t;
            S = __builtin_elementwise_fma(ww * dl, dl, S);
        } else {
            const vsom_f2 t = cc * dl;
            M = M + t;                                 // :864
            vsom_f2 q = ww * dl;
            q = q * dl;
            S = S + q;                                 // :867
        }
    };
    auto steps = [&](int slot) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float c = (u & 1) ? cv[slot][u >> 1].z : cv[slot][u >> 1].x;
            const float w = (u & 1) ? cv[slot][u >> 1].w : cv[slot][u >> 1].y;
            one(x[slot][u], c, w);
        }
    };
    if (nfull > 0) {
#pragma unroll
        for (int k = 0; k < NG - 1; ++k)
            load(k, k);
    }
    int g = 0;
    for (; g + NG <= nfull; g += NG) {
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            load((k + NG - 1) % NG, g + k + NG - 1);   // NG-1 groups ahead of the one consumed next
            steps(k);
        }
    }
#pragma unroll
    for (int k = 0; k < NG - 1; ++k)                   // the (< NG) full groups left are already in their slots
        if (g + k < nfull)
            steps(k);
    for (int j = nfull * U; j < B; ++j) {
        const float2 v = cw[cw2_index(j, ldn, nlc)];
        one(*reinterpret_cast<const vsom_f2 *>(xp + (size_t)j * ldx), v.x, v.y);
    }
    if (valid) {
        const size_t node = (size_t)(n0 + nl);
        const float Wf = weight[node];
        map[node * pitch + d] = M.x;                   // :870
        sigma[node * pitch + d] = sqrtf(S.x / Wf);     // :873
        if (d + 1 < D) {
            map[node * pitch + d + 1] = M.y;
            sigma[node * pitch + d + 1] = sqrtf(S.y / Wf);
        }
    }
}

// StandardMedianEstimator steps of EIGHT consecutive samples for one lane's packed (dim, dim+1) chains, as one
// hand-scheduled block (update_chain3_kernel runs ONE wavefront per SIMD, where a single wavefront issues an
// instruction every ~5.3 cycles and a dependent one after ~10 -- tools/exp/pk_latency_bench.hip -- so what counts is
// the depth of the per-sample dependency chain and having no bubbles in it; hipcc's version of this loop carried
// a 5-deep chain plus s_nop / v_mov padding around the clamp instructions: ~72 cycles per sample).
// Per sample (Transformation.cpp:50, Som.cpp:861-867):   s = sign(x - M) ; M = M + c*s ; S = S + (w*s)*s.
//   t  = fma(M, -2^24, x*2^24)          the sign of t IS the sign of x - M: the scaling by 2^24 is exact, the fused
//                                       difference rounds once and never to zero (|x - M| >= 2^-149 -> |t| >= 2^-125),
//                                       NaN stays NaN, +-inf keeps its sign; x*2^24 is computed off the chain (by the
//                                       pass that stages the block in LDS) and may overflow to +-inf
//                                       only for |x| >= 2^104, where sign(x - M) = sign(x) because the median walk
//                                       keeps |M| <= sum of c <= B; M*2^24 cannot overflow for the same reason
//   p  = clamp(t * 2^127), n = clamp(-t * 2^127)     [t > 0], [t < 0] as 1.0 / 0.0 (DX10_CLAMP off: NaN passes)
//   M  = fma(c, p, M) ; M = fma(-c, n, M) ; S = fma(w, p, S) ; S = fma(w, n, S)
//                                       exact products, one of p / n is zero: every fma rounds where the reference's
//                                       separate multiply and add round (gen_update_asm.py, compute_median)
// -> a 4-deep chain (t, p|n, M, M) with the three other operations in its shadows: 7 instructions per sample.
// cwK = {c, w} of sample K (op_sel picks the half), xK the sample's two values times 2^24.
#define VSOM_MED_STEP(XS, CW)                                                                    \
    "v_pk_fma_f32 %[t], %[M], %[k24], " XS " neg_lo:[0,1,0] neg_hi:[0,1,0]\n\t"                   \
    "v_pk_mul_f32 %[p], %[t], %[k127] clamp\n\t"                                                  \
    "v_pk_mul_f32 %[n], %[t], %[k127] neg_lo:[1,0] neg_hi:[1,0] clamp\n\t"                        \
    "v_pk_fma_f32 %[M], " CW ", %[p], %[M] op_sel_hi:[0,1,1]\n\t"                                 \
    "v_pk_fma_f32 %[S], " CW ", %[p], %[S] op_sel:[1,0,0]\n\t"                                    \
    "v_pk_fma_f32 %[M], " CW ", %[n], %[M] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"   \
    "v_pk_fma_f32 %[S], " CW ", %[n], %[S] op_sel:[1,0,0]\n\t"

// xs[] = the samples' values ALREADY scaled by 2^24 (the staging pass of update_chain3_kernel multiplies once per
// value and workgroup instead of once per value and node row)
__device__ __forceinline__ void vsom_median_steps8(vsom_f2 &M, vsom_f2 &S, const vsom_f2 (&xs)[8], const float4 (&cv)[4])
{
    const vsom_f2 k24 = {0x1.0p24f, 0x1.0p24f}, k127 = {0x1.0p127f, 0x1.0p127f};
    const vsom_f2 c0 = {cv[0].x, cv[0].y}, c1 = {cv[0].z, cv[0].w}, c2 = {cv[1].x, cv[1].y}, c3 = {cv[1].z, cv[1].w},
                  c4 = {cv[2].x, cv[2].y}, c5 = {cv[2].z, cv[2].w}, c6 = {cv[3].x, cv[3].y}, c7 = {cv[3].z, cv[3].w};
    vsom_f2 t, p, n;
    asm volatile(VSOM_MED_STEP("%[x0]", "%[c0]") VSOM_MED_STEP("%[x1]", "%[c1]") VSOM_MED_STEP("%[x2]", "%[c2]")
                 VSOM_MED_STEP("%[x3]", "%[c3]") VSOM_MED_STEP("%[x4]", "%[c4]") VSOM_MED_STEP("%[x5]", "%[c5]")
                 VSOM_MED_STEP("%[x6]", "%[c6]") VSOM_MED_STEP("%[x7]", "%[c7]")
                 : [M] "+v"(M), [S] "+v"(S), [t] "=&v"(t), [p] "=&v"(p), [n] "=&v"(n)
                 : [x0] "v"(xs[0]), [x1] "v"(xs[1]), [x2] "v"(xs[2]), [x3] "v"(xs[3]), [x4] "v"(xs[4]), [x5] "v"(xs[5]),
                   [x6] "v"(xs[6]), [x7] "v"(xs[7]), [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [c4] "v"(c4),
                   [c5] "v"(c5), [c6] "v"(c6), [c7] "v"(c7), [k24] "s"(k24), [k127] "s"(k127));
}

// one sample of the same (chunk tails)
__device__ __forceinline__ void vsom_median_step1(vsom_f2 &M, vsom_f2 &S, vsom_f2 xs, vsom_f2 cw)
{
    const vsom_f2 k24 = {0x1.0p24f, 0x1.0p24f}, k127 = {0x1.0p127f, 0x1.0p127f};
    vsom_f2 t, p, n;
    asm volatile(VSOM_MED_STEP("%[x0]", "%[c0]")
                 : [M] "+v"(M), [S] "+v"(S), [t] "=&v"(t), [p] "=&v"(p), [n] "=&v"(n)
                 : [x0] "v"(xs), [c0] "v"(cw), [k24] "s"(k24), [k127] "s"(k127));
}

// update_chain2_kernel with the operands staged ONCE per workgroup through LDS.  In chain2 every wavefront
// issues, per sample, a 512-byte x load and half a 1-KB (c,w) load whose lanes mostly repeat addresses --
// the vector-memory pipe processes every lane's address and return slot, and that, not arithmetic, bounded
// it (C4: Median and Standard both 0.86 ms).  Here the 256 lanes of a workgroup (NW = 256 >> PLOG nodes x
// PL = 1 << PLOG dim pairs) fetch each block of CT samples with 16-byte loads that touch every byte once
// (x: CT rows of PL pairs; (c,w): CT/2 pair rows of NW nodes), two blocks ahead of the one being consumed
// (registers -> LDS ring of three), and the chains read their operands from LDS as broadcasts.
template <bool MEDIAN, int FMA, int PLOG>
__global__ __launch_bounds__(256, 1) void update_chain3_kernel(const float *__restrict__ Xs, int ldx,
                                                               const float2 *__restrict__ cw, int ldn, int B,
                                                               int n0, int nloc, int D,
                                                               float *__restrict__ map,
                                                               float *__restrict__ sigma, int pitch,
                                                               const float *__restrict__ weight)
{
    constexpr int PL = 1 << PLOG, NW = 256 >> PLOG, CT = 64;
    constexpr int XP = CT * PL / 2 / 256;                 // 16-byte pieces of x per thread and block (PL >= 8)
    constexpr int CP = (CT / 2) * NW / 256 > 0 ? (CT / 2) * NW / 256 : 1;   // ... of (c,w)
    static_assert(PLOG >= 3 && PLOG <= 6, "8..64 dim pairs per node row");
    __shared__ __attribute__((aligned(16))) float xs[3][CT][2 * PL];
    __shared__ __attribute__((aligned(16))) float4 cs[3][CT / 2][NW];
    if (MEDIAN)
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 8, 1), 0");   // DX10_CLAMP off: clamp(NaN) = NaN
    const int tid = threadIdx.x;
    const int lnode = tid >> PLOG, lp = tid & (PL - 1);
    const int nl = blockIdx.x * NW + lnode;
    const int d = 2 * (blockIdx.y * PL + lp);
    const bool valid = nl < nloc && d < D;
    const int nbase = blockIdx.x * NW;                    // first node of the workgroup (local index)
    const float *xbase = Xs + 2 * blockIdx.y * PL;        // first dim of the workgroup
    const int nblocks = (B + CT - 1) / CT;
    const int lastrow = B > 0 ? B - 1 : 0, lastpair = B > 0 ? (B - 1) >> 1 : 0;

    // one thread's share of a block in NAMED registers (arrays behind the lambdas ended up in scratch)
    auto load_x = [&](int blk, int i) {
        const int piece = tid + 256 * i;                  // row-major over [CT][PL/2] 16-byte pieces
        int row = blk * CT + piece / (PL / 2);
        row = row < lastrow ? row : lastrow;              // clamped: re-reads the last row, never past the chunk
        return *reinterpret_cast<const float4 *>(xbase + (size_t)row * ldx + 4 * (piece % (PL / 2)));
    };
    auto load_c = [&](int blk, int i) {
        int piece = tid + 256 * i;                        // row-major over [CT/2][NW]
        piece = piece < (CT / 2) * NW ? piece : (CT / 2) * NW - 1;   // (fewer pieces than threads: PL = 64)
        int pr = blk * (CT / 2) + piece / NW;
        pr = pr < lastpair ? pr : lastpair;
        int node = nbase + piece % NW;
        node = node < nloc ? node : nloc - 1;
        return reinterpret_cast<const float4 *>(cw)[(size_t)pr * ldn + node];
    };
    auto store_x = [&](int slot, int i, float4 v) {
        const int piece = tid + 256 * i;
        if (MEDIAN) {   // the Median chains consume x * 2^24 (vsom_median_steps8): scaled once here, exactly
            v.x = v.x * 0x1.0p24f;
            v.y = v.y * 0x1.0p24f;
            v.z = v.z * 0x1.0p24f;
            v.w = v.w * 0x1.0p24f;
        }
        *reinterpret_cast<float4 *>(&xs[slot][piece / (PL / 2)][4 * (piece % (PL / 2))]) = v;
    };
    auto store_c = [&](int slot, int i, float4 v) {
        const int piece = tid + 256 * i;
        if (piece < (CT / 2) * NW)
            cs[slot][piece / NW][piece % NW] = v;
    };
    float4 gx0, gx1, gx2, gx3, gx4, gx5, gx6, gx7, gc0, gc1, gc2, gc3;
#define VSOM_C3_LOAD(blk)                                                                       \
    do {                                                                                        \
        gx0 = load_x(blk, 0);                                                                   \
        if (XP > 1) gx1 = load_x(blk, 1);                                                       \
        if (XP > 2) { gx2 = load_x(blk, 2); gx3 = load_x(blk, 3); }                             \
        if (XP > 4) { gx4 = load_x(blk, 4); gx5 = load_x(blk, 5); gx6 = load_x(blk, 6); gx7 = load_x(blk, 7); } \
        gc0 = load_c(blk, 0);                                                                   \
        if (CP > 1) gc1 = load_c(blk, 1);                                                       \
        if (CP > 2) { gc2 = load_c(blk, 2); gc3 = load_c(blk, 3); }                             \
    } while (0)
#define VSOM_C3_STORE(slot)                                                                     \
    do {                                                                                        \
        store_x(slot, 0, gx0);                                                                  \
        if (XP > 1) store_x(slot, 1, gx1);                                                      \
        if (XP > 2) { store_x(slot, 2, gx2); store_x(slot, 3, gx3); }                           \
        if (XP > 4) { store_x(slot, 4, gx4); store_x(slot, 5, gx5); store_x(slot, 6, gx6); store_x(slot, 7, gx7); } \
        store_c(slot, 0, gc0);                                                                  \
        if (CP > 1) store_c(slot, 1, gc1);                                                      \
        if (CP > 2) { store_c(slot, 2, gc2); store_c(slot, 3, gc3); }                           \
    } while (0)

    vsom_f2 M = {0.f, 0.f}, S = {0.f, 0.f};               // :843-844
    auto one = [&](vsom_f2 xv, float c, float w) {        // Standard; the same operations as update_chain2_kernel's
        const vsom_f2 cc = {c, c}, ww = {w, w};           // (Median: vsom_median_steps8 / vsom_median_step1 on x * 2^24)
        vsom_f2 dl = xv - M;
        if (MEDIAN) {
            __builtin_trap();                             // never called: xs holds scaled values
        } else if (FMA == 1) {
            M = __builtin_elementwise_fma(cc, dl, M);
            S = __builtin_elementwise_fma(ww * dl, dl, S);
        } else if (FMA == 2) {                         // VSOM_UPDATE_FMA_SIGMA: mean chain strict
            const vsom_f2 t = cc * dl;
            M = M + t;
            S = __builtin_elementwise_fma(ww * dl, dl, S);
        } else {
            const vsom_f2 t = cc * dl;
            M = M + t;
            vsom_f2 q = ww * dl;
            q = q * dl;
            S = S + q;
        }
    };

    // blocks past the end are clamped re-reads of the last rows (loaded and stored, never consumed): the
    // pipeline has no conditional loads
    VSOM_C3_LOAD(0);
    VSOM_C3_STORE(0);
    VSOM_C3_LOAD(1);
    VSOM_C3_STORE(1);
    __syncthreads();
    // The chains run at ONE wavefront per SIMD (C4: 1024 wavefronts), so nothing hides the LDS round trip of
    // the operand reads but the wavefront itself: the operands of the NEXT group of U samples are read from LDS
    // (two register sets used alternately) while the current group's dependent chain executes -- also across
    // the block boundary: slot (blk+1)%3 was completed and made visible by the barrier that ended the previous
    // iteration.  r2 read a group, waited for it, then computed: ~150 exposed cycles per 8 samples of a
    // ~40-cycle-per-sample chain (C4: 0.60 ms).
    constexpr int U = 8, NGRP = CT / U;
    vsom_f2 xa[U], xb[U];
    float4 ca[U / 2], cb[U / 2];
    auto ldsl = [&](vsom_f2 (&xv)[U], float4 (&cv)[U / 2], int slot, int t) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            xv[u] = *reinterpret_cast<const vsom_f2 *>(&xs[slot][t + u][2 * lp]);
#pragma unroll
        for (int u = 0; u < U / 2; ++u)
            cv[u] = cs[slot][(t >> 1) + u][lnode];
    };
    auto steps = [&](const vsom_f2 (&xv)[U], const float4 (&cv)[U / 2]) {
        if (MEDIAN) {
            vsom_median_steps8(M, S, xv, cv);              // hand-scheduled: 4-deep chain, 8 instructions per sample
            return;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            one(xv[u], (u & 1) ? cv[u >> 1].z : cv[u >> 1].x, (u & 1) ? cv[u >> 1].w : cv[u >> 1].y);
    };
    ldsl(xa, ca, 0, 0);
    for (int blk = 0; blk < nblocks; ++blk) {
        const int slot = blk % 3, nslot = (blk + 1) % 3;
        VSOM_C3_LOAD(blk + 2);                            // in flight while this block is consumed
        const int nt = B - blk * CT < CT ? B - blk * CT : CT;
        if (nt == CT) {
#pragma unroll
            for (int g = 0; g < NGRP; g += 2) {
                ldsl(xb, cb, slot, (g + 1) * U);
                steps(xa, ca);
                if (g + 2 < NGRP)
                    ldsl(xa, ca, slot, (g + 2) * U);
                else
                    ldsl(xa, ca, nslot, 0);                // first group of the next block
                steps(xb, cb);
            }
        } else {                                          // the last, partial block
            int t = 0;
            for (; t + U <= nt; t += U) {
                ldsl(xa, ca, slot, t);
                steps(xa, ca);
            }
            for (; t < nt; ++t) {
                const float4 cv = cs[slot][t >> 1][lnode];
                const vsom_f2 xv = *reinterpret_cast<const vsom_f2 *>(&xs[slot][t][2 * lp]);
                if (MEDIAN) {
                    const vsom_f2 cw1 = {(t & 1) ? cv.z : cv.x, (t & 1) ? cv.w : cv.y};
                    vsom_median_step1(M, S, xv, cw1);
                } else {
                    one(xv, (t & 1) ? cv.z : cv.x, (t & 1) ? cv.w : cv.y);
                }
            }
        }
        VSOM_C3_STORE((blk + 2) % 3);                     // slot (blk+2)%3 was last read in iteration blk-1
        __syncthreads();
    }
    if (valid) {
        const size_t node = (size_t)(n0 + nl);
        const float Wf = weight[node];
        map[node * pitch + d] = M.x;                      // :870
        sigma[node * pitch + d] = sqrtf(S.x / Wf);        // :873
        if (d + 1 < D) {
            map[node * pitch + d + 1] = M.y;
            sigma[node * pitch + d + 1] = sqrtf(S.y / Wf);
        }
    }
}

#undef VSOM_C3_LOAD
#undef VSOM_C3_STORE

// CLR: lane = node, RP pairs per lane; model = [A | B] (Transformation.cpp:107-142)
template <int RP>
__global__ __launch_bounds__(256) void update_clr_kernel(const float *__restrict__ XP,
                                                         const float *__restrict__ YP, int ldx,
                                                         const float2 *__restrict__ cw, int ldn,
                                                         int B, int n0, int nloc, int P, int ppitch,
                                                         int nslices, int pbase, float *__restrict__ map,
                                                         float *__restrict__ sigma, int pitch,
                                                         const float *__restrict__ weight)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int slice = blockIdx.y * 4 + wave;
    if (slice >= nslices)
        return;
    const int p0 = pbase + slice * RP;   // pbase: first pair not covered by the assembly kernel
    const int nl = blockIdx.x * 64 + lane;
    const bool valid = nl < nloc;
    const int nlc = valid ? nl : nloc - 1;

    float A[RP], Bv[RP], SA[RP], SB[RP];
#pragma unroll
    for (int k = 0; k < RP; ++k) {
        A[k] = 0.f;
        Bv[k] = 0.f;
        SA[k] = 0.f;
        SB[k] = 0.f;
    }
    vsom_cfp xr = (vsom_cfp)(XP + p0);
    vsom_cfp yr = (vsom_cfp)(YP + p0);
    const float4 *cp = (const float4 *)cw + nlc;
    // same software pipeline as update_kernel: loads of sample pair r+1 in flight while pair r is used
    auto load = [&](float (&xa)[RP], float (&ya)[RP], float (&xb)[RP], float (&yb)[RP], float4 &cv, int r) {
        vsom_cfp px0 = xr + (size_t)(2 * r) * ldx, py0 = yr + (size_t)(2 * r) * ldx;
        vsom_cfp px1 = px0 + ldx, py1 = py0 + ldx;
#pragma unroll
        for (int k = 0; k < RP; ++k) {
            xa[k] = px0[k];
            ya[k] = py0[k];
            xb[k] = px1[k];
            yb[k] = py1[k];
        }
        cv = cp[(size_t)r * ldn];
    };
    auto step = [&](const float (&x)[RP], const float (&y)[RP], float c, float w) {
#pragma unroll
        for (int k = 0; k < RP; ++k) {
            const float xp = x[k], yp = y[k];
            float inner = A[k] * xp;       // A.*x' + B - y'   (Transformation.cpp:129)
            inner = inner + Bv[k];
            inner = inner - yp;
            float m2 = -2.f * inner;       // -2*inner (exact)  :135-136
            float aD = m2 * xp;            // aDelta            :135
            float tA = c * aD;
            float tB = c * m2;
            float uA = w * aD;
            uA = uA * aD;
            float uB = w * m2;
            uB = uB * m2;
            A[k] = A[k] + tA;              // Som.cpp:864
            Bv[k] = Bv[k] + tB;
            SA[k] = SA[k] + uA;            // Som.cpp:867
            SB[k] = SB[k] + uB;
        }
    };
    const int npair = B >> 1;
    float xa0[RP], ya0[RP], xa1[RP], ya1[RP], xb0[RP], yb0[RP], xb1[RP], yb1[RP];
    float4 ca, cb;
    load(xa0, ya0, xa1, ya1, ca, 0);
    int r = 0;
    for (; r + 2 <= npair; r += 2) {
        load(xb0, yb0, xb1, yb1, cb, r + 1);
        step(xa0, ya0, ca.x, ca.y);
        step(xa1, ya1, ca.z, ca.w);
        load(xa0, ya0, xa1, ya1, ca, r + 2);
        step(xb0, yb0, cb.x, cb.y);
        step(xb1, yb1, cb.z, cb.w);
    }
    if (r < npair) {
        load(xb0, yb0, xb1, yb1, cb, r + 1);
        step(xa0, ya0, ca.x, ca.y);
        step(xa1, ya1, ca.z, ca.w);
        if (B & 1)
            step(xb0, yb0, cb.x, cb.y);
    } else if (B & 1) {
        step(xa0, ya0, ca.x, ca.y);
    }
    if (valid) {
        const size_t node = (size_t)(n0 + nl);
        const float Wf = weight[node];
#pragma unroll
        for (int k = 0; k < RP; ++k) {
            if (p0 + k < P) {
                map[node * pitch + p0 + k] = A[k];
                map[node * pitch + ppitch + p0 + k] = Bv[k];
                sigma[node * pitch + p0 + k] = sqrtf(SA[k] / Wf);
                sigma[node * pitch + ppitch + p0 + k] = sqrtf(SB[k] / Wf);
            }
        }
    }
}

// sigmaMap[i] = sqrt(S / W)  (Som.cpp:873) for the columns the assembly kernels left as raw S.
// One workgroup per node row, 8-byte accesses (ncols is even: 14/16 dims or 8 pairs per slice).
// When the last slice ran over the end of the part (ncols > nvalid: it read the zero padding of the
// sample rows) its padding columns of mean and sigma^2 are put back to zero -- the MFMA search
// reads model rows to the padded length and the padding could hold NaN (0/0 weights).
__global__ __launch_bounds__(256) void sigma_finalize_kernel(float *__restrict__ sigma, float *__restrict__ map, int pitch,
                                                             int col0, int ncols, int nvalid, int n0, int nloc,
                                                             const float *__restrict__ weight)
{
    const int nl = blockIdx.x;
    if (nl >= nloc)
        return;
    const size_t node = (size_t)n0 + nl;
    const float Wf = weight[node];
    float2 *p = reinterpret_cast<float2 *>(sigma + node * pitch + col0);
    float2 *pm = reinterpret_cast<float2 *>(map + node * pitch + col0);
    for (int i = threadIdx.x; i < (ncols >> 1); i += 256) {
        float2 v = p[i];
        v.x = sqrtf(v.x / Wf);
        v.y = sqrtf(v.y / Wf);
        if (2 * i + 1 >= nvalid) {             // padding columns (at most one slice wide)
            float2 m = pm[i];
            if (2 * i >= nvalid) {
                v.x = 0.f;
                m.x = 0.f;
            }
            v.y = 0.f;
            m.y = 0.f;
            pm[i] = m;
        }
        p[i] = v;
    }
}

// hand-scheduled gfx950 code object (gen_update_asm.py -> vsom_update_gfx950.s -> .hsaco),
// embedded at build time
static const unsigned char vsom_update_hsaco[] = {
#include "vsom_update_hsaco.inc"
};

// Column split of the Standard assembly update: n16 slices of 16 dims followed by n14 slices of 14,
// covering D (rounded up to even: one padding column) exactly when 16*n16 + 14*n14 has a solution --
// every even count >= 84 has one -- choosing the fewest idle wavefront slots (workgroups carry 4 slices)
// and then the most 14-wide slices (the faster kernel: 784 = 56 * 14 tiles the chip with no tail).
// Otherwise one kernel whose ragged last slice runs into the padding (limit = usable row pitch).
// VSOM_UPD_SPLIT="n16,n14" overrides (development).
static void vsom_update_split(unsigned D, unsigned limit, unsigned &n16, unsigned &n14)
{
    static int env16 = -2, env14 = -2;
    if (env16 == -2) {
        env16 = env14 = -1;
        if (const char *e = getenv("VSOM_UPD_SPLIT"))
            if (sscanf(e, "%d,%d", &env16, &env14) != 2)
                env16 = env14 = -1;
    }
    if (env16 >= 0 && env14 >= 0 && (unsigned)(16 * env16 + 14 * env14) >= D &&
        (unsigned)(16 * env16 + 14 * env14) <= limit) {
        n16 = (unsigned)env16;
        n14 = (unsigned)env14;
        return;
    }
    const unsigned De = (D + 1) & ~1u;
    int best = -1;
    for (unsigned b = 0; 16 * b <= De; ++b) {
        const unsigned rem = De - 16 * b;
        if (rem % 14)
            continue;
        const unsigned a = rem / 14;
        const int idle = (int)((4 - a % 4) % 4 + (4 - b % 4) % 4);
        const int score = 1000000 - idle * 10000 + (int)a;      // fewest idle slots, then most 14-wide slices
        if (score > best) {
            best = score;
            n16 = b;
            n14 = a;
        }
    }
    if (best >= 0 && De <= limit)
        return;
    const unsigned s14 = (D + 13) / 14, s16 = (D + 15) / 16;
    n16 = n14 = 0;
    if (s14 * 14 <= limit && s14 * 14 <= s16 * 16)
        n14 = s14;
    else if (s16 * 16 <= limit)
        n16 = s16;
}

#define VSOM_UPD_LDS_DEFAULT 1

struct UpdAsmArgs {
    const void *xs;
    const void *cw2;
    void *map;
    void *sbuf;
    unsigned ldx_bytes, ldn_bytes, B, nloc, nslices, pitch_bytes, n0, ppitch_bytes;   // ppitch: CLR only
    const void *yp;                  // CLR: y' rows; Standard / Median: live-slice record of the compaction or null
    const void *zmask;               // Standard family: all-zero (sample, slice) bit mask or null (kernarg 80 B)
};

int vsom_load_asm_module(vsom_ctx *c)
{
    if (c->upd_module)
        return VSOM_OK;
    hipModule_t mod;
    if (const char *alt = std::getenv("VSOM_ASM_HSACO")) {   // development: time a variant code object (tools/exp)
        VSOM_HIP_CHECK(hipModuleLoad(&mod, alt));
    } else
    VSOM_HIP_CHECK(hipModuleLoadData(&mod, vsom_update_hsaco));
    hipFunction_t f16, f14, m16, m14, fclr, d16, d14, sf16, sf14, l14[4], l16[4];
    VSOM_HIP_CHECK(hipModuleGetFunction(&f16, mod, "vsom_update_std_rd16_gfx950"));
    VSOM_HIP_CHECK(hipModuleGetFunction(&f14, mod, "vsom_update_std_rd14_gfx950"));
    VSOM_HIP_CHECK(hipModuleGetFunction(&m16, mod, "vsom_update_fma_rd16_gfx950"));
    VSOM_HIP_CHECK(hipModuleGetFunction(&m14, mod, "vsom_update_fma_rd14_gfx950"));
    VSOM_HIP_CHECK(hipModuleGetFunction(&sf16, mod, "vsom_update_sfma_rd16_gfx950"));
    VSOM_HIP_CHECK(hipModuleGetFunction(&sf14, mod, "vsom_update_sfma_rd14_gfx950"));
    static const char *const lds_names[4] = {"std", "fma", "sfma", "med"};
    for (int i = 0; i < 4; ++i) {
        const std::string n14 = std::string("vsom_update_") + lds_names[i] + "_rd14_lds_gfx950",
                          n16 = std::string("vsom_update_") + lds_names[i] + "_rd16_lds_gfx950";
        VSOM_HIP_CHECK(hipModuleGetFunction(&l14[i], mod, n14.c_str()));
        VSOM_HIP_CHECK(hipModuleGetFunction(&l16[i], mod, n16.c_str()));
    }
    VSOM_HIP_CHECK(hipModuleGetFunction(&fclr, mod, "vsom_update_clr_rp8_gfx950"));
    VSOM_HIP_CHECK(hipModuleGetFunction(&d16, mod, "vsom_update_med_rd16_gfx950"));
    VSOM_HIP_CHECK(hipModuleGetFunction(&d14, mod, "vsom_update_med_rd14_gfx950"));
    c->upd_module = mod;
    c->upd_fn16 = f16;
    c->upd_fn14 = f14;
    c->upd_fma16 = m16;
    c->upd_fma14 = m14;
    c->upd_sfma16 = sf16;
    c->upd_sfma14 = sf14;
    for (int i = 0; i < 4; ++i) {
        c->upd_lds14[i] = l14[i];
        c->upd_lds16[i] = l16[i];
    }
    c->upd_clr8 = fclr;
    c->upd_med16 = d16;
    c->upd_med14 = d14;
    static const char *const nq_names[4] = {"std", "fma", "sfma", "med"};
    for (int i = 0; i < 4; ++i) {
        hipFunction_t f;
        const std::string n = std::string("vsom_update_") + nq_names[i] + "_nq32_gfx950";
        VSOM_HIP_CHECK(hipModuleGetFunction(&f, mod, n.c_str()));
        c->upd_nq[i] = f;
        const std::string n2 = std::string("vsom_update_") + nq_names[i] + "_nt4_gfx950";
        VSOM_HIP_CHECK(hipModuleGetFunction(&f, mod, n2.c_str()));
        c->upd_nt[i] = f;
    }
    return VSOM_OK;
}

// Host: tabulate (float)calculateNeighbourhoodWeight over (|dx|,|dy|) (Som.cpp:949-975).
// The argument of exp depends on (cx-bx)^2 and (cy-by)^2 only, so the table is bit-identical
// to per-pair evaluation with the same libm.
double vsom_neighbourhood_weight(size_t cx, size_t cy, size_t bx, size_t by, double sigma)
{
    if (sigma > 1.0) {
        double cxd = (double)cx, cyd = (double)cy, bxd = (double)bx, byd = (double)by;
        return std::exp(-((cxd - bxd) * (cxd - bxd) / 2.0 / sigma / sigma +
                          (cyd - byd) * (cyd - byd) / 2.0 / sigma / sigma));
    } else if (cx == bx && cy == by) {
        return 1.0;
    }
    return 0.0;
}

int ensure_lut(vsom_ctx *c, double sigma)
{
    if (c->lut && c->lut_sigma == sigma)
        return VSOM_OK;
    // largest y that SomIndex(som, idx) can produce is (N-W)/H (Q10)
    uint32_t ymax = c->N ? (c->N - c->W) / c->H : 0;
    uint32_t lh = ymax + 1, lw = c->W;
    size_t need = (size_t)lh * lw;
    if (need > c->lut_cap) {
        VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->lut)
            VSOM_HIP_CHECK(hipFree(c->lut));
        if (c->lut_host)
            VSOM_HIP_CHECK(hipHostFree(c->lut_host));
        c->lut = nullptr;
        c->lut_host = nullptr;
        VSOM_HIP_CHECK(hipMalloc(&c->lut, need * sizeof(float)));
        VSOM_HIP_CHECK(hipHostMalloc(&c->lut_host, 2 * need * sizeof(float)));   // two halves used alternately
        c->lut_cap = need;
        c->lut_ev_valid[0] = c->lut_ev_valid[1] = false;
    }
    // the table changes with every epoch's sigma: stage it through the half of the pinned buffer whose
    // previous copy (two tables ago) has certainly left it -- no stream synchronisation on this path
    const int k = c->lut_slot;
    c->lut_slot ^= 1;
    if (!c->lut_ev[k])
        VSOM_HIP_CHECK(hipEventCreateWithFlags(&c->lut_ev[k], hipEventDisableTiming));
    if (c->lut_ev_valid[k])
        VSOM_HIP_CHECK(hipEventSynchronize(c->lut_ev[k]));
    float *host = c->lut_host + (size_t)k * c->lut_cap;
    for (uint32_t dy = 0; dy < lh; ++dy)
        for (uint32_t dx = 0; dx < lw; ++dx)
            host[(size_t)dy * lw + dx] = (float)vsom_neighbourhood_weight(dx, dy, 0, 0, sigma);
    VSOM_HIP_CHECK(hipMemcpyAsync(c->lut, host, need * sizeof(float), hipMemcpyHostToDevice, c->stream));
    VSOM_HIP_CHECK(hipEventRecord(c->lut_ev[k], c->stream));
    c->lut_ev_valid[k] = true;
    c->lut_sigma = sigma;
    c->lut_w = lw;
    c->lut_h = lh;
    return VSOM_OK;
}

// trainBatchSomEpoch on an EMPTY chunk (the trailing zero-row load of a chunked MnistDataLoader pass,
// MnistDataLoader.cpp:49-55): phase 2 still runs over every node with no samples -- the model vector
// becomes the zero it started from (:843,870), sigmaMap = sqrt(0/0) = NaN (:873), weightMap = 0 (:875)
__global__ void empty_epoch_kernel(float *__restrict__ map, float *__restrict__ sigma, float *__restrict__ weight,
                                   int pitch, int part_pitch, int part_len, int n0, int nloc)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nloc * pitch)
        return;
    const int nl = (int)(i / pitch), col = (int)(i % pitch);
    const size_t at = (size_t)(n0 + nl) * pitch + col;
    const float zero = weight[n0 + nl] * 0.f;          // keeps the division below out of the constant folder
    map[at] = 0.f;
    sigma[at] = (col % part_pitch) < part_len ? sqrtf(0.f / (zero * 0.f + 0.f)) : 0.f;   // pad columns stay zero
}

__global__ void zero_weight_kernel(float *__restrict__ weight, int n0, int nloc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nloc)
        weight[n0 + i] = 0.f;
}

// lane = node assembly kernels need wavefronts: ceil(nodes/64) * ceil(D/14) of them for 1024 SIMDs.  Below
// this many the chain kernels (lane = (node, dim pair), operands through LDS) are faster although they pay
// an LDS read per operand where the assembly kernels take x from SGPRs (measured crossover, DESIGN.md
// section 4; VSOM_CHAIN_MAX_WAVES overrides, development)
static size_t vsom_chain_max_waves()
{
    static long v = -1;
    if (v < 0) {
        const char *e = std::getenv("VSOM_CHAIN_MAX_WAVES");
        v = e ? std::atol(e) : VSOM_CHAIN_MAX_WAVES;
    }
    return (size_t)v;
}

int launch_phase2(vsom_ctx *c, double sigma, size_t n0, size_t n1)
{
    if (n1 <= n0)
        return vsom_join_aux(c);
    if (c->B == 0) {
        TimerScope ts(c, VSOM_T_UPDATE);
        const size_t nloc = n1 - n0, tot = nloc * c->pitch;
        hipLaunchKernelGGL(empty_epoch_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, c->map,
                           c->sigma, c->weight, (int)c->pitch, (int)c->part_pitch, (int)c->part_len, (int)n0, (int)nloc);
        hipLaunchKernelGGL(zero_weight_kernel, dim3((unsigned)((nloc + 255) / 256)), dim3(256), 0, c->stream, c->weight,
                           (int)n0, (int)nloc);
        VSOM_HIP_CHECK(hipGetLastError());
        return vsom_join_aux(c);
    }
    int rc = ensure_lut(c, sigma);
    if (rc)
        return rc;
    const size_t nloc = n1 - n0;
    const size_t ldn = (nloc + 63) / 64 * 64;
    // pair rows: ceil(B/2) + the prefetch ring of the assembly kernel (4) + slack
    const size_t prow = (c->B + 1) / 2 + 24;   // (the LDS-sharing kernels fetch three groups of four pair-rows ahead)
    const size_t need = prow * ldn * 2;   // float2 elements
    if (need > c->cw_cap) {
        if (c->cw)
            VSOM_HIP_CHECK(hipFree(c->cw));
        c->cw = nullptr;
        VSOM_HIP_CHECK(hipMalloc(&c->cw, need * sizeof(float2)));
        // rows beyond B are prefetched (never used): keep them initialised
        VSOM_HIP_CHECK(hipMemsetAsync(c->cw, 0, need * sizeof(float2), c->stream));
        c->cw_cap = need;
    }
    {
        TimerScope ts(c, VSOM_T_CW);
        hipLaunchKernelGGL(bxy_kernel, dim3((unsigned)((c->B + 255) / 256)), dim3(256), 0, c->stream,
                           c->lastbmu, (int)c->B, (int)c->W, (int)c->H, c->bxy);
        const size_t lut_bytes = (size_t)c->lut_w * c->lut_h * sizeof(float);
        bool piped = false;
        if (c->cw_mode == 0) {
            // role-split kernel: NW nodes per workgroup so that the grid still covers the CUs
            // 16 nodes per workgroup and the table in LDS only while it is small: what counts is how
            // many workgroups (worker wavefronts) a CU holds -- 40 KB of tiles each; a 64-KB table
            // (128x128 map) would leave one per CU.  Measured at C3: 64 nodes + LDS table 0.153 ms,
            // 32 + global 0.142, 16 + global 0.121, 16 + LDS 0.20.
            bool lds = lut_bytes <= 24 * 1024;
            int nw = 16;
            if (const char *e = std::getenv("VSOM_CW_NW"))      // development knobs
                nw = std::atoi(e) == 64 ? 64 : (std::atoi(e) == 32 ? 32 : 16);
            if (const char *e = std::getenv("VSOM_CW_LUT_GLOBAL"))
                lds = e[0] == '1' ? false : (e[0] == '0' ? lut_bytes <= 96 * 1024 : lds);
            const int T = 2048 / nw;
            const size_t smem = (size_t)5 * 2048 * sizeof(float) + (size_t)3 * T * sizeof(int2) + (lds ? lut_bytes : 0);
            const void *fn = nw == 64 ? (lds ? (const void *)cwp_kernel<64, 32, true> : (const void *)cwp_kernel<64, 32, false>)
                           : nw == 32 ? (lds ? (const void *)cwp_kernel<32, 64, true> : (const void *)cwp_kernel<32, 64, false>)
                                      : (lds ? (const void *)cwp_kernel<16, 128, true> : (const void *)cwp_kernel<16, 128, false>);
            if (smem <= 64 * 1024 ||
                hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess) {
                const int2 *bxy = c->bxy;
                int B = (int)c->B, in0 = (int)n0, in1 = (int)n1, iW = (int)c->W, iH = (int)c->H, lw = (int)c->lut_w,
                    lh = (int)c->lut_h, ildn = (int)ldn;
                const float *lut = c->lut;
                float2 *cwp = c->cw;
                float *wgt = c->weight;
                void *args[] = {&bxy, &B, &in0, &in1, &iW, &iH, &lut, &lw, &lh, &cwp, &ildn, &wgt};
                VSOM_HIP_CHECK(hipLaunchKernel(fn, dim3((unsigned)((nloc + nw - 1) / nw)), dim3(CWP_THREADS), args, smem,
                                               c->stream));
                piped = true;
            } else {
                (void)hipGetLastError();
            }
        }
        if (!piped) {
            const bool lds = lut_bytes <= 64 * 1024;
            // quads give nloc/16 wavefronts; below one per SIMD use 16 lanes per node (nloc/4 wavefronts)
            const bool wide = c->cw_mode == 2 || (c->cw_mode != 1 && nloc / 16 < 1024 && c->B >= 64);
            auto kern = wide ? (lds ? cw16_kernel<true> : cw16_kernel<false>) : (lds ? cw_kernel<true> : cw_kernel<false>);
            const size_t lanes = nloc * (wide ? 16 : 4);
            hipLaunchKernelGGL(kern, dim3((unsigned)((lanes + 255) / 256)), dim3(256), lds ? lut_bytes : 0, c->stream,
                               c->bxy, (int)c->B, (int)n0, (int)n1, (int)c->W, (int)c->H, c->lut, (int)c->lut_w,
                               (int)c->lut_h, c->cw, (int)ldn, c->weight);
        }
        VSOM_HIP_CHECK(hipGetLastError());
    }
    int sig_cols = 0;   // columns left as raw S by the assembly kernel
    {
        TimerScope ts(c, VSOM_T_UPDATE);
        const unsigned gx = (unsigned)((nloc + 63) / 64);
        if (c->transform == VSOM_CLR) {
            constexpr int RP = 8;
            int pbase = 0;
            if (c->use_asm) {
                // hand-scheduled kernel for the full 8-pair slices (gen_update_asm.py, KC)
                if ((rc = vsom_load_asm_module(c)))
                    return rc;
                const unsigned nfull = (c->part_len + RP - 1) / RP;   // a ragged last slice runs over the padding
                if (nfull * RP <= c->part_pitch) {
                    UpdAsmArgs a;
                    a.xs = c->XP;
                    a.cw2 = c->cw;
                    a.map = c->map;
                    a.sbuf = c->sigma;
                    a.ldx_bytes = c->part_pitch * 4u;
                    a.ldn_bytes = (unsigned)(ldn * 16u);
                    a.B = (unsigned)c->B;
                    a.nloc = (unsigned)nloc;
                    a.nslices = nfull;
                    a.pitch_bytes = c->pitch * 4u;
                    a.n0 = (unsigned)n0;
                    a.ppitch_bytes = c->part_pitch * 4u;
                    a.yp = c->YP;
                    a.zmask = nullptr;
                    size_t sz = 72;          // kernarg segment of the CLR kernel
                    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz,
                                     HIP_LAUNCH_PARAM_END};
                    VSOM_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)c->upd_clr8, 8 * ((nfull + 3) / 4), (gx + 7) / 8, 1,
                                                         256, 1, 1, 0, c->stream, nullptr, extra));
                    pbase = (int)c->part_len;
                    sig_cols = (int)(nfull * RP);
                }
            }
            const int rest = (int)c->part_len - pbase;
            if (rest > 0) {
                const int nsl = (rest + RP - 1) / RP;
                dim3 grid(gx, (unsigned)((nsl + 3) / 4));
                hipLaunchKernelGGL(update_clr_kernel<RP>, grid, dim3(256), 0, c->stream, c->XP, c->YP,
                                   (int)c->part_pitch, c->cw, (int)ldn, (int)c->B, (int)n0, (int)nloc,
                                   (int)c->part_len, (int)c->part_pitch, nsl, pbase, c->map, c->sigma,
                                   (int)c->pitch, c->weight);
            }
        } else if ((size_t)gx * ((c->D + 13) / 14) <= vsom_chain_max_waves() && c->use_chain) {
            // lane = node would leave most SIMDs idle: one lane per (node, dim pair) chain instead
            int dl_log2 = 0;
            while ((1u << dl_log2) < c->D && dl_log2 < 6)
                ++dl_log2;
            const unsigned DL = 1u << dl_log2;
            constexpr int U = 8;
            // lanes = nloc * D chains; when that is no more than two wavefronts per SIMD, two dims per lane
            // with a deep load ring (update_chain2_kernel) beat one dim per lane (VSOM_CHAIN2=0/1 overrides)
            static int chain2_env = -1;
            if (chain2_env < 0) {
                const char *e = std::getenv("VSOM_CHAIN2");
                chain2_env = e ? (e[0] == '1' ? 1 : 0) : 2;
            }
            const bool chain2 = chain2_env != 0;   // VSOM_CHAIN2=0: r1's one-dim-per-lane kernel (development)
            if (chain2) {
                int pl_log2 = 0;                               // dim pairs per node row of the workgroup
                while ((2u << pl_log2) < c->D && pl_log2 < 6)
                    ++pl_log2;
                const unsigned PL = 1u << pl_log2, npairs = (c->D + 1) / 2;
                dim3 grid2((unsigned)((nloc + (256 / PL) - 1) / (256 / PL)), (npairs + PL - 1) / PL);
                constexpr int NG = 4;
                const bool med = c->transform == VSOM_MEDIAN, fma = c->update_mode == VSOM_UPDATE_FMA,
                           sfma = c->update_mode == VSOM_UPDATE_FMA_SIGMA;
                static int chain3_env = -1;                    // VSOM_CHAIN3=0: the register-ring kernel (development)
                if (chain3_env < 0) {
                    const char *e = std::getenv("VSOM_CHAIN3");
                    chain3_env = e && e[0] == '0' ? 0 : 1;
                }
                const void *k3 = nullptr;                      // operands staged through LDS: 8..64 dim pairs per row
#define VSOM_K3(P) (med ? (const void *)update_chain3_kernel<true, 0, P> \
                        : (fma ? (const void *)update_chain3_kernel<false, 1, P>                    \
                               : (sfma ? (const void *)update_chain3_kernel<false, 2, P> : (const void *)update_chain3_kernel<false, 0, P>)))
                if (chain3_env && pl_log2 >= 3 && pl_log2 <= 6)
                    k3 = pl_log2 == 3 ? VSOM_K3(3) : pl_log2 == 4 ? VSOM_K3(4) : pl_log2 == 5 ? VSOM_K3(5) : VSOM_K3(6);
#undef VSOM_K3
                if (k3) {
                    const float *xs_ = c->Xs;
                    const float2 *cw_ = c->cw;
                    int ildx = (int)c->xpitch, ildn = (int)ldn, iB = (int)c->B, in0 = (int)n0, inl = (int)nloc, iD = (int)c->D,
                        ipitch = (int)c->pitch;
                    float *map_ = c->map, *sg_ = c->sigma;
                    const float *wt_ = c->weight;
                    void *args[] = {&xs_, &ildx, &cw_, &ildn, &iB, &in0, &inl, &iD, &map_, &sg_, &ipitch, &wt_};
                    VSOM_HIP_CHECK(hipLaunchKernel(k3, grid2, dim3(256), args, 0, c->stream));
                } else {
                auto kern2 = med ? update_chain2_kernel<true, U, NG, 0>
                                 : (fma ? update_chain2_kernel<false, U, NG, 1>
                                        : (sfma ? update_chain2_kernel<false, U, NG, 2> : update_chain2_kernel<false, U, NG, 0>));
                hipLaunchKernelGGL(kern2, grid2, dim3(256), 0, c->stream, c->Xs, (int)c->xpitch, c->cw, (int)ldn,
                                   (int)c->B, (int)n0, (int)nloc, (int)c->D, pl_log2, c->map, c->sigma,
                                   (int)c->pitch, c->weight);
                }
            } else {
            dim3 grid((unsigned)((nloc + (256 / DL) - 1) / (256 / DL)), (c->D + DL - 1) / DL);
            auto kern = c->transform == VSOM_MEDIAN ? update_chain_kernel<true, U, 0>
                        : (c->update_mode == VSOM_UPDATE_FMA ? update_chain_kernel<false, U, 1>
                           : (c->update_mode == VSOM_UPDATE_FMA_SIGMA ? update_chain_kernel<false, U, 2>
                                                                      : update_chain_kernel<false, U, 0>));
            hipLaunchKernelGGL(kern, grid, dim3(256), 0, c->stream, c->Xs, (int)c->xpitch, c->cw, (int)ldn,
                               (int)c->B, (int)n0, (int)nloc, (int)c->D, dl_log2, c->map, c->sigma,
                               (int)c->pitch, c->weight);
            }
        } else {
            constexpr int RD = 16;
            int dbase = 0;
            if ((c->transform == VSOM_STANDARD || c->transform == VSOM_MEDIAN) && c->use_asm) {
                // hand-scheduled kernels (Standard strict / contracted, Median), 16 or 14 dims per wavefront: the first n16 slices by the
                // 16-wide kernel, the following n14 by the 14-wide one on the side stream (the two run
                // concurrently; vsom_update_split picks the pair).  Columns past D are the zero padding
                // of the rows; sigma_finalize_kernel re-zeroes them afterwards.
                if ((rc = vsom_load_asm_module(c)))
                    return rc;
                unsigned n16 = 0, n14 = 0;
                vsom_update_split(c->D, c->pitch < c->xpitch ? c->pitch : c->xpitch, n16, n14);
                const bool fma = c->update_mode == VSOM_UPDATE_FMA, sfma = c->update_mode == VSOM_UPDATE_FMA_SIGMA;
                const bool med = c->transform == VSOM_MEDIAN;     // its FMAs are exact: one kernel for all modes
                void *fn16 = med ? c->upd_med16 : (fma ? c->upd_fma16 : (sfma ? c->upd_sfma16 : c->upd_fn16));
                void *fn14 = med ? c->upd_med14 : (fma ? c->upd_fma14 : (sfma ? c->upd_sfma14 : c->upd_fn14));
                // 14-dim Standard kernels whose workgroup shares ONE (c,w) stream through LDS (gen_update_asm.py,
                // "lds"): a quarter of the L2 requests (VSOM_UPD_LDS=0/1 overrides the choice, development)
                static int lds_env = -1;
                if (lds_env < 0) {
                    const char *e = std::getenv("VSOM_UPD_LDS");
                    lds_env = e ? (e[0] == '1' ? 1 : 0) : 2;
                }
                // Measured (strict, 784 dims, B = 4096): 16384 nodes (two rounds of 7 wavefronts per SIMD) update 4.56 ->
                // 4.46 ms, sigma-contracted 4.16 -> 3.91, contracted 3.50 -> 3.30, Median 6.37 -> 5.96; 8192 nodes (one
                // round) 2.36 vs 2.37; 4096 nodes 1.50 -> 1.53; 2048 nodes 0.81 -> 0.88 -- the barriers cost more than the
                // L2 requests once a SIMD holds few wavefronts, so the shared stream is used from one full round up.
                const bool use_lds = lds_env == 1 || (lds_env == 2 && VSOM_UPD_LDS_DEFAULT &&
                                                      (size_t)gx * ((c->D + 13) / 14) > 7168);
                if (use_lds) {
                    const int v = med ? 3 : (fma ? 1 : (sfma ? 2 : 0));
                    fn14 = c->upd_lds14[v];
                    fn16 = c->upd_lds16[v];
                }
                // lane = (node, four dims) kernels (gen_nq_asm.py) where lane = node has too few wavefronts to balance
                // (VSOM_UPD_NQ=0/1 overrides the choice: tests/test_gpu_nq_kernels.py runs the suite's shapes through them)
                static int nq_env = -1;
                if (nq_env < 0) {
                    const char *e = std::getenv("VSOM_UPD_NQ");
                    nq_env = e ? (e[0] == '1' ? 1 : 0) : 2;
                }
                // lane = node, four dims per wavefront, x from scalar loads of the transposed chunk (gen_nt_asm.py)
                static int nt_env = -1;
                if (nt_env < 0) {
                    const char *e = std::getenv("VSOM_UPD_NT");
                    nt_env = e ? (e[0] == '1' ? 1 : 0) : 2;
                }
                const bool use_nt = nt_env != 0;
                const bool use_nq = !use_nt && (nq_env == 1 || (nq_env == 2 && (size_t)gx * ((c->D + 13) / 14) <= VSOM_NQ_MAX_WAVES));
                // column compaction (vsom_compact.hip): the chains of the columns that are zero in every row of
                // the chunk are retired -- the 14-wide kernel runs on the gathered live columns (device-side
                // slice count) into dense scratch rows and cc_expand_kernel writes map / sigmaMap back
                const bool compact = c->cc_valid;
                if (compact && (rc = vsom_cc_ensure_update_scratch(c)))
                    return rc;
                // Standard chains: (sample, slice) blocks that are all zero take the form without the subtraction
                const bool zpath = compact && c->transform == VSOM_STANDARD && !use_nq && !use_nt;
                if (zpath && (rc = vsom_cc_ensure_zmask(c)))
                    return rc;
                auto launch = [&](void *fn, unsigned nsl, unsigned col0, hipStream_t st) -> int {
                    UpdAsmArgs a;
                    a.xs = compact ? c->Xc : c->Xs + col0;
                    a.cw2 = c->cw;
                    a.map = compact ? c->Uc_map : c->map + col0;
                    a.sbuf = compact ? c->Uc_S : c->sigma + col0;
                    a.ldx_bytes = (compact ? c->cpitch : c->xpitch) * 4u;
                    a.ldn_bytes = (unsigned)(ldn * 16u);
                    a.B = (unsigned)c->B;
                    a.nloc = (unsigned)nloc;
                    a.nslices = nsl;
                    a.pitch_bytes = (compact ? c->cpitch : c->pitch) * 4u;
                    a.n0 = (unsigned)n0;
                    a.ppitch_bytes = 0;
                    a.yp = compact ? (const void *)c->cc_meta : nullptr;
                    a.zmask = zpath && c->cc_zmask_valid ? (const void *)c->cc_zmask : nullptr;
                    size_t sz = med ? 72 : 80;   // kernarg segments of the Median / Standard-family kernels
                    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz,
                                     HIP_LAUNCH_PARAM_END};
                    // XCD-aware grid: x = 8 * slice quads, y = node groups / 8 (see gen_update_asm.py)
                    // development: VSOM_UPD_WG_CAP=k reserves (unused) LDS so that a CU holds at most k workgroups
                    static int wg_cap = -1;
                    if (wg_cap < 0) {
                        const char *e = std::getenv("VSOM_UPD_WG_CAP");
                        wg_cap = e ? std::atoi(e) : 0;
                    }
                    const unsigned lds_reserve = wg_cap > 0 ? (unsigned)((160 * 1024 / wg_cap) & ~1023) : 0u;
                    VSOM_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)fn, 8 * ((nsl + 3) / 4), (gx + 7) / 8, 1, 256, 1,
                                                         1, lds_reserve, st, nullptr, extra));
                    return VSOM_OK;
                };
                if (compact) {
                    n16 = 0;
                    n14 = (c->D + 13) / 14;      // upper bound; the kernel reads the live count from cc_meta
                }
                if (use_nt) {
                    // workgroup = 64 nodes x 8 column quads (one per wavefront); grid.x = 8 * column blocks (XCD-aware)
                    if ((rc = vsom_xq_ensure(c)))
                        return rc;
                    const unsigned cols = compact ? c->cpitch : c->pitch;
                    const unsigned quads = compact ? c->cpitch / 4 : (c->D + 3) / 4;
                    UpdAsmArgs a;
                    a.xs = c->Xq;
                    a.cw2 = c->cw;
                    a.map = compact ? c->Uc_map : c->map;
                    a.sbuf = compact ? c->Uc_S : c->sigma;
                    a.ldx_bytes = c->xq_bpad * 16u;
                    a.ldn_bytes = (unsigned)(ldn * 16u);
                    a.B = (unsigned)c->B;
                    a.nloc = (unsigned)nloc;
                    a.nslices = quads;
                    a.pitch_bytes = cols * 4u;
                    a.n0 = (unsigned)n0;
                    a.ppitch_bytes = 0;
                    a.yp = compact ? (const void *)c->cc_meta : nullptr;
                    a.zmask = c->zq;
                    size_t sz = 80;
                    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz,
                                     HIP_LAUNCH_PARAM_END};
                    void *fn = c->upd_nt[med ? 3 : (fma ? 1 : (sfma ? 2 : 0))];
                    VSOM_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)fn, 8 * ((quads + 7) / 8), (gx + 7) / 8, 1, 512, 1, 1, 0,
                                                         c->stream, nullptr, extra));
                    dbase = (int)c->D;
                    sig_cols = compact ? -1 : (int)(quads * 4);
                } else
                if (use_nq) {
                    // workgroup = 32 nodes x 32 columns; grid.x = 8 * column blocks (XCD-aware, gen_nq_asm.py); the
                    // last block of a ragged depth runs into the rows' zero padding (pitches are multiples of 32)
                    const unsigned cols = compact ? c->cpitch : c->pitch;
                    const unsigned nb = compact ? c->cpitch / 32 : (c->D + 31) / 32;
                    const unsigned ng = (unsigned)((nloc + 31) / 32);
                    UpdAsmArgs a;
                    a.xs = compact ? c->Xc : c->Xs;
                    a.cw2 = c->cw;
                    a.map = compact ? c->Uc_map : c->map;
                    a.sbuf = compact ? c->Uc_S : c->sigma;
                    a.ldx_bytes = (compact ? c->cpitch : c->xpitch) * 4u;
                    a.ldn_bytes = (unsigned)(ldn * 16u);
                    a.B = (unsigned)c->B;
                    a.nloc = (unsigned)nloc;
                    a.nslices = nb;
                    a.pitch_bytes = cols * 4u;
                    a.n0 = (unsigned)n0;
                    a.ppitch_bytes = 0;
                    a.yp = compact ? (const void *)c->cc_meta : nullptr;
                    a.zmask = nullptr;
                    size_t sz = 80;
                    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz,
                                     HIP_LAUNCH_PARAM_END};
                    void *fn = c->upd_nq[med ? 3 : (fma ? 1 : (sfma ? 2 : 0))];
                    VSOM_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)fn, 8 * nb, (ng + 7) / 8, 1, 256, 1, 1, 0, c->stream,
                                                         nullptr, extra));
                    dbase = (int)c->D;
                    sig_cols = compact ? -1 : (int)(nb * 32);
                } else
                if (n16 + n14 > 0) {
                    const bool both = n16 > 0 && n14 > 0;
                    if (both) {   // fork: the 14-wide part beside the 16-wide one
                        VSOM_HIP_CHECK(hipEventRecord(c->ev_fork, c->stream));
                        VSOM_HIP_CHECK(hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
                    }
                    if (n16 > 0 && (rc = launch(fn16, n16, 0, c->stream)))
                        return rc;
                    if (n14 > 0 && (rc = launch(fn14, n14, n16 * 16, both ? c->aux_stream : c->stream)))
                        return rc;
                    if (both) {
                        VSOM_HIP_CHECK(hipEventRecord(c->ev_join, c->aux_stream));
                        c->aux_pending = true;
                        if ((rc = vsom_join_aux(c)))
                            return rc;
                    }
                    dbase = (int)c->D;
                    sig_cols = compact ? -1 : (int)(n16 * 16 + n14 * 14);
                }
            }
            const int rest = (int)c->D - dbase;
            if (rest > 0) {
                const int nsl = (rest + RD - 1) / RD;
                dim3 grid(gx, (unsigned)((nsl + 3) / 4));
                if (c->transform == VSOM_MEDIAN)
                    hipLaunchKernelGGL((update_kernel<RD, true>), grid, dim3(256), 0, c->stream, c->Xs,
                                       (int)c->xpitch, c->cw, (int)ldn, (int)c->B, (int)n0, (int)nloc,
                                       (int)c->D, nsl, dbase, c->map, c->sigma, (int)c->pitch, c->weight);
                else
                    hipLaunchKernelGGL((update_kernel<RD, false>), grid, dim3(256), 0, c->stream, c->Xs,
                                       (int)c->xpitch, c->cw, (int)ldn, (int)c->B, (int)n0, (int)nloc,
                                       (int)c->D, nsl, dbase, c->map, c->sigma, (int)c->pitch, c->weight);
            }
        }
        VSOM_HIP_CHECK(hipGetLastError());
    }
    if (sig_cols < 0) {
        TimerScope ts(c, VSOM_T_SIGMA);
        if ((rc = vsom_cc_expand(c, n0, nloc)))
            return rc;
    }
    if (sig_cols > 0) {
        TimerScope ts(c, VSOM_T_SIGMA);
        // the assembly kernels left raw S in those columns (CLR: in the A part and in the B part)
        for (uint32_t part = 0; part < c->nparts; ++part)
            hipLaunchKernelGGL(sigma_finalize_kernel, dim3((unsigned)nloc), dim3(256), 0, c->stream,
                               c->sigma, c->map, (int)c->pitch, (int)(part * c->part_pitch), sig_cols,
                               (int)c->part_len, (int)n0, (int)nloc, c->weight);
        VSOM_HIP_CHECK(hipGetLastError());
    }
    return vsom_join_aux(c);   // the MSE sum forked by launch_finish ran beside the kernels above
}
