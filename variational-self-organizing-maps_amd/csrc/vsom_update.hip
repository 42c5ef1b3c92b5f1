// vsom_update.hip -- phase 2 of Som::trainBatchSomEpoch (Som.cpp:809-876) on gfx950.
//
// For every node i (independent) and the samples j of the chunk in load order:
//     w = (float)calculateNeighbourhoodWeight(node, bmu_j, sigma)       Som.cpp:851, 949-975
//     W += w ; c = w / W                                                  :857, :864
//     delta = Stepper(x_j, M) ; M = M + c*delta ; S = S + (w*delta)*delta  :861-867
//   map[i] = M ; sigmaMap[i] = sqrt(S / W) ; weightMap[i] = W             :870-875
//
// Split into
//   cwp_kernel   : SomIndex(*this, lastBMU[j]) (:847-849) and the neighbourhood chain per node: LUT lookup of w (the double exp is tabulated on
//                  the host over (|dx|,|dy|), bit-identical), the serial fp32 prefix sum W and
//                  c = w/W -> cw[j][i] = (c, w).  Role-split workgroups: one wavefront only adds,
//                  eight look up / divide / store.
//   the N*D chains, hand-scheduled code object (one rounding per fp32 operation as the reference's SSE2
//   build rounds it, so bit-identical; VSOM_UPDATE_FMA / _FMA_SIGMA opt out for Standard, include/vsom_hip.h):
//                  - vsom_update_{std,sfma,fma,med}_nt4_gfx950 (gen_nt_asm.py): Standard / Median, lane = node,
//                    one column quad per wavefront, x from scalar loads of the transposed chunk (vsom_xq.hip)
//                    as SGPR operands of v_pk_*, (c,w) staged once per workgroup through LDS, the 5-operation
//                    step for all-zero quads.
//                  - vsom_update_clr_rp8_gfx950 (gen_update_asm.py): CLR, lane = node, 8 parameter pairs per lane.
//                  - update_chain3_kernel (HIP, below): maps too small to fill the chip with lane = node
//                    (C4: 64x64x32): one lane per (node, dim pair) chain, operands staged through LDS.
//   sigma_finalize_kernel : sigmaMap = sqrt(S / W) for the columns the assembly kernels left as S
//                  (+ zeroes of the padding columns a ragged last slice wrote).
#include "vsom_device.hpp"
#include <cmath>
#include <cstdlib>
#include <mutex>

// (c,w) layout: pair-interleaved, one float4 {c_j, w_j, c_j+1, w_j+1} per node and sample pair at
// [(j>>1)][node] -- the chain kernels stage it with 16-byte loads.

// Role-split neighbourhood chain: one workgroup (9 wavefronts) serves NW nodes and walks the chunk
// in tiles of T samples (NW*T = 2048) through LDS.  Wavefront 0 does nothing but the serial part --
// the fp32 prefix W_j = W_{j-1} + w_j of tile t (one LDS read, one add, one LDS write per sample);
// wavefronts 1..8 meanwhile look up w for tile t+1 and turn tile t-1 into (c = w/W, w) pairs,
// stored as one float4 per node and sample pair.  No redundant chain work, so the time per
// workgroup is the chain's own ~B dependent adds instead of B x (lookup + division + chain).
#define CWP_WT 512                               // worker threads
#define CWP_THREADS (64 + CWP_WT)
template <int NW, int T, bool LUT_LDS>
__global__ __launch_bounds__(CWP_THREADS) void cwp_kernel(const u64 *__restrict__ lastbmu, int B, int n0, int n1,
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
            int bx = 0, by = 0;
            if (j < B)
                vsom_somindex(lastbmu[j], (u64)W, (u64)H, bx, by);   // SomIndex(*this, lastBMU[j]) (Som.cpp:847-849)
            bL[(t % 3) * T + wt] = make_int2(bx, by);
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

typedef float vsom_f2 __attribute__((ext_vector_type(2)));

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

// Chains of the maps too small for lane = node (C4: 64x64x32 = 65 536 packed chains): lane = (node, dim pair),
// packed arithmetic, operands staged ONCE per workgroup through LDS.  (Loading them per lane straight from
// global memory -- a 512-byte x load and half a 1-KB (c,w) load per wavefront and sample whose lanes mostly
// repeat addresses -- was bound by the vector-memory pipe, not by arithmetic: C4 0.86 ms.)  The 256 lanes of a workgroup (NW = 256 >> PLOG nodes x
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
    auto one = [&](vsom_f2 xv, float c, float w) {        // Standard, one sample
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

// hand-scheduled gfx950 code object (gen_update_asm.py + gen_nt_asm.py -> vsom_update_gfx950.s -> .hsaco),
// embedded at build time
static const unsigned char vsom_update_hsaco[] = {
#include "vsom_update_hsaco.inc"
};

// kernarg segment shared by the hand-scheduled kernels (80 bytes; the CLR kernel reads the first 72)
struct UpdAsmArgs {
    const void *xs;                  // nt: Xq (vsom_xq.hip); CLR: x' rows
    const void *cw2;
    void *map;
    void *sbuf;
    unsigned ldx_bytes, ldn_bytes, B, nloc, nslices, pitch_bytes, n0, ppitch_bytes;   // ppitch: CLR only
    const void *yp;                  // CLR: y' rows; nt: live-column record of the compaction or null
    const void *zq;                  // nt: all-zero (sample, quad) bit mask
};

int vsom_load_asm_module(vsom_ctx *c)
{
    if (c->upd_module)
        return VSOM_OK;
    hipModule_t mod;
#ifdef VSOM_DEVELOPMENT
    if (const char *alt = std::getenv("VSOM_ASM_HSACO")) {   // time a variant code object (tools/exp/mk_variant.sh)
        VSOM_HIP_CHECK(hipModuleLoad(&mod, alt));
    } else
#endif
    VSOM_HIP_CHECK(hipModuleLoadData(&mod, vsom_update_hsaco));
    hipFunction_t f;
    VSOM_HIP_CHECK(hipModuleGetFunction(&f, mod, "vsom_update_clr_rp8_gfx950"));
    c->upd_clr8 = f;
    static const char *const nt_names[4] = {"std", "fma", "sfma", "med"};
    for (int i = 0; i < 4; ++i) {
        const std::string n = std::string("vsom_update_") + nt_names[i] + "_nt4_gfx950";
        VSOM_HIP_CHECK(hipModuleGetFunction(&f, mod, n.c_str()));
        c->upd_nt[i] = f;
    }
    c->upd_module = mod;
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

// Maps whose node shard times depth is small (ceil(nodes/64) * ceil(D/14) <= VSOM_CHAIN_MAX_WAVES, the measured
// crossover against lane = node kernels, vsom_internal.hpp) take the small-map chain kernel, lane = (node, dim
// pair): C4's 64x64x32 map would give the quad kernels half a wavefront per SIMD.
static bool vsom_use_chain(const vsom_ctx *c, size_t nloc)
{
    if (!c->use_chain || c->transform == VSOM_CLR)
        return false;
    return ((nloc + 63) / 64) * ((c->D + 13) / 14) <= VSOM_CHAIN_MAX_WAVES;
}

extern "C" int vsom_small_map_chains(const vsom_ctx *c, size_t nodes) { return c && vsom_use_chain(c, nodes) ? 1 : 0; }

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
    // pair rows: ceil(B/2) + what the kernels' staging reads ahead (one block of 16 pair-rows) + slack
    const size_t prow = (c->B + 1) / 2 + 24;
    const size_t need = prow * ldn * 2;   // float2 elements
    if (need > c->cw_cap) {
        if (c->cw)
            VSOM_HIP_CHECK(hipFree(c->cw));
        c->cw = nullptr;
        VSOM_HIP_CHECK(hipMalloc(&c->cw, need * sizeof(float2)));
        // rows beyond B are staged (never used): keep them initialised
        VSOM_HIP_CHECK(hipMemsetAsync(c->cw, 0, need * sizeof(float2), c->stream));
        c->cw_cap = need;
    }
    {
        TimerScope ts(c, VSOM_T_CW);
        // role-split kernel, 16 nodes per workgroup; the table in LDS only while it is small: what counts is how
        // many workgroups (worker wavefronts) a CU holds -- 40 KB of tiles each; a 64-KB table (128x128 map) would
        // leave one per CU.  Measured at C3: 64 nodes + LDS table 0.153 ms, 32 + global 0.142, 16 + global 0.121,
        // 16 + LDS 0.20.
        const size_t lut_bytes = (size_t)c->lut_w * c->lut_h * sizeof(float);
        const bool lds = lut_bytes <= 24 * 1024;
        constexpr int nw = 16, T = 2048 / nw;
        const size_t smem = (size_t)5 * 2048 * sizeof(float) + (size_t)3 * T * sizeof(int2) + (lds ? lut_bytes : 0);
        const void *fn = lds ? (const void *)cwp_kernel<nw, T, true> : (const void *)cwp_kernel<nw, T, false>;
        if (smem > 64 * 1024)
            VSOM_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const u64 *bxy = c->lastbmu;
        int B = (int)c->B, in0 = (int)n0, in1 = (int)n1, iW = (int)c->W, iH = (int)c->H, lw = (int)c->lut_w,
            lh = (int)c->lut_h, ildn = (int)ldn;
        const float *lut = c->lut;
        float2 *cwp = c->cw;
        float *wgt = c->weight;
        void *args[] = {&bxy, &B, &in0, &in1, &iW, &iH, &lut, &lw, &lh, &cwp, &ildn, &wgt};
        VSOM_HIP_CHECK(hipLaunchKernel(fn, dim3((unsigned)((nloc + nw - 1) / nw)), dim3(CWP_THREADS), args, smem, c->stream));
    }
    int sig_cols = 0;   // > 0: columns left as raw S by the kernel; < 0: the compaction's scratch rows hold M and raw S
    {
        TimerScope ts(c, VSOM_T_UPDATE);
        const unsigned gx = (unsigned)((nloc + 63) / 64);
        if ((rc = vsom_load_asm_module(c)))
            return rc;
        if (c->transform == VSOM_CLR) {
            // lane = node, 8 parameter pairs per wavefront (gen_update_asm.py); a ragged last slice runs over the
            // parts' zero padding (part_pitch is a multiple of 32)
            constexpr unsigned RP = 8;
            const unsigned nsl = (c->part_len + RP - 1) / RP;
            UpdAsmArgs a;
            a.xs = c->XP;
            a.cw2 = c->cw;
            a.map = c->map;
            a.sbuf = c->sigma;
            a.ldx_bytes = c->part_pitch * 4u;
            a.ldn_bytes = (unsigned)(ldn * 16u);
            a.B = (unsigned)c->B;
            a.nloc = (unsigned)nloc;
            a.nslices = nsl;
            a.pitch_bytes = c->pitch * 4u;
            a.n0 = (unsigned)n0;
            a.ppitch_bytes = c->part_pitch * 4u;
            a.yp = c->YP;
            a.zq = nullptr;
            size_t sz = 72;
            void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            VSOM_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)c->upd_clr8, 8 * ((nsl + 3) / 4), (gx + 7) / 8, 1, 256, 1, 1, 0,
                                                 c->stream, nullptr, extra));
            sig_cols = (int)(nsl * RP);
        } else if (vsom_use_chain(c, nloc)) {
            // small maps: one lane per (node, dim pair) chain, operands staged through LDS (update_chain3_kernel);
            // at least 8 dim pairs per node row (lanes past the depth idle)
            int pl_log2 = 3;
            while ((2u << pl_log2) < c->D && pl_log2 < 6)
                ++pl_log2;
            const unsigned PL = 1u << pl_log2, npairs = (c->D + 1) / 2;
            dim3 grid2((unsigned)((nloc + (256 / PL) - 1) / (256 / PL)), (npairs + PL - 1) / PL);
            const bool med = c->transform == VSOM_MEDIAN, fma = c->update_mode == VSOM_UPDATE_FMA,
                       sfma = c->update_mode == VSOM_UPDATE_FMA_SIGMA;
#define VSOM_K3(P) (med ? (const void *)update_chain3_kernel<true, 0, P> \
                        : (fma ? (const void *)update_chain3_kernel<false, 1, P>                    \
                               : (sfma ? (const void *)update_chain3_kernel<false, 2, P> : (const void *)update_chain3_kernel<false, 0, P>)))
            const void *k3 = pl_log2 == 3 ? VSOM_K3(3) : pl_log2 == 4 ? VSOM_K3(4) : pl_log2 == 5 ? VSOM_K3(5) : VSOM_K3(6);
#undef VSOM_K3
            const float *xs_ = c->Xs;
            const float2 *cw_ = c->cw;
            int ildx = (int)c->xpitch, ildn = (int)ldn, iB = (int)c->B, in0 = (int)n0, inl = (int)nloc, iD = (int)c->D,
                ipitch = (int)c->pitch;
            float *map_ = c->map, *sg_ = c->sigma;
            const float *wt_ = c->weight;
            void *args[] = {&xs_, &ildx, &cw_, &ildn, &iB, &in0, &inl, &iD, &map_, &sg_, &ipitch, &wt_};
            VSOM_HIP_CHECK(hipLaunchKernel(k3, grid2, dim3(256), args, 0, c->stream));
        } else {
            // Standard / Median: lane = node, one column quad per wavefront, workgroup = 64 nodes x 8 quads, x from
            // scalar loads of the transposed chunk (gen_nt_asm.py, vsom_xq.hip); grid.x = 8 * column blocks
            // (XCD-aware).  With the column compaction (vsom_compact.hip) the kernel runs on the gathered live
            // columns -- their count is a device value the kernel reads from the record -- into dense scratch rows
            // that cc_expand_kernel writes back; otherwise a last quad of a ragged depth runs into the rows' zero
            // padding and sigma_finalize_kernel re-zeroes it.
            if ((rc = vsom_xq_ensure(c)))
                return rc;
            // nothing below reads the staged rows (Xs / Xc / int8 images) of this chunk any more: the next chunk's
            // staging kernels may overwrite them beside the chains (vsom_prefetch_chunk / vsom_stage_next_device)
            // -- where that pays: kernels running beside the chain kernel take slots and clock from it, and measured
            // (tools/exp/ab_stage.py, interleaved A/B, strict, B = 4096 x 784) the step gains 2.7 % / 2.3 % on 48x48 /
            // 64x64 maps (the chain launch leaves slots idle: 1.3 rounds of the 1024 resident workgroups), nothing
            // at 80x80 / 96x96 and LOSES 1 % at 112x112 / 128x128 (six full rounds: 0.05 ms of staging kernels cost the
            // chains 0.09 ms).  So the event is offered only when the chain launch is at most two rounds.
            const size_t chain_wgs = (size_t)gx * (((c->cc_valid ? c->cpitch / 4 : (c->D + 3) / 4) + 7) / 8);
            if (chain_wgs <= 2048) {
                VSOM_HIP_CHECK(hipEventRecord(c->ev_rows_free, c->stream));
                c->rows_free_valid = true;
            }
            const bool compact = c->cc_valid;
            if (compact && (rc = vsom_cc_ensure_update_scratch(c)))
                return rc;
            const bool fma = c->update_mode == VSOM_UPDATE_FMA, sfma = c->update_mode == VSOM_UPDATE_FMA_SIGMA;
            const bool med = c->transform == VSOM_MEDIAN;     // its FMAs are exact: one kernel for all modes
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
            a.zq = c->zq;
            size_t sz = 80;
            void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            void *fn = c->upd_nt[med ? 3 : (fma ? 1 : (sfma ? 2 : 0))];
            VSOM_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)fn, 8 * ((quads + 7) / 8), (gx + 7) / 8, 1, 512, 1, 1, 0,
                                                 c->stream, nullptr, extra));
            sig_cols = compact ? -1 : (int)(quads * 4);
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
