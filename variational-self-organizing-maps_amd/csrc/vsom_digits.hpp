// vsom_digits.hpp -- the base-128 digit grid of the integer shortlist contraction (vsom_sl_i8.hip), shared by the
// kernels that quantise model rows and sample rows.
//
// A row with largest finite magnitude mx gets the scale s = 2^E, E = exponent(mx) - 5, so that |v| / s < 64, and
// every value is written as   v = s (d1 + d2/128 + d3/16384) + r   with integer digits d_l in [-64, 64] and
// |r| <= s 2^-15.  The residuals between the steps are differences of a value and its rounding to a coarser power-of-two
// grid: exact in fp32.  Rows too small for the grid (E < -50) keep E = -50; their bound is the magnitude itself.  (With
// both scales >= 2^-50 the product t_s s_n 2^-13 of a sample's and a model row's scale is a normal fp32 number: the fp32
// epilogues multiply them, sl_k64_kernel.)
#pragma once
#include "vsom_device.hpp"
#include <type_traits>

// slots of one counter set `scal` (unsigned words; layout: vsom_shortlist.hip): 32 line-sized slots per quantity
#define SLI_NMAX(slot) (1024 + 32 * (slot))     // max |M_n|^2 over the finite rows
#define SLI_EMAX(slot) (2048 + 32 * (slot))     // max eps_n: digit residual bound of a model row (>= s_n 2^-15)
#define SLI_L1MAX(slot) (3072 + 32 * (slot))    // max |M_n|_1 over the contracted columns
#define SLI_NONZERO 6                           // some model value is not +-0 (NaN counts)

#define SL_EFLOOR (-50)
// scale 2^E, its inverse and the residual bound eps of a row whose largest finite magnitude is mx
__device__ __forceinline__ void sl_row_scale(float mx, float &s1, float &is1, float &eps)
{
    int e = mx > 0.f ? (int)((__float_as_uint(mx) >> 23) & 0xFF) - 127 : -100;
    e = mx > 0.f && ((__float_as_uint(mx) >> 23) & 0xFF) == 0 ? -126 : e;      // denormal maximum
    int E = e - 5;                                       // mx < 2^(e+1)  ->  mx / 2^E < 64
    E = E < SL_EFLOOR ? SL_EFLOOR : E;
    s1 = __uint_as_float((unsigned)(E + 127) << 23);
    is1 = __uint_as_float((unsigned)(127 - E) << 23);
    // eps >= s 2^-15 in every case (callers bound s by eps 2^15): a row below the grid is bounded by its magnitude
    eps = e - 5 < SL_EFLOOR ? fmaxf(mx, 0x1.0p-65f) : s1 * 0x1.0p-15f;
}

// the three digits of v on the grid of scale s1 (non-finite values count as 0: such a row is excluded / redone)
__device__ __forceinline__ void sl_digits3(float v, float s1, float is1, int &a, int &b, int &c)
{
    const float s2 = s1 * 0.0078125f, is2 = is1 * 128.f, is3 = is2 * 128.f;      // s / 128 (exact), 128 / s, 16384 / s
    v = fabsf(v) <= 3.0e38f ? v : 0.f;
    float t = rintf(v * is1);
    t = fminf(fmaxf(t, -64.f), 64.f);
    const float ra = v - t * s1;                         // exact
    float t2 = rintf(ra * is2);
    t2 = fminf(fmaxf(t2, -64.f), 64.f);
    const float rb = ra - t2 * s2;                       // exact
    float t3 = rintf(rb * is3);
    t3 = fminf(fmaxf(t3, -64.f), 64.f);
    a = (int)t;
    b = (int)t2;
    c = (int)t3;
}

// Minimum over the 32 lanes that share `lane & ~31`, with DPP row operations on the vector pipe (five v_min_f32_dpp);
// the result is valid in lanes 31 and 63.  Inputs must not be NaN.  (The same reduction with __shfl_xor is five
// ds_bpermute_b32 round trips through the LDS crossbar, each waited for: 160 of them per wavefront were 40 % of the
// integer contraction kernel's time -- 257 -> see profiles/EXPERIMENTS.md.)
__device__ __forceinline__ float sl_min32_dpp(float v)
{
    auto step = [](float x, auto ctrl, auto rowmask) {
        const int b = __float_as_int(x);
        const int o = __builtin_amdgcn_update_dpp(b, b, decltype(ctrl)::value, decltype(rowmask)::value, 0xF, false);
        return fminf(x, __int_as_float(o));
    };
    v = step(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});     // quad_perm [1,0,3,2]
    v = step(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});     // quad_perm [2,3,0,1]
    v = step(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{});    // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{});    // row_mirror: every lane of a row holds the row's minimum
    v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});    // row_bcast:15 into rows 1 and 3
    return v;
}

// Feedback of a shortlist search: {redo samples, candidates} of this call into the host-visible words (a hint for the
// pause policy, read without synchronising), and the counters of the OTHER counter set cleared for the next search (the
// two sets alternate: nobody touches the other one during this search).  Done by the first wavefront of the redo pass's
// exact kernel (bmu_tile_kernel, vsom_bmu.hip) -- it runs after the refinement kernel anyway; as a launch of its own
// this was 6 us per search.
struct SlFeedback {
    const unsigned *scal;        // this search's counter set (null: no feedback)
    unsigned *host_fb;
    unsigned nrows;
    const unsigned *xflag;       // the chunk's kind word or null
    unsigned *scal_next;
};
__device__ __forceinline__ void sl_feedback_write(const SlFeedback &f)
{
    const unsigned t = threadIdx.x;          // < 64
    if (t < 8)
        f.scal_next[t] = 0u;
    if (t < 32) {
        f.scal_next[16 + 32 * t] = 0u;       // candidate-count slots
        f.scal_next[SLI_NMAX(t)] = 0u;       // max |M|^2 / eps / |M|_1 slots of the integer contraction (vsom_sl_i8.hip)
        f.scal_next[SLI_EMAX(t)] = 0u;
        f.scal_next[SLI_L1MAX(t)] = 0u;
    }
    unsigned cand = t < 32 ? f.scal[16 + 32 * t] : 0u;
    for (int off = 16; off > 0; off >>= 1)
        cand += (unsigned)__shfl_xor((int)cand, off);
    if (t != 0)
        return;
    f.host_fb[0] = f.scal[4];
    f.host_fb[1] = cand;
    f.host_fb[2] = f.nrows;
    f.host_fb[4] = f.xflag ? f.xflag[0] : 0u;
    // (no fence before the sequence word: a torn read costs at most one misjudged search, and a system-scope fence here
    // waited 2-3 us for the writes to cross the bus)
    f.host_fb[3] = f.host_fb[3] + 1u;
}
