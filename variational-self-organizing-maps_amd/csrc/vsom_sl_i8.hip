// vsom_sl_i8.hip -- the shortlist contraction of Som::findBmu (Som.cpp:291-309; vsom_shortlist.hip) on the INTEGER
// matrix pipe, for chunks whose samples are small non-negative integers (MNIST: MnistDataLoader.cpp:73-75 yields the raw
// pixel values 0..255 as floats).
//
// The matrix pipe only PRUNES: G[s][n] ~ |M_n|^2 - 2 <x_s, M_n> with a proven error bound, every returned index and
// distance comes from the exact-order evaluation of sl_select_kernel.  Here the approximation is computed in exact
// integer arithmetic instead of an fp32 MFMA chain, 10x faster on the matrix pipe (v_mfma_i32_32x32x32_i8 runs 32x the
// multiply-adds per cycle of v_mfma_f32_32x32x2_f32; three of them per product) and with a TIGHTER bound:
//   * model rows: M_nk = s_n (q1 + q2/128 + q3/16384)_nk + r_nk, s_n a power of two >= 2^-5 max_k |M_nk|... chosen so
//     that q1 = rint(M/s) lies in [-64, 64]; the residuals are formed exactly in fp32 (differences of a value and its
//     rounding to a coarser grid), q2, q3 in [-64, 64], |r_nk| <= s_n 2^-15 =: eps_n            (sl_prepare_i8_kernel)
//   * samples: x_sk integer in [0, 255]  ->  int8 (x - 128); the offset is put back with the row sums of q (exact)
//   * <x, M^> = s_n 2^-14 [ 16384 I1 + 128 I2 + I3 ],  I_l = sum_k x_k q_l,nk exact in int32; the combination, the
//     offset term, the scale and |M|^2 - 2 <x, M^> are evaluated in fp64 (exact up to the final rounding to fp32)
// Bound:  |G - (|M_n|^2 - 2 <x, M_n>)| <= 2 |x|_1 eps_n + u (|M_n|^2 from its fp64 sum) + u |G|
//                                      <= 2 |x|_1 eps_max + 3.1 u (nMmax + |x|^2)               (u = 2^-24)
// which sl_select_kernel uses in place of the fp32 chain's 2 g1 (nMmax + |x|^2), g1 = (32 + K/32 + 3) u: at C3
// (K = 672 live columns, |x|_1 ~ 2e4, max |M| ~ 255) about 5 against 70.
// A chunk that is not of that kind (quant_x raises a flag) is searched by the exact-order kernel this once, and the
// host returns to the fp32 contraction for the context (vsom_shortlist.hip).
#include "vsom_device.hpp"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// scal slots (unsigned words): see vsom_shortlist.hip; 32 line-sized slots each for max nrm / max eps
#define SLI_NMAX(slot) (1024 + 32 * (slot))
#define SLI_EMAX(slot) (1024 + 1024 + 32 * (slot))

// ---- samples: int8 image, |x|_1, "is this uint8 data" --------------------------------------------------------------
// one wavefront per row; xi row pitch kp8 bytes, columns past the row's length hold x = 0 (-128)
__global__ __launch_bounds__(256) void sl_quant_x_kernel(const float *__restrict__ X, int ldx, int B, int kp8,
                                                         signed char *__restrict__ xi, float *__restrict__ l1,
                                                         unsigned *__restrict__ xflag)
{
    const int row = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B)
        return;
    const float *x = X + (size_t)row * ldx;
    signed char *o = xi + (size_t)row * kp8;
    float sum = 0.f;
    bool bad = false;
    for (int k4 = lane * 4; k4 < kp8; k4 += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k4 < ldx)                                   // ldx is a multiple of 32
            v = *reinterpret_cast<const float4 *>(x + k4);
        const float vv[4] = {v.x, v.y, v.z, v.w};
        char4 q;
        signed char *qq = reinterpret_cast<signed char *>(&q);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float f = vv[u];
            const bool ok = f >= 0.f && f <= 255.f && f == rintf(f);      // NaN fails the comparisons
            bad |= !ok;
            const int iv = ok ? (int)f : 0;
            sum += (float)iv;                            // exact: integers, total < 2^24 for rows up to 65536 values
            qq[u] = (signed char)(iv - 128);
        }
        *reinterpret_cast<char4 *>(o + k4) = q;
    }
    for (int off = 32; off > 0; off >>= 1)
        sum += __shfl_xor(sum, off);
    if (__ballot(bad) && lane == 0)
        atomicOr(xflag, 1u);
    if (lane == 0)
        l1[row] = sum;
}

// ---- model rows: |M|^2 (fp64 sum), live columns gathered, three 7-bit digits, row sums -------------------------------
// one WAVEFRONT per node (4 per workgroup), no LDS, no barriers.  idx = live-column list of the compaction (null:
// identity), kp = contraction length (device value kp_dev[2] when compacted).  q planes: [3][N][kp8].
__global__ __launch_bounds__(256) void sl_prepare_i8_kernel(const float *__restrict__ map, int ldm, int Dp, int N,
                                                            const int *__restrict__ idx, int kp, const unsigned *__restrict__ kp_dev,
                                                            int kp8, signed char *__restrict__ q, float *__restrict__ nrm,
                                                            double *__restrict__ qscale, double *__restrict__ qcorr,
                                                            unsigned *__restrict__ scal)
{
    const int n = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N)
        return;
    if (kp_dev)
        kp = (int)kp_dev[2];
    const float *src = map + (size_t)n * ldm;
    double ss = 0.0;
    for (int d = lane * 4; d < Dp; d += 256) {          // rows are zero padded to Dp, a multiple of 32
        const float4 v = *reinterpret_cast<const float4 *>(src + d);
        ss += (double)v.x * (double)v.x + (double)v.y * (double)v.y;
        ss += (double)v.z * (double)v.z + (double)v.w * (double)v.w;
    }
    for (int off = 32; off > 0; off >>= 1)
        ss += __shfl_xor(ss, off);
    const float nf = (float)ss;                          // NaN rows stay NaN, overflow -> inf
    // the row's largest live magnitude (non-finite values count as 0: such a row is excluded / redone anyway)
    float mx = 0.f;
    for (int k = lane; k < kp; k += 64) {
        const int c = idx ? idx[k] : k;
        float v = c >= 0 && c < Dp ? src[c] : 0.f;
        v = fabsf(v);
        mx = (v <= 3.0e38f && v > mx) ? v : mx;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
    }
    // s = 2^E with |M| / s < 64:  E = exponent(mx) - 5  (mx < 2^(exponent+1)); tiny rows: E >= -100
    int e = mx > 0.f ? (int)((__float_as_uint(mx) >> 23) & 0xFF) - 127 : -100;
    e = mx > 0.f && ((__float_as_uint(mx) >> 23) & 0xFF) == 0 ? -126 : e;      // denormal maximum
    int E = e - 5;
    E = E < -100 ? -100 : E;
    const float s1 = __uint_as_float((unsigned)(E + 127) << 23), is1 = __uint_as_float((unsigned)(127 - E) << 23);
    const float s2 = s1 * 0.0078125f, s3 = s2 * 0.0078125f;                     // s / 128, s / 16384 (exact)
    const float is2 = is1 * 128.f, is3 = is2 * 128.f;
    int r1 = 0, r2 = 0, r3 = 0;
    const size_t plane = (size_t)N * kp8;
    signed char *q1 = q + (size_t)n * kp8, *q2 = q1 + plane, *q3 = q2 + plane;
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
                float v = c >= 0 && c < Dp ? src[c] : 0.f;
                v = fabsf(v) <= 3.0e38f ? v : 0.f;
                float t = rintf(v * is1);                // |v| / s < 64 unless the row is tiny (E clamped): clamp
                t = fminf(fmaxf(t, -64.f), 64.f);
                const float ra = v - t * s1;             // exact
                float t2 = rintf(ra * is2);
                t2 = fminf(fmaxf(t2, -64.f), 64.f);
                const float rb = ra - t2 * s2;           // exact
                float t3 = rintf(rb * is3);
                t3 = fminf(fmaxf(t3, -64.f), 64.f);
                a = (int)t;
                b = (int)t2;
                c3 = (int)t3;
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
    for (int off = 32; off > 0; off >>= 1) {
        r1 += __shfl_xor(r1, off);
        r2 += __shfl_xor(r2, off);
        r3 += __shfl_xor(r3, off);
    }
    if (lane == 0) {
        nrm[n] = nf;
        qscale[n] = (double)s3;                          // s 2^-14: multiplies 16384 I1 + 128 I2 + I3
        qcorr[n] = 128.0 * (double)((long long)r1 * 16384 + (long long)r2 * 128 + (long long)r3);   // the (x - 128) offset put back
        const int slot = n & 31;
        if (nf == nf) {
            if (nf > 3.0e38f)
                atomicOr(&scal[1], 1u);                  // an inf somewhere: the bound does not apply
            else
                atomicMax(&scal[SLI_NMAX(slot)], __float_as_uint(nf));
        }
        // eps_n = s 2^-15, or the whole magnitude of a row too small for the digit grid (E clamped)
        const float eps = e - 5 < -100 ? mx : s1 * 3.0517578125e-05f;
        atomicMax(&scal[SLI_EMAX(slot)], __float_as_uint(eps));
    }
}

// ---- G = |M|^2 - 2 <x, M^>, tile minima -----------------------------------------------------------------------------
// workgroup tile (64 MI) samples x 64 nodes, wavefront tile (32 MI) x 32 (MI MFMA tiles of 32 x 32), three int32
// accumulator sets (the digits), K streamed through LDS in chunks of 64 bytes with the next chunk's global loads in
// flight.  The contraction itself is ~0.06 ms of matrix-pipe time at C3; what the kernel waits for is its staging
// loads (1.8 GB out of L2 at MI = 2, one chunk of 20 KB per workgroup in flight): measured at C3 -- 128 x 64 tiles, two
// workgroups per CU 0.37 ms, three (154 VGPRs) 0.27 ms; 64 x 64 tiles, five per CU (2.8 GB) 0.30 ms; the fp64
// epilogue and the tile minima cost nothing measurable there.  (Big problems take the LDS-DMA ring kernel below.)
#define IT_N 64
#define IK 64
#define ILD 80      // LDS row stride in bytes (64 + 16: conflict-free 16-byte fragment reads)
template <int MI>
__global__ __launch_bounds__(256, MI == 1 ? 5 : 3) void sl_gemm_i8_kernel(const signed char *__restrict__ xi, int s0, int s1, const signed char *__restrict__ q,
                                                            int N, int kp, const unsigned *__restrict__ kp_dev, int kp8,
                                                            const float *__restrict__ nrm, const double *__restrict__ qscale,
                                                            const double *__restrict__ qcorr, float *__restrict__ G, int ldg,
                                                            float *__restrict__ tmin, int ntm)
{
    if (kp_dev)
        kp = (int)kp_dev[2];
    const int k64 = (kp + IK - 1) / IK * IK;             // <= kp8; columns past kp hold q = 0
    constexpr int IT_S = 64 * MI;
    __shared__ __attribute__((aligned(16))) signed char As[IT_S * ILD];
    __shared__ __attribute__((aligned(16))) signed char Bs[3][IT_N * ILD];
    __shared__ float smin[2][IT_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int sbase = s0 + blockIdx.y * IT_S, nbase = blockIdx.x * IT_N;
    const size_t plane = (size_t)N * kp8;

    v16i acc[3][MI];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[l][i][r] = 0;

    // staging: A tile (64 MI) rows x 64 B = 256 MI pieces of 16 B (MI per thread); B tiles 3 x 64 rows x 64 B = 768 pieces (3)
    v4i pa[MI], pb[3];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int f = tid + 256 * i, r = f >> 2, c = (f & 3) * 16;
            const int s = sbase + r;
            pa[i] = s < s1 ? *reinterpret_cast<const v4i *>(xi + (size_t)s * kp8 + k0 + c) : v4i{0, 0, 0, 0};
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
        for (int i = 0; i < MI; ++i) {
            const int f = tid + 256 * i, r = f >> 2, c = (f & 3) * 16;
            *reinterpret_cast<v4i *>(&As[r * ILD + c]) = pa[i];
        }
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const int r = tid >> 2, c = (tid & 3) * 16;
            *reinterpret_cast<v4i *>(&Bs[l][r * ILD + c]) = pb[l];
        }
        __syncthreads();
        if (k0 + IK < k64)
            gload(k0 + IK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v4i a[MI], b[3];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                a[i] = *reinterpret_cast<const v4i *>(&As[(wm * 32 * MI + i * 32 + lr) * ILD + ks * 32 + 16 * lh]);
#pragma unroll
            for (int l = 0; l < 3; ++l)
                b[l] = *reinterpret_cast<const v4i *>(&Bs[l][(wn * 32 + lr) * ILD + ks * 32 + 16 * lh]);
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    acc[l][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[l], acc[l][i], 0, 0, 0);
        }
    }
    // epilogue (C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)), in fp64: exact up to the
    // final rounding to fp32
    const float inf = __uint_as_float(0x7F800000u);
    const int col = nbase + wn * 32 + lr;
    const bool cok = col < N;
    const double nm = cok ? (double)nrm[col] : 0.0, sc = cok ? qscale[col] : 0.0, cr = cok ? qcorr[col] : 0.0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lrow = wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int row = sbase + lrow;
            const double t = (double)acc[0][i][r] * 16384.0 + (double)acc[1][i][r] * 128.0 + (double)acc[2][i][r] + cr;
            const float g = (float)(nm - 2.0 * (sc * t));
            if (row < s1 && cok)
                G[(size_t)(row - s0) * ldg + col] = g;
            // exact minimum of the finite entries of this row over the workgroup's 64 nodes: NaN -> +inf
            float mn = (cok && g == g) ? g : inf;
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) {     // the 32 lanes that share this row
                const float o = __shfl_xor(mn, off);
                mn = o < mn ? o : mn;
            }
            if (lr == 0)
                smin[wn][lrow] = mn;
        }
    }
    __syncthreads();
    if (tid < IT_S) {
        const int row = sbase + tid;
        if (row < s1) {
            const float a0 = smin[0][tid], a1 = smin[1][tid];
            tmin[(size_t)(row - s0) * ntm + blockIdx.x] = a1 < a0 ? a1 : a0;
        }
    }
}

// ---- the same contraction with big tiles and a ring of LDS stages filled by LDS-DMA ---------------------------------------
// What the register-staged kernel above waits for is its staging loads (1.8 GB out of L2 at C3 with ONE 20 KB chunk per
// workgroup in flight).  Here: workgroup tile 256 samples x 128 nodes (0.9 GB), 8 wavefronts of 64 x 64 (2 x 2 MFMA tiles x
// three digits = 192 accumulator registers), K in chunks of 64 bytes through a ring of THREE 40 KB stages that
// global_load_lds_dwordx4 fills without passing through registers -- two chunks in flight while one is consumed, one
// barrier per chunk.  Measured at C3: 0.26 ms against 0.28 (`SQ_VALU_MFMA_BUSY_CYCLES`: 32 cycles per MFMA, the matrix pipe
// 24 % busy; with one workgroup per CU the K loop and the fp64 epilogue -- ~2400 of the 2900 VALU instructions per
// wavefront -- no longer overlap across workgroups, which eats most of what the staging gains).
// A wave-instruction deposits 1 KB contiguously (16 rows x 64 B); the 16-byte pieces of a row are
// stored XOR-swizzled (slot = piece ^ ((row >> 2) & 3), applied on the GLOBAL side: each lane picks the piece that belongs
// into its slot), which makes the 16-byte fragment reads of 32 consecutive rows conflict-free without row padding.
#define RT_S 256
#define RT_N 128
#define RNS 3
#define RSTAGE 40960      // A 256 x 64 B | q1 128 x 64 B | q2 | q3
__global__ __launch_bounds__(512, 1) void sl_gemm_i8_ring_kernel(const signed char *__restrict__ xi, int s0, int s1,
                                                                 const signed char *__restrict__ q, int N, int kp,
                                                                 const unsigned *__restrict__ kp_dev, int kp8,
                                                                 const float *__restrict__ nrm, const double *__restrict__ qscale,
                                                                 const double *__restrict__ qcorr, float *__restrict__ G, int ldg,
                                                                 float *__restrict__ tmin, int ntm)
{
    if (kp_dev)
        kp = (int)kp_dev[2];
    const int nchunks = (kp + IK - 1) / IK;             // columns past kp hold q = 0
    extern __shared__ __attribute__((aligned(1024))) signed char ring[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int sbase = s0 + blockIdx.y * RT_S, nbase = blockIdx.x * RT_N;
    const size_t plane = (size_t)N * kp8;

    // loader: 40 units of 1 KB per stage (A: 16 units of 16 rows, each digit plane: 8), 5 per wavefront
    const signed char *src[5];
    int dst[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int id = wave * 5 + i;
        const int rin = lane >> 2, slot = lane & 3;
        if (id < 16) {
            const int row = id * 16 + rin;
            int sr = sbase + row;
            sr = sr < s1 ? sr : s1 - 1;                  // rows past the chunk: re-read the last one (never stored)
            src[i] = xi + (size_t)sr * kp8 + ((slot ^ ((row >> 2) & 3)) << 4);
            dst[i] = id * 1024;
        } else {
            const int pl = (id - 16) >> 3, u = (id - 16) & 7;
            const int row = u * 16 + rin;
            int n = nbase + row;
            n = n < N ? n : N - 1;
            src[i] = q + pl * plane + (size_t)n * kp8 + ((slot ^ ((row >> 2) & 3)) << 4);
            dst[i] = 16384 + pl * 8192 + u * 1024;
        }
    }
    auto issue = [&](int c) {                            // chunk c -> stage c % RNS
        const int sb = (c % RNS) * RSTAGE;
#pragma unroll
        for (int i = 0; i < 5; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[i] + (size_t)c * IK),
                                             (__attribute__((address_space(3))) void *)(ring + sb + dst[i]), 16, 0, 0);
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

    issue(0);
    if (nchunks > 1)
        issue(1);
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks)
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");     // my 5 pieces of chunk c have landed (chunk c+1's may be in flight)
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                 // everybody's pieces of chunk c; everybody is done with chunk c-1
        if (c + 2 < nchunks)
            issue(c + 2);                                // into the stage chunk c-1 was read from
        const signed char *st = ring + (c % RNS) * RSTAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            v4i a[2], b[3][2];
            const int P = ks * 2 + lh;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wm * 64 + i * 32 + lr;
                a[i] = *reinterpret_cast<const v4i *>(st + r * 64 + ((P ^ ((r >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int r = wn * 64 + j * 32 + lr;
                    b[l][j] = *reinterpret_cast<const v4i *>(st + 16384 + l * 8192 + r * 64 + ((P ^ ((r >> 2) & 3)) << 4));
                }
#pragma unroll
            for (int l = 0; l < 3; ++l)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[l][i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[l][j], acc[l][i][j], 0, 0, 0);
        }
    }
    // epilogue as above; a wavefront's 64 columns are exactly one 64-node tile of `tmin`
    const float inf = __uint_as_float(0x7F800000u);
    double nm[2], sc[2], cr[2];
    int col[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        col[j] = nbase + wn * 64 + j * 32 + lr;
        const bool ok = col[j] < N;
        nm[j] = ok ? (double)nrm[col[j]] : 0.0;
        sc[j] = ok ? qscale[col[j]] : 0.0;
        cr[j] = ok ? qcorr[col[j]] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = sbase + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float mn = inf;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double t = (double)acc[0][i][j][r] * 16384.0 + (double)acc[1][i][j][r] * 128.0 + (double)acc[2][i][j][r] + cr[j];
                const float g = (float)(nm[j] - 2.0 * (sc[j] * t));
                if (row < s1 && col[j] < N)
                    G[(size_t)(row - s0) * ldg + col[j]] = g;
                const float m = (col[j] < N && g == g) ? g : inf;       // exact minimum of the finite entries: NaN -> +inf
                mn = m < mn ? m : mn;
            }
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) {     // the 32 lanes that share this row
                const float o = __shfl_xor(mn, off);
                mn = o < mn ? o : mn;
            }
            if (lr == 0 && row < s1)
                tmin[(size_t)(row - s0) * ntm + blockIdx.x * 2 + wn] = mn;
        }
    }
}

// ---- host --------------------------------------------------------------------------------------------------------------
// prepares the int8 images and computes G / tmin for samples [s0, s1) (ldg, ntm as the fp32 path lays them out: 64-node
// tile minima); scal must have been reset
int launch_sl_i8(vsom_ctx *c, size_t s0, size_t s1, size_t ldg, size_t ntm, unsigned *xflag)
{
    const bool compact = c->cc_valid;
    const uint32_t kmax = compact ? c->cpitch : c->xpitch;
    const uint32_t kp8 = (kmax + 63) / 64 * 64;
    if (!c->sl_q || c->sl_kp8 != kp8) {
        if (c->sl_q) {
            VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
            VSOM_HIP_CHECK(hipFree(c->sl_q));
            VSOM_HIP_CHECK(hipFree(c->sl_qscale));
            VSOM_HIP_CHECK(hipFree(c->sl_qcorr));
        }
        c->sl_q = nullptr;
        VSOM_HIP_CHECK(hipMalloc(&c->sl_q, (size_t)3 * c->N * kp8));
        VSOM_HIP_CHECK(hipMalloc(&c->sl_qscale, (size_t)c->N * sizeof(double)));
        VSOM_HIP_CHECK(hipMalloc(&c->sl_qcorr, (size_t)c->N * sizeof(double)));
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
        VSOM_HIP_CHECK(hipMalloc(&c->sl_xi, (size_t)c->Bcap * kp8));
        VSOM_HIP_CHECK(hipMalloc(&c->sl_l1, (size_t)c->Bcap * sizeof(float)));
        c->sl_xi_cap = (size_t)c->Bcap * kp8;
        c->xi_valid = false;
    }
    unsigned *scal = c->sl_scal;
    if (!c->xi_valid) {      // once per staged chunk
        hipLaunchKernelGGL(sl_quant_x_kernel, dim3((unsigned)((c->B + 3) / 4)), dim3(256), 0, c->stream,
                           compact ? c->Xc : c->Xs, (int)kmax, (int)c->B, (int)kp8, c->sl_xi, c->sl_l1, xflag);
        c->xi_valid = true;
    }
    const unsigned *kp_dev = compact ? (const unsigned *)c->cc_meta : nullptr;
    hipLaunchKernelGGL(sl_prepare_i8_kernel, dim3((unsigned)((c->N + 3) / 4)), dim3(256), 0, c->stream, c->map, (int)c->pitch,
                       (int)c->part_pitch, (int)c->N, compact ? (const int *)c->cc_idx : (const int *)nullptr, (int)kmax, kp_dev,
                       (int)kp8, c->sl_q, c->sl_nrm, c->sl_qscale, c->sl_qcorr, scal);
    // big maps and chunks: 256 x 128 tiles through the LDS-DMA ring; otherwise (few tiles: they would not fill the
    // chip) 128 x 64 tiles staged through registers
    const size_t big_tiles = ((size_t)c->N + RT_N - 1) / RT_N * ((s1 - s0 + RT_S - 1) / RT_S);
    static bool ring_ok = hipFuncSetAttribute((const void *)sl_gemm_i8_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              RNS * RSTAGE) == hipSuccess;
    if (ring_ok && big_tiles >= 1024) {
        dim3 grid((unsigned)(ntm / 2), (unsigned)((s1 - s0 + RT_S - 1) / RT_S));   // ntm = 2 ceil(N / 128) 64-node tiles
        hipLaunchKernelGGL(sl_gemm_i8_ring_kernel, grid, dim3(512), RNS * RSTAGE, c->stream, c->sl_xi, (int)s0, (int)s1, c->sl_q,
                           (int)c->N, (int)kmax, kp_dev, (int)kp8, c->sl_nrm, c->sl_qscale, c->sl_qcorr, c->sl_G, (int)ldg,
                           c->sl_tmin, (int)ntm);
    } else {
        constexpr int MI = 2;
        dim3 grid((unsigned)ntm, (unsigned)((s1 - s0 + 64 * MI - 1) / (64 * MI)));   // ntm 64-node tiles (the last may lie past N: minima +inf)
        hipLaunchKernelGGL(sl_gemm_i8_kernel<MI>, grid, dim3(256), 0, c->stream, c->sl_xi, (int)s0, (int)s1, c->sl_q, (int)c->N,
                           (int)kmax, kp_dev, (int)kp8, c->sl_nrm, c->sl_qscale, c->sl_qcorr, c->sl_G, (int)ldg, c->sl_tmin, (int)ntm);
    }
    VSOM_HIP_CHECK(hipGetLastError());
    return VSOM_OK;
}
