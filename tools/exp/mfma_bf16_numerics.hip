// mfma_bf16_numerics.hip -- development experiment: how does v_mfma_f32_32x32x16_bf16 round?
// A rigorous shortlist bound over a bf16-split contraction (three exact bf16 parts of an fp32 operand,
// 16x the fp32 MFMA rate) needs to know what one instruction does with its 16 exact products and C.
// For random operands with a wide exponent spread, D = A(32x16) * B(16x32) + C is compared bit for bit with
//   (a) the exact sum (double: exponents are confined so that it is exact) rounded ONCE to nearest-even,
//   (b) the exact sum TRUNCATED towards zero,
//   (c) a sequential fmaf chain over k = 0..15 starting from C,
//   (d) two sequential chains (k 0..7 and 8..15, the two lane halves) added to C.
// Build: hipcc --offload-arch=gfx950 -O2 tools/exp/mfma_bf16_numerics.hip -o tools/exp/mfma_bf16_numerics
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const uint16_t *A, const uint16_t *B, const float *C, float *D)
{
    // lane l (r = l&31, h = l>>5) holds A[row r][k = 8h + j] and B[k = 8h + j][col r], j = 0..7
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    bf16x8 a, b;
    uint16_t ta[8], tb[8];
    for (int j = 0; j < 8; ++j) {
        ta[j] = A[r * 16 + 8 * h + j];
        tb[j] = B[(8 * h + j) * 32 + r];
    }
    memcpy(&a, ta, 16);
    memcpy(&b, tb, 16);
    f32x16 c;
    for (int i = 0; i < 16; ++i)
        c[i] = C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r];     // row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i)
        D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}

static float bf(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t rnd_bf16(int spread)
{
    const int e = 127 + (rand() % (2 * spread + 1)) - spread;
    return (uint16_t)(((rand() & 1) << 15) | (e << 7) | (rand() & 0x7F));
}
static float trunc_to_f32(double v)
{
    float f = (float)v;                      // nearest
    if (std::fabs((double)f) > std::fabs(v)) // went away from zero: step back
        f = std::nextafterf(f, 0.f);
    return f;
}

int main()
{
    uint16_t *dA, *dB;
    float *dC, *dD;
    hipMalloc(&dA, 32 * 16 * 2); hipMalloc(&dB, 16 * 32 * 2); hipMalloc(&dC, 32 * 32 * 4); hipMalloc(&dD, 32 * 32 * 4);
    for (int spread : {0, 2, 6, 12}) {
        for (int withC = 0; withC < 2; ++withC) {
            long n = 0, ma = 0, mb = 0, mc = 0, md = 0;
            for (int trial = 0; trial < 200; ++trial) {
                std::vector<uint16_t> A(32 * 16), B(16 * 32);
                std::vector<float> C(32 * 32, 0.f), D(32 * 32);
                for (auto &v : A) v = rnd_bf16(spread);
                for (auto &v : B) v = rnd_bf16(spread);
                if (withC)
                    for (auto &v : C) v = bf(rnd_bf16(spread)) * (float)(rand() % 1000) / 64.f;
                hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
                hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
                hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
                hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
                for (int i = 0; i < 32; ++i)
                    for (int j = 0; j < 32; ++j) {
                        double ex = (double)C[i * 32 + j];
                        float seq = C[i * 32 + j], h0 = 0.f, h1 = 0.f;
                        for (int kk = 0; kk < 16; ++kk) {
                            const float a = bf(A[i * 16 + kk]), b = bf(B[kk * 32 + j]);
                            ex += (double)a * (double)b;
                            seq = fmaf(a, b, seq);
                            if (kk < 8) h0 = fmaf(a, b, h0); else h1 = fmaf(a, b, h1);
                        }
                        const float got = D[i * 32 + j];
                        uint32_t g; memcpy(&g, &got, 4);
                        auto same = [&](float f) { uint32_t u; memcpy(&u, &f, 4); return u == g; };
                        ++n;
                        ma += same((float)ex);
                        mb += same(trunc_to_f32(ex));
                        mc += same(seq);
                        md += same((h0 + h1) + C[i * 32 + j]);
                    }
            }
            printf("spread 2^+-%d, C %s: of %ld outputs  exact-sum RN %ld  exact-sum truncated %ld  fmaf chain %ld  two half chains %ld\n",
                   spread, withC ? "random" : "zero", n, ma, mb, mc, md);
        }
    }
    return 0;
}
