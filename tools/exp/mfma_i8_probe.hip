// Probe: operand lane map of v_mfma_i32_32x32x32_i8 on gfx950 with exact integer data.
// Assumed: lane l (r = l & 31, h = l >> 5) holds A[row r][k = 16 h + j] and B[k = 16 h + j][col r] in byte j
// (j = 0..15) of its 128-bit fragment; C/D: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
// build: hipcc --offload-arch=gfx950 -O2 tools/exp/mfma_i8_probe.hip -o tools/exp/bin/mfma_i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void probe(const signed char *A, const signed char *B, int *D)
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v4i a, b;
    memcpy(&a, A + r * 32 + 16 * h, 16);            // A row-major [32][32]
    signed char bb[16];
    for (int j = 0; j < 16; ++j)
        bb[j] = B[(16 * h + j) * 32 + r];           // B row-major [k][col]
    memcpy(&b, bb, 16);
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        D[row * 32 + r] = c[reg];
    }
}

int main()
{
    signed char hA[1024], hB[1024];
    int hD[1024], ref[1024];
    srand(1);
    for (int i = 0; i < 1024; ++i) {
        hA[i] = (signed char)(rand() % 255 - 127);
        hB[i] = (signed char)(rand() % 255 - 127);
    }
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            int s = 0;
            for (int k = 0; k < 32; ++k)
                s += (int)hA[i * 32 + k] * (int)hB[k * 32 + j];
            ref[i * 32 + j] = s;
        }
    signed char *dA, *dB;
    int *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i)
        bad += hD[i] != ref[i];
    printf("mfma_i32_32x32x32_i8 assumed lane map: %s (%d mismatches)\n", bad ? "WRONG" : "CONFIRMED", bad);
    return bad != 0;
}
