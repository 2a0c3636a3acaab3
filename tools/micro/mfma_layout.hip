// Developer check of the operand / result layout of v_mfma_f32_32x32x2_f32 on gfx950 as k_nn_wave's stream stage assumes it:
//   A: lane l holds A[i = l % 32][k = l / 32],  B: lane l holds B[k = l / 32][j = l % 32],
//   D: lane l, register r holds D[i = 8 * (r / 4) + 4 * (l / 32) + (r % 4)][j = l % 32].
// build: hipcc -O2 --offload-arch=gfx950 -o tools/micro/mfma_layout tools/micro/mfma_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(const float *A, const float *B, float *D)      // A[32][2], B[2][32], D[32][32]
{
    const int l = threadIdx.x;
    const float a = A[(l % 32) * 2 + l / 32], b = B[(l / 32) * 32 + l % 32];
    floatx16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[(8 * (r / 4) + 4 * (l / 32) + (r % 4)) * 32 + l % 32] = acc[r];
}
int main()
{
    float hA[64], hB[64], hD[1024], *dA, *dB, *dD;
    for (int i = 0; i < 64; ++i) { hA[i] = (float)(1 + i); hB[i] = (float)(100 + 3 * i); }
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        const float want = hA[i * 2] * hB[j] + hA[i * 2 + 1] * hB[32 + j];
        if (hD[i * 32 + j] != want) { if (bad < 5) printf("D[%d][%d] = %g, want %g\n", i, j, hD[i * 32 + j], want); ++bad; }
    }
    printf("mfma_f32_32x32x2f32 layout: %s (%d mismatches)\n", bad ? "WRONG" : "as assumed", bad);
    return bad != 0;
}
