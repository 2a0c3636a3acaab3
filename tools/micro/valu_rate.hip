// Micro-benchmark: sustained VALU issue rate on MI355X for the instruction mix of the NN inner loop.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_rate tools/micro/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float sx, float sy)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // 8 independent v_fma_f32 per round x 8
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sx), "v"(sy));
            }
        } else if (MODE == 1) {   // v_pk_fma_f32 on register pairs
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6) : "v"(*(double *)&sx), "v"(*(double *)&sy));
            }
        } else if (MODE == 2) {   // min / med3 / cmp+cndmask mix
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_med3_f32 %0, %0, %1, %2\n v_min_f32 %1, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %3, %3, %4, vcc\n"
                             "v_med3_f32 %4, %4, %5, %6\n v_min_f32 %5, %5, %6\n v_cmp_lt_f32 vcc, %6, %7\n v_cndmask_b32 %7, %7, %0, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            }
        } else if (MODE == 3) {   // fma with an SGPR operand
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                             "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sx));
            }
        } else if (MODE == 4) {   // v_sub / v_mul / v_add plain VOP2
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_sub_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sx));
            }
        } else if (MODE == 5) {   // v_pk_add_f32 with an SGPR pair operand (as the compiler emits for two candidates)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             "v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3\n"
                             : "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6) : "s"(*(double *)&sx));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
void run(const char *name, float *d, int blocks_per_cu)
{
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks_per_cu waves per SIMD (4 waves per block, one per SIMD), 64 instr per iteration
    const double instr_per_simd = (double)blocks_per_cu * iters * 64.0;
    printf("%-28s waves/SIMD %d : %.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}

int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32 (vgpr)", d, w);
        run<3>("v_fma_f32 (sgpr operand)", d, w);
        run<4>("v_sub/mul/add VOP2", d, w);
        run<1>("v_pk_fma_f32", d, w);
        run<5>("v_pk_add(sgpr)/v_pk_mul", d, w);
        run<2>("med3/min/cmp/cndmask", d, w);
    }
    return 0;
}
