// Wave-wide reductions and the inclusive scan on the vector ALU alone (DPP + v_permlane*_swap, gfx950) against a host loop.
// hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o wr wave_reduce_dpp.hip && ./wr
#include <hip/hip_runtime.h>
template <class Op> __device__ __forceinline__ int allreduce_i(int x, Op op)
{
    x = op(x, __builtin_amdgcn_mov_dpp(x, 0xb1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
    x = op(x, __builtin_amdgcn_mov_dpp(x, 0x4e, 0xf, 0xf, false));      // quad_perm [2,3,0,1]
    x = op(x, __builtin_amdgcn_mov_dpp(x, 0x141, 0xf, 0xf, false));     // row_half_mirror
    x = op(x, __builtin_amdgcn_mov_dpp(x, 0x140, 0xf, 0xf, false));     // row_mirror
    auto a = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
    x = op((int)a[0], (int)a[1]);
    auto b = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
    x = op((int)b[0], (int)b[1]);
    return x;
}
__device__ __forceinline__ int incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
    return v;
}
__global__ void k2(int *out, const int *in) { out[threadIdx.x] = incl_scan(in[threadIdx.x]); }
__global__ void k(int *out, const int *in)
{
    int v = in[threadIdx.x];
    out[threadIdx.x] = allreduce_i(v, [](int a, int b) { return a < b ? a : b; });
    out[64 + threadIdx.x] = allreduce_i(v, [](int a, int b) { return a + b; });
}
int main()
{
    int h[64], *di, *dout, ho[128];
    for (int i = 0; i < 64; ++i) h[i] = (i * 37 + 11) % 101 - 50;
    hipMalloc(&di, 256); hipMalloc(&dout, 512);
    hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, di);
    hipMemcpy(ho, dout, 512, hipMemcpyDeviceToHost);
    int mn = 1 << 30, sm = 0; for (int i = 0; i < 64; ++i) { mn = h[i] < mn ? h[i] : mn; sm += h[i]; }
    int bad = 0; for (int i = 0; i < 64; ++i) bad += (ho[i] != mn) + (ho[64 + i] != sm);
    hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, dout, di);
    hipMemcpy(ho, dout, 256, hipMemcpyDeviceToHost);
    int run = 0, badscan = 0; for (int i = 0; i < 64; ++i) { run += h[i]; badscan += ho[i] != run; }
    printf("min %d sum %d bad %d, inclusive scan bad %d\n", mn, sm, bad, badscan);
    return bad != 0 || badscan != 0;
}
