#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
// hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o /tmp/wsum tools/micro/wave_sum_f64_dpp.hip && /tmp/wsum
#include "../../ch_shrinkwrap_amd/csrc/nw_device.h"
__global__ void k(double *out, const double *in) { out[threadIdx.x] = nw_wave_sum(in[threadIdx.x]); }
int main()
{
    double h[64], ho[64], *di, *dout;
    for (int i = 0; i < 64; ++i) h[i] = std::sin(i * 1.7) * 1e3 + i;
    hipMalloc(&di, 512); hipMalloc(&dout, 512);
    hipMemcpy(di, h, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, di);
    hipMemcpy(ho, dout, 512, hipMemcpyDeviceToHost);
    double sm = 0; for (int i = 0; i < 64; ++i) sm += h[i];
    int bad = 0; for (int i = 0; i < 64; ++i) bad += (fabs(ho[i] - sm) > 1e-9 * fabs(sm)) + (ho[i] != ho[0]);
    printf("sum %.12g got %.12g bad %d\n", sm, ho[0], bad);
    return bad != 0;
}
