// Device-side helpers shared by the NanoWrap kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NW_BLOCK 256
#define NW_WAVE 64

// ---- uniform grid ------------------------------------------------------------------------------------------
// Face centroids are binned into the cells of edge h of one dense lattice (x fastest: a (z,y) row of cells is one contiguous
// candidate range of the cell-sorted centroid array).  The localizations are NOT binned: they are Morton-sorted once (nw_nn.h).
struct NwGrid {
    float ox, oy, oz;      // origin (lower corner)
    float h, inv_h;        // cell edge
    float eps;             // safety slack for the rounding of the cell assignment
    int gx, gy, gz;        // cells per axis
    int ncell;
};

__device__ __forceinline__ int nw_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ void nw_cell_coords(const NwGrid &g, float x, float y, float z, int &ix, int &iy, int &iz)
{
    ix = nw_clampi((int)floorf((x - g.ox) * g.inv_h), 0, g.gx - 1);
    iy = nw_clampi((int)floorf((y - g.oy) * g.inv_h), 0, g.gy - 1);
    iz = nw_clampi((int)floorf((z - g.oz) * g.inv_h), 0, g.gz - 1);
}

__device__ __forceinline__ int nw_cell_index(const NwGrid &g, int ix, int iy, int iz)
{
    return ix + g.gx * (iy + g.gy * iz);
}

// ---- wave / block reductions (wave64 shuffles; no LDS for the intra-wave part) ------------------------------
__device__ __forceinline__ double nw_wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ int nw_wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// Normal-equation sums are DETERMINISTIC: every workgroup reduces its NV partial sums in a fixed order (wave shuffles, then the
// four waves in order) and stores them as plain doubles into its own row part[blockIdx][0..NV); k_reduce_scalars then adds the rows
// in a fixed order.  No floating-point atomics: two runs of the same inputs give bit-identical sums, whatever the order in which
// the workgroups happen to finish.
template <int NV>
__device__ __forceinline__ void nw_block_reduce_store(double (&v)[NV], double *__restrict__ part, double *s_part /* [NV*4] */)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double s = nw_wave_sum(v[k]);
        if (lane == 0) s_part[k * 4 + wv] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV)
        part[(int64_t)blockIdx.x * NV + threadIdx.x] = (s_part[threadIdx.x * 4 + 0] + s_part[threadIdx.x * 4 + 1]) + (s_part[threadIdx.x * 4 + 2] + s_part[threadIdx.x * 4 + 3]);
}

// The same sums through LDS MEMORY: every thread parks its NV values, wave 0 adds the four values of each lane position (one per
// wave, fixed order) and does the only wave reduction.  The shuffles of a wave reduction are LDS-pipe instructions (12 per double),
// and that pipe is shared by the CU's four SIMDs: a quarter of them is what is left.  s_red: NV * 256 doubles.
template <int NV>
__device__ __forceinline__ void nw_block_reduce_store_lds(double (&v)[NV], double *__restrict__ part, double *s_red)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < NV; ++k) s_red[k * NW_BLOCK + tid] = v[k];
    __syncthreads();
    if (tid < 64) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const double *c = s_red + k * NW_BLOCK + tid;
            const double s = nw_wave_sum((c[0] + c[64]) + (c[128] + c[192]));
            if (tid == 0) part[(int64_t)blockIdx.x * NV + k] = s;
        }
    }
}

// plain (non-replicated) variant with float64 atomics for the set-up reductions (mesh area, weight sums): not on the iteration path
template <int NV>
__device__ __forceinline__ void nw_block_reduce_atomic(double (&v)[NV], double *out, double *s_part /* [NV*4] */)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double s = nw_wave_sum(v[k]);
        if (lane == 0) s_part[k * 4 + wv] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV)
        atomicAdd(out + threadIdx.x, (s_part[threadIdx.x * 4 + 0] + s_part[threadIdx.x * 4 + 1]) + (s_part[threadIdx.x * 4 + 2] + s_part[threadIdx.x * 4 + 3]));
    __syncthreads();
}

// XCD-aware remap of a 1-D grid: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 names the
// group that shares an L2), so hand each XCD one CONTIGUOUS chunk of the work list: neighbouring work items
// (spatially adjacent grid cells, consecutive vertices) then share candidates / gathered rows in one L2.
// Launch 8*ceil(n/8) blocks; returns -1 for the padding blocks.  Speed only, never correctness.
__device__ __forceinline__ int nw_xcd_remap(int b, int n)
{
    const int chunk = (n + 7) >> 3;
    const int i = (b & 7) * chunk + (b >> 3);
    return i < n ? i : -1;
}
