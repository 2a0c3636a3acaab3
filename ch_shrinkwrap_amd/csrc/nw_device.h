// Device-side helpers shared by the NanoWrap kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NW_BLOCK 256
#define NW_WAVE 64

// wide loads / stores at 4-byte alignment (gfx950 global memory instructions take any dword-aligned address; a row of S is 36 bytes)
typedef float nw_f4u __attribute__((ext_vector_type(4), aligned(4)));        // 16-byte load at 4-byte alignment (a row of S is 36 bytes)
typedef float nw_f3u __attribute__((ext_vector_type(3), aligned(4)));
typedef int nw_i3u __attribute__((ext_vector_type(3), aligned(4)));

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

// ---- wave / block reductions ---------------------------------------------------------------------------------------------------------
// Wave-wide sum of a double on the VECTOR ALU alone: a butterfly of DPP moves inside each row of 16 lanes (quad permutes, row_half_mirror,
// row_mirror), then v_permlane16_swap / v_permlane32_swap (gfx950) between the rows.  __shfl_xor compiles to ds_bpermute_b32 -- an
// instruction of the LDS pipe, which a CU's four SIMDs share, twelve per double: the fourteen sums of k_prior_directions were ~170 of them
// per wave and bounded that kernel once its gathers had gone (round 5: 14.2 -> 7 us).  Every lane ends with the same total; the
// association is a fixed tree, so the sums stay deterministic.
__device__ __forceinline__ double nw_dpp_xadd(double v, int ctrl_id)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (ctrl_id) {
    case 0: lo = __builtin_amdgcn_mov_dpp(lo, 0xb1, 0xf, 0xf, false); hi = __builtin_amdgcn_mov_dpp(hi, 0xb1, 0xf, 0xf, false); break;       // quad_perm [1,0,3,2]
    case 1: lo = __builtin_amdgcn_mov_dpp(lo, 0x4e, 0xf, 0xf, false); hi = __builtin_amdgcn_mov_dpp(hi, 0x4e, 0xf, 0xf, false); break;       // quad_perm [2,3,0,1]
    case 2: lo = __builtin_amdgcn_mov_dpp(lo, 0x141, 0xf, 0xf, false); hi = __builtin_amdgcn_mov_dpp(hi, 0x141, 0xf, 0xf, false); break;     // row_half_mirror
    default: lo = __builtin_amdgcn_mov_dpp(lo, 0x140, 0xf, 0xf, false); hi = __builtin_amdgcn_mov_dpp(hi, 0x140, 0xf, 0xf, false); break;    // row_mirror
    }
    return v + __hiloint2double(hi, lo);
}

__device__ __forceinline__ double nw_wave_sum(double v)
{
    v = nw_dpp_xadd(v, 0);
    v = nw_dpp_xadd(v, 1);
    v = nw_dpp_xadd(v, 2);
    v = nw_dpp_xadd(v, 3);
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
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

// The workgroup's NV sums: wave sums on the vector ALU, 4 NV doubles through LDS, the row stored by wave 0 (k_reduce_scalars adds the rows of
// all workgroups in a fixed order).  s_w: NV * 4 doubles.  All threads call it.
template <int NV>
__device__ __forceinline__ void nw_block_sums_store(double (&v)[NV], double *__restrict__ part, double *s_w)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double s = nw_wave_sum(v[k]);
        if (lane == 0) s_w[k * 4 + wv] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV)
        part[(int64_t)blockIdx.x * NV + threadIdx.x] = (s_w[threadIdx.x * 4 + 0] + s_w[threadIdx.x * 4 + 1]) + (s_w[threadIdx.x * 4 + 2] + s_w[threadIdx.x * 4 + 3]);
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

// ---- the ring half of the curvature prior ----------------------------------------------------------------------------------------------
// _ncc() (mesh_conj_grad.py:770-820) needs, per vertex, the 1-ring centroid (float32 sum in slot order, :782-788) and the normal-consistency
// sum (:796-800) -- both functions of the MESH positions and the block-stale normals alone, i.e. known before the iteration's query and
// scatter have run; only the gate min(pi^2, 1) (:807-814) needs this iteration's A^T 1.  The ring half is therefore computed apart
// (round 5) by workgroups APPENDED to the query launch's grid (k_nn_wave): they are dispatched when the query's own workgroups have all
// started, i.e. they run in the launch's long drain, on compute units the query no longer fills, and the 1-ring gathers (20 x 4 + ~6 x 24
// bytes a vertex) leave the iteration's critical path.  k_prior_directions then streams {ring sums, accumulator, estimate, own normal}.
// Output per vertex: ring4 = {sum x, sum y, sum z (float32), number of neighbours (int bits)}, ring_a = the normal-consistency sum (f64).
struct NwRingArgs {
    int M, maxdeg;
    const int *nbr_t;                 // slot-major 1-ring table [NB][M]
    const float *meshpos, *nrm;
    float4 *ring4;
    double *ring_a;
};

// one vertex; the arithmetic and its order are exactly the former first half of k_prior_directions.  Written for FEW registers (it shares
// a kernel with the query, whose 80 VGPRs it must not exceed): the ring is walked four slots at a time, positions are fetched again for
// the second pass (they are in the CU's cache by then).
__device__ __forceinline__ void nw_prior_ring_vertex(const NwRingArgs &R, int v)
{
    const int M = R.M, maxdeg = R.maxdeg;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    int ms = 0;
    for (int s0 = 0; s0 < maxdeg; s0 += 4) {
        int nb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) nb[s] = (s0 + s < maxdeg) ? R.nbr_t[(int64_t)(s0 + s) * M + v] : -1;
        float q[4][3];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int n = nb[s] >= 0 ? nb[s] : v;
            q[s][0] = R.meshpos[3 * n]; q[s][1] = R.meshpos[3 * n + 1]; q[s][2] = R.meshpos[3 * n + 2];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (nb[s] >= 0) { sx += q[s][0]; sy += q[s][1]; sz += q[s][2]; ++ms; }
    }
    double asum = 0.0;
    if (ms > 0) {
        const double vcx = (double)sx / ms, vcy = (double)sy / ms, vcz = (double)sz / ms;
        const float Nx = R.nrm[3 * v], Ny = R.nrm[3 * v + 1], Nz = R.nrm[3 * v + 2];
        for (int s0 = 0; s0 < maxdeg; s0 += 4) {
            int nb[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) nb[s] = (s0 + s < maxdeg) ? R.nbr_t[(int64_t)(s0 + s) * M + v] : -1;
            float q[4][3], u[4][3];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int n = nb[s] >= 0 ? nb[s] : v;
                q[s][0] = R.meshpos[3 * n]; q[s][1] = R.meshpos[3 * n + 1]; q[s][2] = R.meshpos[3 * n + 2];
                u[s][0] = R.nrm[3 * n]; u[s][1] = R.nrm[3 * n + 1]; u[s][2] = R.nrm[3 * n + 2];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (nb[s] >= 0) {
                    const double cnx = (double)q[s][0] - vcx, cny = (double)q[s][1] - vcy, cnz = (double)q[s][2] - vcz;
                    const double cdot = (cnx * (double)u[s][0] + cny * (double)u[s][1]) + cnz * (double)u[s][2];
                    const float ndn = (u[s][0] * Nx + u[s][1] * Ny) + u[s][2] * Nz;
                    const float den = sqrtf(2.0f * (fmaxf(ndn, 0.0f) + 1.0f));
                    asum += cdot / (double)den;
                }
            }
        }
    }
    R.ring4[v] = make_float4(sx, sy, sz, __int_as_float(ms));
    R.ring_a[v] = asum;
}
