// The per-localization attraction step of the iteration (k_attract, nw_kernels.h), kept apart from the kernel's workgroup plumbing: the
// step was also tried in the TAIL of the nearest-face query (a wave that has just found its 64 localizations' faces going straight on to
// their weights, residuals and scatter through a table of its own): 163 us for the combined launch against 129 + 38 apart -- the step's
// dependent gathers lengthen every latency-bound wave of the query instead of filling its idle issue slots (tools/experiments/r03_notes.md).
#pragma once
#include "nw_device.h"

// K4b: weight matrix row, A f, weighted + de-weighted residual, and the A^T scatter -- one thread per
// localization (cell-sorted order, so the three vertices of neighbouring threads sit in the same L2 lines).
//   v_idx = faces[face]; d_j = |f[v_j] - p| (f32); w_j = 1/max(d_j,1e-6), row-normalised    mesh_conj_grad.py:488-510
//   Af = sum_j f[v_j] w_j (f32, corner order)                                                :544-545
//   res = weights*(p - Af); res *= 1/(d*sigma_inv/2 + 1) (float64 factor, float32 store)      :222,231,248
//   vacc[v_j] += {w_j*res, w_j}  -> S0 = A^T res and A^T 1 in ONE pass                        :253, conj_grad_utils.c:153-162,
//                                                                                            _membrane_mesh.pyx:1633
// Scatter: the localizations are brick-sorted, so the 256 points of a workgroup reference only a few hundred distinct
// vertices.  Contributions are first summed per vertex in an LDS hash table, then flushed with ONE global float atomic
// per (vertex, component), four adjacent lanes covering the vertex's contiguous float4, i.e. one memory-side atomic
// request per touched vertex instead of twelve per point (MI355X_MICROARCH.md "Global float atomics": scattered
// single-dword atomics run ~17x below the contiguous rate).
// The accumulators are 64-bit FIXED POINT, in LDS (ds_add_u64) and in HBM (`vacc`, global_atomic_add_x2), not float: measured on
// MI355X, the twelve ds_add_f32 per point cost 67 us per launch at 1M points (LDS float atomics retire ~1 lane every 3 cycles
// per CU), twelve ds_add_u64 25 us.  The quanta are powers of two (scaling a float32 product by one is exact): for {w res} 2^-36
// of (cloud extent x largest weight), a bound on |res| -- ~10^-11 relative resolution, far below the float32 rounding of each
// product, 2^27 such terms of headroom; for {w} 2^-40 (w <= 1).  Integer addition is associative, so the sums are EXACT and
// independent of the order in which lanes, waves and workgroups arrive: the scatter is bitwise reproducible (the reference's is a
// serial, deterministic loop, conj_grad_utils.c:153-162); k_prior_directions rounds each sum once to float32.
// round-to-nearest-even of |x| < 2^51 to a 64-bit integer: adding 1.5 * 2^52 leaves the integer in the low mantissa bits (one f64 add
// and a 64-bit subtract instead of the dozen instructions of the f64 -> i64 conversion sequence)
__device__ __forceinline__ long long nw_round_to_i64(double x)
{
    return __double_as_longlong(x + 6755399441055744.0) - 0x4338000000000000LL;
}


// everything the step reads and writes (passed by value: kernel arguments are preloaded into scalar registers; fetched from device
// memory at the start of every workgroup the same struct cost k_attract 3.5 us)
struct NwAttractArgs {
    int F;
    float sinv_scalar, w_scalar;
    const float4 *cent_by_face;
    const int *faces;
    const float *pos, *sinv, *wnorm, *dat;      // dat: target of the residual if it is not the localizations (nw_set_data), else NULL
    const unsigned char *mask;
    float *dist, *wout, *res;
    int *vidx;
    long long *vacc;
    double *part;                                // rows of 5: sum res^2, masked sum res^2, sum d, count, max d
    double inv_q, inv_qw;
    const float *d0;                             // sharded mesh with per-localization halos (nw_halo_set_reference): the nearest distances when the shares
    const int *perm;                             // were cut, caller order (perm: sorted slot -> caller index); the "max d" column then holds max (d - d0)
};

// One localization i (position P, nearest face f in [0, F)): stores its row of the weight matrix, residual and distance, adds its
// twelve contributions {w_j res, w_j} to the LDS table of SLOTS = 2^BITS slots (s_key[SLOTS], s_val[4][SLOTS], component-major; a
// contribution that finds no slot within PROBES probes goes to HBM directly -- integer sums: the path taken does not change the result)
// and its share to the caller's partial sums.  `rot` in {0,1,2} rotates the corner order (neighbouring lanes mostly share their
// face: three different accumulators at a time instead of one three times over).
__device__ __forceinline__ void nw_attract_point(const NwAttractArgs &A, int i, const float4 P, int f, int (&vout)[3], unsigned long long (&q)[3][4],
                                                 double (&red)[4], float &dmax, bool &bad)
{
    const float p[3] = {P.x, P.y, P.z};
    int v[3];
    float w[3], fv[3][3];
    float wsum;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        v[j] = A.faces[3 * f + j];
        float dd = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            fv[j][k] = A.pos[3 * v[j] + k];
            const float d = fv[j][k] - p[k];
            const float sq = d * d;
            dd = (k == 0) ? sq : dd + sq;
        }
        w[j] = 1.0f / fmaxf(sqrtf(dd), 1e-6f);
    }
    wsum = (w[0] + w[1]) + w[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) { w[j] = w[j] / wsum; bad |= isnan(w[j]); }
    // dmean: float64 Euclidean distance to the nearest centroid (what cKDTree returns), rounded once to float32
    float d;
    {
        const float4 C = A.cent_by_face[f];
        const double ddx = (double)P.x - (double)C.x, ddy = (double)P.y - (double)C.y, ddz = (double)P.z - (double)C.z;
        d = (float)sqrt(fma(ddz, ddz, fma(ddy, ddy, ddx * ddx)));
        A.dist[i] = d;
    }
    const unsigned m = A.mask[i];
    float r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float af = 0.0f + fv[0][k] * w[0];
        af = af + fv[1][k] * w[1];
        af = af + fv[2][k] * w[2];
        bad |= isnan(af);
        const float wt = A.wnorm ? A.wnorm[3 * i + k] : A.w_scalar;
        const float r0 = wt * ((A.dat ? A.dat[3 * i + k] : p[k]) - af);
        const double si = A.sinv ? (double)A.sinv[3 * i + k] : (double)A.sinv_scalar;
        const double wd = 1.0 / ((double)d * si / 2.0 + 1.0);
        r[k] = (float)((double)r0 * wd);
        const double r2 = (double)r[k] * (double)r[k];
        red[0] += r2;
        if (m & (1u << k)) red[1] += r2;
    }
    red[2] += (double)d;
    red[3] += 1.0;
    dmax = fmaxf(dmax, A.d0 ? d - A.d0[A.perm[i]] : d);
    __builtin_memcpy(A.res + 3 * (int64_t)i, r, 12);           // three 12-byte stores instead of nine dword stores
    __builtin_memcpy(A.vidx + 3 * (int64_t)i, v, 12);
    __builtin_memcpy(A.wout + 3 * (int64_t)i, w, 12);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        vout[j] = v[j];
        const float c[4] = {w[j] * r[0], w[j] * r[1], w[j] * r[2], w[j]};     // float32 products, as the reference forms them
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double x = (double)c[k] * (k < 3 ? A.inv_q : A.inv_qw);           // exact scaling (powers of two)
            bad |= !(fabs(x) < 7.0e13);                                             // far beyond the bound behind the quantum / inf / NaN: raise the NaN status
            q[j][k] = (unsigned long long)nw_round_to_i64(x);
        }
    }
}

// Runs of lanes with the same nearest face (the localizations are sorted by their foot point: 2.6 consecutive lanes share a face on
// average at 2.5 localizations per face, a wave's 64 localizations touch ~25 distinct vertices) are added up on the vector ALU before
// anything goes to the LDS table (round 5): a segmented inclusive scan over the rows of 16 lanes (DPP row_shr 1, 2, 4, 8), after which
// the LAST lane of every run holds the run's twelve sums and is the only one that probes the table and issues ds_add_u64.  The sums are
// integers, so the regrouping does not change a bit of the result.  Call from uniform control flow (every lane of the wave): `run` is the
// lane's face, or any value unique in its row for a lane without a localization (whose q must be zero).  Returns whether the lane is
// the last of its run.
__device__ __forceinline__ bool nw_run_sums(unsigned long long (&q)[3][4], int run)
{
    const int lane = threadIdx.x & 63;
    // head: first lane of its row of 16, or a different run than the lane before
    const int prev = __builtin_amdgcn_update_dpp(~run, run, 0x111, 0xf, 0xf, false);        // row_shr:1 (lanes without a source keep ~run: a head)
    int flag = (prev != run) ? 1 : 0;
    const int next_head = __builtin_amdgcn_update_dpp(1, flag, 0x101, 0xf, 0xf, false);    // row_shl:1 (the row's last lane: 1)
#pragma unroll
    for (int step = 0; step < 4; ++step) {
        const int ctrl = step == 0 ? 0x111 : (step == 1 ? 0x112 : (step == 2 ? 0x114 : 0x118));     // row_shr:1/2/4/8
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int lo = (int)(unsigned)(q[j][k] & 0xffffffffull), hi = (int)(unsigned)(q[j][k] >> 32);
                int tlo, thi;
                if (step == 0) { tlo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true); thi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true); }
                else if (step == 1) { tlo = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xf, 0xf, true); thi = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xf, 0xf, true); }
                else if (step == 2) { tlo = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xf, 0xf, true); thi = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xf, true); }
                else { tlo = __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xf, 0xf, true); thi = __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xf, 0xf, true); }
                const unsigned long long t = ((unsigned long long)(unsigned)thi << 32) | (unsigned long long)(unsigned)tlo;
                q[j][k] += flag ? 0ull : t;
            }
        int tf;
        if (step == 0) tf = __builtin_amdgcn_update_dpp(1, flag, 0x111, 0xf, 0xf, false);
        else if (step == 1) tf = __builtin_amdgcn_update_dpp(1, flag, 0x112, 0xf, 0xf, false);
        else if (step == 2) tf = __builtin_amdgcn_update_dpp(1, flag, 0x114, 0xf, 0xf, false);
        else tf = __builtin_amdgcn_update_dpp(1, flag, 0x118, 0xf, 0xf, false);
        flag |= tf;
        (void)ctrl;
    }
    (void)lane;
    return next_head != 0;
}

// the run's sums into the workgroup's table (the last lane of a run only)
// (The corner order was rotated per lane while every lane scattered -- neighbouring lanes shared their face; the last lanes of different
// runs do not, and a run-time corner index would put q[][] into scratch memory: the build's resource guard refuses that.)
template <int SLOTS, int BITS, int PROBES>
__device__ __forceinline__ void nw_attract_scatter(const NwAttractArgs &A, const int (&v)[3], const unsigned long long (&q)[3][4], int *s_key, unsigned long long *s_val)
{
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int vj = v[j];
        unsigned hsh = ((unsigned)vj * 2654435761u) >> (32 - BITS);
        bool slot = false;
        for (int probe = 0; probe < PROBES; ++probe) {
            const int old = atomicCAS(&s_key[hsh], -1, vj);
            if (old == -1 || old == vj) { slot = true; break; }
            hsh = (hsh + 1) & (SLOTS - 1);
        }
        unsigned long long *a = s_val + hsh;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long x = q[j][k];
            if (slot) atomicAdd(a + k * SLOTS, x);
            else atomicAdd(reinterpret_cast<unsigned long long *>(A.vacc) + 4 * (int64_t)vj + k, x);
        }
    }
}

// The attraction step of one workgroup's localizations when the caller is NOT k_attract: the workgroups k_nn_wave appends to its grid
// (round 5) -- one wave per work item of the query, the item's localizations one per lane.  The caller has initialised the table
// (s_key = -1, s_val = 0) and passed a barrier.  `ok`: the lane has a localization i with a face f_raw inside [0, F).  The workgroup's
// row of partial sums goes to A.part[row] (k_reduce_scalars adds the rows).  blockDim.x = 64 * (waves), at most 4 waves.
template <int HT, int HT_BITS, int PROBES>
__device__ __forceinline__ void nw_attract_workgroup(const NwAttractArgs &A, const float4 *__restrict__ pts, int i, bool ok, int f_raw, NwDevState *__restrict__ st,
                                                     int *s_key, unsigned long long *s_val, double *s_w /* [20] */, int row)
{
    double red[4] = {0.0, 0.0, 0.0, 0.0};
    float dmax = 0.0f;
    unsigned long long q[3][4];
    int v[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) q[j][k] = 0ull;
    if (ok) {
        bool bad = false;
        nw_attract_point(A, i, pts[i], f_raw, v, q, red, dmax, bad);
        if (bad) atomicCAS(&st->status, 0, -3 /* NW_ERR_NAN */);
    }
    const bool last_of_run = nw_run_sums(q, ok ? f_raw : -2 - (int)(threadIdx.x & 15));
    if (ok && last_of_run) nw_attract_scatter<HT, HT_BITS, PROBES>(A, v, q, s_key, s_val);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < 20) s_w[threadIdx.x] = 0.0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double sk = nw_wave_sum(red[k]);
        if (lane == 0) s_w[k * 4 + wv] = sk;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off, 64));
    if (lane == 0) s_w[16 + wv] = (double)dmax;
    __syncthreads();
    if (threadIdx.x < 4) A.part[(int64_t)row * 5 + threadIdx.x] = (s_w[threadIdx.x * 4 + 0] + s_w[threadIdx.x * 4 + 1]) + (s_w[threadIdx.x * 4 + 2] + s_w[threadIdx.x * 4 + 3]);
    else if (threadIdx.x == 4) A.part[(int64_t)row * 5 + 4] = fmax(fmax(s_w[16], s_w[17]), fmax(s_w[18], s_w[19]));
    for (int t = threadIdx.x; t < HT * 4; t += (int)blockDim.x) {
        const int key = s_key[t >> 2];                     // four adjacent lanes flush the four components of one vertex
        if (key >= 0) atomicAdd(reinterpret_cast<unsigned long long *>(A.vacc) + 4 * (int64_t)key + (t & 3), s_val[(t & 3) * HT + (t >> 2)]);
    }
}
