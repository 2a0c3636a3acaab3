// NanoWrap inner-loop kernels for MI355X (gfx950).  HBM/latency-bound irregular work: no MFMA anywhere.
// Reference arithmetic being reproduced is cited per kernel (paths relative to /root/reference/ch_shrinkwrap/).
#pragma once
#include "nw_device.h"

// ---- scalar slots (double): final sums of one iteration, written by k_reduce_scalars ---------------------------------
enum {
    // point-side partial sums (all-reduced across ranks in multi-GPU runs)
    SC_RES2 = 0,     // sum res^2            -> ress log          mesh_conj_grad.py:270
    SC_C0 = 1,       // sum_mask res^2       -> c0                conj_grad.py:189
    SC_SUMD = 2,     // sum of NN distances  -> grid cell-size control
    SC_NPTS = 3,     // number of points contributing
    SC_HC = 4,       // 6: AS^T AS (00,01,02,11,12,22)            conj_grad.py:202
    SC_GC = 10,      // 3: AS^T res                               conj_grad.py:203
    SC_STATUS = 13,  // > 0: a status was raised in this iteration (on this rank, or -- summed over the ranks -- on any): every rank stops with it
    SC_NPOINT = 14,
    // vertex-side sums (replicated on every rank)
    SC_SS = 14,      // 6: S^T S = Hw                             conj_grad.py:211
    SC_SP = 20,      // 3: S_k . prefs64 (Gw = -SP)               conj_grad.py:212
    SC_PP64 = 23,    // sum prefs64^2 -> wpreds                   conj_grad.py:192
    SC_PP32 = 24,    // sum prefs32^2 -> prefs log                mesh_conj_grad.py:271
    SC_T = 25,       // 3: S0.S0, S0.S1, S1.S1 of the RAW directions -> test statistic (mesh_conj_grad.py:262-265); equal to SC_SS
                     //    entries 0, 1, 3 unless a regulariser other than the identity scales the directions (wfunc)
    SC_MAXD = 28,    // largest NN distance (a MAX, not a sum: k_reduce_scalars / k_solve_update treat this slot accordingly)
    SC_COUNT = 29
};

#define NW_SPARTS 32      // ordered partial sums per scalar slot (k_reduce_scalars)

struct NwDevState {
    int stop_at;          // first iteration index (global, per ctx) that must not execute (stop condition)
    int ntests;           // number of test statistics recorded so far (history survives search() calls)
    float tests[3];       // last three test statistics, oldest first
    int status;           // sticky nw_status raised on the device
    int nn_max_ring;
    int iter_base;        // global index of the current block's first iteration (k_set_iter_base): kernels get the block-relative index,
                          // so a captured block (hipGraph) can be replayed for later blocks
    int grid_escape;      // a centroid lay more than a cell outside the cell grid in this block (the query stays exact -- the outermost
                          // cells hold what lies beyond them -- but gets slower: the host lays a new grid before the next block)
};

__global__ void k_set_iter_base(NwDevState *st, int base) { st->iter_base = base; st->grid_escape = 0; }

// ============================================================================================================
// generic exclusive scan of int32 counts (3 launches): out[0..n] with out[n] = total
// ============================================================================================================
#define NW_SCAN_TILE 2048   // 256 threads x 8

__global__ __launch_bounds__(NW_BLOCK) void k_scan_tile_sums(const int *__restrict__ in, int n, int *__restrict__ bsum)
{
    __shared__ int s_w[4];
    const int base = blockIdx.x * NW_SCAN_TILE + threadIdx.x * 8;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += (base + k < n) ? in[base + k] : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(1024) void k_scan_bsums(int *__restrict__ bsum, int nb)
{
    // single workgroup: in-place exclusive scan of the tile sums
    __shared__ int s_w[16];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? bsum[i] : 0;
        int inc = nw_wave_incl_scan(v, lane);
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wv; ++w) woff += s_w[w];
        const int carry = s_carry;
        if (i < nb) bsum[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + woff + inc;
        __syncthreads();
    }
}

// bsum_is_raw: `bsum` holds the raw tile sums (k_scan_bsums was skipped); the workgroup adds the sums of the tiles before its own
// itself (tile counts up to a few thousand: one load per thread and pass) -- two launches instead of three.
__global__ __launch_bounds__(NW_BLOCK) void k_scan_final(int *__restrict__ in, int n, const int *__restrict__ bsum, int *__restrict__ out, int zero_input, int bsum_is_raw)
{
    __shared__ int s_w[4];
    __shared__ int s_pre[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int base = blockIdx.x * NW_SCAN_TILE + threadIdx.x * 8;
    int v[8];
    int s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
    if (zero_input)          // the histogram is consumed: leave it zeroed for the next iteration (saves a memset launch)
#pragma unroll
        for (int k = 0; k < 8; ++k) if (base + k < n) in[base + k] = 0;
    const int inc = nw_wave_incl_scan(s, lane);
    if (lane == 63) s_w[wv] = inc;
    int pre = 0;
    if (bsum_is_raw) {
        for (int b = threadIdx.x; b < (int)blockIdx.x; b += NW_BLOCK) pre += bsum[b];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o, 64);
        if (lane == 0) s_pre[wv] = pre;
    }
    __syncthreads();
    int off = (bsum_is_raw ? (s_pre[0] + s_pre[1]) + (s_pre[2] + s_pre[3]) : bsum[blockIdx.x]) + inc - s;
    for (int w = 0; w < wv; ++w) off += s_w[w];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (base + k < n) out[base + k] = off;
        off += v[k];
        if (base + k == n - 1) out[n] = off;
    }
}

// ============================================================================================================
// set-up kernels (once per nw_set_points / grid change)
// ============================================================================================================
__global__ __launch_bounds__(NW_BLOCK) void k_minmax3(const float *__restrict__ xyz, int64_t n, float *__restrict__ mm /* [6]: min xyz, max xyz */, int *__restrict__ nonfinite)
{
    __shared__ float s_lo[3][4], s_hi[3][4];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    int bad = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        for (int k = 0; k < 3; ++k) {
            const float v = xyz[3 * i + k];
            if (!isfinite(v)) bad = 1;
            lo[k] = fminf(lo[k], v);
            hi[k] = fmaxf(hi[k], v);
        }
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
        }
        if ((threadIdx.x & 63) == 0) { s_lo[k][threadIdx.x >> 6] = lo[k]; s_hi[k][threadIdx.x >> 6] = hi[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        // one pair of atomics per workgroup and axis (six same-address atomics per WAVE serialised: 225 us for 12 MB);
        // float atomic min/max through the monotone int mapping
        const int k = threadIdx.x;
        const float l = fminf(fminf(s_lo[k][0], s_lo[k][1]), fminf(s_lo[k][2], s_lo[k][3]));
        const float h = fmaxf(fmaxf(s_hi[k][0], s_hi[k][1]), fmaxf(s_hi[k][2], s_hi[k][3]));
        atomicMin((int *)&mm[k], l >= 0 ? __float_as_int(l) : (int)(0x80000000u - (unsigned)__float_as_int(l)));
        atomicMax((int *)&mm[3 + k], h >= 0 ? __float_as_int(h) : (int)(0x80000000u - (unsigned)__float_as_int(h)));
    }
    if (bad) atomicOr(nonfinite, 1);
}

// count / sum / sum of squares per axis of the points inside a box, on coordinates quantised to 16 bits of the cloud's extent and
// accumulated in integers: the result does not depend on the order of the atomics (it decides the localizations' sort order, and with
// it the order of every ordered sum downstream -- it has to be the same in every run).  out[7]: n, sx sy sz, sxx syy szz.
__global__ __launch_bounds__(NW_BLOCK) void k_box_moments(const float *__restrict__ xyz, int64_t n, float ox, float oy, float oz, float inv_q,
                                                         float lx, float ly, float lz, float hx, float hy, float hz, unsigned long long *__restrict__ out)
{
    __shared__ unsigned long long s_part[7][4];
    unsigned long long v[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        if (x < lx || x > hx || y < ly || y > hy || z < lz || z > hz) continue;
        const unsigned long long qx = (unsigned)nw_clampi((int)((x - ox) * inv_q), 0, 65535), qy = (unsigned)nw_clampi((int)((y - oy) * inv_q), 0, 65535),
                                 qz = (unsigned)nw_clampi((int)((z - oz) * inv_q), 0, 65535);
        v[0] += 1; v[1] += qx; v[2] += qy; v[3] += qz; v[4] += qx * qx; v[5] += qy * qy; v[6] += qz * qz;
    }
    for (int k = 0; k < 7; ++k) {
        unsigned long long t = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        if ((threadIdx.x & 63) == 0) s_part[k][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    if (threadIdx.x < 7) atomicAdd(out + threadIdx.x, s_part[threadIdx.x][0] + s_part[threadIdx.x][1] + s_part[threadIdx.x][2] + s_part[threadIdx.x][3]);
}

// sum of a float array in float64 (weights.mean(), mesh_conj_grad.py:162)
__global__ __launch_bounds__(NW_BLOCK) void k_sum_f64(const float *__restrict__ x, int64_t n, double *__restrict__ out)
{
    __shared__ double s_part[4];
    double s[1] = {0.0};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s[0] += (double)x[i];
    nw_block_reduce_atomic<1>(s, out, s_part);
}

// largest |x| of a float array (order-independent: integer atomicMax on the bits of the non-negative values)
__global__ __launch_bounds__(NW_BLOCK) void k_absmax_f32(const float *__restrict__ x, int64_t n, int *__restrict__ out)
{
    float m = 0.0f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = fabsf(x[i]);
        m = (v > m && v < INFINITY) ? v : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_int(m));
}

// total mesh area (sum over faces of |cross|/2) in float64: sets the centroid spacing for the grid cell size
__global__ __launch_bounds__(NW_BLOCK) void k_mesh_area(const float *__restrict__ pos, const int *__restrict__ faces, int F, double *__restrict__ out)
{
    __shared__ double s_part[4];
    double s[1] = {0.0};
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < F; f += gridDim.x * blockDim.x) {
        const int a = faces[3 * f], b = faces[3 * f + 1], c = faces[3 * f + 2];
        const float ax = pos[3 * a], ay = pos[3 * a + 1], az = pos[3 * a + 2];
        const float ux = pos[3 * b] - ax, uy = pos[3 * b + 1] - ay, uz = pos[3 * b + 2] - az;
        const float vx = pos[3 * c] - ax, vy = pos[3 * c + 1] - ay, vz = pos[3 * c + 2] - az;
        const float cx = uy * vz - uz * vy, cy = uz * vx - ux * vz, cz = ux * vy - uy * vx;
        s[0] += 0.5 * sqrt((double)cx * cx + (double)cy * cy + (double)cz * cz);
    }
    nw_block_reduce_atomic<1>(s, out, s_part);
}

// brute-force NN distance of a strided sample of the points against ALL face centroids (calibration of the
// grid cell size before the first query): one workgroup per sample point.
__global__ __launch_bounds__(NW_BLOCK) void k_sample_nn(const float *__restrict__ pts_xyz, int64_t N, int nsample, const float *__restrict__ pos,
                                                       const int *__restrict__ faces, int F, double *__restrict__ out_sum)
{
    __shared__ float s_min[4];
    const int64_t i = (int64_t)blockIdx.x * (N / nsample);
    const float px = pts_xyz[3 * i], py = pts_xyz[3 * i + 1], pz = pts_xyz[3 * i + 2];
    float best = INFINITY;
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        const int a = faces[3 * f], b = faces[3 * f + 1], c = faces[3 * f + 2];
        const float cx = (pos[3 * a] + pos[3 * b] + pos[3 * c]) * (1.0f / 3.0f) - px;
        const float cy = (pos[3 * a + 1] + pos[3 * b + 1] + pos[3 * c + 1]) * (1.0f / 3.0f) - py;
        const float cz = (pos[3 * a + 2] + pos[3 * b + 2] + pos[3 * c + 2]) * (1.0f / 3.0f) - pz;
        best = fminf(best, cx * cx + cy * cy + cz * cz);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = fminf(best, __shfl_xor(best, off, 64));
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out_sum, (double)sqrtf(fminf(fminf(s_min[0], s_min[1]), fminf(s_min[2], s_min[3]))));
}

// 1-ring table from the faces array alone, pass 1: every face corner (a -> b, then c, counter-clockwise)
// deposits {b, c, half-edge id 3f+k} in a slot of its vertex a.  err[0] counts vertices whose degree exceeds the table width.
// face_orig (may be NULL): the faces array is in the library's internal order (nw_set_mesh sorts it); the half-edge id that decides where a
// fan starts is the CALLER's, 3 * face_orig[f] + k.  The fourth component keeps the internal face (its area: k_ring_order's face_out).
__global__ void k_ring_collect(const int *__restrict__ faces, int F, int M, int NB, int *__restrict__ cnt, int4 *__restrict__ pairs, int *__restrict__ err,
                               const int *__restrict__ face_orig)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * F) return;
    const int f = i / 3, k = i - 3 * f;
    const int a = faces[3 * f + k], b = faces[3 * f + (k + 1) % 3], c = faces[3 * f + (k + 2) % 3];
    if ((unsigned)a >= (unsigned)M || (unsigned)b >= (unsigned)M || (unsigned)c >= (unsigned)M) { atomicAdd(err + 1, 1); return; }
    const int slot = atomicAdd(&cnt[a], 1);
    if (slot < NB) pairs[(int64_t)a * NB + slot] = make_int4(b, c, face_orig ? 3 * face_orig[f] + k : i, f);
    else atomicAdd(err, 1);
}

// pass 2: order each vertex's fan exactly as the host substrate does (trimesh.TriMesh._build_rings): start at the
// lowest-numbered outgoing half-edge -- for a boundary vertex at the outgoing half-edge that has no twin -- and walk
// counter-clockwise (the neighbour after b is the third corner c of the face that holds a -> b) until the fan closes or ends.
// Open fans list their outgoing half-edges only.  valid[a] = the vertex has at least one face.
// next_out / face_out (may be NULL): per slot, the vertex the NEXT half-edge of the slot's half-edge a -> b points to (the face's third
// corner) and the face itself -- what c_curvature_grad reads through the half-edge records (membrane_mesh_utils.c:1099-1104).
__global__ void k_ring_order(int M, int NB, const int *__restrict__ cnt, const int4 *__restrict__ pairs, int *__restrict__ nbr, unsigned char *__restrict__ valid,
                             int *__restrict__ next_out, int *__restrict__ face_out)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= M) return;
    const int n = min(cnt[a], NB);
    const int4 *P = pairs + (int64_t)a * NB;
    int *out = nbr + (int64_t)a * NB;
    if (valid) valid[a] = n > 0;
    int start = -1, start_he = 0x7fffffff, bstart = -1, bstart_he = -1;
    for (int i = 0; i < n; ++i) {
        const int4 p = P[i];
        if (p.z < start_he) { start_he = p.z; start = i; }
        bool has_twin = false;                       // a -> b has a twin iff some face of the fan ends in b (c_j == b)
        for (int j = 0; j < n; ++j) has_twin |= (P[j].y == p.x);
        if (!has_twin && p.z > bstart_he) { bstart_he = p.z; bstart = i; }
    }
    if (bstart >= 0) start = bstart;
    int cur = start, s = 0;
    while (cur >= 0 && s < NB) {
        const int4 p = P[cur];
        if (next_out) next_out[(int64_t)a * NB + s] = p.y;
        if (face_out) face_out[(int64_t)a * NB + s] = p.w;
        out[s++] = p.x;
        int nxt = -1;
        for (int j = 0; j < n; ++j) if (P[j].x == p.y) { nxt = j; break; }
        if (nxt == start) break;
        cur = nxt;
    }
    for (; s < NB; ++s) {
        out[s] = -1;
        if (next_out) next_out[(int64_t)a * NB + s] = -1;
        if (face_out) face_out[(int64_t)a * NB + s] = -1;
    }
}

// area of the face behind every ring slot: 0.5 |(p1 - p0) x (p2 - p0)| with the corners in the face's own order, every float32 operation
// rounded on its own -- trimesh.TriMesh.update_geometry / nwr_mesh_geometry step for step, so the table equals the host substrate's bit for bit
__global__ void k_slot_area(int64_t n_slots, const int *__restrict__ slot_face, const int *__restrict__ faces, const float *__restrict__ pos, float *__restrict__ area)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    const int f = slot_face[i];
    if (f < 0) { area[i] = 0.0f; return; }
    const int v0 = faces[3 * f], v1 = faces[3 * f + 1], v2 = faces[3 * f + 2];
    const float a0 = pos[3 * v1] - pos[3 * v0], a1 = pos[3 * v1 + 1] - pos[3 * v0 + 1], a2 = pos[3 * v1 + 2] - pos[3 * v0 + 2];
    const float b0 = pos[3 * v2] - pos[3 * v0], b1 = pos[3 * v2 + 1] - pos[3 * v0 + 1], b2 = pos[3 * v2 + 2] - pos[3 * v0 + 2];
    const float c0 = a1 * b2 - a2 * b1, c1 = a2 * b0 - a0 * b2, c2 = a0 * b1 - a1 * b0;      // (-ffp-contract=off: products rounded before the difference)
    const float n = sqrtf((c0 * c0 + c1 * c1) + c2 * c2);
    area[i] = 0.5f * n;
}

__global__ void k_count_mismatch(int64_t n, const int *__restrict__ a, const int *__restrict__ b, int *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicAdd(out, 1);
}

// neighbour table (M, NB) row-major -> slot-major ELL nbr_t[s*M + v] (coalesced over vertices) + max degree
__global__ void k_nbr_transpose(const int *__restrict__ nbr, int M, int NB, int *__restrict__ nbr_t, int *__restrict__ maxdeg)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= M) return;
    int last = 0;
    for (int s = 0; s < NB; ++s) {
        const int n = nbr[(int64_t)v * NB + s];
        nbr_t[(int64_t)s * M + v] = n;
        if (n >= 0) last = s + 1;
    }
    atomicMax(maxdeg, last);
}

// ============================================================================================================
// per-iteration kernels
// ============================================================================================================

// K1: face centroids + cell histogram.  centroid = ((v0+v1)+v2)/3 in float32 = numpy's fv[faces].mean(1)
// (mesh_conj_grad.py:443).  The rank of a face inside its cell (needed by the scatter) comes from the histogram's returning
// atomic; 4e5 returning global atomics cost ~20 us, so the 256 faces of a workgroup -- neighbours on the surface for any
// reasonably ordered mesh -- are first counted per cell in an LDS hash table (integer LDS atomics), and only one global
// atomic per (workgroup, distinct cell) reserves the range.  The same flush adds the workgroup's count to the SCAN TILE the cell
// belongs to (`tile_sums`, a fire-and-forget atomic): the scan's first pass (k_scan_tile_sums, a launch that re-reads the whole
// histogram) is not needed; k_centroid_scatter leaves the tile sums zeroed for the next iteration.
#define NW_FC_HT 512
#define NW_FC_TT 64        // scan tiles a workgroup's cells may lie in before its tile counts go to memory one by one
__global__ __launch_bounds__(NW_BLOCK) void k_face_centroids(NwGrid g, const float *__restrict__ pos, const int *__restrict__ faces, int F,
                                                            float4 *__restrict__ cent_tmp, int *__restrict__ fcell, int *__restrict__ frank, int *__restrict__ count,
                                                            int *__restrict__ tile_sums, int *__restrict__ ambig_count, NwDevState *__restrict__ st, int it,
                                                            int *__restrict__ query_serial)
{
    if (st->iter_base + it >= st->stop_at) return;       // `it` = iteration of this search() block, iter_base = iterations before it
    __shared__ int s_key[NW_FC_HT], s_cnt[NW_FC_HT];
    __shared__ int s_tkey[NW_FC_TT], s_tcnt[NW_FC_TT];
    for (int t = threadIdx.x; t < NW_FC_HT; t += NW_BLOCK) { s_key[t] = -1; s_cnt[t] = 0; }
    if (threadIdx.x < NW_FC_TT) { s_tkey[threadIdx.x] = -1; s_tcnt[threadIdx.x] = 0; }
    __syncthreads();
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f == 0) { *ambig_count = 0; *query_serial += 1; }      // (the serial number of the query this grid is built for: k_nn_wave's in-launch hand-off)
    int slot = 0, local = 0;
    if (f < F) {
        const int a = faces[3 * f], b = faces[3 * f + 1], c = faces[3 * f + 2];
        const float x = ((pos[3 * a] + pos[3 * b]) + pos[3 * c]) / 3.0f;
        const float y = ((pos[3 * a + 1] + pos[3 * b + 1]) + pos[3 * c + 1]) / 3.0f;
        const float z = ((pos[3 * a + 2] + pos[3 * b + 2]) + pos[3 * c + 2]) / 3.0f;
        int ix, iy, iz;
        nw_cell_coords(g, x, y, z, ix, iy, iz);
        const int cell = nw_cell_index(g, ix, iy, iz);
        const float fx = (x - g.ox) * g.inv_h, fy = (y - g.oy) * g.inv_h, fz = (z - g.oz) * g.inv_h;
        if (fx < -1.0f || fy < -1.0f || fz < -1.0f || fx > (float)g.gx + 1.0f || fy > (float)g.gy + 1.0f || fz > (float)g.gz + 1.0f) st->grid_escape = 1;
        cent_tmp[f] = make_float4(x, y, z, __int_as_float(f));
        fcell[f] = cell;
        slot = (int)(((unsigned)cell * 2654435761u) >> 23);            // 9 bits
        for (;;) {
            const int old = atomicCAS(&s_key[slot], -1, cell);
            if (old == -1 || old == cell) break;
            slot = (slot + 1) & (NW_FC_HT - 1);
        }
        local = atomicAdd(&s_cnt[slot], 1);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < NW_FC_HT; t += NW_BLOCK) {
        const int key = s_key[t];
        if (key >= 0) {
            const int c = s_cnt[t];
            // (per tile first in LDS: a workgroup's cells lie in a handful of tiles, and same-address global atomics run one after the other)
            const int tile = key / NW_SCAN_TILE;
            int ts = tile & (NW_FC_TT - 1), probes = 0;
            for (; tile_sums;) {
                const int old = atomicCAS(&s_tkey[ts], -1, tile);
                if (old == -1 || old == tile) { atomicAdd(&s_tcnt[ts], c); break; }
                if (++probes == NW_FC_TT) { atomicAdd(&tile_sums[tile], c); break; }
                ts = (ts + 1) & (NW_FC_TT - 1);
            }
            s_cnt[t] = atomicAdd(&count[key], c);                       // base of this workgroup's faces in the cell
        }
    }
    __syncthreads();
    if (tile_sums && threadIdx.x < NW_FC_TT && s_tkey[threadIdx.x] >= 0) atomicAdd(&tile_sums[s_tkey[threadIdx.x]], s_tcnt[threadIdx.x]);
    __syncthreads();
    if (f < F) frank[f] = s_cnt[slot] + local;
}

// K3: scatter centroids into cell order
__global__ __launch_bounds__(NW_BLOCK) void k_centroid_scatter(int F, const float4 *__restrict__ cent_tmp, const int *__restrict__ fcell, const int *__restrict__ frank,
                                                              const int *__restrict__ start, float4 *__restrict__ cent,
                                                              const NwDevState *__restrict__ st, int it, int *__restrict__ tile_sums, int ntiles)
{
    if (st->iter_base + it >= st->stop_at) return;       // `it` = iteration of this search() block, iter_base = iterations before it
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    for (int t = f; t < ntiles; t += (int)(gridDim.x * blockDim.x)) tile_sums[t] = 0;      // consumed by the scan: zero for the next k_face_centroids
    if (f >= F) return;
    cent[start[fcell[f]] + frank[f]] = cent_tmp[f];
}

// K4a: the exact nearest-face query (k_nn_wave) and the set-up kernels of its work list; K4b, the attraction step, in its tail
#include "nw_attract.h"
#include "nw_nn.h"

// exact float64 re-resolution of the ambiguous points (runner-up inside the float32 error band of the best) as its own launch: one
// wave per point, nw_fixup_point (nw_nn.h).  Only used when the query kernel does not resolve them itself (NW_FUSE_FIXUP=0).
__global__ __launch_bounds__(NW_BLOCK) void k_nn_fixup(NwGrid g, const int *__restrict__ ambig_list, const int *__restrict__ ambig_count, const float4 *__restrict__ pts,
                                                      const int *__restrict__ cstart, const float4 *__restrict__ cent, const float4 *__restrict__ cent_by_face,
                                                      int *__restrict__ face_out, const NwDevState *__restrict__ st, int it, const int *__restrict__ face_orig)
{
    if (st->iter_base + it >= st->stop_at) return;       // `it` = iteration of this search() block, iter_base = iterations before it
    const int na = *ambig_count;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int a = wave; a < na; a += nwaves) {
        const int gi = ambig_list[a];
        const float4 P = pts[gi];
        const int bf = nw_fixup_point(g, P.x, P.y, P.z, cent_by_face[face_out[gi]], cstart, cent, lane, face_orig);
        if (lane == 0) face_out[gi] = bf;
    }
}

// K4b: the attraction step, one thread per localization (nw_attract_point, nw_attract.h), the workgroup's 256 localizations share one
// table.  (Running it in the tail of the query kernel was measured in round 3 and not kept: tools/experiments/r03_notes.md.)
#define NW_HT 512          // slots of the per-workgroup table: 18 KB of LDS, 8 workgroups per CU (1024 slots: 36 KB, 4 per CU, 10 us slower)
#define NW_HT_BITS 9
#define NW_ATTRACT_PPT 1      // (2: 39.7 us against 36.5 -- the two dependent load chains of a thread do not overlap)
#define NW_HT_PROBES 48    // a contribution that finds no slot within this many probes goes to HBM directly (unsorted input only)

__global__ __launch_bounds__(NW_BLOCK) void k_attract(int N, const float4 *__restrict__ pts, const int *__restrict__ face, const NwAttractArgs A,
                                                     NwDevState *__restrict__ st, int it)
{
    if (st->iter_base + it >= st->stop_at) return;       // `it` = iteration of this search() block, iter_base = iterations before it
    __shared__ int s_key[NW_HT];
    __shared__ unsigned long long s_val[NW_HT * 4];        // component-major [4][NW_HT], two's-complement fixed point
    for (int t = threadIdx.x; t < NW_HT; t += NW_BLOCK) s_key[t] = -1;
    for (int t = threadIdx.x; t < NW_HT * 4; t += NW_BLOCK) s_val[t] = 0ull;
    __syncthreads();
    // each XCD takes one contiguous range of the brick-sorted localizations: the vertices its workgroups gather and the
    // accumulator lines their atomics touch then live in ONE L2 instead of being spread round-robin over all eight
    const int blk = nw_xcd_remap(blockIdx.x, (N + NW_BLOCK * NW_ATTRACT_PPT - 1) / (NW_BLOCK * NW_ATTRACT_PPT));
    double red[4] = {0.0, 0.0, 0.0, 0.0};
    float dmax = 0.0f;
#pragma unroll 1
    for (int pp = 0; pp < NW_ATTRACT_PPT; ++pp) {
        const int i = blk < 0 ? N : (blk * NW_ATTRACT_PPT + pp) * (int)blockDim.x + (int)threadIdx.x;
        const int f_raw = i < N ? face[i] : 0;
        // a face id outside [0, F) can only come from a bug in the NN query: never dereference it (a faulting kernel can
        // take the whole node down), raise the internal-error status instead
        if (i < N && (unsigned)f_raw >= (unsigned)A.F) atomicCAS(&st->status, 0, -7 /* NW_ERR_INTERNAL */);
        const bool ok = i < N && (unsigned)f_raw < (unsigned)A.F;
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
        // (uniform control flow from here: the run sums move data between lanes)
        const bool last_of_run = nw_run_sums(q, ok ? f_raw : -2 - (int)(threadIdx.x & 15));
        if (ok && last_of_run) nw_attract_scatter<NW_HT, NW_HT_BITS, NW_HT_PROBES>(A, v, q, s_key, s_val);
    }
    __syncthreads();
    // per-workgroup partial sums (row of 5: four sums + the largest NN distance of the workgroup, which a sharded run checks against
    // its halo radius; a same-address atomicMax from every wave serialised: 40 us at 1M localizations, 0.7 ms at 5M).  Wave sums on the
    // vector ALU (nw_wave_sum).
    {
        __shared__ double s_w[5 * 4];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double sk = nw_wave_sum(red[k]);
            if (lane == 0) s_w[k * 4 + wv] = sk;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off, 64));
        if (lane == 0) s_w[16 + wv] = (double)dmax;
        __syncthreads();
        if (threadIdx.x < 4) A.part[(int64_t)blockIdx.x * 5 + threadIdx.x] = (s_w[threadIdx.x * 4 + 0] + s_w[threadIdx.x * 4 + 1]) + (s_w[threadIdx.x * 4 + 2] + s_w[threadIdx.x * 4 + 3]);
        else if (threadIdx.x == 4) A.part[(int64_t)blockIdx.x * 5 + 4] = fmax(fmax(s_w[16], s_w[17]), fmax(s_w[18], s_w[19]));
    }
    for (int t = threadIdx.x; t < NW_HT * 4; t += NW_BLOCK) {
        const int key = s_key[t >> 2];                     // four adjacent lanes flush the four components of one vertex
        if (key >= 0) atomicAdd(reinterpret_cast<unsigned long long *>(A.vacc) + 4 * (int64_t)key + (t & 3), s_val[(t & 3) * NW_HT + (t >> 2)]);
    }
}

// K5: curvature prior + search directions S0, S1 + all vertex-side dot products.  One thread per vertex;
// the 1-ring comes from the slot-major neighbour table (coalesced over vertices).
//   S0 = vacc.xyz (A^T res)                                                            mesh_conj_grad.py:253
//   pi = sqrt(3 (sum w)^2) in f32, gate min(pi^2,1)                                    _membrane_mesh.pyx:1633-1634, :807-814
//   fdef = _ncc(): 1-ring centroid (f32 slot-ordered sum -> /ms in f64), alpha from neighbour normals (block-stale)
//          and current mesh positions; isolated vertices keep their position           :770-820
//   prefs = f - fdef (f64), stored f32; S1 = -prefs                                    :257-258
__global__ __launch_bounds__(NW_BLOCK) void k_prior_ring(const NwRingArgs R, const NwDevState *__restrict__ st, int it)
{
    // the ring half as a launch of its own (NW_RING_IN_NN=0, and a query launched without its ring workgroups): see nw_device.h
    if (st->iter_base + it >= st->stop_at) return;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < R.M) nw_prior_ring_vertex(R, v);
}

__global__ __launch_bounds__(NW_BLOCK) void k_prior_directions(int M, const float4 *__restrict__ ring4, const double *__restrict__ ring_a, const float *__restrict__ pos,
                                                              const float *__restrict__ meshpos, const float *__restrict__ nrm,
                                                              const long long *__restrict__ vacc, float *__restrict__ S, float *__restrict__ fdef_out,
                                                              float *__restrict__ pi_out, double *__restrict__ part, NwDevState *__restrict__ st, int it, int n_search,
                                                              double q, double qw, const float *__restrict__ wv, const unsigned char *__restrict__ owned)
{
    if (st->iter_base + it >= st->stop_at) return;       // `it` = iteration of this search() block, iter_base = iterations before it
    __shared__ double s_part[14 * 4];
    double red[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) red[k] = 0.0;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < M; v += gridDim.x * blockDim.x) {
        // everything this vertex needs, requested at once (the ring half -- the 1-ring gathers -- has run beside the query: nw_device.h)
        const longlong2 a01 = *reinterpret_cast<const longlong2 *>(vacc + 4 * (int64_t)v), a23 = *reinterpret_cast<const longlong2 *>(vacc + 4 * (int64_t)v + 2);
        const float4 rg = ring4[v];
        const double asum = ring_a[v];
        const nw_f3u Nn = *reinterpret_cast<const nw_f3u *>(nrm + 3 * (int64_t)v);
        const nw_f3u Pm = *reinterpret_cast<const nw_f3u *>(meshpos + 3 * (int64_t)v);
        const nw_f3u Pf = *reinterpret_cast<const nw_f3u *>(pos + 3 * (int64_t)v);
        float *Srow = S + (int64_t)v * 9;
        const float s2c[3] = {n_search > 2 ? Srow[2] : 0.0f, n_search > 2 ? Srow[5] : 0.0f, n_search > 2 ? Srow[8] : 0.0f};
        const float lw = wv ? wv[v] : 1.0f;
        const bool mine = owned ? owned[v] != 0 : true;
        const float4 acc = make_float4((float)((double)a01.x * q), (float)((double)a01.y * q), (float)((double)a23.x * q), (float)((double)a23.y * qw));
        const float sw = acc.w;
        const float pi = sqrtf((sw * sw + sw * sw) + sw * sw);
        const float gate = fminf(pi * pi, 1.0f);
        pi_out[v] = pi;
        const int ms = __float_as_int(rg.w);
        double fd[3];
        if (ms > 0) {
            const double vcx = (double)rg.x / ms, vcy = (double)rg.y / ms, vcz = (double)rg.z / ms;
            const double alpha = (asum / ms) * (double)gate;
            fd[0] = vcx + alpha * (double)Nn.x; fd[1] = vcy + alpha * (double)Nn.y; fd[2] = vcz + alpha * (double)Nn.z;
        } else {
            fd[0] = Pm.x; fd[1] = Pm.y; fd[2] = Pm.z;
        }
        const float pf[3] = {Pf.x, Pf.y, Pf.z};
        const float s0[3] = {acc.x, acc.y, acc.z};
        // regulariser: identity (Lfuncs = ["I"], mesh_conj_grad.py:38) or the diagonal 'wfunc' (:724-735), L x = x * wv
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double p64 = (double)pf[c] - fd[c];
            if (wv) p64 = p64 * (double)lw;                       // wfunc(f - fdef): float64 * float32 -> float64 (:257, conj_grad.py:191)
            const float p32 = (float)p64;                         // prefs[:, 0] is a float32 array
            const float s1 = wv ? -1.0f * (p32 * lw) : -1.0f * p32;      // S[:, 1] = -Lh(prefs) (:258)
            float *row = Srow + 3 * c;
            const float s2 = s2c[c];
            row[0] = s0[c];
            row[1] = s1;
            fdef_out[3 * v + c] = (float)fd[c];
            // LS_k = L(S_k) in float32 (conj_grad.py:199); Hw = LS^T LS, Gw = -LS^T prefs64 (:211-212).  A sharded mesh counts every
            // vertex once: on the rank that owns it
            if (!mine) continue;
            const float l0 = wv ? s0[c] * lw : s0[c], l1 = wv ? s1 * lw : s1, l2 = wv ? s2 * lw : s2;
            red[0] += (double)l0 * l0;
            red[1] += (double)l0 * l1;
            red[2] += (double)l0 * l2;
            red[3] += (double)l1 * l1;
            red[4] += (double)l1 * l2;
            red[5] += (double)l2 * l2;
            red[6] += (double)l0 * p64;
            red[7] += (double)l1 * p64;
            red[8] += (double)l2 * p64;
            red[9] += p64 * p64;
            red[10] += (double)p32 * (double)p32;
            red[11] += (double)s0[c] * s0[c];
            red[12] += (double)s0[c] * s1;
            red[13] += (double)s1 * s1;
        }
    }
    nw_block_sums_store<14>(red, part, s_part);
}

// weights of the 'wfunc' regulariser from the CURRENT estimate f (mesh_conj_grad.py:733 -> vertex_area_weights,
// conj_grad_utils.c:500-548): 1/sqrt(sum_n |f_n - f_i|^2 + 1), float32 sums in slot order; 0 for a vertex without neighbours.
__global__ __launch_bounds__(NW_BLOCK) void k_vertex_area_weights(int M, int NB, const int *__restrict__ nbr, const float *__restrict__ f, float *__restrict__ wv,
                                                                 const NwDevState *__restrict__ st, int it)
{
    if (st->iter_base + it >= st->stop_at) return;       // `it` = iteration of this search() block, iter_base = iterations before it
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int *row = nbr + (int64_t)i * NB;
    float w = 0;
    for (int k = 0; k < NB; ++k) {
        const int n = row[k]; if (n == -1) break;
        float d2 = 0;
        for (int j = 0; j < 3; ++j) { const float dd = f[3 * n + j] - f[3 * i + j]; d2 += dd * dd; }
        w += d2;
    }
    wv[i] = (w > 0) ? (float)(1.0 / (double)sqrtf(w + 1)) : 0.0f;
}

#define NW_SUBSPACE_PPT 4

// K6: A.S_k for the n_search directions and the point-side normal-equation sums, never materialising AS:
//   AS_k[i] = sum_j w_ij S_k[v_ij] (f32, corner order, mesh_conj_grad.py:544-545 via conj_grad.py:198)
//   Hc = AS^T AS, Gc = AS^T res over the masked entries (conj_grad.py:198-203)
// S is stored (3M,3) row-major exactly as the reference's `cg.S`, so the 9 floats of one vertex are contiguous.
// Memory-level parallelism (round 5): a thread's NW_SUBSPACE_PPT localizations are independent, but written as a loop with the index
// load, the row gather and the masked residual load of each corner inside conditionals the compiler issued them one after the other --
// ~28 dependent round trips per thread, SQ_WAIT_ANY 78 % of the wave cycles, 20 us for 37 MB.  Now: (1) the vertex ids, weights,
// residuals and masks of ALL the thread's localizations are loaded unconditionally (a thread beyond N reads localization 0 and
// contributes through an empty mask), (2) the twelve 36-byte rows of S are gathered two localizations at a time (3 wide loads a row).
// Three round trips in all; the arithmetic and its order are unchanged (bit-identical sums).
// (Round 5 also tried ONE gather per run of lanes that share their face -- 2.6 lanes on average; the first lane of a run gathers the three
// rows, parks them in the wave's LDS buffer, every lane of the run reads them back: bit-identical, 22.8 us against 22.0 for gather +
// reduce -- the kernel is bound by neither the gathers' address rate nor their latency any more; not kept.)
__global__ __launch_bounds__(NW_BLOCK) void k_subspace_point_sums(int N, int M, const int *__restrict__ vidx, const float *__restrict__ w, const float *__restrict__ res,
                                                                 const unsigned char *__restrict__ mask, const float *__restrict__ S, double *__restrict__ part,
                                                                 const NwDevState *__restrict__ st, int it, int n_search)
{
    if (st->iter_base + it >= st->stop_at) return;       // `it` = iteration of this search() block, iter_base = iterations before it
    __shared__ double s_w[9 * 4];
    // NW_SUBSPACE_PPT localizations per thread (consecutive tiles of 256): the nine sums are reduced over the workgroup once per
    // 1024 localizations -- the shuffles of that reduction go through the CU's LDS pipe, which bounded the kernel at one per thread
    const int blk = nw_xcd_remap(blockIdx.x, (N + NW_BLOCK * NW_SUBSPACE_PPT - 1) / (NW_BLOCK * NW_SUBSPACE_PPT));      // see k_attract
    double red[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) red[k] = 0.0;
    nw_i3u vi[NW_SUBSPACE_PPT];
    nw_f3u wj[NW_SUBSPACE_PPT], rr[NW_SUBSPACE_PPT];
    unsigned msk[NW_SUBSPACE_PPT];
#pragma unroll
    for (int p = 0; p < NW_SUBSPACE_PPT; ++p) {
        const int64_t i = blk < 0 ? (int64_t)N : ((int64_t)blk * NW_SUBSPACE_PPT + p) * NW_BLOCK + threadIdx.x;
        const int64_t ic = i < N ? i : 0;
        vi[p] = *reinterpret_cast<const nw_i3u *>(vidx + 3 * ic);
        wj[p] = *reinterpret_cast<const nw_f3u *>(w + 3 * ic);
        rr[p] = *reinterpret_cast<const nw_f3u *>(res + 3 * ic);
        const unsigned m = mask[ic];
        msk[p] = i < N ? m : 0u;
    }
#pragma unroll
    for (int h = 0; h < NW_SUBSPACE_PPT; h += 2) {
        float row[2][3][9];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                // (an id outside the mesh can only be the leftover of an attraction step that raised the internal-error status: never dereferenced)
                const unsigned v = min((unsigned)vi[h + q][j], (unsigned)(M - 1));
                const float *src = S + (int64_t)v * 9;
                const nw_f4u a = *reinterpret_cast<const nw_f4u *>(src), b = *reinterpret_cast<const nw_f4u *>(src + 4);
                row[q][j][0] = a.x; row[q][j][1] = a.y; row[q][j][2] = a.z; row[q][j][3] = a.w;
                row[q][j][4] = b.x; row[q][j][5] = b.y; row[q][j][6] = b.z; row[q][j][7] = b.w;
                row[q][j][8] = src[8];
            }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float as[3][3];   // [direction][component]
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) as[k][c] = 0.0f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float wq = wj[h + q][j];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    as[0][c] = as[0][c] + row[q][j][3 * c + 0] * wq;
                    as[1][c] = as[1][c] + row[q][j][3 * c + 1] * wq;
                    if (n_search > 2) as[2][c] = as[2][c] + row[q][j][3 * c + 2] * wq;
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (msk[h + q] & (1u << c)) {
                    const double a0 = as[0][c], a1 = as[1][c], a2 = as[2][c], r = rr[h + q][c];
                    red[0] += a0 * a0; red[1] += a0 * a1; red[2] += a0 * a2;
                    red[3] += a1 * a1; red[4] += a1 * a2; red[5] += a2 * a2;
                    red[6] += a0 * r;  red[7] += a1 * r;  red[8] += a2 * r;
                }
            }
        }
    }
    nw_block_sums_store<9>(red, part, s_w);
}

// Start of a block (search() call) in ONE launch: the estimate restarts from the mesh positions (fs = vertices.copy(),
// mesh_conj_grad.py:170, :1002-1007), the log records are cleared, and the arrays the reference zero-initialises are zeroed WHERE THE
// BLOCK WILL NOT OVERWRITE THEM BEFORE ANYTHING READS THEM:
//   res = 0 * data (:181): k_attract writes every entry in the block's first iteration -> zeroed only if that iteration will not run
//         (the stop condition already holds, or the call asks for no iteration);
//   S = zeros (:207): columns 0 and 1 are written by k_prior_directions, column 2 by k_solve_update before it is first read (n_search is 2
//         in a block's first iteration) -> zeroed only if the first iteration will not run or last_step is off (column 2 then stays zero);
//   the scatter accumulator: every k_solve_update leaves it zeroed for the next scatter -> zeroed here only when the host knows something
//         else has used it since (nw_refresh_normals borrows it; a fresh allocation).
// Five memset / copy nodes of 2-12 MB each were ~30 us of launch latency per block.
__global__ __launch_bounds__(NW_BLOCK) void k_block_begin(int M, int64_t n_res, float *__restrict__ pos, const float *__restrict__ meshpos, float *__restrict__ S,
                                                         float *__restrict__ res, long long *__restrict__ vacc, unsigned *__restrict__ logs_words, int n_log_words,
                                                         const NwDevState *__restrict__ st, int no_iteration_or_no_last_step, int zero_vacc)
{
    const bool first_runs = st->iter_base < st->stop_at;      // (k_set_iter_base ran before this launch)
    const bool zero_res = !first_runs || (no_iteration_or_no_last_step & 1);
    const bool zero_S = !first_runs || no_iteration_or_no_last_step != 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < 3 * (int64_t)M; t += stride) pos[t] = meshpos[t];
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_log_words; t += stride) logs_words[t] = 0u;
    if (zero_S) for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < 9 * (int64_t)M; t += stride) S[t] = 0.0f;
    if (zero_res) for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_res; t += stride) res[t] = 0.0f;
    if (zero_vacc) for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < 4 * (int64_t)M; t += stride) vacc[t] = 0ll;
}

// End of a block whose result goes straight to the host (round 5): the block's logs and the device state are copied into pinned host
// memory by this one-workgroup launch behind the block's last kernel, then a FLAG word in pinned memory takes the next value of a device
// counter.  The host thread that waits for the block spins on that word (a load from its own cache until the PCIe write invalidates the
// line) instead of sleeping in hipStreamSynchronize, whose wake-up comes 40-50 us after the last kernel (completion interrupt); the
// library's copy threads are woken shortly before the block's expected end and spin on the same word.  Stream order makes the flag
// the last thing the block writes: the result (written by k_solve_update) is complete when it shows.
__global__ __launch_bounds__(NW_BLOCK) void k_block_done(const unsigned *__restrict__ logs_words, int n_words, const NwDevState *__restrict__ st,
                                                        unsigned *__restrict__ pin_words, unsigned *__restrict__ pin_state, int *__restrict__ counter, int *host_flag)
{
    for (int t = threadIdx.x; t < n_words; t += NW_BLOCK) pin_words[t] = logs_words[t];
    if (threadIdx.x < (int)(sizeof(NwDevState) / 4)) pin_state[threadIdx.x] = reinterpret_cast<const unsigned *>(st)[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const int v = *counter + 1;
        *counter = v;
        __hip_atomic_store(host_flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// A slice of a large result into the pinned staging buffer (results above 4 MB: the last update kernel writing 10-20 MB over PCIe itself
// would hold the host's copy threads back until its end).  The launch first tells the host that everything queued before it has run --
// the flag word takes the counter's next value, see k_block_done -- so the copy threads take slice s while slice s + 1 is in flight.
__global__ __launch_bounds__(NW_BLOCK) void k_copy_slice(const float *__restrict__ src, float *__restrict__ dst, int64_t n_floats, int *__restrict__ counter, int *host_flag)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int v = *counter + 1;
        *counter = v;
        __hip_atomic_store(host_flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const int64_t n4 = n_floats >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) d4[i] = s4[i];
    if (blockIdx.x == 0 && threadIdx.x < (int)(n_floats & 3)) dst[4 * n4 + threadIdx.x] = src[4 * n4 + threadIdx.x];
}

// K7: <=3x3 regularised normal equations (every workgroup solves them redundantly from the reduced sums),
// f += S c, last step -> S2, write-back, per-iteration log, stop condition.
//   H = Hc + lam^2 Hw accumulated in place in float32, G likewise (conj_grad.py:208-215); float32 LU with
//   partial pivoting (LAPACK sgesv through numpy.linalg.solve, :219); fnew = f0 + S.c in float32 (:227);
//   S[:,2] = fnew - f; f = fnew; mesh positions updated at valid vertices only (mesh_conj_grad.py:281-289).
struct NwSolve { float c[3]; float H[9]; float G[3]; int singular; };
__device__ const unsigned char nw_byte_one = 1;

// index of (r,c) in the packed upper triangle {00,01,02,11,12,22}
__device__ __forceinline__ constexpr int nw_tri(int r, int c) { return r <= c ? (r == 0 ? c : (r == 1 ? 2 + c : 5)) : (c == 0 ? r : (c == 1 ? 2 + r : 5)); }

// Fully unrolled (compile-time indices only -> registers, no scratch): the regularised system and its float32 LU solve.
template <int N>
__device__ __forceinline__ void nw_solve_small(const double *__restrict__ sc, float lam, NwSolve &o)
{
    const double l2 = (double)lam * (double)lam;
    float A[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float hc = (float)sc[SC_HC + nw_tri(r, c)];
            const float hw = (float)sc[SC_SS + nw_tri(r, c)];
            const float h = (float)((double)hc + l2 * (double)hw);
            o.H[3 * r + c] = h;
            A[r][c] = h;
        }
        const float gc = (float)sc[SC_GC + r];
        const double gw = -sc[SC_SP + r];
        const float gg = (float)((double)gc + l2 * gw);
        o.G[r] = gg;
        A[r][3] = gg;
    }
    bool singular = false;
    // Gaussian elimination with partial pivoting, float32; row swaps as selects so every index stays a constant
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float mx = fabsf(A[k][k]);
#pragma unroll
        for (int r = k + 1; r < N; ++r) {
            const bool sw = fabsf(A[r][k]) > mx;
            mx = sw ? fabsf(A[r][k]) : mx;
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float t = A[k][c]; A[k][c] = sw ? A[r][c] : t; A[r][c] = sw ? t : A[r][c]; }
        }
        singular = singular || !(mx > 0.0f);
#pragma unroll
        for (int r = k + 1; r < N; ++r) {
            const float l = A[r][k] / A[k][k];
#pragma unroll
            for (int c = k; c < 4; ++c) A[r][c] = A[r][c] - l * A[k][c];
        }
    }
    float x[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
        float sacc = A[k][3];
#pragma unroll
        for (int c = k + 1; c < N; ++c) sacc = sacc - A[k][c] * x[c];
        x[k] = sacc / A[k][k];
    }
    o.singular = singular ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) o.c[k] = singular ? 0.0f : x[k];
}

struct NwIterLogDev {   // mirrors nw_iter_log in include/nanowrap.h
    double test, res_norm, prefs_norm, cpred, wpred, c[3], H[9], G[3], mean_dist, max_dist;
    int n_search, nn_max_ring, status, executed;
};

__global__ __launch_bounds__(NW_BLOCK) void k_solve_update(int M, float lam, int n_search, unsigned flags, const unsigned char *__restrict__ valid,
                                                          float *__restrict__ pos, float *__restrict__ meshpos, float *__restrict__ S, long long *__restrict__ vacc,
                                                          const double *__restrict__ sc_parts, NwDevState *__restrict__ st,
                                                          NwIterLogDev *__restrict__ logrec, int it, float *__restrict__ host_out)
{
    // host_out (last iteration of a block only): pinned host memory that receives the block's result straight from this kernel --
    // the PCIe writes overlap the kernel instead of following it as a separate device-to-host copy
    if (st->iter_base + it >= st->stop_at) return;       // `it` = iteration of this search() block, iter_base = iterations before it
    __shared__ NwSolve s_sol;
    __shared__ double s_sc[SC_COUNT];
    // Round 5: the vertex data of the thread's first two vertices (S rows, estimate, valid flag) are requested BEFORE the sums and the
    // solve, so that their latency runs beside that serial prologue; all 32 ordered parts of a sum are requested at once (one round
    // trip instead of four).  Per vertex: three wide loads of the 36-byte S row + one of the position instead of a load-wait-store
    // chain per component.  Arithmetic unchanged.
    const int stride = (int)(gridDim.x * blockDim.x);
    const int v0 = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int NPRE = 2;
    float pre_row[NPRE][9], pre_pos[NPRE][3];
    unsigned char pre_ok[NPRE];
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
        const int v = v0 + u * stride;
        const int vc = v < M ? v : 0;
        const float *src = S + (int64_t)vc * 9;
        const nw_f4u a = *reinterpret_cast<const nw_f4u *>(src), b = *reinterpret_cast<const nw_f4u *>(src + 4);
        pre_row[u][0] = a.x; pre_row[u][1] = a.y; pre_row[u][2] = a.z; pre_row[u][3] = a.w;
        pre_row[u][4] = b.x; pre_row[u][5] = b.y; pre_row[u][6] = b.z; pre_row[u][7] = b.w;
        pre_row[u][8] = src[8];
        const nw_f3u p3 = *reinterpret_cast<const nw_f3u *>(pos + 3 * (int64_t)vc);
        pre_pos[u][0] = p3.x; pre_pos[u][1] = p3.y; pre_pos[u][2] = p3.z;
        pre_ok[u] = *(valid ? valid + vc : &nw_byte_one);          // (one unconditional load: a load inside a branch would end in a full wait)
    }
    if (threadIdx.x < SC_COUNT) {
        double pp[NW_SPARTS];
#pragma unroll
        for (int b = 0; b < NW_SPARTS; ++b) pp[b] = sc_parts[threadIdx.x * NW_SPARTS + b];
        double t = 0.0;
        if (threadIdx.x == SC_MAXD) {
#pragma unroll
            for (int b = 0; b < NW_SPARTS; ++b) t = fmax(t, pp[b]);
        } else {
#pragma unroll
            for (int b = 0; b < NW_SPARTS; ++b) t += pp[b];
        }
        s_sc[threadIdx.x] = t;
    }
    __syncthreads();
    const double *sc = s_sc;
    if (threadIdx.x == 0) { if (n_search > 2) nw_solve_small<3>(sc, lam, s_sol); else nw_solve_small<2>(sc, lam, s_sol); }
    __syncthreads();
    const NwSolve sol = s_sol;
    // a status raised earlier in this iteration (NaN in the weight matrix / A f / A^T r: the reference asserts BEFORE `self.f[:] = fnew`,
    // mesh_conj_grad.py:514,548,580 vs :288) leaves the estimate, the mesh positions and the staged result at the last good iterate
    const bool failed = st->status != 0 || sc[SC_STATUS] > 0.0;     // (the second: another rank's, summed in with the normal-equation sums)
    const bool go = !sol.singular && !failed;
    auto update_vertex = [&](int v, const float (&row)[9], const float (&f0)[3], bool ok) {
        float fn[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float step = row[3 * c + 0] * sol.c[0];
            step = step + row[3 * c + 1] * sol.c[1];
            if (n_search > 2) step = step + row[3 * c + 2] * sol.c[2];
            fn[c] = f0[c] + step;
            if ((flags & 1u) && !(fn[c] > 0.0f)) fn[c] = fn[c] * 0.0f;       // fnew*(fnew > 0), mesh_conj_grad.py:277-278
        }
        if (go) {
            if (!(flags & 2u)) {
#pragma unroll
                for (int c = 0; c < 3; ++c) S[(int64_t)v * 9 + 3 * c + 2] = fn[c] - f0[c];
            }
            nw_f3u o; o.x = fn[0]; o.y = fn[1]; o.z = fn[2];
            *reinterpret_cast<nw_f3u *>(pos + 3 * (int64_t)v) = o;
            if (ok) *reinterpret_cast<nw_f3u *>(meshpos + 3 * (int64_t)v) = o;
            if (host_out) *reinterpret_cast<nw_f3u *>(host_out + 3 * (int64_t)v) = o;
        } else if (host_out) {
            nw_f3u o; o.x = f0[0]; o.y = f0[1]; o.z = f0[2];
            *reinterpret_cast<nw_f3u *>(host_out + 3 * (int64_t)v) = o;
        }
        *reinterpret_cast<longlong2 *>(vacc + 4 * (int64_t)v) = make_longlong2(0, 0);          // ready for the next scatter
        *reinterpret_cast<longlong2 *>(vacc + 4 * (int64_t)v + 2) = make_longlong2(0, 0);
    };
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
        const int v = v0 + u * stride;
        if (v < M) update_vertex(v, pre_row[u], pre_pos[u], pre_ok[u] != 0);
    }
    for (int v = v0 + NPRE * stride; v < M; v += stride) {
        float row[9], f0[3];
        const float *src = S + (int64_t)v * 9;
        const nw_f4u a = *reinterpret_cast<const nw_f4u *>(src), b = *reinterpret_cast<const nw_f4u *>(src + 4);
        row[0] = a.x; row[1] = a.y; row[2] = a.z; row[3] = a.w; row[4] = b.x; row[5] = b.y; row[6] = b.z; row[7] = b.w; row[8] = src[8];
        const nw_f3u p3 = *reinterpret_cast<const nw_f3u *>(pos + 3 * (int64_t)v);
        f0[0] = p3.x; f0[1] = p3.y; f0[2] = p3.z;
        update_vertex(v, row, f0, valid ? valid[v] != 0 : true);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // logs (mesh_conj_grad.py:262-274)
        const double s00 = sc[SC_T + 0], s01 = sc[SC_T + 1], s11 = sc[SC_T + 2];
        const float test = 1.0f - (float)(fabs(s01) / (sqrt(s00) * sqrt(s11)));
        NwIterLogDev L;
        L.test = test;
        L.res_norm = sqrt(sc[SC_RES2]);
        L.prefs_norm = sqrt(sc[SC_PP32]);
        double cHc = 0, cG = 0, cHwc = 0, cGw = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (r < n_search) {
                cG += (double)sol.c[r] * sol.G[r];
                cGw += (double)sol.c[r] * (-sc[SC_SP + r]);
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < n_search) {
                        cHc += (double)sol.c[r] * sol.H[3 * r + c] * sol.c[c];
                        cHwc += (double)sol.c[r] * (double)(float)sc[SC_SS + nw_tri(r, c)] * sol.c[c];
                    }
            }
        }
        L.cpred = sc[SC_C0] + cHc - cG;          // Hc/Gc alias the regularised H/G in the reference (conj_grad.py:208,223)
        L.wpred = sc[SC_PP64] + cHwc - cGw;      // conj_grad.py:225
        for (int k = 0; k < 3; ++k) { L.c[k] = sol.c[k]; L.G[k] = sol.G[k]; }
        for (int k = 0; k < 9; ++k) L.H[k] = sol.H[k];
        L.mean_dist = sc[SC_NPTS] > 0 ? sc[SC_SUMD] / sc[SC_NPTS] : 0.0;
        L.max_dist = sc[SC_MAXD];
        L.n_search = n_search;
        L.nn_max_ring = st->nn_max_ring;
        if (sol.singular) atomicCAS(&st->status, 0, -4 /* NW_ERR_SINGULAR */);
        if (sc[SC_STATUS] > 0.0) atomicCAS(&st->status, 0, -8 /* NW_ERR_REMOTE */);
        L.status = st->status;
        L.executed = (failed || sol.singular) ? 0 : 1;       // (the reference raised inside this iteration: it does not count, and the result is copied out the ordinary way)
        *logrec = L;
        // history + stop condition for the NEXT iteration (mesh_conj_grad.py:1009-1016)
        float a = st->tests[1], b = st->tests[2];
        st->tests[0] = a; st->tests[1] = b; st->tests[2] = test;
        st->ntests += 1;
        if (st->ntests >= 3 && (test < b) && (b < a) && (a < 1e-6f)) st->stop_at = st->iter_base + it + 1;
        if (st->status != 0) st->stop_at = st->iter_base + it + 1;
    }
}

// The per-workgroup partial sums of the three reduction kernels (rows of 4 / 9 / 11 doubles, ~0.45 MB at 1M localizations) are
// added in a FIXED order in two steps (deterministic, no atomics): NW_SPARTS workgroups each add one contiguous share of the rows
// (one workgroup alone is limited by the ~25 GB/s a single CU pulls: 23 us) into sc[slot * NW_SPARTS + part]; the consumer
// (k_solve_update, every workgroup for itself) adds the NW_SPARTS values of a slot in order.  Each share is read flat and
// coalesced by a number of threads that is a multiple of the row length, so a thread stays on one column.
template <int NV, int NT, int MAXCOL, int MAXSLOT>
__device__ __forceinline__ void nw_reduce_columns(const double *__restrict__ part, int nblk, double *s_acc /* [NT] */, double *__restrict__ sc, int slot0, int bid)
{
    static_assert(NT % NV == 0 && NT <= NW_BLOCK, "thread count must be a multiple of the row length");
    const int t = threadIdx.x;
    const int per = (nblk + NW_SPARTS - 1) / NW_SPARTS;
    const int r0 = min(bid * per, nblk), r1 = min(r0 + per, nblk);
    const double *__restrict__ p = part + (int64_t)r0 * NV;
    const int64_t total = (int64_t)(r1 - r0) * NV;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;       // four independent chains: loads in flight, fixed association
    if (t < NT) {
        int64_t e = t;
        if (MAXCOL >= 0 && t % NV == MAXCOL) {           // this thread's column holds maxima (all values >= 0)
            for (; e < total; e += NT) a0 = fmax(a0, p[e]);
            s_acc[t] = a0;
        } else {
            for (; e + 3 * NT < total; e += 4 * NT) { a0 += p[e]; a1 += p[e + NT]; a2 += p[e + 2 * NT]; a3 += p[e + 3 * NT]; }
            for (; e < total; e += NT) a0 += p[e];
            s_acc[t] = (a0 + a1) + (a2 + a3);
        }
    }
    __syncthreads();
    if (t < NV) {
        double s = 0.0;
        if (MAXCOL >= 0 && t == MAXCOL) { for (int k = t; k < NT; k += NV) s = fmax(s, s_acc[k]); }
        else { for (int k = t; k < NT; k += NV) s += s_acc[k]; }
        sc[((MAXCOL >= 0 && t == MAXCOL) ? MAXSLOT : slot0 + t) * NW_SPARTS + bid] = s;
    }
    __syncthreads();
}

__global__ __launch_bounds__(NW_BLOCK) void k_reduce_scalars(const double *__restrict__ part_a, int nblk_a, const double *__restrict__ part_p, int nblk_p,
                                                            const double *__restrict__ part_s, int nblk_s, double *__restrict__ sc,
                                                            const NwDevState *__restrict__ st, int it)
{
    if (st->iter_base + it >= st->stop_at) return;       // `it` = iteration of this search() block, iter_base = iterations before it
    __shared__ double s_acc[NW_BLOCK];
    // 3 x NW_SPARTS workgroups: the three tables are reduced side by side (one memory round trip instead of three in a row)
    const int table = (int)blockIdx.x / NW_SPARTS, bid = (int)blockIdx.x % NW_SPARTS;
    if (table == 0 && threadIdx.x == 0) sc[SC_STATUS * NW_SPARTS + bid] = (bid == 0 && st->status != 0) ? 1.0 : 0.0;
    if (table == 0) nw_reduce_columns<5, 255, 4, SC_MAXD>(part_a, nblk_a, s_acc, sc, SC_RES2, bid);   // k_attract: res^2, masked res^2, sum d, count; max d
    else if (table == 1) nw_reduce_columns<9, 252, -1, 0>(part_s, nblk_s, s_acc, sc, SC_HC, bid);     // k_subspace_point_sums: Hc (6), Gc (3)
    else nw_reduce_columns<14, 252, -1, 0>(part_p, nblk_p, s_acc, sc, SC_SS, bid);                    // k_prior_directions: LS^T LS (6), LS.prefs (3), |prefs|^2 (2), raw S.S (3)
}

// ============================================================================================================
// operators / read-back helpers
// ============================================================================================================
// y[perm[i]] = A x   (Afunc with the cached weight matrix)
__global__ void k_apply_A(int N, const int *__restrict__ perm, const int *__restrict__ vidx, const float *__restrict__ w, const float *__restrict__ x, float *__restrict__ y)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int o = perm[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float a = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) a = a + x[3 * vidx[3 * i + j] + c] * w[3 * i + j];
        y[3 * o + c] = a;
    }
}

// z += A^T r, r given in the caller's point order
__global__ void k_apply_At(int N, const int *__restrict__ perm, const int *__restrict__ vidx, const float *__restrict__ w, const float *__restrict__ r, float *__restrict__ z)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int o = perm[i];
    const float r0 = r[3 * o], r1 = r[3 * o + 1], r2 = r[3 * o + 2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float *a = z + 3 * (int64_t)vidx[3 * i + j];
        const float wj = w[3 * i + j];
        atomicAdd(a + 0, wj * r0);
        atomicAdd(a + 1, wj * r1);
        atomicAdd(a + 2, wj * r2);
    }
}

// un-permute per-point arrays (width floats or ints per point) back to the caller's point order
__global__ void k_unpermute(int N, int width, const int *__restrict__ perm, const uint32_t *__restrict__ in, uint32_t *__restrict__ out)
{
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= (int64_t)N * width) return;
    const int i = (int)(t / width), k = (int)(t % width);
    out[(int64_t)perm[i] * width + k] = in[t];
}

// ---- sharded mesh ('halo' mode): boundary rows <-> one dense buffer over the GLOBAL boundary list --------
// A boundary vertex (held by more than one rank) has one row in the dense buffer every rank all-reduces.  slot2local[s] = this rank's
// local vertex for global boundary slot s, or -1; b_local[k] / b_slot[k] = local vertex / slot of the k-th boundary vertex held here.
// pack: one thread per slot (rows this rank does not hold are zero); unpack: one thread per held boundary vertex.
__global__ __launch_bounds__(NW_BLOCK) void k_halo_pack_acc(int nslot, const int *__restrict__ slot2local, const long long *__restrict__ vacc, long long *__restrict__ buf)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslot) return;
    const int l = slot2local[s];
    longlong2 a = make_longlong2(0, 0), b = make_longlong2(0, 0);
    if (l >= 0) { a = *reinterpret_cast<const longlong2 *>(vacc + 4 * (int64_t)l); b = *reinterpret_cast<const longlong2 *>(vacc + 4 * (int64_t)l + 2); }
    *reinterpret_cast<longlong2 *>(buf + 4 * (int64_t)s) = a;
    *reinterpret_cast<longlong2 *>(buf + 4 * (int64_t)s + 2) = b;
}

__global__ __launch_bounds__(NW_BLOCK) void k_halo_unpack_acc(int n, const int *__restrict__ b_local, const int *__restrict__ b_slot, const long long *__restrict__ buf,
                                                             long long *__restrict__ vacc)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int l = b_local[k], s = b_slot[k];
    *reinterpret_cast<longlong2 *>(vacc + 4 * (int64_t)l) = *reinterpret_cast<const longlong2 *>(buf + 4 * (int64_t)s);
    *reinterpret_cast<longlong2 *>(vacc + 4 * (int64_t)l + 2) = *reinterpret_cast<const longlong2 *>(buf + 4 * (int64_t)s + 2);
}

// rows of a (M,3) float array (positions, normals): only the OWNER's value goes into the buffer, so that the sum over ranks is the
// owner's row and every holder takes it -- ghosts get their update, computed copies cannot drift
__global__ __launch_bounds__(NW_BLOCK) void k_halo_pack_rows(int nslot, const int *__restrict__ slot2local, const unsigned char *__restrict__ owned,
                                                            const float *__restrict__ rows, float *__restrict__ buf)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslot) return;
    const int l = slot2local[s];
    const bool mine = l >= 0 && owned[l];
    buf[3 * (int64_t)s] = mine ? rows[3 * (int64_t)l] : 0.0f;
    buf[3 * (int64_t)s + 1] = mine ? rows[3 * (int64_t)l + 1] : 0.0f;
    buf[3 * (int64_t)s + 2] = mine ? rows[3 * (int64_t)l + 2] : 0.0f;
}

__global__ __launch_bounds__(NW_BLOCK) void k_halo_unpack_rows(int n, const int *__restrict__ b_local, const int *__restrict__ b_slot, const float *__restrict__ buf,
                                                              float *__restrict__ dst, float *__restrict__ dst2 /* second copy (mesh positions) or NULL */)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int l = b_local[k], s = b_slot[k];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = buf[3 * (int64_t)s + c];
        dst[3 * (int64_t)l + c] = v;
        if (dst2) dst2[3 * (int64_t)l + c] = v;
    }
}

// ---- owner-wise exchange (nw_set_boundary with peers): rows of the exchange buffers are (peer, vertex) pairs, laid out peer after peer ----
// the partial sums a peer's copies collected for vertices this rank owns: added to the owner's rows (integers: any order gives the same sum;
// a vertex held by several peers has a row per peer)
__global__ __launch_bounds__(NW_BLOCK) void k_px_add_acc(int n, const int *__restrict__ ids, const long long *__restrict__ buf, long long *__restrict__ vacc)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t l = ids[k];
#pragma unroll
    for (int c = 0; c < 4; ++c) atomicAdd(reinterpret_cast<unsigned long long *>(vacc + 4 * l + c), (unsigned long long)buf[4 * (int64_t)k + c]);
}

__global__ __launch_bounds__(NW_BLOCK) void k_px_take_acc(int n, const int *__restrict__ ids, const long long *__restrict__ buf, long long *__restrict__ vacc)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t l = ids[k];
    *reinterpret_cast<longlong2 *>(vacc + 4 * l) = *reinterpret_cast<const longlong2 *>(buf + 4 * (int64_t)k);
    *reinterpret_cast<longlong2 *>(vacc + 4 * l + 2) = *reinterpret_cast<const longlong2 *>(buf + 4 * (int64_t)k + 2);
}

// The owners' SUMS go back as the four float32 the kernels convert an accumulator row to ((float)(sum * quantum): 16 B instead of 32).
// The copy stores the integer that converts to exactly that float: the quanta are powers of two, so f / q is an integer (a sum rounded
// to 24 bits is still a multiple of the quantum) and (float)((double)(f / q) * q) == f -- every holder computes with the same numbers.
__global__ __launch_bounds__(NW_BLOCK) void k_px_pack_acc_f32(int n, const int *__restrict__ ids, const long long *__restrict__ vacc, double q, double qw, float4 *__restrict__ buf)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t l = ids[k];
    const longlong2 a01 = *reinterpret_cast<const longlong2 *>(vacc + 4 * l), a23 = *reinterpret_cast<const longlong2 *>(vacc + 4 * l + 2);
    buf[k] = make_float4((float)((double)a01.x * q), (float)((double)a01.y * q), (float)((double)a23.x * q), (float)((double)a23.y * qw));
}

__global__ __launch_bounds__(NW_BLOCK) void k_px_take_acc_f32(int n, const int *__restrict__ ids, const float4 *__restrict__ buf, double inv_q, double inv_qw, long long *__restrict__ vacc)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t l = ids[k];
    const float4 f = buf[k];
    *reinterpret_cast<longlong2 *>(vacc + 4 * l) = make_longlong2((long long)((double)f.x * inv_q), (long long)((double)f.y * inv_q));
    *reinterpret_cast<longlong2 *>(vacc + 4 * l + 2) = make_longlong2((long long)((double)f.z * inv_q), (long long)((double)f.w * inv_qw));
}

__global__ __launch_bounds__(NW_BLOCK) void k_px_pack_rows(int n, const int *__restrict__ ids, const float *__restrict__ rows, float *__restrict__ buf)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t l = ids[k];
    buf[3 * (int64_t)k] = rows[3 * l]; buf[3 * (int64_t)k + 1] = rows[3 * l + 1]; buf[3 * (int64_t)k + 2] = rows[3 * l + 2];
}

__global__ __launch_bounds__(NW_BLOCK) void k_px_take_rows(int n, const int *__restrict__ ids, const float *__restrict__ buf, float *__restrict__ dst,
                                                          float *__restrict__ dst2 /* second copy (mesh positions) or NULL */)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t l = ids[k];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = buf[3 * (int64_t)k + c];
        dst[3 * l + c] = v;
        if (dst2) dst2[3 * l + c] = v;
    }
}

// the owners' rows of a (M_local,3) array into the (M_global,3) array that one all-reduce per block turns into the whole mesh
// (`full` zeroed beforehand; gv = global id of every local vertex)
__global__ __launch_bounds__(NW_BLOCK) void k_halo_gather_owned(int M, const int *__restrict__ gv, const unsigned char *__restrict__ owned, const float *__restrict__ rows,
                                                               float *__restrict__ full)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= M || !owned[v]) return;
    const int64_t g = gv[v];
    full[3 * g] = rows[3 * (int64_t)v]; full[3 * g + 1] = rows[3 * (int64_t)v + 1]; full[3 * g + 2] = rows[3 * (int64_t)v + 2];
}

// End-of-block statistics of a sharded mesh in one launch: stats = {largest nearest distance of the block (from the host's logs), this
// rank's accumulator quantum, max over the WHOLE mesh of |full - ref|^2} as float32 (a power-of-two quantum is exact in float32), for one
// MAX all-reduce over the ranks.  `full` = the whole mesh after the all-reduce of the owners' rows, `ref` = where the mesh was when the
// shares were cut.  stats[2] must be zero before the launch (non-negative floats order like unsigned integers: atomicMax on the bits).
// logs != NULL: max_dist is taken from the block's own iteration records on the device (the largest of the executed iterations) instead of
// from the host -- the block's tail is then enqueued behind its last iteration without waiting for the logs (nw_search on a sharded mesh)
// host_out (pinned host memory or NULL): the whole mesh is written there on the way -- the PCIe writes overlap this kernel's reads instead
// of a device-to-host copy queued behind it (the same trade as the single-GPU block's last update kernel, DESIGN section 3 "Block tail")
__global__ __launch_bounds__(NW_BLOCK) void k_halo_block_stats(int64_t n_global, const float *__restrict__ full, const float *__restrict__ ref, float max_dist, float quantum,
                                                              float *__restrict__ stats, const NwIterLogDev *__restrict__ logs, int nlogs, float *__restrict__ host_out)
{
    float m = 0.0f;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_global; v += (int64_t)gridDim.x * blockDim.x) {
        const float fx = full[3 * v], fy = full[3 * v + 1], fz = full[3 * v + 2];
        if (host_out) { host_out[3 * v] = fx; host_out[3 * v + 1] = fy; host_out[3 * v + 2] = fz; }
        const float dx = fx - ref[3 * v], dy = fy - ref[3 * v + 1], dz = fz - ref[3 * v + 2];
        const float d2 = dx * dx + dy * dy + dz * dz;
        m = d2 > m ? d2 : m;                               // (a NaN never wins: the block's NaN status is raised elsewhere)
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(reinterpret_cast<unsigned *>(stats) + 2, __float_as_uint(m));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (logs) { max_dist = 0.0f; for (int i = 0; i < nlogs; ++i) if (logs[i].executed) max_dist = fmaxf(max_dist, (float)logs[i].max_dist); }
        stats[0] = max_dist; stats[1] = quantum;
    }
}

// ---- block-boundary geometry refresh (the reference's `self.face_normals; self.vertex_neighbors` after a block,
// _membrane_mesh.pyx:1524-1527, for an unchanged topology): area-weighted vertex normals from the CURRENT device positions.
// Definition (this build's, PYME's is unpinned -- trimesh.py): n_v = normalise( sum over incident faces of
// (v1-v0) x (v2-v0) ), zero for vertices without faces.  Faces scatter into a 64-bit FIXED-POINT accumulator (quantum q, a power of
// two: 2^-44 of the squared scene extent, which bounds a cross product), so the sums are exact and do not depend on the order in
// which the faces arrive -- the normals feed the next block's curvature prior, and a fit must be reproducible bit for bit.
__global__ __launch_bounds__(NW_BLOCK) void k_normals_scatter(int F, const int *__restrict__ faces, const float *__restrict__ pos, long long *__restrict__ acc, double inv_q)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int a = faces[3 * f], b = faces[3 * f + 1], c = faces[3 * f + 2];
    const float ax = pos[3 * a], ay = pos[3 * a + 1], az = pos[3 * a + 2];
    const float ux = pos[3 * b] - ax, uy = pos[3 * b + 1] - ay, uz = pos[3 * b + 2] - az;
    const float vx = pos[3 * c] - ax, vy = pos[3 * c + 1] - ay, vz = pos[3 * c + 2] - az;
    const float cx = uy * vz - uz * vy, cy = uz * vx - ux * vz, cz = ux * vy - uy * vx;
    const long long qx = __double2ll_rn((double)cx * inv_q), qy = __double2ll_rn((double)cy * inv_q), qz = __double2ll_rn((double)cz * inv_q);
    const int vv[3] = {a, b, c};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        atomicAdd((unsigned long long *)&acc[3 * (int64_t)vv[k]], (unsigned long long)qx);
        atomicAdd((unsigned long long *)&acc[3 * (int64_t)vv[k] + 1], (unsigned long long)qy);
        atomicAdd((unsigned long long *)&acc[3 * (int64_t)vv[k] + 2], (unsigned long long)qz);
    }
}

__global__ __launch_bounds__(NW_BLOCK) void k_normals_finish(int M, const long long *__restrict__ acc, float *__restrict__ nrm)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= M) return;
    const double x = (double)acc[3 * (int64_t)v], y = (double)acc[3 * (int64_t)v + 1], z = (double)acc[3 * (int64_t)v + 2];      // (the quantum cancels)
    const double l = sqrt(x * x + y * y + z * z);
    const bool ok = l > 0.0 && isfinite(l);
    nrm[3 * v] = ok ? (float)(x / l) : 0.0f;
    nrm[3 * v + 1] = ok ? (float)(y / l) : 0.0f;
    nrm[3 * v + 2] = ok ? (float)(z / l) : 0.0f;
}

// alternate regularisers (gather forms are one thread per vertex; the scatter forms use float atomics except
// kind 1, whose in-place division makes it order dependent and is run by a single thread to keep the
// reference's serial semantics).
__global__ void k_lfunc_gather(int kind, int M, int NB, const int *__restrict__ nbr, const float *__restrict__ x, const float *__restrict__ f0, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int *row = nbr + (int64_t)i * NB;
    if (row[0] == -1) return;
    if (kind == 0) {           // conj_grad_utils.c:286-302
        for (int j = 0; j < 3; ++j) {
            float d = out[3 * i + j];
            int n_ = 0;
            for (int k = 0; k < NB; ++k) { const int n = row[k]; if (n == -1) break; d += (x[3 * n + j] - x[3 * i + j]); ++n_; }
            out[3 * i + j] = d / n_;
        }
    } else if (kind == 2) {    // conj_grad_utils.c:412-491
        float w = 0; int n_ = 0;
        for (int k = 0; k < NB; ++k) {
            const int n = row[k]; if (n == -1) break;
            float d2 = 0;
            for (int j = 0; j < 3; ++j) { const float dd = f0[3 * n + j] - f0[3 * i + j]; d2 += dd * dd; }
            w += d2; ++n_;
        }
        if (w > 0) {
            const float sw = sqrtf(w);
            for (int k = 0; k < n_; ++k) { const int n = row[k]; for (int j = 0; j < 3; ++j) out[3 * i + j] += (x[3 * n + j] - x[3 * i + j]) / sw; }
        }
    } else if (kind == 4) {    // conj_grad_utils.c:500-548
        float w = 0;
        for (int k = 0; k < NB; ++k) {
            const int n = row[k]; if (n == -1) break;
            float d2 = 0;
            for (int j = 0; j < 3; ++j) { const float dd = x[3 * n + j] - x[3 * i + j]; d2 += dd * dd; }
            w += d2;
        }
        w = (w > 0) ? (float)(1.0 / (double)sqrtf(w + 1)) : 0.0f;
        out[3 * i] = w; out[3 * i + 1] = w; out[3 * i + 2] = w;
    }
}

__global__ void k_lfunc_lhw(int M, int NB, const int *__restrict__ nbr, const float *__restrict__ x, const float *__restrict__ f0, float *__restrict__ out)
{
    // conj_grad_utils.c:628-697, scatter form
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int *row = nbr + (int64_t)i * NB;
    if (row[0] == -1) return;
    float w = 0; int n_ = 0;
    for (int k = 0; k < NB; ++k) {
        const int n = row[k]; if (n == -1) break;
        float d2 = 0;
        for (int j = 0; j < 3; ++j) { const float dd = f0[3 * i + j] - f0[3 * n + j]; d2 += dd * dd; }
        w += d2; ++n_;
    }
    if (w > 0) {
        const float sw = sqrtf(w);
        for (int k = 0; k < n_; ++k) { const int n = row[k]; for (int j = 0; j < 3; ++j) atomicAdd(&out[3 * n + j], (x[3 * i + j] - x[3 * n + j]) / sw); }
    }
}

// The same operator, one thread per TARGET vertex.  In the serial loop below a vertex n is only ever written while one of the vertices
// i that list it is visited -- out[n] = (out[n] + x[i] - x[n]) / deg(i) -- and those visits come in increasing i: the value of n is a
// fold over the vertices listing it, in index order, independent of every other vertex.  With a symmetric neighbour table (a manifold
// mesh whose rings fit the table) "the vertices listing n" is n's own row: every vertex folds for itself, same operations, same order,
// bit-identical to the serial walk.  A row that is not mirrored (or lists a vertex twice) raises *asym: the host falls back to the
// serial kernel.
__global__ __launch_bounds__(NW_BLOCK) void k_lfunc_lh_gather(int M, int NB, const int *__restrict__ nbr, const float *__restrict__ x, const float *__restrict__ out_in,
                                                             float *__restrict__ out, int *__restrict__ asym)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= M) return;
    float v[3] = {out_in[3 * n], out_in[3 * n + 1], out_in[3 * n + 2]};
    const int *row = nbr + (int64_t)n * NB;
    int ids[64], cnt = 0;
    for (int k = 0; k < NB && k < 64; ++k) { const int i = row[k]; if (i == -1) break; ids[cnt++] = i; }
    if (NB > 64) *asym = 1;
    for (int a = 1; a < cnt; ++a) {                       // ascending visiting order
        const int key = ids[a];
        int b = a - 1;
        while (b >= 0 && ids[b] > key) { ids[b + 1] = ids[b]; --b; }
        ids[b + 1] = key;
    }
    const float xn[3] = {x[3 * n], x[3 * n + 1], x[3 * n + 2]};
    for (int a = 0; a < cnt; ++a) {
        const int i = ids[a];
        if ((unsigned)i >= (unsigned)M || i == n || (a > 0 && ids[a - 1] == i)) { *asym = 1; continue; }
        const int *ri = nbr + (int64_t)i * NB;
        int deg = 0, hits = 0;
        for (int k = 0; k < NB; ++k) { const int m = ri[k]; if (m == -1) break; ++deg; hits += (m == n); }
        if (hits != 1) { *asym = 1; continue; }
#pragma unroll
        for (int j = 0; j < 3; ++j) { v[j] += (x[3 * i + j] - xn[j]); v[j] /= deg; }
    }
    out[3 * n] = v[0]; out[3 * n + 1] = v[1]; out[3 * n + 2] = v[2];
}

__global__ void k_lfunc_lh_serial(int M, int NB, const int *__restrict__ nbr, const float *__restrict__ x, float *__restrict__ out)
{
    // conj_grad_utils.c:344-364: the per-vertex in-place "/= N" on the neighbours makes the result depend on the
    // vertex visiting order, so the reference's serial order is kept (not on the default path; M-sized, one lane).
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (int i = 0; i < M; ++i) {
        const int *row = nbr + (int64_t)i * NB;
        if (row[0] == -1) continue;
        for (int j = 0; j < 3; ++j) {
            int n_ = 0;
            for (int k = 0; k < NB; ++k) { const int n = row[k]; if (n == -1) break; out[3 * n + j] += (x[3 * i + j] - x[3 * n + j]); ++n_; }
            for (int k = 0; k < n_; ++k) { const int n = row[k]; out[3 * n + j] /= n_; }
        }
    }
}

// ============================================================================================================
// Block-boundary kernel: per-vertex curvature tensor / Canham-Helfrich energy (c_curvature_grad,
// membrane_mesh_utils.c:915-1250; Householder+Givens eigen-solve :618-720; 2x2 pseudo-inverse :841-890).
// One thread per vertex, three passes over the <= 20 1-ring neighbours, float64 internals / float32 outputs exactly
// where the reference has them.  Deviations from the reference's arithmetic (all at the 1e-15 level): the least-squares
// system A^T A, A^T b is accumulated on the fly instead of materialising A (no 2x20 scratch arrays), and the random
// jitter (rand(), :1017) is either supplied by the caller or replaced by a counter-based hash of (vertex, axis).
// ============================================================================================================
__device__ __forceinline__ double nw_jitter_unit(unsigned v, unsigned k)
{
    unsigned x = v * 0x9E3779B1u + k * 0x85EBCA77u + 0x165667B1u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (double)(x >> 8) / 16777216.0;
}

__device__ __forceinline__ double nw_norm3d(double a, double b, double c) { return sqrt((a * a + b * b) + c * c); }
__device__ __forceinline__ double nw_safe_div(double x, double y) { return (fabs(y) < 1e-15) ? 0.0 : x / y; }

__device__ __forceinline__ void nw_proj_matrix(float vx, float vy, float vz, double coef, double *m)
{
    const double v0 = vx, v1 = vy, v2 = vz;
    const float xy = (float)(-1.0 * coef * v0 * v1), xz = (float)(-1.0 * coef * v0 * v2), yz = (float)(-1.0 * coef * v1 * v2);   // float in the reference (:233)
    m[0] = 1.0 - coef * v0 * v0; m[1] = xy; m[2] = xz;
    m[3] = xy; m[4] = 1.0 - coef * v1 * v1; m[5] = yz;
    m[6] = xz; m[7] = yz; m[8] = 1.0 - coef * v2 * v2;
}

__device__ __forceinline__ double nw_normal_diff(double q)
{
    q = q * q;
    return (q > 1.0) ? sqrt(2.0) : sqrt(2.0 - 2.0 * sqrt(1.0 - q));
}

__global__ __launch_bounds__(128) void k_curvature(int M, int NB, const float *__restrict__ pos, const float *__restrict__ nrm, const unsigned char *__restrict__ valid,
                                                  const int *__restrict__ nbr, const int *__restrict__ nbr_next, const float *__restrict__ nbr_area,
                                                  const double *__restrict__ jitter, float dN, float kc, float kg, float c0,
                                                  float *__restrict__ k_0, float *__restrict__ k_1, float *__restrict__ e_0, float *__restrict__ e_1,
                                                  float *__restrict__ H, float *__restrict__ K, float *__restrict__ dH, float *__restrict__ dK,
                                                  float *__restrict__ E, float *__restrict__ pE, float *__restrict__ dE_nb, float *__restrict__ dEdN)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    if (valid && !valid[i]) {
        H[i] = K[i] = dH[i] = dK[i] = dE_nb[i] = E[i] = pE[i] = 0.0f;
        dEdN[3 * i] = dEdN[3 * i + 1] = dEdN[3 * i + 2] = 0.0f;
        return;
    }
    const float vix = pos[3 * i], viy = pos[3 * i + 1], viz = pos[3 * i + 2];
    const float Nx = nrm[3 * i], Ny = nrm[3 * i + 1], Nz = nrm[3 * i + 2];
    const int *row = nbr + (int64_t)i * NB;
    // pass 1: radial weights, neighbour centroid, jitter width
    float cx = 0.f, cy = 0.f, cz = 0.f;
    double r_sum = 0.0, jw = 10000000000000000.0;
    int n = 0;
    while (n < NB && row[n] != -1) {
        const int v = row[n];
        const float x = pos[3 * v], y = pos[3 * v + 1], z = pos[3 * v + 2];
        cx += x; cy += y; cz += z;
        const double dn = nw_norm3d((double)x - (double)vix, (double)y - (double)viy, (double)z - (double)viz);
        if (dn > 1e-15) r_sum += 1.0 / dn;
        if (dn < jw) jw = dn;
        ++n;
    }
    cx /= n; cy /= n; cz /= n;
    {
        const double j0 = jitter ? jitter[3 * i] : nw_jitter_unit((unsigned)i, 0u);
        const double j1 = jitter ? jitter[3 * i + 1] : nw_jitter_unit((unsigned)i, 1u);
        const double j2 = jitter ? jitter[3 * i + 2] : nw_jitter_unit((unsigned)i, 2u);
        cx = (float)((double)cx + jw * (j0 - 0.5));
        cy = (float)((double)cy + jw * (j1 - 0.5));
        cz = (float)((double)cz + jw * (j2 - 0.5));
    }
    float ux = cx - vix, uy = cy - viy, uz = cz - viz;                  // vivj
    float un = 0.0f;
    un += ux * ux; un += uy * uy; un += uz * uz;
    const float vivj_norm = (float)sqrt((double)un);
    if (vivj_norm > 0.0f) { ux /= vivj_norm; uy /= vivj_norm; uz /= vivj_norm; } else { ux = uy = uz = 0.0f; }
    const double sx = (double)ux * (double)dN, sy = (double)uy * (double)dN, sz = (double)uz * (double)dN;   // NvidN
    const double wx = (double)vix - sx, wy = (double)viy - sy, wz = (double)viz - sz;                       // viNvidN
    double p[9];
    nw_proj_matrix(Nx, Ny, Nz, 1.0, p);
    double M00 = 0, M01 = 0, M02 = 0, M10 = 0, M11 = 0, M12 = 0, M20 = 0, M21 = 0, M22 = 0;
    double dareas = 0.0, areas = 0.0;
    double hx = 0, hy = 0, hz = 0, gx = 0, gy = 0, gz = 0;            // dv_hat, dv_1_hat persist across neighbours (:1059-1062)
    float den = 0.0f;
    // pass 2: curvature tensor
    for (int j = 0; j < n; ++j) {
        const int v = row[j];
        const double dx = (double)pos[3 * v] - (double)vix, dy = (double)pos[3 * v + 1] - (double)viy, dz = (double)pos[3 * v + 2] - (double)viz;
        const double d1x = dx - sx, d1y = dy - sy, d1z = dz - sz;
        const double dn = nw_norm3d(dx, dy, dz), dn1 = nw_norm3d(d1x, d1y, d1z);
        if (dn > 1e-15) { hx = dx / dn; hy = dy / dn; hz = dz / dn; }
        if (dn1 > 1e-15) { gx = d1x / dn1; gy = d1y / dn1; gz = d1z / dn1; }
        const double nx_ = dx * -1.0, ny_ = dy * -1.0, nz_ = dz * -1.0;
        const double Tx = (p[0] * nx_ + p[1] * ny_) + p[2] * nz_, Ty = (p[3] * nx_ + p[4] * ny_) + p[5] * nz_, Tz = (p[6] * nx_ + p[7] * ny_) + p[8] * nz_;
        const double Tn = nw_norm3d(Tx, Ty, Tz);
        double tx = 0, ty = 0, tz = 0;
        if (Tn > 1e-15) { tx = Tx / Tn; ty = Ty / Tn; tz = Tz / Tn; }
        const float jx = nrm[3 * v], jy = nrm[3 * v + 1], jz = nrm[3 * v + 2];
        const double Ni_diff = nw_normal_diff(((double)Nx * hx + (double)Ny * hy) + (double)Nz * hz);
        const double Nj_diff = nw_normal_diff(((double)jx * hx + (double)jy * hy) + (double)jz * hz);
        const double Nj1_diff = nw_normal_diff(((double)jx * gx + (double)jy * gy) + (double)jz * gz);
        const double kj = nw_safe_div(2.0 * Nj_diff, dn), kj1 = nw_safe_div(2.0 * Nj1_diff, dn1);
        const double w = nw_safe_div(nw_safe_div(1.0, dn), r_sum);
        const double sgn = ((((double)Nx * nx_ + (double)Ny * ny_) + (double)Nz * nz_) < 0) ? -1.0 : 1.0;
        const double k = nw_safe_div(2.0 * sgn * Ni_diff, dn);
        const double Aj = nbr_area[(int64_t)i * NB + j];
        const int vn = nbr_next[(int64_t)i * NB + j];
        const double ex = (double)pos[3 * vn] - wx, ey = (double)pos[3 * vn + 1] - wy, ez = (double)pos[3 * vn + 2] - wz;
        const double crx = d1y * ez - d1z * ey, cry = d1z * ex - d1x * ez, crz = d1x * ey - d1y * ex;
        const double dAj = 0.5 * nw_norm3d(crx, cry, crz);
        dareas += dAj;
        areas += Aj;
        const double a0 = 2.0 * kj - (double)c0, a1 = 2.0 * kj1 - (double)c0;
        den += ((float)((Aj * w * 0.5 * (double)kc * (a0 * a0) - dAj * w * 0.5 * (double)kc * (a1 * a1)))) / dN;
        const double wk = w * k;
        M00 += (tx * tx) * wk; M01 += (tx * ty) * wk; M02 += (tx * tz) * wk;
        M10 += (ty * tx) * wk; M11 += (ty * ty) * wk; M12 += (ty * tz) * wk;
        M20 += (tz * tx) * wk; M21 += (tz * ty) * wk; M22 += (tz * tz) * wk;
    }
    dE_nb[i] = den;
    // eigen-solve: Householder reflection taking the normal to +-e1, then one Givens rotation on the 2x2 minor
    double l1, l2, v1x, v1y, v1z, v2x, v2y, v2z;
    {
        const float sxf = 1.0f - Nx, syf = 0.0f - Ny, szf = 0.0f - Nz, axf = 1.0f + Nx, ayf = 0.0f + Ny, azf = 0.0f + Nz;
        float ns = 0.0f, na = 0.0f;
        ns += sxf * sxf; ns += syf * syf; ns += szf * szf;
        na += axf * axf; na += ayf * ayf; na += azf * azf;
        const float nsf = (float)sqrt((double)ns), naf = (float)sqrt((double)na);
        float Wx, Wy, Wz;
        if (nsf > naf) { Wx = sxf / nsf; Wy = syf / nsf; Wz = szf / nsf; } else { Wx = axf / naf; Wy = ayf / naf; Wz = azf / naf; }
        double Q[9];
        nw_proj_matrix(Wx, Wy, Wz, 2.0, Q);
        const double Mv[9] = {M00, M01, M02, M10, M11, M12, M20, M21, M22};
        double QM[9], R[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) { double a = 0.0; for (int kk = 0; kk < 3; ++kk) a += Q[3 * r + kk] * Mv[3 * kk + c]; QM[3 * r + c] = a; }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) { double a = 0.0; for (int kk = 0; kk < 3; ++kk) a += QM[3 * r + kk] * Q[3 * c + kk]; R[3 * r + c] = a; }   // * Q^T
        const double tau = nw_safe_div(R[8] - R[4], 2.0 * R[5]);
        const double t = ((tau < 0) ? -1.0 : 1.0) / (fabs(tau) + sqrt(1 + tau * tau));
        l1 = R[4] - t * R[5];
        l2 = R[8] + t * R[5];
        const double cs = 1.0 / sqrt(1 + t * t), sn = t * cs;
        // QviT[3i+1] = Q[3+i], QviT[3i+2] = Q[6+i]
        const double ax = cs * Q[3] - sn * Q[6], ay = cs * Q[4] - sn * Q[7], az = cs * Q[5] - sn * Q[8];
        const double bx = sn * Q[3] + cs * Q[6], by = sn * Q[4] + cs * Q[7], bz = sn * Q[5] + cs * Q[8];
        if (l1 > l2) { v1x = ax; v1y = ay; v1z = az; v2x = bx; v2y = by; v2z = bz; }
        else { const double tmp = l1; l1 = l2; l2 = tmp; v2x = ax; v2y = ay; v2z = az; v1x = bx; v1y = by; v1z = bz; }
    }
    float k0f, k1f;
    if (isnan(l1)) { k0f = 0.0f; k1f = 0.0f; v1x = v1y = v1z = v2x = v2y = v2z = 0.0; }
    else { k0f = (float)(3.0 * l1 - l2); k1f = (float)(3.0 * l2 - l1); }
    k_0[i] = k0f; k_1[i] = k1f;
    e_0[3 * i] = (float)v1x; e_0[3 * i + 1] = (float)v1y; e_0[3 * i + 2] = (float)v1z;
    e_1[3 * i] = (float)v2x; e_1[3 * i + 1] = (float)v2y; e_1[3 * i + 2] = (float)v2z;
    const float Hf = (float)(0.5 * (double)(k0f + k1f));
    const float Kf = (float)(k0f * k1f);
    H[i] = Hf; K[i] = Kf;
    // pass 3: quadric fit in the (e0,e1) plane displaced by dN -> principal curvatures after the shift.  Same order of operations as
    // the reference's matmul chain (membrane_mesh_utils.c:1165-1187): A^T A by rows of A in neighbour order, its 2x2 pseudo-inverse,
    // then (pinv A^T)[.][j] b[j] summed in neighbour order (second walk over the ring; the unused slots of the fixed 2x20 arrays
    // contribute exact zeros there).
    double S00 = 0, S01 = 0, S11 = 0;
    for (int j = 0; j < n; ++j) {
        const int v = row[j];
        const double dx = (double)pos[3 * v] - (double)vix, dy = (double)pos[3 * v + 1] - (double)viy, dz = (double)pos[3 * v + 2] - (double)viz;
        const double t0 = (dx * v1x + dy * v1y) + dz * v1z, t1 = (dx * v2x + dy * v2y) + dz * v2z;
        const double A0 = t0 * t0, A1 = t1 * t1;
        S00 += A0 * A0; S01 += A0 * A1; S11 += A1 * A1;
    }
    double kp0, kp1;
    {
        const double a = S00, b = S01, c = S01, d = S11;
        const double a2 = a * a, b2 = b * b, c2 = c * c, d2 = d * d;
        const double a2b2 = a2 + b2, c2d2 = c2 + d2, dif = a2b2 - c2d2, tacbd = 2 * (a * c + b * d);
        const double theta = 0.5 * atan2(2 * (a * b + c * d), a2 + c2 - b2 - d2), phi = 0.5 * atan2(tacbd, dif);
        const double ct = cos(theta), cp = cos(phi), st = sin(theta), sp = sin(phi);
        const double ctcp = ct * cp, ctsp = ct * sp, stcp = st * cp, stsp = st * sp;
        const double sign0 = ((ctcp * a + ctsp * c + stcp * b + stsp * d) < 0) ? -1.0 : 1.0;
        const double sign1 = ((stsp * a - stcp * c - ctsp * b + ctcp * d) < 0) ? -1.0 : 1.0;
        const double ss = a2b2 + c2d2, sd = sqrt(dif * dif + tacbd * tacbd);
        const double sig0 = sqrt((ss + sd) / 2.0), sssd = ss - sd;
        const double sig1 = (sssd > 0) ? sqrt(sssd / 2.0) : 0.0;
        const double thresh = (1e-8) * 0.5 * sqrt(5.0) * sig0;
        const double si0 = (sig0 < thresh) ? 0.0 : (1.0 / sig0), si1 = (sig1 < thresh) ? 0.0 : (1.0 / sig1);
        const double s0 = sign0 * si0, s1 = sign1 * si1;
        const double i00 = ctcp * s0 + stsp * s1, i01 = ctsp * s0 - stcp * s1, i10 = stcp * s0 - ctsp * s1, i11 = stsp * s0 + ctcp * s1;
        kp0 = 0.0; kp1 = 0.0;
        for (int j = 0; j < n; ++j) {
            const int v = row[j];
            const double dx = (double)pos[3 * v] - (double)vix, dy = (double)pos[3 * v + 1] - (double)viy, dz = (double)pos[3 * v + 2] - (double)viz;
            const double t0 = (dx * v1x + dy * v1y) + dz * v1z, t1 = (dx * v2x + dy * v2y) + dz * v2z;
            const double A0 = t0 * t0, A1 = t1 * t1;
            const double b = A0 * (double)k0f + A1 * (double)k1f - (double)dN;
            kp0 += (i00 * A0 + i01 * A1) * b;
            kp1 += (i10 * A0 + i11 * A1) * b;
        }
    }
    const float dHf = (float)(0.5 * (kp0 + kp1)), dKf = (float)(kp0 * kp1);
    dH[i] = dHf; dK[i] = dKf;
    const double hh = 2.0 * (double)Hf - (double)c0;
    const float Ef = (float)(areas * ((0.5 * (double)kc * (hh * hh) + (double)kg * (double)Kf)));
    E[i] = Ef;
    const float pEf = (float)exp(-(1.0 / 0.0257) * (double)Ef);
    pE[i] = pEf;
    const double dh = 2.0 * (double)dHf - (double)c0;
    const double dEdN_H = dareas * ((0.5 * (double)kc * (dh * dh) + (double)kg * (double)dKf));
    const double dsum = ((double)Ef - dEdN_H) / (double)dN + (double)den;
    const double lo = -0.5 * (double)vivj_norm, hi = 0.5 * (double)vivj_norm;
    const float dEdNs = (float)(-1.0 * (double)((float)((dsum > hi) ? hi : ((dsum < lo) ? lo : dsum))) * (1.0 - (double)pEf));
    dEdN[3 * i] = dEdNs * ux; dEdN[3 * i + 1] = dEdNs * uy; dEdN[3 * i + 2] = dEdNs * uz;
}
