// Exact nearest-face-centroid query, wave-autonomous form (replaces cKDTree build + query, mesh_conj_grad.py:451-454:
// exact Euclidean 1-NN in float64).  MI355X (gfx950), wave64.  Wave-private LDS only, no workgroup barriers, no MFMA.
//
// Layout.  Localizations are sorted ONCE (nw_set_points) by the 30-bit Morton code of their position inside the cloud's
// bounding cube, and once more after the first block by the Morton code of their FOOT POINT (centroid of the nearest face):
// every contiguous run of the list is then compact along the surface whatever the height above it.  A work item is a run of
// <= 64 consecutive localizations that does not leave one aligned Morton block (edge ~4 fine cells); one WAVE owns one item,
// lane = localization.  Face centroids are binned per iteration into the fine cells of the uniform grid (cell-sorted float4
// {x, y, z, face id}, x fastest: a (z,y) row of cells is one contiguous candidate range, `cstart` = dense cell table).
//
// Walk.  Every lane keeps its best key b1 (and runner-up b2).  The ball of radius sqrt(d1) around the localization bounds
// where a closer centroid can be; the wave visits the bounding box (in cells) of its lanes' balls:
//   * lane = SEGMENT of NW_SEG cells of one (z,y) row: nine gathers from `cstart` (all in flight together) tell which of the
//     segment's cells hold centroids that have not been visited yet;
//   * lane = LOCALIZATION again: a non-empty segment is kept only if some lane's ball reaches the row's (y,z) square, and a
//     non-empty CELL only if some lane's ball reaches the cell box (exact box-distance tests against the lane's CURRENT best);
//     surviving cells join into runs and their candidate ranges are COLLECTED in a wave-private LDS list (NwWaveLds::rs/pre);
//   * nw_stream evaluates the list in batches of 64: lane = i-th candidate of the list (binary search in the prefix counts,
//     one independent load per lane, the next batch's loads in flight during the current batch), staged in LDS in expanded
//     form, then lane = localization reads every staged candidate (broadcast ds_read_b128) and evaluates it.
// Warm start: the previous iteration's nearest face (the sorted order never changes) gives every lane a tight starting radius,
// so one visit of the box settles the wave.  Cold start (first iteration, new topology): own cell +- 1, then the ball box of
// what that found; lanes that have seen no candidate yet double their margin until the box covers the grid.
// A lane is final once its ball lies inside the visited box; the visited region only grows and the radii only shrink.
//
// Arithmetic.  Wave-local coordinates p' = p - O, c' = c - O (O = mean of the wave's localizations), K = max |p'|^2.  The staged
// candidate is {-2x', -2y', -2z', |c'|^2 + K}:  key = fma(px', X, fma(py', Y, fma(pz', Z, W))) = |p - c|^2 + (K - |p'|^2) >= 0,
// kept as an unsigned integer (float bits, low 4 mantissa bits = position in the 16-chunk): v_and_or_b32, v_med3_u32, v_min_u32.
// Cells are culled against the best + tol (+ the rounding slack of the cell assignment), so every centroid whose key is
// within the error band of the winner IS evaluated and is seen by the runner-up; a lane whose runner-up lies within 2 tol of
// its best is re-resolved in float64 (< 1 % of the lanes) at the end of the kernel: the whole wave over that one ball, lane =
// candidate (nw_fixup_point).  The result is the float64 argmin for every localization (lowest face id on exact ties; SciPy's tie
// order is unspecified).
//
// Outliers.  Up to NW_OUTLIERS lanes of a wave whose warm radius is far above the wave's (background noise far from the surface)
// are taken out of the walk and out of the local frame and resolved like the ambiguous ones.
//
// Work list.  The kernel times every item once; the host then puts the heavy items first (two classes, each in spatial order) and,
// for clouds too small to fill the GPU twice, cuts them into pieces (order_items_by_cost, nanowrap.hip).
#pragma once
#include "nw_device.h"

// Error bound of a key (expanded form on wave-local coordinates, |p'|^2 <= K, candidate within d of p): every intermediate of
//   fma(px', X, fma(py', Y, fma(pz', Z, W))),  W = |c'|^2 + K,   is bounded by |c'|^2 + K + 2 |p'| |c'| <= 5 K + 4 d^2 <= 5 K + 4 key,
// three fused roundings + the roundings of the staged operands (2^-24 each, ~8 in all) give < 2.4e-6 K + 1.9e-6 key, the index bits
// another 16 ulp = 1.9e-6 key.  With margin:   tol(key) = NW_NN_TOLK * K + NW_NN_TOL * key   (absolute, in units of d^2).
// NB the K term: a lane at the edge of its wave has key ~ d^2 << K, so a tolerance relative to the key alone is NOT a bound
// (seen as a few hundred float64 ties resolved to the wrong face id at 5M localizations).
#define NW_NN_TOL 6e-6f
#define NW_NN_TOLK 4e-6f
#define NW_NN_CULL 4e-6f          // relative slack of the cell-culling test (rounding of the box distance)
#define NW_ITEM_POINTS 64
#define NW_OUTLIERS 8               // lanes per wave that may be resolved on their own (k_nn_wave)
#define NW_NNA_HT 256                // slots of the scatter table of an attraction workgroup inside the query launch (two items: <= 128 localizations)
#define NW_NNA_HT_BITS 8

struct NwItem { int p0, n; };

// ---- set-up: Morton keys, work items ------------------------------------------------------------------------
__device__ __forceinline__ unsigned nw_spread10(unsigned v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// key = Morton code of the position quantised to 1024^3 over the cloud's bounding cube (edge `ext`, lower corner lo)
__global__ void k_morton_keys(const float *__restrict__ xyz, int N, float lox, float loy, float loz, float inv_unit, unsigned *__restrict__ key, int *__restrict__ idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int qx = nw_clampi((int)((xyz[3 * i] - lox) * inv_unit), 0, 1023);
    const int qy = nw_clampi((int)((xyz[3 * i + 1] - loy) * inv_unit), 0, 1023);
    const int qz = nw_clampi((int)((xyz[3 * i + 2] - loz) * inv_unit), 0, 1023);
    key[i] = nw_spread10((unsigned)qx) | (nw_spread10((unsigned)qy) << 1) | (nw_spread10((unsigned)qz) << 2);
    idx[i] = i;
}

// Internal order of the faces (nw_set_mesh): key = Morton code of the face centroid, like the localizations'.  The caller's face ids are
// whatever a mesher or seven remeshing passes left behind; every kernel that walks faces or gathers by face id (centroid binning, the
// warm start of the query, the attraction step) wants neighbours in space to be neighbours in memory.
__global__ void k_face_morton_keys(const float *__restrict__ pos, const int *__restrict__ faces, int F, int M, float lox, float loy, float loz, float inv_unit,
                                   unsigned *__restrict__ key, int *__restrict__ idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F) return;
    float c[3] = {0.0f, 0.0f, 0.0f};
    for (int k = 0; k < 3; ++k) {
        const int v = faces[3 * i + k];
        if ((unsigned)v < (unsigned)M)
            for (int a = 0; a < 3; ++a) c[a] += pos[3 * v + a];
    }
    const float third = 1.0f / 3.0f;
    const float fx = (c[0] * third - lox) * inv_unit, fy = (c[1] * third - loy) * inv_unit, fz = (c[2] * third - loz) * inv_unit;
    // (a non-finite centroid sorts to the front: any order is a valid one)
    const int qx = fx == fx ? nw_clampi((int)fminf(fmaxf(fx, 0.0f), 1023.0f), 0, 1023) : 0;
    const int qy = fy == fy ? nw_clampi((int)fminf(fmaxf(fy, 0.0f), 1023.0f), 0, 1023) : 0;
    const int qz = fz == fz ? nw_clampi((int)fminf(fmaxf(fz, 0.0f), 1023.0f), 0, 1023) : 0;
    key[i] = nw_spread10((unsigned)qx) | (nw_spread10((unsigned)qy) << 1) | (nw_spread10((unsigned)qz) << 2);
    idx[i] = i;
}

__global__ void k_face_regather(int F, const int *__restrict__ order, const int *__restrict__ faces_in, int *__restrict__ faces_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F) return;
    const int o = order[i];
    faces_out[3 * i] = faces_in[3 * o]; faces_out[3 * i + 1] = faces_in[3 * o + 1]; faces_out[3 * i + 2] = faces_in[3 * o + 2];
}

// internal face ids -> the caller's (nw_get(NW_ARR_FACE)); ids outside [0, F) are left alone
__global__ void k_translate_ids(int n, int *__restrict__ ids, const int *__restrict__ orig, int F)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int f = ids[i];
    if ((unsigned)f < (unsigned)F) ids[i] = orig[f];
}

// "lower face id" of the tie rule, in the CALLER's numbering (orig = internal -> caller, NULL = the same); the sentinel loses
__device__ __forceinline__ bool nw_id_less(int a, int b, const int *__restrict__ orig)
{
    if (orig && a != 0x7fffffff && b != 0x7fffffff) return orig[a] < orig[b];
    return a < b;
}

// gather the localizations into sorted order, baking the residual weighting of search() (mesh_conj_grad.py:156-164):
// weights = weights / weights.mean(), mask = weights > 0 (array) or isfinite(data) (scalar)
__global__ void k_point_gather(int N, const float *__restrict__ xyz, const int *__restrict__ order,
                               const float *__restrict__ sinv_in, const float *__restrict__ w_in, const double *__restrict__ wsum, int w_is_array,
                               float4 *__restrict__ pts, float *__restrict__ sinv, float *__restrict__ wnorm, unsigned char *__restrict__ mask)
{
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= N) return;
    const int i = order[slot];
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    pts[slot] = make_float4(x, y, z, __int_as_float(i));
    if (sinv_in)
        for (int k = 0; k < 3; ++k) sinv[3 * slot + k] = sinv_in[3 * i + k];
    unsigned m = 0;
    if (w_is_array) {
        const float mean = (float)(wsum[0] / (3.0 * (double)N));
        for (int k = 0; k < 3; ++k) {
            const float w = w_in[3 * i + k];
            wnorm[3 * slot + k] = (w_is_array == 2) ? w : w / mean;      // 2: already divided by the global mean
            m |= (w > 0.0f) ? (1u << k) : 0u;
        }
    } else {
        m = (isfinite(x) ? 1u : 0u) | (isfinite(y) ? 2u : 0u) | (isfinite(z) ? 4u : 0u);
    }
    mask[slot] = (unsigned char)m;
}

// Projection order (second sort, once per cloud after the first completed query): key = Morton code of the CENTROID of the
// localization's nearest face, i.e. of its foot point on the surface.  Localizations over the same patch of surface become
// neighbours in the list whatever their height above it, so a wave's 64 localizations share one small set of candidate cells
// (in the plain 3-D Morton order a wave's box is as wide as its localizations are scattered in height).  Speed only.
__global__ void k_projection_keys(int N, const int *__restrict__ face, int F, const float4 *__restrict__ cent_by_face, const float4 *__restrict__ pts,
                                  float lox, float loy, float loz, float inv_unit, unsigned *__restrict__ key, int *__restrict__ idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int f = face[i];
    const float4 C = ((unsigned)f < (unsigned)F) ? cent_by_face[f] : pts[i];
    const int qx = nw_clampi((int)((C.x - lox) * inv_unit), 0, 1023);
    const int qy = nw_clampi((int)((C.y - loy) * inv_unit), 0, 1023);
    const int qz = nw_clampi((int)((C.z - loz) * inv_unit), 0, 1023);
    key[i] = nw_spread10((unsigned)qx) | (nw_spread10((unsigned)qy) << 1) | (nw_spread10((unsigned)qz) << 2);
    idx[i] = i;
}

// move the per-localization inputs (and the warm-start faces) from the old sorted order to the new one: new slot s <- old slot order[s]
__global__ void k_point_regather(int N, const int *__restrict__ order, const float4 *__restrict__ pts_o, const int *__restrict__ perm_o,
                                 const float *__restrict__ sinv_o, const float *__restrict__ wnorm_o, const unsigned char *__restrict__ mask_o, const int *__restrict__ face_o,
                                 float4 *__restrict__ pts, int *__restrict__ perm, float *__restrict__ sinv, float *__restrict__ wnorm, unsigned char *__restrict__ mask,
                                 int *__restrict__ face)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const int o = order[s];
    pts[s] = pts_o[o];
    perm[s] = perm_o[o];
    mask[s] = mask_o[o];
    face[s] = face_o[o];
    if (sinv_o)
        for (int k = 0; k < 3; ++k) sinv[3 * s + k] = sinv_o[3 * o + k];
    if (wnorm_o)
        for (int k = 0; k < 3; ++k) wnorm[3 * s + k] = wnorm_o[3 * o + k];
}

// search(data, ...) with `data` other than the localizations (mesh_conj_grad.py:164, 180-181, 222): the residual's target in sorted
// order, and -- for scalar weights -- the mask isfinite(data) (:164).  data_in == NULL restores the mask of the localizations.
__global__ void k_data_gather(int N, const int *__restrict__ perm, const float *__restrict__ data_in, const float4 *__restrict__ pts, float *__restrict__ dat,
                              unsigned char *__restrict__ mask /* NULL: weights array decides */)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= N) return;
    const int o = perm[s];
    float d[3];
    if (data_in) { d[0] = data_in[3 * o]; d[1] = data_in[3 * o + 1]; d[2] = data_in[3 * o + 2]; dat[3 * s] = d[0]; dat[3 * s + 1] = d[1]; dat[3 * s + 2] = d[2]; }
    else { const float4 P = pts[s]; d[0] = P.x; d[1] = P.y; d[2] = P.z; }
    if (mask) mask[s] = (unsigned char)((isfinite(d[0]) ? 1u : 0u) | (isfinite(d[1]) ? 2u : 0u) | (isfinite(d[2]) ? 4u : 0u));
}

// block heads of the sorted key list: head[i] = 1 where the aligned Morton block (key >> shift) changes
__global__ void k_block_heads(const unsigned *__restrict__ key, int N, int shift, int *__restrict__ head)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    head[i] = (i == 0 || (key[i] >> shift) != (key[i - 1] >> shift)) ? 1 : 0;
}

// bid = exclusive scan of head (+ head) - 1; every head records where its block starts
__global__ void k_block_starts(const int *__restrict__ head, const int *__restrict__ scan, int N, int *__restrict__ bstart)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (head[i]) bstart[scan[i]] = i;
    if (i == N - 1) bstart[scan[i] + head[i]] = N;
}

__global__ void k_block_item_counts(const int *__restrict__ bstart, int nblocks, int *__restrict__ nitems, int item_points)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    nitems[b] = (bstart[b + 1] - bstart[b] + item_points - 1) / item_points;
}

__global__ void k_block_fill_items(const int *__restrict__ bstart, const int *__restrict__ istart, int nblocks, NwItem *__restrict__ items, int item_points)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const int p0 = bstart[b], p1 = bstart[b + 1];
    const int n = (p1 - p0 + item_points - 1) / item_points;        // a crowded block is cut into n EQUAL runs
    if (n <= 0) return;
    const int per = (p1 - p0 + n - 1) / n;
    int o = istart[b];
    for (int p = p0; p < p1; p += per) {
        NwItem w;
        w.p0 = p; w.n = min(per, p1 - p);
        items[o++] = w;
    }
}

// work list ordered by measured cost, longest first: sort keys (descending cost = ascending key), then the gather
__global__ void k_item_cost_keys(const unsigned *__restrict__ cost, int n, unsigned *__restrict__ key, int *__restrict__ idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    key[i] = 0xffffffffu - cost[i];
    idx[i] = i;
}

__global__ void k_items_gather(const NwItem *__restrict__ in, const int *__restrict__ order, int n, NwItem *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[order[i]];
}

// ---- the query ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int nw_wave_min_i(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int nw_wave_max_i(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

// developer counters of one launch (nw_debug, what = 0): candidates evaluated, rows listed / visited, cells tested / visited, ...
enum { NWS_CAND = 0, NWS_ROWS_NONEMPTY, NWS_ROWS_PASS, NWS_CELLS_TESTED, NWS_CELLS_PASS, NWS_BOX_ROWS, NWS_ROUNDS, NWS_T_WAVE_MAX, NWS_T_STREAM, NWS_T_WAVE, NWS_T_PRO, NWS_T_TAIL, NWS_LANE_CELLS, NWS_LANE_CELLS_MAX, NWS_LANE_CAND, NWS_LANE_CAND_MAX, NWS_COUNT };   // LANE_*: cells / candidates of the collected ranges that the lane's OWN ball reaches, summed over the lanes; *_MAX: the wave's largest lane, summed over the waves   // T_*: s_memtime ticks / 16 (PRO: before the walk, TAIL: after it)
#define NWS_COPIES 1024             // sets of counters the waves spread their atomics over (one set: 17 000 same-address atomics took 3 ms per launch)
struct NwStats { int v[NWS_COUNT]; bool timed; };

// Best / runner-up are kept as integer KEYS: the float32 bits of d^2 (non-negative, so they order like unsigned integers)
// with the low 4 mantissa bits replaced by the candidate's position inside its chunk of 16.  One v_and_or + v_med3_u32 +
// v_min_u32 per candidate, all on vector registers, replace compare + select + move-from-scalar (measured on MI355X: a VALU
// instruction that reads a scalar register or writes/reads VCC issues at ~4.3 cycles per wave, a pure-VGPR one at ~2.6).
// The truncation (16 ulp = 1.9e-6 relative) is inside the ambiguity band: such near-ties go to the float64 fix-up.
#define NW_KEY_BITS 4
#define NW_KEY_MASK 0xfffffff0u
#define NW_KEY_INF 0x7f800000u

struct NwLane {
    float px, py, pz;        // the localization RELATIVE TO THE WAVE'S ORIGIN (mean of its localizations)
    float dkp;               // K - |p'|^2: key value = d^2 + dkp (K = the wave's bias, >= every lane's |p'|^2, keeps keys non-negative)
    int kbits;               // bits of the wave's bias K (uniform): NW_NN_TOLK * K is the part of a key's error bound that does not shrink with the key
    float ux, uy, uz;        // the same in cell units: (p - origin) * inv_h
    float ax, ay, az;        // u + eps      } distance (cell units) from u to the slab [k - eps, k + 1 + eps] of cell index k:
    float bx, by, bz;        // u - 1 - eps  }   max(k - a, b - k, 0)
    unsigned b1, b2;         // best / runner-up key
    int bgi;                 // position (in the list being streamed) of the 16-chunk the best candidate came from, -1: unchanged
    int bslot;               // slot of the best candidate in the cell-sorted centroid array (-1: none yet)
    unsigned keymask;        // NW_KEY_MASK, pinned in a vector register
};

// conservative (upper) estimate of the lane's best squared distance from its key: key = d^2 + (K - |p'|^2) up to NW_NN_TOL of the
// key's magnitude
__device__ __forceinline__ float nw_best_d2(const NwLane &L)
{
    const float kf = __uint_as_float(L.b1);
    return fmaxf(kf - L.dkp, 0.0f) + (NW_NN_TOL * kf + NW_NN_TOLK * __int_as_float(L.kbits));
}

// one candidate (the same for all lanes: an LDS broadcast read) against the lanes' localizations; KI = its position in the
// 16-chunk (an inline constant).  The candidate is staged in EXPANDED form relative to the wave's origin,
//     C = {-2x', -2y', -2z', |c'|^2 + K},   key value = |p' - c'|^2 - |p'|^2 + K = fma(px', X, fma(py', Y, fma(pz', Z, W))):
// three fused multiply-adds (the conversion of a candidate is done once, by the lane that stages it).
template <int KI>
__device__ __forceinline__ void nw_eval(NwLane &L, const float4 C)
{
    const float d = fmaf(L.px, C.x, fmaf(L.py, C.y, fmaf(L.pz, C.z, C.w)));
    unsigned key, b2;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(d), "v"(L.keymask), "n"(KI));
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(b2) : "v"(L.b1), "v"(L.b2), "v"(key));      // b1 <= b2: the runner-up is the median of {b1, b2, key}
    L.b2 = b2;
    L.b1 = min(L.b1, key);
}

// candidate as stored -> staged (expanded) form; O = the wave's origin, K = its bias
__device__ __forceinline__ float4 nw_expand(const float4 C, float Ox, float Oy, float Oz, float K)
{
    const float x = C.x - Ox, y = C.y - Oy, z = C.z - Oz;
    return make_float4(-2.0f * x, -2.0f * y, -2.0f * z, fmaf(z, z, fmaf(y, y, x * x)) + K);
}

// distance (cell units) from the lane's coordinate to the slab [k - eps, k + 1 + eps] of cell index k (a, b: see NwLane)
__device__ __forceinline__ float nw_slab_d(float a, float b, float kf)
{
    return fmaxf(fmaxf(kf - a, b - kf), 0.0f);
}

// ---- wave-private LDS: the list of candidate ranges collected by the walk, and one batch of 64 staged candidates ---------
#define NW_XCD_RUN 16         // consecutive workgroups of the work list per XCD turn (interleaved-runs mapping)
#define NW_SEG 8             // cells per row segment (one lane fetches the 9 cell starts of a segment)
#define NW_RNG_MAX 48        // ranges collected before they are streamed
struct NwWaveLds {
    float4 cand[64];                 // one batch of candidates: lane i stages candidate i, every lane then reads them all (broadcast)
    int rs[NW_RNG_MAX];              // first slot of range r
    int pre[NW_RNG_MAX + 1];         // candidates before range r (pre[nr] = total)
};

// LDS traffic inside ONE wave is in program order; this only stops the compiler from moving accesses across the point
__device__ __forceinline__ void nw_wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// lane -> slot of the g-th candidate of the collected list (binary search in the prefix counts; nr <= 48 < 64: six steps)
__device__ __forceinline__ int nw_list_slot(const NwWaveLds *W, int nr, int g)
{
    int lo = 0, hi = nr;                      // invariant: pre[lo] <= g < pre[hi]
#pragma unroll
    for (int step = 0; step < 6; ++step) {
        const int mid = (lo + hi) >> 1;
        const bool right = W->pre[mid] <= g;
        lo = right ? mid : lo;
        hi = right ? hi : mid;
    }
    return W->rs[lo] + (g - W->pre[lo]);
}

// Stream the collected ranges: 64 candidates per batch are fetched by the 64 lanes (one coalesced-ish global load each), staged in
// LDS and evaluated by every lane against its localization; the fetch of batch b+1 is in flight while batch b is evaluated.
template <bool STATS>
__device__ __forceinline__ void nw_stream(NwLane &L, NwWaveLds *W, const float4 *__restrict__ cent, int nr, int total, int lane, NwStats &S,
                                          float Ox, float Oy, float Oz, float K)
{
    if (total <= 0) return;
    const unsigned long long t_in = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
    if (STATS) S.v[NWS_CAND] += (total + 3) & ~3;
    W->pre[nr] = total;
    nw_wave_lds_sync();
    // lanes past the end of the list stage a far-away dummy (the last group of four of a batch is evaluated whole)
    const float4 far = make_float4(0.0f, 0.0f, 0.0f, 1e30f);
    float4 C = nw_expand(cent[nw_list_slot(W, nr, min(lane, total - 1))], Ox, Oy, Oz, K);
    if (lane >= total) C = far;
    for (int g0 = 0; g0 < total; g0 += 64) {
        W->cand[lane] = C;
        nw_wave_lds_sync();
        const int gn = g0 + 64 + lane;
        if (g0 + 64 < total) {                                                            // next batch: in flight during the evaluation
            C = nw_expand(cent[nw_list_slot(W, nr, min(gn, total - 1))], Ox, Oy, Oz, K);
            if (gn >= total) C = far;
        }
        const int cnt = min(64, total - g0);
        for (int c = 0; c < cnt; c += 16) {
            const unsigned before = L.b1;
            const float4 *q = W->cand + c;
            nw_eval<0>(L, q[0]); nw_eval<1>(L, q[1]); nw_eval<2>(L, q[2]); nw_eval<3>(L, q[3]);
            if (c + 4 < cnt) { nw_eval<4>(L, q[4]); nw_eval<5>(L, q[5]); nw_eval<6>(L, q[6]); nw_eval<7>(L, q[7]); }
            if (c + 8 < cnt) { nw_eval<8>(L, q[8]); nw_eval<9>(L, q[9]); nw_eval<10>(L, q[10]); nw_eval<11>(L, q[11]); }
            if (c + 12 < cnt) { nw_eval<12>(L, q[12]); nw_eval<13>(L, q[13]); nw_eval<14>(L, q[14]); nw_eval<15>(L, q[15]); }
            L.bgi = (L.b1 != before) ? g0 + c : L.bgi;
        }
        nw_wave_lds_sync();                   // all reads of this batch precede the next staging write
    }
    // lanes whose best changed: position in the list -> slot in the centroid array (the list is about to be reused)
    if (L.bgi >= 0) L.bslot = nw_list_slot(W, nr, L.bgi + (int)(L.b1 & 15u));
    L.bgi = -1;
    nw_wave_lds_sync();
    if (STATS) S.v[NWS_T_STREAM] += (int)((__builtin_amdgcn_s_memtime() - t_in) >> 4);
}

// lane j's k-th cell start (k wave-uniform, 0..8): the nine values live in nine registers, picked by a scalar branch
__device__ __forceinline__ int nw_pick(const int (&cs)[NW_SEG + 1], int k, int j)
{
    switch (k) {
    case 0: return __builtin_amdgcn_readlane(cs[0], j);
    case 1: return __builtin_amdgcn_readlane(cs[1], j);
    case 2: return __builtin_amdgcn_readlane(cs[2], j);
    case 3: return __builtin_amdgcn_readlane(cs[3], j);
    case 4: return __builtin_amdgcn_readlane(cs[4], j);
    case 5: return __builtin_amdgcn_readlane(cs[5], j);
    case 6: return __builtin_amdgcn_readlane(cs[6], j);
    case 7: return __builtin_amdgcn_readlane(cs[7], j);
    default: return __builtin_amdgcn_readlane(cs[8], j);
    }
}

__device__ __forceinline__ float nw_readlane_f(float v, int j) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j)); }

// Exact float64 re-resolution of ONE localization P by a whole wave: C0 = a centroid known to be at most tol further than the nearest
// (the float32 winner).  Phase 1: every (z,y) row of fine cells of the box around the ball of radius |P - C0| (1 + 1e-4) + eps gets a
// lane that fetches its candidate range; a wave scan lays the ranges end to end.  Phase 2: the candidates are spread over the lanes
// (binary search in the scanned offsets through shuffles), evaluated in float64, and reduced (lowest face id on exact ties).
// Every lane returns the face id.
__device__ __forceinline__ int nw_fixup_point(const NwGrid &g, float Px, float Py, float Pz, const float4 C0, const int *__restrict__ cstart,
                                           const float4 *__restrict__ cent, int lane, const int *__restrict__ face_orig)
{
    const float r = sqrtf((Px - C0.x) * (Px - C0.x) + (Py - C0.y) * (Py - C0.y) + (Pz - C0.z) * (Pz - C0.z)) * (1.0f + 1e-4f) + g.eps;
    int lx, ly, lz, hx, hy, hz;
    nw_cell_coords(g, Px - r, Py - r, Pz - r, lx, ly, lz);
    nw_cell_coords(g, Px + r, Py + r, Pz + r, hx, hy, hz);
    const int ny = hy - ly + 1, nrow = ny * (hz - lz + 1);
    // a row contributes the cells its chord of the ball covers, not the whole width of the box (cell units; the slack of the cell
    // assignment widens every cell; the outermost cells of the grid also hold what lies beyond it)
    const float epsu = g.eps * g.inv_h;
    const float ux = (Px - g.ox) * g.inv_h, uy = (Py - g.oy) * g.inv_h, uz = (Pz - g.oz) * g.inv_h;
    const float ru = r * g.inv_h * (1.0f + 1e-5f) + epsu, ru2 = ru * ru;
    double best = INFINITY;
    int bf = 0x7fffffff;
    for (int rb = 0; rb < nrow; rb += 64) {
        int start = 0, len = 0;
        const int rr = rb + lane;
        if (rr < nrow) {
            const int z = lz + rr / ny, y = ly + rr % ny;
            const float dyl = y == 0 ? 0.0f : (float)y - (uy + epsu), dyh = y == g.gy - 1 ? 0.0f : (uy - 1.0f - epsu) - (float)y;
            const float dzl = z == 0 ? 0.0f : (float)z - (uz + epsu), dzh = z == g.gz - 1 ? 0.0f : (uz - 1.0f - epsu) - (float)z;
            const float dy = fmaxf(fmaxf(dyl, dyh), 0.0f), dz = fmaxf(fmaxf(dzl, dzh), 0.0f);
            const float rem = ru2 - fmaf(dz, dz, dy * dy);
            if (rem >= 0.0f) {
                const float sx = sqrtf(rem) * (1.0f + 1e-5f) + epsu + 1e-3f;
                const int xl = nw_clampi((int)floorf(ux - sx), lx, hx), xh = nw_clampi((int)floorf(ux + sx), lx, hx);
                const int c0 = nw_cell_index(g, xl, y, z);
                start = cstart[c0];
                len = cstart[c0 + (xh - xl) + 1] - start;
            }
        }
        const int inc = nw_wave_incl_scan(len, lane);           // candidates up to and including this lane's row
        const int total = __shfl(inc, 63, 64);
        for (int base = 0; base < total; base += 64) {           // uniform trip count: every lane takes part in the shuffles
            const int e = min(base + lane, total - 1);
            int lo = 0;                                          // first row whose inclusive count exceeds e
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) {
                const int probe = __shfl(inc, lo + step - 1, 64);
                if (probe <= e) lo += step;
            }
            const int row_start = __shfl(start, lo, 64);
            const int row_excl = __shfl(inc, lo, 64) - __shfl(len, lo, 64);
            if (base + lane < total) {
                const float4 C = cent[row_start + (e - row_excl)];
                const double dx = (double)Px - (double)C.x, dy = (double)Py - (double)C.y, dz = (double)Pz - (double)C.z;
                const double d2 = fma(dz, dz, fma(dy, dy, dx * dx));
                const int fid = __float_as_int(C.w);
                if (d2 < best || (d2 == best && nw_id_less(fid, bf, face_orig))) { best = d2; bf = fid; }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double od = __shfl_xor(best, off, 64);
        const int of = __shfl_xor(bf, off, 64);
        if (od < best || (od == best && nw_id_less(of, bf, face_orig))) { best = od; bf = of; }
    }
    return bf;
}

// STATS: the developer counters of nw_debug, what = 0 (compiled out of the production variant: they cost registers)
template <bool STATS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_nn_wave(NwGrid g, const NwItem *__restrict__ items, int nitems, const float4 *__restrict__ pts,
                                                 const int *__restrict__ cstart, const float4 *__restrict__ cent, const float4 *__restrict__ cent_by_face, int F,
                                                 int *__restrict__ face_io, int warm, int *__restrict__ ambig_list, int *__restrict__ ambig_count,
                                                 NwDevState *__restrict__ st, int it, unsigned long long *__restrict__ stats,
                                                 unsigned *__restrict__ item_cost, const int *__restrict__ face_orig, float outl_f, int outl_max,
                                                 const NwRingArgs R, int nb_query, int nb_ring, const NwAttractArgs A, int *__restrict__ item_done, const int *__restrict__ query_serial)
{
    if (st->iter_base + it >= st->stop_at) return;       // `it` = iteration of this search() block, iter_base = iterations before it
    // One LDS block for the three kinds of workgroup of this launch: the query's wave-private lists, or the attraction step's table.
    constexpr int LDS_QUERY = 4 * (int)sizeof(NwWaveLds), LDS_ATTRACT = NW_NNA_HT * 4 + NW_NNA_HT * 4 * 8 + 20 * 8;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[LDS_QUERY > LDS_ATTRACT ? LDS_QUERY : LDS_ATTRACT];
    // what item_done[] reads once THIS query has left an item: the number of grid builds so far (k_face_centroids counts them; never reset,
    // so a word left by an earlier block, mesh or optimiser cannot be mistaken for this query's)
    const int epoch = item_done ? *query_serial : 0;
    // Workgroups behind the query's own: they are dispatched last, so they run while the query's last waves drain (the launch's last third:
    // 6 100 waves in flight at 10 % of it, 2 000 at 80 %, 200 at 90 %).
    //   [nb_query, nb_query + nb_ring): the ring half of the curvature prior, one thread per vertex (nw_device.h);
    //   behind them, one workgroup per workgroup of the query: the ATTRACTION STEP of the same work items (nw_attract.h), a wave per item.
    if ((int)blockIdx.x >= nb_query && (int)blockIdx.x < nb_query + nb_ring) {
        const int v = ((int)blockIdx.x - nb_query) * (int)blockDim.x + (int)threadIdx.x;
        if (v < R.M) nw_prior_ring_vertex(R, v);
        return;
    }
    if ((int)blockIdx.x >= nb_query + nb_ring) {
        // The attraction step of an item needs the item's nearest faces: the query's wave stores them write-through (agent-scope stores),
        // waits for the stores, and one lane then stores the iteration's number into item_done[item]; the wave here polls that word and reads
        // the faces with agent-scope loads (MI355X_MICROARCH.md, hand-offs with sc1 loads).  Nothing in the query waits for anything, and a
        // workgroup of this kind is dispatched only after every workgroup of the query on its XCD has been (in-order dispatch; observed, not
        // promised) -- so the wait ends.  Should dispatch ever be out of order and the wait not end within 200 ms, the wave gives up and
        // raises NW_ERR_HANDOFF: the block fails loudly instead of hanging or using an old face, and the host runs the two steps as separate
        // launches from then on.
        int *s_key = reinterpret_cast<int *>(s_raw);
        unsigned long long *s_val = reinterpret_cast<unsigned long long *>(s_raw + NW_NNA_HT * 4);
        double *s_w = reinterpret_cast<double *>(s_raw + NW_NNA_HT * 4 + NW_NNA_HT * 4 * 8);
        for (int t = threadIdx.x; t < NW_NNA_HT; t += (int)blockDim.x) s_key[t] = -1;
        for (int t = threadIdx.x; t < NW_NNA_HT * 4; t += (int)blockDim.x) s_val[t] = 0ull;
        __syncthreads();
        const int ab = (int)blockIdx.x - nb_query - nb_ring;
        const int wpb_a = (int)blockDim.x >> 6, nwb_a = (nitems + wpb_a - 1) / wpb_a;
        int wb_a;
        if (warm & 4) {                                   // (the query's own mapping: the item's faces and vertices are in this XCD's L2)
            const int x = ab & 7, l = ab >> 3;
            wb_a = ((l / NW_XCD_RUN) * 8 + x) * NW_XCD_RUN + (l % NW_XCD_RUN);
            if (wb_a >= nwb_a) wb_a = -1;
        } else if (warm & 2) wb_a = ab < nwb_a ? ab : -1;
        else wb_a = nw_xcd_remap(ab, nwb_a);
        const int wi_a = __builtin_amdgcn_readfirstlane(wb_a * wpb_a + (int)(threadIdx.x >> 6));
        const int lane_a = threadIdx.x & 63;
        bool have = wb_a >= 0 && wi_a < nitems;
        NwItem item_a; item_a.p0 = 0; item_a.n = 0;
        if (have) {
            item_a = items[wi_a];
            int seen = 0;
            if (lane_a == 0) {
                const unsigned long long t0 = wall_clock64();
                for (;;) {
                    seen = __hip_atomic_load(item_done + wi_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (seen == epoch) break;
                    __builtin_amdgcn_s_sleep(16);
                    if (wall_clock64() - t0 > 20000000ull) break;        // 200 ms of the 100 MHz clock: no warm item takes a hundredth of that
                }
            }
            seen = __builtin_amdgcn_readfirstlane(seen);
            if (seen != epoch) { if (lane_a == 0) atomicCAS(&st->status, 0, -9 /* NW_ERR_HANDOFF */); have = false; }
        }
        const bool act = have && lane_a < item_a.n;
        const int gi_a = item_a.p0 + (act ? lane_a : 0);
        int f_raw = 0;
        if (act) {
            f_raw = __hip_atomic_load(face_io + gi_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)f_raw >= (unsigned)A.F) atomicCAS(&st->status, 0, -7 /* NW_ERR_INTERNAL */);
        }
        nw_attract_workgroup<NW_NNA_HT, NW_NNA_HT_BITS, 48>(A, pts, gi_a, act && (unsigned)f_raw < (unsigned)A.F, f_raw, st, s_key, s_val, s_w, ab);
        return;
    }
    NwWaveLds *s_wave = reinterpret_cast<NwWaveLds *>(s_raw);
    NwStats S;
#pragma unroll
    for (int k = 0; k < NWS_COUNT; ++k) S.v[k] = 0;
    S.timed = STATS;
    const unsigned long long t_wave = (STATS || item_cost) ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long t_real = (item_cost && (warm & 64)) ? wall_clock64() : 0ull;      // developer aid (NW_ITEM_TIMES): the 100 MHz clock all XCDs share
    const int lane = threadIdx.x & 63;
    // Workgroup -> work-list position.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 names the L2 they share).
    //   (warm & 2) plain: consecutive list positions land on different XCDs: every L2 pulls the whole centroid / cell tables;
    //   (warm & 4) interleaved runs: XCD x takes runs of NW_XCD_RUN consecutive positions, the runs of the 8 XCDs interleaved along
    //              the list: neighbours in space (= in the list) share an L2, and all XCDs advance through the list together;
    //   else       one contiguous slab of the list per XCD (nw_xcd_remap): best locality, but the slabs' work differs and the
    //              launch waits for the slowest XCD.
    const int wpb = (int)blockDim.x >> 6, nwb = (nitems + wpb - 1) / wpb;       // waves (= work items) per workgroup, workgroups with work
    int wb;
    if (warm & 4) {
        const int x = (int)blockIdx.x & 7, l = (int)blockIdx.x >> 3;
        wb = ((l / NW_XCD_RUN) * 8 + x) * NW_XCD_RUN + (l % NW_XCD_RUN);
        if (wb >= nwb) wb = -1;
    } else if (warm & 2) wb = (int)blockIdx.x < nwb ? (int)blockIdx.x : -1;
    else wb = nw_xcd_remap(blockIdx.x, nwb);
    const int wi = __builtin_amdgcn_readfirstlane(wb * wpb + (int)(threadIdx.x >> 6));
    if (wi < 0 || wi >= nitems) return;
    NwWaveLds *W = &s_wave[threadIdx.x >> 6];
    const NwItem item = items[wi];
    const bool active = lane < item.n;
    const int gi = item.p0 + (active ? lane : 0);                        // idle lanes shadow lane 0 (they never write)
    NwLane L;
    const float4 P = pts[gi];
    // Cell-unit coordinates for the WALK (own cell, ball box, slab distances): those of the localization's projection onto the grid's
    // box.  The outermost cells hold whatever lies beyond them (nw_cell_coords clamps), so the distance from P to a centroid filed
    // in a cell is at least the distance between their projections (projection onto a convex box does not expand distances), hence
    // at least the projection's distance to the cell: the culling stays a valid lower bound for a localization outside the grid
    // (background far from the structure) and for a centroid that left it since the grid was laid.  The keys use P itself.
    L.ux = fminf(fmaxf((P.x - g.ox) * g.inv_h, 0.0f), (float)g.gx);
    L.uy = fminf(fmaxf((P.y - g.oy) * g.inv_h, 0.0f), (float)g.gy);
    L.uz = fminf(fmaxf((P.z - g.oz) * g.inv_h, 0.0f), (float)g.gz);
    const float epsu = g.eps * g.inv_h;                                  // rounding slack of the cell assignment, cell units
    L.ax = L.ux + epsu; L.ay = L.uy + epsu; L.az = L.uz + epsu;
    L.bx = L.ux - 1.0f - epsu; L.by = L.uy - 1.0f - epsu; L.bz = L.uz - 1.0f - epsu;
    // warm start: the previous nearest face, and its centroid
    int prev = -1;
    if (warm & 1) {
        prev = face_io[gi];
        if ((unsigned)prev >= (unsigned)F) prev = -1;
    }
    const float cullk = g.inv_h * g.inv_h * (1.0f + NW_NN_CULL);
    // Outliers.  A localization far from the surface has a ball many cells wide, and every candidate in it would be evaluated by all
    // 64 lanes; it also stretches the wave's local frame (origin, bias K) and with it every lane's error band.  Up to NW_OUTLIERS
    // lanes whose warm-start radius is far above the wave's (2.5 x its rms, and more than 1.5 cells) are taken out of the walk -- no
    // share in the box, the frame or the culling -- and resolved at the end like the ambiguous ones: the whole wave over that one
    // ball, float64, exact (nw_fixup_point needs only a centroid that bounds the distance: the warm-start face).
    unsigned long long outl_mask = 0ull;
    if ((warm & 25) == 9) {                                          // warm start, in-kernel fix-up, not switched off (16)
        float r2 = 0.0f;
        if (active && prev >= 0) {                                   // (the centroid is fetched again for the warm start below: four registers less across the reductions)
            const float4 Cw = cent_by_face[prev];
            const float wx = P.x - Cw.x, wy = P.y - Cw.y, wz = P.z - Cw.z;
            r2 = fmaf(wz, wz, fmaf(wy, wy, wx * wx));
        }
        float sum = r2;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        const float mean = nw_readlane_f(sum, 0) / (float)item.n;
        const unsigned long long m = __ballot(active && prev >= 0 && r2 > outl_f * mean && r2 * cullk > 2.25f);
        if (__popcll(m) <= outl_max) outl_mask = m;
    }
    const bool wact = active && !((outl_mask >> lane) & 1ull);         // takes part in the walk
    // wave origin = mean of the walk's localizations (idle and outlier lanes stand in for the first of them), bias K = largest
    // |p'|^2: the distances are evaluated in the expanded form on LOCAL coordinates, so their float32 error scales with the
    // wave's size, not with the coordinate offset
    const int first = __builtin_ctzll(__ballot(wact));
    const float Qx = wact ? P.x : nw_readlane_f(P.x, first), Qy = wact ? P.y : nw_readlane_f(P.y, first), Qz = wact ? P.z : nw_readlane_f(P.z, first);
    float Ox = Qx, Oy = Qy, Oz = Qz;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { Ox += __shfl_xor(Ox, off, 64); Oy += __shfl_xor(Oy, off, 64); Oz += __shfl_xor(Oz, off, 64); }
    Ox = nw_readlane_f(Ox * (1.0f / 64.0f), 0); Oy = nw_readlane_f(Oy * (1.0f / 64.0f), 0); Oz = nw_readlane_f(Oz * (1.0f / 64.0f), 0);
    L.px = P.x - Ox; L.py = P.y - Oy; L.pz = P.z - Oz;
    const float pn = fmaf(L.pz, L.pz, fmaf(L.py, L.py, L.px * L.px));
    const float qx = Qx - Ox, qy = Qy - Oy, qz = Qz - Oz;
    float Kb = fmaf(qz, qz, fmaf(qy, qy, qx * qx));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) Kb = fmaxf(Kb, __shfl_xor(Kb, off, 64));
    Kb = nw_readlane_f(Kb, 0) * (1.0f + 1e-5f) + 1e-6f;
    L.dkp = Kb - pn;                                                   // (negative for an outlier: its keys are never used)
    L.kbits = __builtin_amdgcn_readfirstlane(__float_as_int(Kb));          // (a scalar register: nothing to keep alive per lane)
    const float tolk = NW_NN_TOLK * __int_as_float(L.kbits);
    L.b1 = NW_KEY_INF; L.b2 = NW_KEY_INF; L.bgi = -1; L.bslot = -1;
    L.keymask = NW_KEY_MASK;
    asm volatile("" : "+v"(L.keymask));
    if (prev >= 0) {
        const float4 C = nw_expand(cent_by_face[prev], Ox, Oy, Oz, Kb);
        const float d = fmaf(L.px, C.x, fmaf(L.py, C.y, fmaf(L.pz, C.z, C.w)));
        // strictly above the key the walk will compute for this very centroid, so the walk re-finds it (and its slot); the bump
        // also becomes the runner-up until a real one is seen, so it must lie outside the ambiguity band (2 NW_NN_TOL)
        L.b1 = __float_as_uint(d * (1.0f + 8.0f * NW_NN_TOL) + 8.0f * tolk + 1e-30f) | 15u;
    }
    // own cell (clamped like nw_cell_coords)
    const int cx = nw_clampi((int)floorf(L.ux), 0, g.gx - 1), cy = nw_clampi((int)floorf(L.uy), 0, g.gy - 1), cz = nw_clampi((int)floorf(L.uz), 0, g.gz - 1);
    int Exl = 0, Exh = -1, Eyl = 0, Eyh = -1, Ezl = 0, Ezh = -1;         // visited box (cells), empty
    int margin = 1, rounds = 0;
    int lane_cells = 0, lane_cand = 0;                                   // (STATS only)
    if (STATS) S.v[NWS_T_PRO] += (int)((__builtin_amdgcn_s_memtime() - t_wave) >> 4);
    for (;;) {
        // ---- the box this round must cover: the lanes' balls (finite b1) or their own cell +- margin (nothing seen yet)
        const bool seen = L.b1 < NW_KEY_INF;
        const bool any_unseen = __any(wact && !seen);
        int lxl, lxh, lyl, lyh, lzl, lzh;
        if (seen) {
            const float ru = fminf(sqrtf(nw_best_d2(L) * cullk) + 2.0f * epsu + 1e-3f, 4096.0f);     // (a lane that has only seen pad entries)
            lxl = (int)floorf(L.ux - ru); lxh = (int)floorf(L.ux + ru);
            lyl = (int)floorf(L.uy - ru); lyh = (int)floorf(L.uy + ru);
            lzl = (int)floorf(L.uz - ru); lzh = (int)floorf(L.uz + ru);
        } else {
            lxl = cx - margin; lxh = cx + margin; lyl = cy - margin; lyh = cy + margin; lzl = cz - margin; lzh = cz + margin;
        }
        if (!wact) { lxl = lyl = lzl = 0x7fffffff; lxh = lyh = lzh = -0x7fffffff; }
        int Nxl = nw_clampi(nw_wave_min_i(lxl), 0, g.gx - 1), Nxh = nw_clampi(nw_wave_max_i(lxh), 0, g.gx - 1);
        int Nyl = nw_clampi(nw_wave_min_i(lyl), 0, g.gy - 1), Nyh = nw_clampi(nw_wave_max_i(lyh), 0, g.gy - 1);
        int Nzl = nw_clampi(nw_wave_min_i(lzl), 0, g.gz - 1), Nzh = nw_clampi(nw_wave_max_i(lzh), 0, g.gz - 1);
        const bool Eok = Exh >= Exl;
        if (Eok) {
            Nxl = min(Nxl, Exl); Nxh = max(Nxh, Exh); Nyl = min(Nyl, Eyl); Nyh = max(Nyh, Eyh); Nzl = min(Nzl, Ezl); Nzh = max(Nzh, Ezh);
            if (Nxl == Exl && Nxh == Exh && Nyl == Eyl && Nyh == Eyh && Nzl == Ezl && Nzh == Ezh) {
                if (!any_unseen) break;                                  // every ball lies inside the visited box
                const bool whole = Exl == 0 && Eyl == 0 && Ezl == 0 && Exh == g.gx - 1 && Eyh == g.gy - 1 && Ezh == g.gz - 1;
                if (whole) break;                                        // no centroid anywhere (cannot happen with F >= 1)
                margin <<= 1;
                continue;
            }
        }
        ++rounds;
        // ---- walk N \ E.  lane = one SEGMENT of NW_SEG cells of one (z,y) row of N: it fetches the segment's nine cell starts (all
        // requests of 64 segments in flight together: one round trip), and knows which of its cells hold centroids and are new.
        const int ny = Nyh - Nyl + 1, nrows = ny * (Nzh - Nzl + 1);
        const int nchunk = (Nxh - Nxl + NW_SEG) / NW_SEG, nseg = nrows * nchunk;
        const float inv_ny = 1.0f / (float)ny, inv_nc = 1.0f / (float)nchunk;
        const float r2u = wact ? nw_best_d2(L) * cullk + epsu : -1.0f;   // culling radius^2 (cell units) for this round; idle and outlier lanes reach nothing
        if (STATS) S.v[NWS_BOX_ROWS] += nrows;
        int nr = 0, total = 0;                                           // collected ranges (wave-uniform)
        for (int sb = 0; sb < nseg; sb += 64) {
            const int sidx = sb + lane;
            const bool ok = sidx < nseg;
            const int r = (int)(((float)sidx + 0.5f) * inv_nc);           // exact sidx / nchunk for the small ints involved
            const int ch = sidx - r * nchunk;
            const int rz = (int)(((float)r + 0.5f) * inv_ny);
            const int y = Nyl + (r - rz * ny), z = Nzl + rz;
            const int xa = Nxl + NW_SEG * ch;
            const int nxh = min(NW_SEG, Nxh - xa + 1);                    // cells of this segment
            const int rowbase = g.gx * (y + g.gy * z);
            int cs[NW_SEG + 1];
#pragma unroll
            for (int k = 0; k <= NW_SEG; ++k) cs[k] = ok ? cstart[rowbase + xa + min(k, nxh)] : 0;
            unsigned m = 0;
#pragma unroll
            for (int k = 0; k < NW_SEG; ++k) m |= (cs[k + 1] > cs[k]) ? (1u << k) : 0u;
            if (Eok && y >= Eyl && y <= Eyh && z >= Ezl && z <= Ezh) {
                // cells of the visited box are done: drop the bits of x in [Exl, Exh]
                const int k0 = max(Exl - xa, 0), k1 = min(Exh - xa, NW_SEG - 1);
                if (k1 >= k0) m &= ~(((2u << k1) - 1u) & ~((1u << k0) - 1u));
            }
            const float yf = (float)y, zf = (float)z, xaf = (float)xa;
            unsigned long long segs = __ballot(ok && m != 0u);
            if (STATS) S.v[NWS_ROWS_NONEMPTY] += __popcll(segs);
            while (segs) {
                const int j = __builtin_ctzll(segs);
                segs &= segs - 1ull;
                // row test: does any lane's ball reach the (y,z) square of this row?
                const float dy = nw_slab_d(L.ay, L.by, nw_readlane_f(yf, j)), dz = nw_slab_d(L.az, L.bz, nw_readlane_f(zf, j));
                const float dyz2 = fmaf(dz, dz, dy * dy);
                if (!__any(dyz2 <= r2u)) continue;
                if (STATS) S.v[NWS_ROWS_PASS] += 1;
                const unsigned mj = (unsigned)__builtin_amdgcn_readlane((int)m, j);
                const float xj = nw_readlane_f(xaf, j);
                unsigned pm = 0, t = mj;
                while (t) {
                    const int k = __builtin_ctz(t);
                    t &= t - 1u;
                    const float dx = nw_slab_d(L.ax, L.bx, xj + (float)k);
                    if (STATS) S.v[NWS_CELLS_TESTED] += 1;
                    const bool mine = fmaf(dx, dx, dyz2) <= r2u;
                    if (STATS && mine) { lane_cells += 1; lane_cand += nw_pick(cs, k + 1, j) - nw_pick(cs, k, j); }
                    if (__any(mine)) { pm |= 1u << k; if (STATS) S.v[NWS_CELLS_PASS] += 1; }
                }
                // runs of surviving cells (empty or culled-empty cells in between join for free) -> candidate ranges
                const unsigned joinable = pm | ~mj;
                while (pm) {
                    const int k0 = __builtin_ctz(pm);
                    const unsigned stop = (~joinable >> k0) & ((1u << (NW_SEG - k0)) - 1u);     // first cell >= k0 that is non-empty and culled
                    const int k1 = stop ? k0 + __builtin_ctz(stop) : NW_SEG;
                    const int c0 = nw_pick(cs, k0, j), c1 = nw_pick(cs, k1, j);
                    if (c1 > c0) {
                        W->rs[nr] = c0;
                        W->pre[nr] = total;
                        ++nr; total += c1 - c0;
                        if (nr == NW_RNG_MAX) { nw_stream<STATS>(L, W, cent, nr, total, lane, S, Ox, Oy, Oz, Kb); nr = 0; total = 0; }
                    }
                    pm = (k1 >= 32) ? 0u : (pm >> k1) << k1;
                }
            }
        }
        nw_stream<STATS>(L, W, cent, nr, total, lane, S, Ox, Oy, Oz, Kb);
        Exl = Nxl; Exh = Nxh; Eyl = Nyl; Eyh = Nyh; Ezl = Nzl; Ezh = Nzh;
        if (!any_unseen) break;          // the box was built from every lane's ball and b1 only shrinks: all lanes are final
    }
    const unsigned long long t_tail = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
    {
        // (the item, the lane's localization index and its previous face are re-derived here rather than kept alive across the walk:
        // with them in registers the kernel does not fit the 80 VGPRs of six waves per SIMD and spills 20 bytes per lane)
        const NwItem item_t = items[wi];
        const bool active = lane < item_t.n;
        const int gi = item_t.p0 + (active ? lane : 0);
        int fid = (warm & 1) ? face_io[gi] : -1;
        if ((unsigned)fid >= (unsigned)F) fid = -1;
        unsigned long long om0 = outl_mask;
        asm volatile("" : "+s"(om0));
        if (active && L.bslot >= 0 && !((om0 >> lane) & 1ull)) fid = __float_as_int(cent[L.bslot].w);      // (an outlier's keys mean nothing: its bound stays the warm-start face)
        // runner-up inside the error band (or the walk did not re-find the warm-start face): float64 re-resolution
        // both keys carry the same offset K - |p'|^2: their difference is a difference of squared distances, each known to tol(key)
        const float d1 = __uint_as_float(L.b1), d2 = __uint_as_float(L.b2);
        unsigned long long om = outl_mask;
        asm volatile("" : "+s"(om));                               // (re-derive the lane's bit here instead of keeping it alive across the walk)
        const bool outl = (om >> lane) & 1ull;
        const bool amb = active && (outl || L.bslot < 0 || d2 - d1 <= 2.0f * (NW_NN_TOL * d2 + NW_NN_TOLK * __int_as_float(L.kbits)));
        if (warm & 8) {
            // resolved here, by the whole wave, one ambiguous localization after the other (0.4 per wave on average): their cells are
            // in this CU's caches, and the iteration has one launch and one list less
            unsigned long long todo = __ballot(amb && (unsigned)fid < (unsigned)F);
            const float4 Pw = pts[gi];
            while (todo) {
                const int j = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                const float4 C0 = cent_by_face[__builtin_amdgcn_readlane(fid, j)];
                const int res = nw_fixup_point(g, nw_readlane_f(Pw.x, j), nw_readlane_f(Pw.y, j), nw_readlane_f(Pw.z, j), C0, cstart, cent, lane, face_orig);
                if (lane == j) fid = res;
            }
            if (active) {
                if (item_done) __hip_atomic_store(face_io + gi, fid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // write-through: read by another CU in this launch
                else face_io[gi] = fid;
            }
        } else if (active) {
            face_io[gi] = fid;
            if (amb) { const int k = atomicAdd(ambig_count, 1); ambig_list[k] = gi; }
        }
        if (item_done) {
            // the item's faces are out: wait for the stores, then say so (one lane; the attraction workgroups of this launch poll the word)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(item_done + wi, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (STATS) S.v[NWS_T_TAIL] += (int)((__builtin_amdgcn_s_memtime() - t_tail) >> 4);
    if (lane == 0 && rounds > 1) atomicMax(&st->nn_max_ring, rounds);
    // how long this item took: the work list is ordered by it once (longest first), so that the launch does not end on its heaviest waves
    if (item_cost && lane == 0) {
        item_cost[wi] = (unsigned)min((unsigned long long)0xffffffffu, (__builtin_amdgcn_s_memtime() - t_wave) >> 4);
        // developer aid (NW_ITEM_TIMES): start and duration of every item in 10 ns ticks, for the launch's time line (tools/nn_costs.py)
        if (warm & 64) { item_cost[nitems + wi] = (unsigned)t_real; item_cost[wi] = (unsigned)(wall_clock64() - t_real); }
    }
    if (STATS) {
        int cs_ = lane_cells, cm_ = lane_cells, ns_ = lane_cand, nm_ = lane_cand;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            cs_ += __shfl_xor(cs_, off, 64); ns_ += __shfl_xor(ns_, off, 64);
            cm_ = max(cm_, __shfl_xor(cm_, off, 64)); nm_ = max(nm_, __shfl_xor(nm_, off, 64));
        }
        S.v[NWS_LANE_CELLS] = cs_; S.v[NWS_LANE_CELLS_MAX] = cm_; S.v[NWS_LANE_CAND] = ns_; S.v[NWS_LANE_CAND_MAX] = nm_;
    }
    if (STATS && stats && lane == 0) {
        S.v[NWS_ROUNDS] = rounds;
        S.v[NWS_T_WAVE] = (int)((__builtin_amdgcn_s_memtime() - t_wave) >> 4);
        unsigned long long *mine = stats + (size_t)(wi & (NWS_COPIES - 1)) * NWS_COUNT;
#pragma unroll
        for (int k = 0; k < NWS_COUNT; ++k) atomicAdd(mine + k, (unsigned long long)S.v[k]);
        atomicMax(mine + NWS_T_WAVE_MAX, (unsigned long long)S.v[NWS_T_WAVE]);
    }
}
