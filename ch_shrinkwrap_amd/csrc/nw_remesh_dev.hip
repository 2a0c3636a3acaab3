// Block-boundary isotropic remesher ON THE DEVICE (MI355X, gfx950): nw_remesh_device of include/nanowrap.h.
//
// What it stands in for: PYME's TriangleMesh.remesh(5, target, 0.5, n_relax=0), which the reference calls between optimiser blocks
// (/root/reference/ch_shrinkwrap/_membrane_mesh.pyx:1546, schedule :1443-1455) and which is not part of the reference tree (SURVEY.md 8c:
// parity unpinned).  Same published algorithm as the host remesher (csrc/remesh.cpp; Botsch & Kobbelt 2004) and the same admission tests,
// operation by operation: per iteration split the edges longer than 4/3 L, collapse those shorter than 4/5 L, flip towards degree six, then
// n_relax steps of tangential relaxation (the block boundary asks for none).
//
// How a chain of local operations becomes kernels.  An operation reads and rewrites the 1-rings of a handful of vertices, its FOOTPRINT:
//     split a-b (opposite c, d)    {a, b, c, d}
//     collapse a into b            {a, b} and every neighbour of a
//     flip a-b to c-d              {a, b, c, d}
// Two operations whose footprints share no vertex touch disjoint faces, twin links and vertex records and can run at once.  A pass
// (split, collapse or flip of one iteration) first scans all half-edges with the cheap half of its test and compacts the hits into a
// candidate LIST (a few per cent of the edges); then it is a loop of ROUNDS over that list; a round is
//     bid    one thread per entry tests the edge exactly like the host code does (read-only: nothing is being changed), and a candidate
//            writes its 64-bit key -- (round number, hash of the half-edge and the round, half-edge id): see rm_key for why a hash and
//            not the edge's length -- into every footprint vertex with atomicMax;
//     apply  a candidate that finds its own key in ALL its footprint vertices owns them and performs the operation; the others wait for
//            the next round.  The candidate with the largest key always wins, so every round makes progress; the round number on top of the
//            key makes every earlier round's keys lose, so nothing is ever cleared.
// Nothing depends on which thread runs when: atomicMax commutes, and the ids of what a split creates come from a prefix sum over the
// round's winners in list order.  Two runs give the same arrays.  The host launches at most two rounds ahead of the GPU's reports (every
// round writes its number of bidders to pinned memory) and ends a pass when a round had none, or almost none.
//
// Layout (all in HBM for the whole call): half-edge 3f+k runs from F[3f+k] to F[3f+(k+1)%3] -- next / previous / face of a half-edge are
// arithmetic, only the origin `F` and `twin` are stored; positions in float64 (as the host code); vhe[v] = one outgoing half-edge, val[v] =
// degree, bnd[v] = vertex on an unmatched edge or a bow-tie: frozen.  A face that dies has F = -1.  Capacities are fixed per attempt from
// the number of faces the edge lengths call for; an attempt that runs out reports it and the call starts again with twice the room.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <climits>
#include <vector>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <algorithm>

#include "../../include/nanowrap.h"

#define NW_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

typedef unsigned long long u64;

struct D3 { double x, y, z; };
__device__ __forceinline__ D3 operator-(const D3 &a, const D3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 operator+(const D3 &a, const D3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ D3 operator*(const D3 &a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ double dot(const D3 &a, const D3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 cross(const D3 &a, const D3 &b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double norm2(const D3 &a) { return dot(a, a); }

enum { RC_CAND = 0, RC_SPLIT, RC_COLLAPSE, RC_FLIP, RC_CORRUPT, RC_BADARG, RC_NONMANIFOLD, RC_MAXVAL, RC_NV, RC_NF, RC_NLIST, RC_OVERFLOW, RC_VCAP, RC_FCAP, RC_COUNT };

struct RM {
    D3 *pos;
    int *F, *twin;                 // per half-edge
    int *vhe, *val;                // per vertex
    unsigned char *bnd;
    u64 *owner;                    // per vertex: the largest key bid so far (a key starts with its round's number: a later round outbids every earlier one, nothing is cleared)
    int *round_cand;               // per round: candidates that bid
    int *host_cand;                // the same in pinned host memory, written when the round's bids are in: the host stops a pass without a stall
    u64 *ckey;                     // per half-edge: the key its operation bid with, 0 = no candidate
    int *win, *scan;               // per half-edge (splits): winner flag and its prefix sum
    int *cnt;                      // RC_* counters
    int max_valence;
    double high2, low2;
};

__device__ __forceinline__ int rm_next(int h) { return (h % 3 == 2) ? h - 2 : h + 1; }
__device__ __forceinline__ int rm_prev(int h) { return (h % 3 == 0) ? h + 2 : h - 1; }

struct Quad { int h, t, hn, hp, tn, tp, a, b, c, d; };
// h: a->b in face (a, b, c); t = twin: b->a in face (b, a, d)
__device__ __forceinline__ bool rm_quad(const RM &m, int h, Quad &q)
{
    q.h = h; q.t = m.twin[h];
    if (q.t < 0) return false;
    q.hn = rm_next(h); q.hp = rm_prev(h); q.tn = rm_next(q.t); q.tp = rm_prev(q.t);
    q.a = m.F[h]; q.b = m.F[q.hn]; q.c = m.F[q.hp]; q.d = m.F[q.tp];
    return true;
}

__device__ __forceinline__ D3 rm_face_normal(const RM &m, int a, int b, int c) { return cross(m.pos[b] - m.pos[a], m.pos[c] - m.pos[a]); }

// outgoing half-edges of an interior vertex, counter-clockwise; false if the fan does not close within 64 steps
#define RM_RING_MAX 64
template <class Fn>
__device__ __forceinline__ bool rm_ring(const RM &m, int v, Fn fn)
{
    const int h0 = m.vhe[v];
    if (h0 < 0) return false;
    int h = h0, guard = 0;
    do {
        fn(h);
        h = m.twin[rm_prev(h)];
        if (++guard > RM_RING_MAX) return false;
    } while (h != h0 && h >= 0);
    return h == h0;
}

// ---- set-up -------------------------------------------------------------------------------------------------------------------------
__global__ void k_rm_load(const float *__restrict__ v, int nv, const int *__restrict__ f, int nf, RM m)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nv) {
        const float x = v[3 * i], y = v[3 * i + 1], z = v[3 * i + 2];
        if (!isfinite(x) || !isfinite(y) || !isfinite(z)) m.cnt[RC_BADARG] = 1;
        m.pos[i] = {(double)x, (double)y, (double)z};
        m.vhe[i] = INT_MAX; m.val[i] = 0; m.bnd[i] = 0;
    }
    if (i < nf) {
        const int a = f[3 * i], b = f[3 * i + 1], c = f[3 * i + 2];
        if (a < 0 || a >= nv || b < 0 || b >= nv || c < 0 || c >= nv || a == b || b == c || c == a) m.cnt[RC_BADARG] = 1;
        m.F[3 * i] = a; m.F[3 * i + 1] = b; m.F[3 * i + 2] = c;
    }
}

// how many faces the target length calls for: a face with longest edges l1 >= l2 ends up as about (l1 / high + 1)(l2 / high + 1) pieces
// (csrc/remesh.cpp, run_passes); block sums, added on the host in block order
__global__ void k_rm_pieces(RM m, int nf, double high, double *__restrict__ part)
{
    __shared__ double s[256];
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    double p = 0.0;
    if (f < nf) {
        const int a = m.F[3 * f], b = m.F[3 * f + 1], c = m.F[3 * f + 2];
        double e0 = sqrt(norm2(m.pos[a] - m.pos[b])), e1 = sqrt(norm2(m.pos[b] - m.pos[c])), e2 = sqrt(norm2(m.pos[c] - m.pos[a]));
        const double lo = fmin(e0, fmin(e1, e2)), hi = fmax(e0, fmax(e1, e2)), mid = e0 + e1 + e2 - lo - hi;
        p = (hi / high + 1.0) * (mid / high + 1.0);
    }
    s[threadIdx.x] = p;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = s[0];
}

#define RM_EMPTY 0xffffffffffffffffull
__device__ __forceinline__ unsigned rm_hash(u64 k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return (unsigned)k; }

// directed edge -> half-edge; a direction that is already there makes the mesh non-manifold (three faces on an edge, or two of one sense)
__global__ void k_rm_hash_insert(RM m, int nh, u64 *__restrict__ keys, int *__restrict__ vals, unsigned mask)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= nh) return;
    const u64 key = ((u64)(unsigned)m.F[h] << 32) | (unsigned)m.F[rm_next(h)];
    unsigned s = rm_hash(key) & mask;
    for (unsigned probe = 0; probe <= mask; ++probe) {
        const u64 old = atomicCAS(&keys[s], RM_EMPTY, key);
        if (old == RM_EMPTY) { vals[s] = h; return; }
        if (old == key) { m.cnt[RC_NONMANIFOLD] = 1; return; }
        s = (s + 1) & mask;
    }
}

__global__ void k_rm_hash_lookup(RM m, int nh, const u64 *__restrict__ keys, const int *__restrict__ vals, unsigned mask)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= nh) return;
    const int o = m.F[h], t = m.F[rm_next(h)];
    const u64 key = ((u64)(unsigned)t << 32) | (unsigned)o;
    unsigned s = rm_hash(key) & mask;
    int tw = -1;
    for (unsigned probe = 0; probe <= mask; ++probe) {
        const u64 k = keys[s];
        if (k == RM_EMPTY) break;
        if (k == key) { tw = vals[s]; break; }
        s = (s + 1) & mask;
    }
    m.twin[h] = tw;
    atomicAdd(&m.val[o], 1);
    atomicMin(&m.vhe[o], h);
    if (tw < 0) { m.bnd[o] = 1; m.bnd[t] = 1; }
}

// unreferenced slots; a vertex whose fan does not close after `val` steps is a bow-tie: left alone like a boundary vertex
__global__ void k_rm_vertex_flags(RM m, int nv)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    if (m.vhe[v] == INT_MAX) { m.vhe[v] = -1; return; }
    if (m.bnd[v]) return;
    int n = 0;
    const bool closed = rm_ring(m, v, [&](int) { ++n; });
    if (!closed || n != m.val[v]) m.bnd[v] = 1;
}

// ---- split ----------------------------------------------------------------------------------------------------------------------------
// The priority of a candidate is a hash of its half-edge and the round, not the edge's length: with "longest first" a patch of edges that all
// want splitting is a monotone chain of keys in which only local maxima win -- one per round along the chain (64 rounds did not finish a
// pass) --, with hashed keys a round takes a constant share of the candidates (Luby's independent set: 5-8 rounds).
// `round` (the launch's number in the call) only orders the rounds; `seed` (pass, round within the pass) feeds the hash.  The host launches a
// round or two more than a pass needs, depending on how far ahead of the GPU it happens to be: such rounds find nothing to do, and must
// not shift the hashes of the rounds that follow.
__device__ __forceinline__ u64 rm_key(unsigned round, unsigned seed, int h)
{
    unsigned x = (unsigned)h * 2654435761u ^ (seed * 0x9e3779b9u);
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return ((u64)(round & 0xffffu) << 48) | ((u64)(x & 0xffffu) << 32) | (unsigned)h;
}

// the cheap half of a pass's test, over ALL half-edges: which edges may be candidates at all (kind 0: too long, 1: too short, 2: a flip would
// bring the degrees closer to six).  The flags are compacted into the pass's candidate LIST (prefix sum: in half-edge order), and the
// rounds of the pass then look at the list only -- a few per cent of the edges.  What an operation of this pass turns into a candidate is
// seen by the next pass of its kind, as in the host code (whose sweeps look at an edge once).
template <int KIND>
__global__ void k_rm_candidates(RM m, int *__restrict__ flag)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= 3 * m.cnt[RC_NF]) return;
    int f = 0;
    Quad q;
    if (m.F[h] >= 0 && rm_quad(m, h, q) && h < q.t && q.c != q.d) {
        if (KIND == 2) {
            const int before = abs(m.val[q.a] - 6) + abs(m.val[q.b] - 6) + abs(m.val[q.c] - 6) + abs(m.val[q.d] - 6);
            const int after = abs(m.val[q.a] - 7) + abs(m.val[q.b] - 7) + abs(m.val[q.c] - 5) + abs(m.val[q.d] - 5);
            f = after < before;
        } else {
            const double l2 = norm2(m.pos[q.a] - m.pos[q.b]);
            f = KIND == 0 ? (l2 > m.high2 && isfinite(l2) && !(m.bnd[q.a] && m.bnd[q.b])) : (l2 < m.low2);
        }
    }
    flag[h] = f;
}

__global__ void k_rm_compact(RM m, const int *__restrict__ flag, const int *__restrict__ at, int *__restrict__ list)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    const int nh = 3 * m.cnt[RC_NF];
    if (h >= nh) return;
    if (flag[h]) list[at[h]] = h;
    if (h == nh - 1) m.cnt[RC_NLIST] = at[h] + flag[h];
}


__global__ void k_rm_split_bid(RM m, const int *__restrict__ list, unsigned round, unsigned seed)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m.cnt[RC_NLIST]) return;
    const int h = list[i];
    u64 key = 0;
    Quad q;
    // (h < twin: operations of earlier rounds re-pair half-edges, and both halves of an edge may have come to stand in the list -- an
    // edge is the business of its lower half-edge, here and in the apply kernels)
    if (m.F[h] >= 0 && rm_quad(m, h, q) && h < q.t && q.c != q.d && !(m.bnd[q.a] && m.bnd[q.b])) {
        const double l2 = norm2(m.pos[q.a] - m.pos[q.b]);
        if (l2 > m.high2 && isfinite(l2)) {
            key = rm_key(round, seed, h);
            atomicMax(&m.owner[q.a], key); atomicMax(&m.owner[q.b], key); atomicMax(&m.owner[q.c], key); atomicMax(&m.owner[q.d], key);
            atomicAdd(&m.round_cand[round], 1);
        }
    }
    m.ckey[h] = key;
}

__global__ void k_rm_split_mark(RM m, const int *__restrict__ list)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m.cnt[RC_NLIST]) return;
    const int h = list[i];
    const u64 key = m.ckey[h];
    int w = 0;
    if (key) {
        Quad q;
        rm_quad(m, h, q);
        w = m.owner[q.a] == key && m.owner[q.b] == key && m.owner[q.c] == key && m.owner[q.d] == key;
    }
    m.win[i] = w;
}

// after a pass's first round: only the entries that bid stay in the list.  (Four in five of the edges below the collapse length fail an
// admission test -- mostly "would create a long edge" -- and would fail it again in every round; the host code looks at an edge once per
// sweep, too.)  The first round's bid kernel flags its bidders (bid_flag) -> prefix sum -> k_rm_relist.
__global__ void k_rm_relist(RM m, const int *__restrict__ list, const int *__restrict__ flag, const int *__restrict__ at, int *__restrict__ list_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m.cnt[RC_NLIST] && flag[i]) list_out[at[i]] = list[i];
}

__global__ void k_rm_relist_commit(RM m, const int *__restrict__ flag, const int *__restrict__ at)
{
    const int n = m.cnt[RC_NLIST];
    if (n > 0) m.cnt[RC_NLIST] = at[n - 1] + flag[n - 1];
}

// the round's winners get their new vertex / faces from the prefix sum over the list; the sizes move on when all of them are in
__global__ void k_rm_split_commit(RM m, unsigned round)
{
    *(volatile int *)(m.host_cand + round) = m.round_cand[round];
    const int n = m.cnt[RC_NLIST];
    if (n <= 0) return;
    const int W = m.scan[n - 1] + m.win[n - 1];
    if (m.cnt[RC_OVERFLOW]) return;
    m.cnt[RC_NV] += W; m.cnt[RC_NF] += 2 * W;
}

// a -> b becomes a -> m -> b; faces (a, b, c), (b, a, d) become (a, m, c), (m, b, c), (b, m, d), (m, a, d).  The two old faces keep their
// slots (and the half-edges c->a, d->b their ids); the two new ones take the slots the prefix sum assigns.
__global__ void k_rm_split_apply(RM m, const int *__restrict__ list)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = m.cnt[RC_NLIST];
    if (i >= n || !m.win[i]) return;
    const int W = m.scan[n - 1] + m.win[n - 1];
    const int nv0 = m.cnt[RC_NV], nf0 = m.cnt[RC_NF];
    if (nv0 + W > m.cnt[RC_VCAP] || nf0 + 2 * W > m.cnt[RC_FCAP]) { m.cnt[RC_OVERFLOW] = 1; return; }      // (every winner sees the same: none is applied)
    const int h = list[i];
    Quad q;
    rm_quad(m, h, q);
    const int k = m.scan[i];
    const int mv = nv0 + k, f2 = nf0 + 2 * k, f3 = f2 + 1;
    const int hn_t = m.twin[q.hn], tn_t = m.twin[q.tn];
    m.pos[mv] = (m.pos[q.a] + m.pos[q.b]) * 0.5;
    m.val[mv] = 4; m.bnd[mv] = 0; m.vhe[mv] = 3 * f2;
    // f0 = (a, m, c) in the slots (h, hn, hp); f1 = (b, m, d) in (t, tn, tp)
    m.F[q.hn] = mv; m.F[q.tn] = mv;
    // f2 = (m, b, c), f3 = (m, a, d)
    m.F[3 * f2] = mv; m.F[3 * f2 + 1] = q.b; m.F[3 * f2 + 2] = q.c;
    m.F[3 * f3] = mv; m.F[3 * f3 + 1] = q.a; m.F[3 * f3 + 2] = q.d;
    m.twin[q.h] = 3 * f3;      m.twin[3 * f3] = q.h;              // a->m | m->a
    m.twin[q.t] = 3 * f2;      m.twin[3 * f2] = q.t;              // b->m | m->b
    m.twin[q.hn] = 3 * f2 + 2; m.twin[3 * f2 + 2] = q.hn;         // m->c | c->m
    m.twin[q.tn] = 3 * f3 + 2; m.twin[3 * f3 + 2] = q.tn;         // m->d | d->m
    m.twin[3 * f2 + 1] = hn_t; if (hn_t >= 0) m.twin[hn_t] = 3 * f2 + 1;     // b->c moved to f2
    m.twin[3 * f3 + 1] = tn_t; if (tn_t >= 0) m.twin[tn_t] = 3 * f3 + 1;     // a->d moved to f3
    if (m.vhe[q.b] == q.hn) m.vhe[q.b] = 3 * f2 + 1;
    if (m.vhe[q.a] == q.tn) m.vhe[q.a] = 3 * f3 + 1;
    m.val[q.c] += 1; m.val[q.d] += 1;
    atomicAdd(&m.cnt[RC_SPLIT], 1);
}

// ---- collapse -------------------------------------------------------------------------------------------------------------------------
// a = origin of h into b = its end (b keeps its position): the admission tests of HalfEdgeMesh::collapse (csrc/remesh.cpp), in its order.
// ra / rh: the neighbours of a and the half-edges to them.
__device__ bool rm_collapse_ok(const RM &m, int h, int *ra, int *rh, int &na)
{
    Quad q;
    if (!rm_quad(m, h, q)) return false;
    const int a = q.a, b = q.b, c = q.c, d = q.d;
    if (m.bnd[a] || m.bnd[b] || m.bnd[c] || m.bnd[d]) return false;
    if (c == d || m.val[a] < 3 || m.val[b] < 3 || m.val[c] <= 3 || m.val[d] <= 3) return false;
    if (m.val[a] + m.val[b] - 4 > m.max_valence || m.val[a] + m.val[b] - 4 < 3) return false;
    na = 0;
    bool overflow = false;
    const bool closed = rm_ring(m, a, [&](int o) { if (na < RM_RING_MAX) { rh[na] = o; ra[na++] = m.F[rm_next(o)]; } else overflow = true; });
    if (!closed || overflow || na != m.val[a]) return false;
    // (the tests are the host code's; their order is not: "would create a long edge" turns away four in five of the short edges and needs
    // a's ring only, so it comes before the walk round b)
    const D3 pb = m.pos[b];
    for (int i = 0; i < na; ++i) {
        const int x = ra[i];
        if (x == b) continue;
        if (norm2(m.pos[x] - pb) > m.high2) return false;
    }
    int common = 0;
    const bool closed_b = rm_ring(m, b, [&](int o) { const int x = m.F[rm_next(o)]; for (int i = 0; i < na; ++i) common += (ra[i] == x); });
    if (!closed_b || common != 2) return false;
    for (int i = 0; i < na; ++i) {
        const int o = rh[i];
        const int x = m.F[rm_next(o)], y = m.F[rm_prev(o)];
        if (x == b || y == b) continue;
        const D3 n0 = rm_face_normal(m, a, x, y), n1 = rm_face_normal(m, b, x, y);
        const double d01 = dot(n0, n1);
        if (!(d01 > 0.0) || d01 * d01 < 0.04 * norm2(n0) * norm2(n1)) return false;
    }
    if (m.twin[q.hn] < 0 || m.twin[q.hp] < 0 || m.twin[q.tn] < 0 || m.twin[q.tp] < 0) return false;
    return true;
}

__global__ void k_rm_collapse_bid(RM m, const int *__restrict__ list, unsigned round, unsigned seed, int *__restrict__ bid_flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m.cnt[RC_NLIST]) return;
    const int h = list[i];
    if (bid_flag) bid_flag[i] = 0;
    if (m.F[h] < 0) { m.ckey[h] = 0; return; }           // (collapsed away by an earlier round; its twin's key went with it)
    const int t = m.twin[h];
    if (t < 0) { m.ckey[h] = 0; return; }
    if (h > t) return;                                   // (see k_rm_split_bid; the lower half-edge's thread, if it is listed, writes both keys)
    u64 kh = 0, kt = 0;
    const int a0 = m.F[h], b0 = m.F[rm_next(h)];
    const double l2 = norm2(m.pos[a0] - m.pos[b0]);
    if (l2 < m.low2) {
        int ra[RM_RING_MAX], rh[RM_RING_MAX], na = 0;
        int e = -1;
        if (rm_collapse_ok(m, h, ra, rh, na)) e = h;
        else if (rm_collapse_ok(m, t, ra, rh, na)) e = t;
        if (e >= 0) {
            const u64 key = rm_key(round, seed, e);
            atomicMax(&m.owner[m.F[e]], key);
            for (int i = 0; i < na; ++i) atomicMax(&m.owner[ra[i]], key);                     // (b is one of them)
            atomicAdd(&m.round_cand[round], 1);
            if (e == h) kh = key; else kt = key;
            if (bid_flag) bid_flag[i] = 1;
        }
    }
    m.ckey[h] = kh; m.ckey[t] = kt;
}

__global__ void k_rm_collapse_apply(RM m, const int *__restrict__ list, unsigned round)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *(volatile int *)(m.host_cand + round) = m.round_cand[round];
    if (i >= m.cnt[RC_NLIST]) return;
    int h = list[i];
    if (m.F[h] < 0) return;
    // (the edge is the business of its lower half-edge, as in the bid -- which chose a direction)
    {
        const int t = m.twin[h];
        if (t < 0 || h > t) return;
        u64 k = m.ckey[h];
        if (!k) { h = t; k = m.ckey[h]; }
        if (!k) return;
    }
    const u64 key = m.ckey[h];
    // A candidate that does not own a must not even look at a's ring: the owner may be rewriting it in this very launch (its faces may
    // be dead already: F = -1).  Owning a makes the ring -- and the two faces at the edge -- this thread's to read; the rest of the
    // footprint is then checked vertex by vertex.
    const int a = m.F[h];
    if (a < 0 || m.owner[a] != key) return;
    Quad q;
    if (!rm_quad(m, h, q)) { m.cnt[RC_CORRUPT] = 1; return; }
    int ra[RM_RING_MAX], rh[RM_RING_MAX], na = 0;
    const bool closed = rm_ring(m, q.a, [&](int o) { if (na < RM_RING_MAX) { rh[na] = o; ra[na++] = m.F[rm_next(o)]; } });
    if (!closed) { m.cnt[RC_CORRUPT] = 1; return; }
    bool mine = true;
    for (int i = 0; i < na; ++i) mine = mine && m.owner[ra[i]] == key;
    if (!mine) return;
    const int hn_t = m.twin[q.hn], hp_t = m.twin[q.hp], tn_t = m.twin[q.tn], tp_t = m.twin[q.tp];
    // every half-edge that left a now leaves b (so every half-edge that pointed to a points to b)
    for (int i = 0; i < na; ++i) m.F[rh[i]] = q.b;
    m.twin[hn_t] = hp_t; m.twin[hp_t] = hn_t;
    m.twin[tn_t] = tp_t; m.twin[tp_t] = tn_t;
    m.vhe[q.b] = tp_t;                                   // b->d
    if (m.vhe[q.c] == q.hp) m.vhe[q.c] = hn_t;           // c->b
    if (m.vhe[q.d] == q.tp) m.vhe[q.d] = tn_t;           // d->(a, now b)
    const int dead[6] = {q.h, q.hn, q.hp, q.t, q.tn, q.tp};
    for (int k = 0; k < 6; ++k) { m.F[dead[k]] = -1; m.twin[dead[k]] = -1; }
    m.val[q.b] = m.val[q.a] + m.val[q.b] - 4; m.val[q.c] -= 1; m.val[q.d] -= 1;
    m.val[q.a] = 0; m.vhe[q.a] = -1;
    atomicAdd(&m.cnt[RC_COLLAPSE], 1);
}

// ---- flip -----------------------------------------------------------------------------------------------------------------------------
// the admission tests of HalfEdgeMesh::flip (csrc/remesh.cpp; with the area test of its strict variant); returns the gain in degree error
__device__ int rm_flip_gain(const RM &m, const Quad &q)
{
    const int a = q.a, b = q.b, c = q.c, d = q.d;
    if (m.bnd[a] || m.bnd[b] || m.bnd[c] || m.bnd[d] || c == d) return 0;
    if (m.val[a] <= 3 || m.val[b] <= 3 || m.val[c] + 1 > m.max_valence || m.val[d] + 1 > m.max_valence) return 0;
    const int before = abs(m.val[a] - 6) + abs(m.val[b] - 6) + abs(m.val[c] - 6) + abs(m.val[d] - 6);
    const int after = abs(m.val[a] - 7) + abs(m.val[b] - 7) + abs(m.val[c] - 5) + abs(m.val[d] - 5);
    if (after >= before) return 0;
    bool hit = false;
    const bool closed = rm_ring(m, c, [&](int o) { hit |= (m.F[rm_next(o)] == d); });
    if (!closed || hit) return 0;
    const D3 n0 = rm_face_normal(m, a, b, c), n1 = rm_face_normal(m, b, a, d);
    const double l0 = norm2(n0), l1 = norm2(n1);
    if (!(l0 > 0) || !(l1 > 0)) return 0;
    if (dot(n0, n1) < 0.3 * sqrt(l0 * l1)) return 0;
    const D3 m0 = rm_face_normal(m, a, d, c), m1 = rm_face_normal(m, d, b, c);
    const D3 navg = n0 * (1.0 / sqrt(l0)) + n1 * (1.0 / sqrt(l1));
    const double q0 = dot(m0, navg), q1 = dot(m1, navg);
    if (!(q0 > 0) || !(q1 > 0)) return 0;
    if (q0 * q0 < 0.04 * norm2(m0) * norm2(navg) || q1 * q1 < 0.04 * norm2(m1) * norm2(navg)) return 0;
    const double amin = 0.01 * fmin(l0, l1);
    if (norm2(m0) < amin || norm2(m1) < amin) return 0;
    return before - after;
}

__global__ void k_rm_flip_bid(RM m, const int *__restrict__ list, unsigned round, unsigned seed)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m.cnt[RC_NLIST]) return;
    const int h = list[i];
    u64 key = 0;
    Quad q;
    if (m.F[h] >= 0 && rm_quad(m, h, q) && h < q.t) {
        const int gain = rm_flip_gain(m, q);
        if (gain > 0) {
            key = rm_key(round, seed, h);
            atomicMax(&m.owner[q.a], key); atomicMax(&m.owner[q.b], key); atomicMax(&m.owner[q.c], key); atomicMax(&m.owner[q.d], key);
            atomicAdd(&m.round_cand[round], 1);
        }
    }
    m.ckey[h] = key;
}

// faces (a, b, c), (b, a, d) become (a, d, c) in the slots (h, hn, hp) and (b, c, d) in (t, tn, tp): c->a and d->b keep their ids
__global__ void k_rm_flip_apply(RM m, const int *__restrict__ list, unsigned round)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *(volatile int *)(m.host_cand + round) = m.round_cand[round];
    if (i >= m.cnt[RC_NLIST]) return;
    const int h = list[i];
    const u64 key = m.ckey[h];
    if (!key) return;
    // (a candidate that owns a and b has the two faces at the edge to itself; one that does not may read them half-rewritten, but whatever
    // it reads there are vertices of the owner's footprint, never its own: it cannot mistake itself for the winner)
    Quad q;
    if (!rm_quad(m, h, q) || h > q.t) return;
    if (!(m.owner[q.a] == key && m.owner[q.b] == key && m.owner[q.c] == key && m.owner[q.d] == key)) return;
    const int hn_t = m.twin[q.hn], tn_t = m.twin[q.tn];
    m.F[q.hn] = q.d;                                      // h: a->d, hn: d->c, hp: c->a
    m.F[q.tn] = q.c;                                      // t: b->c, tn: c->d, tp: d->b
    m.twin[q.h] = tn_t; if (tn_t >= 0) m.twin[tn_t] = q.h;
    m.twin[q.t] = hn_t; if (hn_t >= 0) m.twin[hn_t] = q.t;
    m.twin[q.hn] = q.tn; m.twin[q.tn] = q.hn;
    if (m.vhe[q.a] == q.tn) m.vhe[q.a] = q.h;
    if (m.vhe[q.b] == q.hn) m.vhe[q.b] = q.t;
    m.val[q.a] -= 1; m.val[q.b] -= 1; m.val[q.c] += 1; m.val[q.d] += 1;
    atomicAdd(&m.cnt[RC_FLIP], 1);
}

// ---- tangential relaxation (HalfEdgeMesh::relax, csrc/remesh.cpp) --------------------------------------------------------------------
// every interior vertex moves by l x (the tangential part of the vector to the centroid of its ring); all vertices at once (Jacobi: new
// positions go to `upd`, k_rm_relax_commit copies them back).  The normal is the sum of the ring's face normals (area weights).
__global__ void k_rm_relax(RM m, double l, D3 *__restrict__ upd)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= m.cnt[RC_NV]) return;
    const D3 p = m.pos[v];
    D3 out = p;
    if (m.vhe[v] >= 0 && !m.bnd[v] && m.val[v] >= 3) {
        D3 g{0, 0, 0}, nrm{0, 0, 0};
        int n = 0;
        const bool closed = rm_ring(m, v, [&](int o) {
            const D3 x = m.pos[m.F[rm_next(o)]], y = m.pos[m.F[rm_prev(o)]];
            g = g + x;
            nrm = nrm + cross(x - p, y - p);
            ++n;
        });
        if (closed && n > 0) {
            const D3 d = g * (1.0 / n) - p;
            const double nn = norm2(nrm);
            D3 tang = d;
            if (nn > 0) tang = d - nrm * (dot(d, nrm) / nn);
            out = p + tang * l;
        }
    }
    upd[v] = out;
}

__global__ void k_rm_relax_commit(RM m, const D3 *__restrict__ upd)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < m.cnt[RC_NV]) m.pos[v] = upd[v];
}

// ---- result ---------------------------------------------------------------------------------------------------------------------------
__global__ void k_rm_mark_used(RM m, int nf, int *__restrict__ face_alive, int *__restrict__ vert_used)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    const int a = m.F[3 * f];
    face_alive[f] = a >= 0;
    if (a >= 0) { vert_used[a] = 1; vert_used[m.F[3 * f + 1]] = 1; vert_used[m.F[3 * f + 2]] = 1; }
}

// The vertices of the result are numbered in Morton order of their positions (30-bit key over the input's bounding cube, which holds every
// midpoint; equal keys keep their slot order): what the splits append would otherwise sit at the end of the array in the order of the
// rounds, and every kernel of the next block that gathers by vertex id -- attraction step, 1-rings -- pays for neighbours in space that are
// not neighbours in memory (2-4 % of an iteration after seven remeshing steps, 9 % on a shuffled mesh: tools/experiments/r05_vertex_order.py).
__device__ __forceinline__ unsigned rm_spread10(unsigned v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu; v = (v | (v << 8)) & 0x0300f00fu; v = (v | (v << 4)) & 0x030c30c3u; v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void k_rm_vertex_keys(RM m, int nv, const int *__restrict__ vert_used, const int *__restrict__ vert_at, double lox, double loy, double loz, double inv_unit,
                                 unsigned *__restrict__ key, int *__restrict__ idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv || !vert_used[i]) return;
    const int o = vert_at[i];
    const D3 p = m.pos[i];
    const unsigned qx = (unsigned)fmin(1023.0, fmax(0.0, (p.x - lox) * inv_unit)), qy = (unsigned)fmin(1023.0, fmax(0.0, (p.y - loy) * inv_unit)),
                   qz = (unsigned)fmin(1023.0, fmax(0.0, (p.z - loz) * inv_unit));
    key[o] = rm_spread10(qx) | (rm_spread10(qy) << 1) | (rm_spread10(qz) << 2);
    idx[o] = o;
}

__global__ void k_rm_rank(const int *__restrict__ order, int n, int *__restrict__ rank)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) rank[order[r]] = r;
}

__global__ void k_rm_write_out(RM m, int nv, int nf, const int *__restrict__ face_alive, const int *__restrict__ face_at, const int *__restrict__ vert_used,
                               const int *__restrict__ vert_slot, const int *__restrict__ rank, float *__restrict__ ov, int *__restrict__ of, double *__restrict__ part)
{
    __shared__ double s[256];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nv && vert_used[i]) {
        const int o = rank[vert_slot[i]];
        ov[3 * o] = (float)m.pos[i].x; ov[3 * o + 1] = (float)m.pos[i].y; ov[3 * o + 2] = (float)m.pos[i].z;
        atomicMax(&m.cnt[RC_MAXVAL], m.val[i]);
    }
    double len = 0.0;
    if (i < nf && face_alive[i]) {
        const int o = face_at[i];
        const int a = m.F[3 * i], b = m.F[3 * i + 1], c = m.F[3 * i + 2];
        of[3 * o] = rank[vert_slot[a]]; of[3 * o + 1] = rank[vert_slot[b]]; of[3 * o + 2] = rank[vert_slot[c]];
        len = sqrt(norm2(m.pos[a] - m.pos[b])) + sqrt(norm2(m.pos[b] - m.pos[c])) + sqrt(norm2(m.pos[c] - m.pos[a]));
    }
    s[threadIdx.x] = len;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = s[0];
}

// ---- host side --------------------------------------------------------------------------------------------------------------------------
// Device memory of a call is carved out of two blocks that stay allocated between calls (a call makes some thirty arrays; hipMalloc and
// hipFree of a few MB each took longer than the kernels).  A block grows to what the last call wanted; what does not fit is allocated
// and freed the slow way.  The blocks, the stream and the pinned words belong to one device at a time and are guarded by g_lock (calls are
// serialised).
struct Cache { void *p = nullptr; size_t cap = 0, want = 0; };
struct DevBuf {
    Cache &c;
    size_t used = 0;
    std::vector<void *> extra;
    explicit DevBuf(Cache &cache) : c(cache)
    {
        if (c.want > c.cap) {
            if (c.p) (void)hipFree(c.p);
            c.p = nullptr; c.cap = 0;
            const size_t n = c.want + c.want / 4;
            if (n <= ((size_t)4 << 30) && hipMalloc(&c.p, n) == hipSuccess) c.cap = n;
        }
    }
    template <class T> T *get(size_t n)
    {
        const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
        const size_t at = used;
        used += bytes;
        if (at + bytes <= c.cap) return (T *)((char *)c.p + at);
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        extra.push_back(p);
        return (T *)p;
    }
    ~DevBuf()
    {
        for (void *p : extra) (void)hipFree(p);
        c.want = std::max(c.want, used);
    }
};
struct Session {
    int device = -1;
    Cache main, temp;
    hipStream_t stream = nullptr;
    int *h_cand = nullptr;
};
Session g_session;
std::mutex g_lock;

// (inside attempt(): an early exit first drains the stream -- asynchronous copies into this frame's arrays or the caller's vectors may be in flight)
#define RM_RETURN(code) do { (void)hipStreamSynchronize(g_session.stream); return (code); } while (0)
#define RM_HIP(x) do { if ((x) != hipSuccess) RM_RETURN(NW_ERR_HIP); } while (0)
#define RM_GRID(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256)

enum { RM_RETRY = 1 };        // an attempt ran out of room
enum { ROUNDS_CAP = 4096 };   // rounds of one call (numbered through; a call that needs more stops early)

static int exclusive_scan(const int *in, int *out, int n, void *tmp, size_t tmp_bytes, hipStream_t s)
{
    return hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, in, out, n, s) == hipSuccess ? NW_OK : NW_ERR_HIP;
}

static int attempt(const float *vertices, int64_t nv_in, const int32_t *faces, int64_t nf_in, int n_iterations, double L, double relax_lambda, int n_relax, int max_valence, double room,
                   std::vector<float> &ov, std::vector<int32_t> &of, nw_remesh_stats *stats, bool verbose)
{
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    hipStream_t st = g_session.stream;
    DevBuf B(g_session.main);
    const double high = 4.0 / 3.0 * L, low = 4.0 / 5.0 * L;
    // inputs on the device, the number of faces the lengths call for
    float *d_vin = B.get<float>(3 * (size_t)nv_in);
    int *d_fin = B.get<int>(3 * (size_t)nf_in);
    int *d_cnt = B.get<int>(RC_COUNT);
    const int nblk_f = (int)((nf_in + 255) / 256);
    double *d_part0 = B.get<double>((size_t)nblk_f);
    if (!d_vin || !d_fin || !d_cnt || !d_part0) RM_RETURN(NW_ERR_NOMEM);
    RM_HIP(hipMemcpyAsync(d_vin, vertices, sizeof(float) * 3 * (size_t)nv_in, hipMemcpyHostToDevice, st));
    RM_HIP(hipMemcpyAsync(d_fin, faces, sizeof(int) * 3 * (size_t)nf_in, hipMemcpyHostToDevice, st));
    RM_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int) * RC_COUNT, st));
    // a first, small mesh record just to measure the input (positions + faces)
    RM m{};
    m.cnt = d_cnt; m.max_valence = max_valence; m.high2 = high * high; m.low2 = low * low;
    // capacity is decided after the pieces are counted; the load needs pos / F / vhe / val / bnd of the input's size at least, so the
    // count runs on temporaries of that size
    {
        DevBuf T(g_session.temp);
        RM t = m;
        t.pos = T.get<D3>((size_t)nv_in); t.F = T.get<int>(3 * (size_t)nf_in); t.vhe = T.get<int>((size_t)nv_in); t.val = T.get<int>((size_t)nv_in);
        t.bnd = T.get<unsigned char>((size_t)nv_in);
        if (!t.pos || !t.F || !t.vhe || !t.val || !t.bnd) RM_RETURN(NW_ERR_NOMEM);
        hipLaunchKernelGGL(k_rm_load, RM_GRID(std::max(nv_in, nf_in)), 0, st, d_vin, (int)nv_in, d_fin, (int)nf_in, t);
        hipLaunchKernelGGL(k_rm_pieces, dim3(nblk_f), dim3(256), 0, st, t, (int)nf_in, high, d_part0);
        std::vector<double> part((size_t)nblk_f);
        int cnt[RC_COUNT];
        RM_HIP(hipMemcpyAsync(part.data(), d_part0, sizeof(double) * (size_t)nblk_f, hipMemcpyDeviceToHost, st));
        RM_HIP(hipMemcpyAsync(cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, st));
        RM_HIP(hipStreamSynchronize(st));
        if (cnt[RC_BADARG]) RM_RETURN(NW_ERR_BADARG);
        double pieces = 0.0;
        for (double p : part) pieces += p;
        if (!(pieces < 67108864.0)) RM_RETURN(NW_ERR_BADARG);                      // (the host code's "runaway": a vertex flung far away)
        m.high2 = high * high;
        // room for the faces: what the lengths call for (or the input, if that is more), times `room`
        const double want = std::max(pieces, (double)nf_in) * room + 8192.0;
        if (want > 5.0e8) RM_RETURN(NW_ERR_NOMEM);
        const size_t Fcap = (size_t)want, Vcap = (size_t)nv_in + (Fcap - (size_t)nf_in) / 2 + 1024, Hcap = 3 * Fcap;
        m.pos = B.get<D3>(Vcap); m.F = B.get<int>(Hcap); m.twin = B.get<int>(Hcap); m.vhe = B.get<int>(Vcap); m.val = B.get<int>(Vcap);
        m.bnd = B.get<unsigned char>(Vcap); m.owner = B.get<u64>(Vcap); m.ckey = B.get<u64>(Hcap); m.win = B.get<int>(Hcap); m.scan = B.get<int>(Hcap);
        if (!m.pos || !m.F || !m.twin || !m.vhe || !m.val || !m.bnd || !m.owner || !m.ckey || !m.win || !m.scan) RM_RETURN(NW_ERR_NOMEM);
        size_t tmp_bytes = 0;
        if (hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, m.win, m.scan, (int)Hcap, st) != hipSuccess) RM_RETURN(NW_ERR_HIP);
        void *d_tmp = B.get<unsigned char>(tmp_bytes + 16);
        if (!d_tmp) RM_RETURN(NW_ERR_NOMEM);
        hipLaunchKernelGGL(k_rm_load, RM_GRID(std::max(nv_in, nf_in)), 0, st, d_vin, (int)nv_in, d_fin, (int)nf_in, m);
        // twins through a hash table of the directed edges
        unsigned hcap = 1;
        while (hcap < 2 * 3 * (size_t)nf_in) hcap <<= 1;
        u64 *d_keys = T.get<u64>(hcap);
        int *d_vals = T.get<int>(hcap);
        if (!d_keys || !d_vals) RM_RETURN(NW_ERR_NOMEM);
        RM_HIP(hipMemsetAsync(d_keys, 0xff, sizeof(u64) * hcap, st));
        const int nh_in = (int)(3 * nf_in);
        hipLaunchKernelGGL(k_rm_hash_insert, RM_GRID(nh_in), 0, st, m, nh_in, d_keys, d_vals, hcap - 1);
        hipLaunchKernelGGL(k_rm_hash_lookup, RM_GRID(nh_in), 0, st, m, nh_in, d_keys, d_vals, hcap - 1);
        hipLaunchKernelGGL(k_rm_vertex_flags, RM_GRID(nv_in), 0, st, m, (int)nv_in);
        RM_HIP(hipMemcpyAsync(cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, st));
        RM_HIP(hipStreamSynchronize(st));
        if (cnt[RC_NONMANIFOLD]) RM_RETURN(NW_ERR_NONMANIFOLD);
        const auto t1 = now();

        // sizes live on the device from here on (the splits move them); the host reads them where it needs a grid size
        int *d_list = T.get<int>(Hcap / 2 + 64), *d_list2 = T.get<int>(Hcap / 2 + 64);
        D3 *d_upd = n_relax > 0 ? T.get<D3>(Vcap) : nullptr;
        if (!d_list || !d_list2 || (n_relax > 0 && !d_upd)) RM_RETURN(NW_ERR_NOMEM);
        {
            int init[RC_COUNT] = {0};
            init[RC_NV] = (int)nv_in; init[RC_NF] = (int)nf_in; init[RC_VCAP] = (int)Vcap; init[RC_FCAP] = (int)Fcap;
            RM_HIP(hipMemcpyAsync(d_cnt, init, sizeof(init), hipMemcpyHostToDevice, st));
            RM_HIP(hipStreamSynchronize(st));
        }
        // rounds are numbered through the whole call; how many candidates bid in a round is written to pinned host memory by the round's
        // apply kernel, so the host -- a round or two ahead with its launches -- sees a pass run dry without waiting for anything
        m.round_cand = T.get<int>(ROUNDS_CAP);
        int *h_cand = g_session.h_cand;
        for (int i = 0; i < ROUNDS_CAP; ++i) h_cand[i] = -1;
        m.host_cand = h_cand;
        if (!m.round_cand) RM_RETURN(NW_ERR_NOMEM);
        RM_HIP(hipMemsetAsync(m.round_cand, 0, sizeof(int) * ROUNDS_CAP, st));
        RM_HIP(hipMemsetAsync(m.owner, 0, sizeof(u64) * Vcap, st));
        int nv = (int)nv_in, nf = (int)nf_in;
        int rounds[3] = {0, 0, 0};
        unsigned round_id = 0, pass_seq = 0;
        auto read_cnt = [&](int *c) -> int {
            if (hipMemcpyAsync(c, d_cnt, sizeof(int) * RC_COUNT, hipMemcpyDeviceToHost, st) != hipSuccess) RM_RETURN(NW_ERR_HIP);
            return hipStreamSynchronize(st) == hipSuccess ? NW_OK : NW_ERR_HIP;
        };
        // one pass of one kind: the candidate list from a scan over all half-edges, then rounds over the list.  Returns the list's length.
        const int R_MAX[3] = {24, 32, 24};
        auto pass = [&](int kind, int *n_list_out) -> int {
            const int nh = 3 * nf;
            if (kind == 0) hipLaunchKernelGGL(k_rm_candidates<0>, RM_GRID(nh), 0, st, m, m.win);
            else if (kind == 1) hipLaunchKernelGGL(k_rm_candidates<1>, RM_GRID(nh), 0, st, m, m.win);
            else hipLaunchKernelGGL(k_rm_candidates<2>, RM_GRID(nh), 0, st, m, m.win);
            if (exclusive_scan(m.win, m.scan, nh, d_tmp, tmp_bytes, st) != NW_OK) RM_RETURN(NW_ERR_HIP);
            hipLaunchKernelGGL(k_rm_compact, RM_GRID(nh), 0, st, m, m.win, m.scan, d_list);
            int c[RC_COUNT];
            if (read_cnt(c) != NW_OK) RM_RETURN(NW_ERR_HIP);
            const int n = c[RC_NLIST];
            *n_list_out = n;
            if (n == 0) return NW_OK;
            if ((size_t)n > Hcap / 2 + 64) RM_RETURN(NW_ERR_INTERNAL);
            const unsigned first = round_id + 1;
            const unsigned pass_no = ++pass_seq;
            const int *list = d_list;
            int first_bids = -1;
            for (int r = 0; r < R_MAX[kind]; ++r) {
                // The host may run at most two rounds ahead of the GPU's reports: it waits (a spin on pinned memory) for the count of the round
                // before last, and stops the pass when a reported round had no bidder -- or so few that the next iteration may as well
                // have them (a tail of a handful of candidates that keep losing to each other took as many rounds as all the others).
                if (r >= 2) {
                    const volatile int *w = h_cand + (first + (unsigned)r - 2);
                    const auto t_w = now();
                    while (*w < 0 && ms(t_w, now()) < 2000.0) {}
                    if (*w < 0) RM_RETURN(NW_ERR_INTERNAL);
                }
                bool dry = false;
                // (an empty round may be noticed whenever its report happens to be in -- the rounds behind it do nothing either way --, but the
                // "so few" rule decides whether real work is done: it looks at the round before last only, whose report the host has waited for)
                for (unsigned k = first; k <= round_id; ++k) {
                    const int b = *(volatile int *)(h_cand + k);
                    if (k == first && b >= 0) first_bids = b;
                    dry = dry || b == 0 || (r >= 2 && k + 2 <= first + (unsigned)r && b > 0 && first_bids > 0 && b < 8 && b * 500 < first_bids);
                }
                if (dry || round_id + 1 >= (unsigned)ROUNDS_CAP) break;
                const unsigned round = ++round_id, seed = pass_no * 64u + (unsigned)r;
                if (kind == 0) {
                    hipLaunchKernelGGL(k_rm_split_bid, RM_GRID(n), 0, st, m, list, round, seed);
                    hipLaunchKernelGGL(k_rm_split_mark, RM_GRID(n), 0, st, m, list);
                    if (exclusive_scan(m.win, m.scan, n, d_tmp, tmp_bytes, st) != NW_OK) RM_RETURN(NW_ERR_HIP);
                    hipLaunchKernelGGL(k_rm_split_apply, RM_GRID(n), 0, st, m, list);
                    hipLaunchKernelGGL(k_rm_split_commit, dim3(1), dim3(1), 0, st, m, round);
                } else if (kind == 1) {
                    hipLaunchKernelGGL(k_rm_collapse_bid, RM_GRID(n), 0, st, m, list, round, seed, r == 0 ? m.win : (int *)nullptr);
                    hipLaunchKernelGGL(k_rm_collapse_apply, RM_GRID(n), 0, st, m, list, round);
                    if (r == 0) {
                        if (exclusive_scan(m.win, m.scan, n, d_tmp, tmp_bytes, st) != NW_OK) RM_RETURN(NW_ERR_HIP);
                        hipLaunchKernelGGL(k_rm_relist, RM_GRID(n), 0, st, m, list, m.win, m.scan, d_list2);
                        hipLaunchKernelGGL(k_rm_relist_commit, dim3(1), dim3(1), 0, st, m, m.win, m.scan);
                        list = d_list2;
                    }
                } else {
                    hipLaunchKernelGGL(k_rm_flip_bid, RM_GRID(n), 0, st, m, list, round, seed);
                    hipLaunchKernelGGL(k_rm_flip_apply, RM_GRID(n), 0, st, m, list, round);
                }
                ++rounds[kind];
            }
            if (std::getenv("NWR_VERBOSE") && std::atoi(std::getenv("NWR_VERBOSE")) >= 2) {
                (void)hipStreamSynchronize(st);
                std::fprintf(stderr, "[nw_remesh_device]     kind %d: list %d, bids per round:", kind, n);
                for (unsigned k = first; k <= round_id; ++k) std::fprintf(stderr, " %d", h_cand[k]);
                std::fprintf(stderr, "\n");
            }
            return NW_OK;
        };
        for (int it = 0; it < n_iterations; ++it) {
            int before[RC_COUNT], c[RC_COUNT], n_list = 0, rc;
            if (read_cnt(before) != NW_OK) RM_RETURN(NW_ERR_HIP);
            // split: what a pass creates is looked at by the next one (the host code's sweeps: up to 8)
            int first_list = 0;
            for (int sub = 0; sub < 4; ++sub) {
                if ((rc = pass(0, &n_list)) != NW_OK) RM_RETURN(rc);
                if (n_list == 0) break;
                if (sub == 0) first_list = n_list;
                if (read_cnt(c) != NW_OK) RM_RETURN(NW_ERR_HIP);
                if (c[RC_OVERFLOW]) RM_RETURN(RM_RETRY);
                nv = c[RC_NV]; nf = c[RC_NF];
                // (what the splits of a sweep leave too long is the next sweep's; a sweep over a handful of edges -- a scan over all half-edges, a
                // compaction and a few rounds for six edges of 4 10^5 -- is left to the next iteration, as the rounds' tails are)
                if (n_list < 32 && n_list * 200 < first_list) break;
            }
            if ((rc = pass(1, &n_list)) != NW_OK) RM_RETURN(rc);
            if ((rc = pass(2, &n_list)) != NW_OK) RM_RETURN(rc);
            for (int k = 0; k < n_relax; ++k) {
                hipLaunchKernelGGL(k_rm_relax, RM_GRID(nv), 0, st, m, relax_lambda, d_upd);
                hipLaunchKernelGGL(k_rm_relax_commit, RM_GRID(nv), 0, st, m, d_upd);
            }
            if (read_cnt(c) != NW_OK) RM_RETURN(NW_ERR_HIP);
            if (c[RC_CORRUPT]) RM_RETURN(NW_ERR_INTERNAL);
            if (verbose) std::fprintf(stderr, "[nw_remesh_device] iteration %d: %d / %d / %d operations so far, %d vertex slots, %d face slots; rounds so far %d / %d / %d\n", it,
                                      c[RC_SPLIT], c[RC_COLLAPSE], c[RC_FLIP], nv, nf, rounds[0], rounds[1], rounds[2]);
            // a pass that changed nothing would be repeated unchanged by every later iteration
            if (n_relax == 0 && c[RC_SPLIT] == before[RC_SPLIT] && c[RC_COLLAPSE] == before[RC_COLLAPSE] && c[RC_FLIP] == before[RC_FLIP]) break;      // (with relaxation every vertex moves: the next iteration sees another mesh)
        }
        const auto t2 = now();
        // compact: faces that are alive, vertices they refer to (relative order kept)
        int *d_alive = T.get<int>((size_t)nf + 1), *d_face_at = T.get<int>((size_t)nf + 1), *d_used = T.get<int>((size_t)nv + 1), *d_vert_at = T.get<int>((size_t)nv + 1);
        const int nblk = (std::max(nv, nf) + 255) / 256;
        double *d_part = T.get<double>((size_t)nblk);
        if (!d_alive || !d_face_at || !d_used || !d_vert_at || !d_part) RM_RETURN(NW_ERR_NOMEM);
        RM_HIP(hipMemsetAsync(d_used, 0, sizeof(int) * ((size_t)nv + 1), st));
        RM_HIP(hipMemsetAsync(d_alive + nf, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_rm_mark_used, RM_GRID(nf), 0, st, m, nf, d_alive, d_used);
        if (exclusive_scan(d_alive, d_face_at, nf + 1, d_tmp, tmp_bytes, st) != NW_OK) RM_RETURN(NW_ERR_HIP);
        if (exclusive_scan(d_used, d_vert_at, nv + 1, d_tmp, tmp_bytes, st) != NW_OK) RM_RETURN(NW_ERR_HIP);
        int n_out[2];
        RM_HIP(hipMemcpyAsync(&n_out[0], d_face_at + nf, sizeof(int), hipMemcpyDeviceToHost, st));
        RM_HIP(hipMemcpyAsync(&n_out[1], d_vert_at + nv, sizeof(int), hipMemcpyDeviceToHost, st));
        RM_HIP(hipStreamSynchronize(st));
        const int nf_out = n_out[0], nv_out = n_out[1];
        float *d_ov = T.get<float>(3 * (size_t)nv_out);
        int *d_of = T.get<int>(3 * (size_t)nf_out);
        unsigned *d_key = T.get<unsigned>((size_t)nv_out), *d_key2 = T.get<unsigned>((size_t)nv_out);
        int *d_idx = T.get<int>((size_t)nv_out), *d_order = T.get<int>((size_t)nv_out), *d_rank = T.get<int>((size_t)nv_out);
        if (!d_ov || !d_of || !d_key || !d_key2 || !d_idx || !d_order || !d_rank) RM_RETURN(NW_ERR_NOMEM);
        {
            // Morton order of the result's vertices (see k_rm_vertex_keys); the bounding cube is the input's
            double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int64_t v = 0; v < nv_in; ++v) for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], (double)vertices[3 * v + k]); hi[k] = std::max(hi[k], (double)vertices[3 * v + k]); }
            const double ext = std::max({hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], 1e-30});
            hipLaunchKernelGGL(k_rm_vertex_keys, RM_GRID(nv), 0, st, m, nv, d_used, d_vert_at, lo[0], lo[1], lo[2], 1024.0 / ext, d_key, d_idx);
            size_t sort_bytes = 0;
            if (hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, d_key, d_key2, d_idx, d_order, nv_out, 0, 30, st) != hipSuccess) RM_RETURN(NW_ERR_HIP);
            void *d_sort = T.get<unsigned char>(sort_bytes + 16);
            if (!d_sort) RM_RETURN(NW_ERR_NOMEM);
            if (hipcub::DeviceRadixSort::SortPairs(d_sort, sort_bytes, d_key, d_key2, d_idx, d_order, nv_out, 0, 30, st) != hipSuccess) RM_RETURN(NW_ERR_HIP);
            hipLaunchKernelGGL(k_rm_rank, RM_GRID(nv_out), 0, st, d_order, nv_out, d_rank);
        }
        hipLaunchKernelGGL(k_rm_write_out, dim3(nblk), dim3(256), 0, st, m, nv, nf, d_alive, d_face_at, d_used, d_vert_at, d_rank, d_ov, d_of, d_part);
        ov.resize(3 * (size_t)nv_out);
        of.resize(3 * (size_t)nf_out);
        std::vector<double> lens((size_t)nblk);
        int c[RC_COUNT];
        RM_HIP(hipMemcpyAsync(ov.data(), d_ov, sizeof(float) * ov.size(), hipMemcpyDeviceToHost, st));
        RM_HIP(hipMemcpyAsync(of.data(), d_of, sizeof(int) * of.size(), hipMemcpyDeviceToHost, st));
        RM_HIP(hipMemcpyAsync(lens.data(), d_part, sizeof(double) * (size_t)nblk, hipMemcpyDeviceToHost, st));
        RM_HIP(hipMemcpyAsync(c, d_cnt, sizeof(c), hipMemcpyDeviceToHost, st));
        RM_HIP(hipStreamSynchronize(st));
        if (stats) {
            double s = 0.0;
            for (double l : lens) s += l;
            stats->n_split = c[RC_SPLIT]; stats->n_collapse = c[RC_COLLAPSE]; stats->n_flip = c[RC_FLIP];
            stats->mean_edge_length = nf_out ? s / (3.0 * nf_out) : 0.0;
            stats->max_valence = c[RC_MAXVAL];
            stats->rounds_split = rounds[0]; stats->rounds_collapse = rounds[1]; stats->rounds_flip = rounds[2];
        }
        if (verbose) std::fprintf(stderr, "[nw_remesh_device] set-up %.2f ms, passes %.2f ms (%d + %d + %d rounds), result %.2f ms\n", ms(t0, t1), ms(t1, t2), rounds[0], rounds[1],
                                  rounds[2], ms(t2, now()));
    }
    return NW_OK;
}

}  // namespace

NW_EXPORT int nw_remesh_device(int device, const float *vertices, int64_t n_vertices, const int32_t *faces, int64_t n_faces, int n_iterations,
                               float target_edge_length, float relax_lambda, int n_relax, int max_valence, float **out_vertices, int64_t *out_n_vertices,
                               int32_t **out_faces, int64_t *out_n_faces, nw_remesh_stats *stats)
{
    if (!vertices || !faces || !out_vertices || !out_n_vertices || !out_faces || !out_n_faces) return NW_ERR_BADARG;
    if (n_vertices < 3 || n_faces < 1 || n_vertices > (1ll << 28) || n_faces > (1ll << 28) || n_iterations < 0 || n_relax < 0 || n_relax > 1000 || !(target_edge_length > 0.0f)) return NW_ERR_BADARG;
    *out_vertices = nullptr; *out_faces = nullptr; *out_n_vertices = 0; *out_n_faces = 0;
    std::lock_guard<std::mutex> lock(g_lock);
    // (the calling thread's current device is the caller's business: put back on every way out)
    int caller_device = -1;
    if (hipGetDevice(&caller_device) != hipSuccess) return NW_ERR_HIP;
    struct DeviceGuard { int d; ~DeviceGuard() { if (d >= 0) (void)hipSetDevice(d); } } device_guard{caller_device};
    if (hipSetDevice(device) != hipSuccess) return NW_ERR_HIP;
    if (g_session.device != device) {
        // (another device than last time: what was cached belongs to the old one)
        if (g_session.device >= 0 && hipSetDevice(g_session.device) == hipSuccess) {
            if (g_session.main.p) (void)hipFree(g_session.main.p);
            if (g_session.temp.p) (void)hipFree(g_session.temp.p);
            if (g_session.stream) (void)hipStreamDestroy(g_session.stream);
            if (g_session.h_cand) (void)hipHostFree(g_session.h_cand);
            (void)hipSetDevice(device);
        }
        g_session = Session();
        if (hipStreamCreateWithFlags(&g_session.stream, hipStreamNonBlocking) != hipSuccess) return NW_ERR_HIP;
        if (hipHostMalloc((void **)&g_session.h_cand, sizeof(int) * ROUNDS_CAP, hipHostMallocDefault) != hipSuccess) { (void)hipStreamDestroy(g_session.stream); g_session = Session(); return NW_ERR_NOMEM; }
        g_session.device = device;
    }
    const bool verbose = std::getenv("NWR_VERBOSE") != nullptr;
    const int mv = max_valence > 0 ? std::min(max_valence, 60) : 16;
    try {
        std::vector<float> ov;
        std::vector<int32_t> of;
        int rc = RM_RETRY;
        double room = 1.5;
        if (const char *e = std::getenv("NW_REMESH_ROOM")) room = std::max(0.05, std::atof(e));      // (tests: start too small, so that the retry runs)
        // (slots of faces and vertices that die are not used again within a call: a target far below the input's lengths, where the split pass
        // overshoots and the collapses take a third back, needs several times the final size -- the attempt that runs out stops at once)
        for (int tries = 0; tries < 8 && rc == RM_RETRY; ++tries, room *= 2.0) {
            rc = attempt(vertices, n_vertices, faces, n_faces, n_iterations, (double)target_edge_length, (double)relax_lambda, n_relax, mv, room, ov, of, stats, verbose);
            if (rc == RM_RETRY && verbose) std::fprintf(stderr, "[nw_remesh_device] out of room at %.1f x the expected faces: again with twice that\n", room);
        }
        if (rc == RM_RETRY) return NW_ERR_NOMEM;
        if (rc != NW_OK) return rc;
        float *pv = (float *)std::malloc(sizeof(float) * std::max<size_t>(ov.size(), 3));
        int32_t *pf = (int32_t *)std::malloc(sizeof(int32_t) * std::max<size_t>(of.size(), 3));
        if (!pv || !pf) { std::free(pv); std::free(pf); return NW_ERR_NOMEM; }
        std::memcpy(pv, ov.data(), sizeof(float) * ov.size());
        std::memcpy(pf, of.data(), sizeof(int32_t) * of.size());
        *out_vertices = pv; *out_n_vertices = (int64_t)(ov.size() / 3); *out_faces = pf; *out_n_faces = (int64_t)(of.size() / 3);
        return NW_OK;
    } catch (const std::exception &) {
        (void)hipStreamSynchronize(g_session.stream);       // (as RM_RETURN: nothing of the call may still be in flight)
        return NW_ERR_NOMEM;
    }
}

NW_EXPORT void nw_host_free(void *p) { std::free(p); }
