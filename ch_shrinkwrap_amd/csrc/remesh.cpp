// Block-boundary isotropic remesher (host code, C-ABI in include/nw_remesh.h).
//
// Stands in for PYME's TriangleMesh.remesh, which the reference calls between optimiser blocks
// (/root/reference/ch_shrinkwrap/_membrane_mesh.pyx:1546, :1219) but which is not part of the reference tree.  Built from the
// published algorithm (Botsch & Kobbelt 2004): split long edges, collapse short edges, flip towards degree six, optional
// tangential relaxation, on an array-based half-edge structure.  Half-edge h points TO vert[h]; its origin is vert[prev[h]].
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include <chrono>
#include <cstdio>
#include <thread>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <pthread.h>
#include <exception>
#include <system_error>
#include <utility>
#include <climits>

#include "../../include/nw_remesh.h"

#define NWR_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

struct V3 {
    double x, y, z;
};
inline V3 operator-(const V3 &a, const V3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(const V3 &a, const V3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(const V3 &a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3 &a, const V3 &b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm2(const V3 &a) { return dot(a, a); }

int n_threads()
{
    if (const char *e = std::getenv("NW_REMESH_THREADS")) return std::max(1, std::atoi(e));
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(hc ? hc : 1u, 16u));
}

// NWR_VERBOSE=2: wall time of the stages between two marks (a call's serial stretches are what is left to shorten)
struct StageClock {
    const bool on = std::getenv("NWR_VERBOSE") && std::atoi(std::getenv("NWR_VERBOSE")) >= 2;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char *what)
    {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[nw_remesh]     %-28s %.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// Worker threads that live as long as the process (started on first use): a remesh of 4 10^5 faces is a dozen short parallel loops plus two
// rounds of pieces, and starting 15 threads for each of them cost more than some of the loops.  One job at a time; the caller works too.
// A caller that finds the pool taken (two Python threads in the library at once) or that is itself a worker (a loop inside a piece) runs
// its job on its own thread -- every job is written so that its result does not depend on who runs which part.
class Pool {
public:
    // task(i) for every i in [0, n), handed out one at a time, on at most max_threads threads (the caller included).  An exception in a
    // task (std::bad_alloc of a scratch vector) never leaves its thread: the first one is kept and re-thrown on the calling thread.
    template <class Fn>
    void run(int n, int max_threads, Fn task)
    {
        if (n <= 0) return;
        std::exception_ptr err;
        std::mutex err_m;
        auto guarded = [&](int i) {
            try { task(i); }
            catch (...) { std::lock_guard<std::mutex> lk(err_m); if (!err) err = std::current_exception(); }
        };
        const int want = std::min(std::min(max_threads, n), n_threads());
        if (want <= 1 || tl_worker || !busy_.try_lock()) {
            for (int i = 0; i < n; ++i) guarded(i);
        } else {
            std::lock_guard<std::mutex> hold(busy_, std::adopt_lock);
            ensure_threads(want - 1);
            std::atomic<int> next{0};
            auto drain = [&] { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) guarded(i); };
            std::function<void()> job = drain;
            const int helpers = std::min(want - 1, (int)th_.size());
            {
                std::lock_guard<std::mutex> lk(m_);
                job_ = &job; wanted_ = helpers; running_ = 0; ++gen_;
            }
            if (helpers > 0) cv_work_.notify_all();
            drain();
            {
                // no helper may still be inside `job` (or about to enter it) when it goes out of scope
                std::unique_lock<std::mutex> lk(m_);
                wanted_ = 0;                                  // (helpers that have not started yet stay out)
                cv_done_.wait(lk, [&] { return running_ == 0; });
                job_ = nullptr;
            }
        }
        if (err) std::rethrow_exception(err);
    }

    static Pool &get()
    {
        static std::once_flag once;
        std::call_once(once, [] {
            g_pool = new Pool();                              // (never destroyed: its threads wait on a condition variable until the process ends)
            pthread_atfork(nullptr, nullptr, [] { g_pool = new Pool(); });     // a forked child has none of the threads: a new pool, started on first use
        });
        return *g_pool;
    }

private:
    static Pool *g_pool;
    static thread_local bool tl_worker;
    std::mutex busy_, m_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> th_;
    std::function<void()> *job_ = nullptr;
    int wanted_ = 0, running_ = 0;
    uint64_t gen_ = 0;

    void ensure_threads(int n)
    {
        while ((int)th_.size() < n) {
            try { th_.emplace_back([this] { worker(); }); th_.back().detach(); }
            catch (...) { break; }                            // (fewer threads than wanted: the others take the work)
        }
    }
    void worker()
    {
        tl_worker = true;
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_work_.wait(lk, [&] { return gen_ != seen; });
            seen = gen_;
            if (wanted_ <= 0 || !job_) continue;
            --wanted_; ++running_;
            std::function<void()> *job = job_;
            lk.unlock();
            (*job)();
            lk.lock();
            if (--running_ == 0) cv_done_.notify_all();
        }
    }
};
Pool *Pool::g_pool = nullptr;
thread_local bool Pool::tl_worker = false;

// fn(lo, hi) over [0, n) cut into contiguous chunks, one per thread.  Only for loops whose iterations write disjoint outputs: the result
// does not depend on the number of threads.
template <class Fn>
void parallel_for(int64_t n, int64_t min_chunk, Fn fn)
{
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads(), n / std::max<int64_t>(min_chunk, 1)));
    if (T <= 1) { fn((int64_t)0, n); return; }
    Pool::get().run(T, T, [&](int t) { fn(n * t / T, n * (t + 1) / T); });
}

// twin[3f+k] = the half-edge running the other way along edge (faces[f][k], faces[f][k+1]), -1 on a boundary.  Linear time:
// the half-edges are bucketed by their origin vertex, and the twin of a->b is looked up among the handful leaving b.
int match_twins(const int32_t *faces, int64_t nf, int64_t nv, int *twin)
{
    const int64_t nh = 3 * nf;
    StageClock clk;
    std::vector<int> first(nv + 1, 0), out(nh);
    for (int64_t h = 0; h < nh; ++h) first[faces[h] + 1] += 1;
    for (int64_t v = 0; v < nv; ++v) first[v + 1] += first[v];
    if (nf > 100000) clk.mark("twins: count");
    {
        std::vector<int> fill(first.begin(), first.end() - 1);
        for (int64_t h = 0; h < nh; ++h) out[fill[faces[h]]++] = (int)h;
    }
    if (nf > 100000) clk.mark("twins: fill");
    // target vertex of every half-edge, laid out next to the bucket entries (one sequential read per candidate)
    std::vector<int> tgt(nh), out_tgt(nh);
    parallel_for(nf, 1 << 15, [&](int64_t lo, int64_t hi) {
        for (int64_t f = lo; f < hi; ++f) { tgt[3 * f] = faces[3 * f + 1]; tgt[3 * f + 1] = faces[3 * f + 2]; tgt[3 * f + 2] = faces[3 * f]; }
    });
    parallel_for(nh, 1 << 16, [&](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) out_tgt[i] = tgt[out[i]]; });
    std::atomic<int> bad{0};
    parallel_for(nh, 1 << 16, [&](int64_t lo, int64_t hi) {
        for (int64_t h = lo; h < hi; ++h) {
            const int a = faces[h], b = tgt[h];
            int t = -1;
            for (int i = first[b]; i < first[b + 1]; ++i) if (out_tgt[i] == a) { if (t >= 0) bad.store(1, std::memory_order_relaxed); t = out[i]; }
            twin[h] = t;
        }
    });
    if (nf > 100000) clk.mark("twins: match");
    if (bad.load()) return NWR_ERR_NONMANIFOLD;
    // the same directed edge twice: both copies found the same twin, which can point back at only one of them; an unmatched
    // duplicate pair shows up among the half-edges leaving their origin
    parallel_for(nh, 1 << 16, [&](int64_t lo, int64_t hi) {
        for (int64_t h = lo; h < hi; ++h) {
            if (twin[h] >= 0) { if (twin[twin[h]] != (int)h) bad.store(1, std::memory_order_relaxed); continue; }
            const int a = faces[h], b = tgt[h];
            int same = 0;
            for (int i = first[a]; i < first[a + 1]; ++i) same += (out_tgt[i] == b);
            if (same != 1) bad.store(1, std::memory_order_relaxed);
        }
    });
    if (nf > 100000) clk.mark("twins: check");
    return bad.load() ? NWR_ERR_NONMANIFOLD : NWR_OK;
}

struct HalfEdgeMesh {
    std::vector<V3> pos;
    std::vector<int> vhe;                       // an outgoing half-edge of the vertex, -1 = deleted / unreferenced
    std::vector<int> val;                       // vertex degree
    std::vector<unsigned char> boundary;        // vertex touches an unmatched edge: never modified
    std::vector<int> vert, next, prev, twin, face;   // per half-edge; vert == -1 -> deleted
    std::vector<int> fhe;                       // a half-edge of the face, -1 = deleted
    int max_valence = 16;
    int64_t n_split = 0, n_collapse = 0, n_flip = 0;
    bool corrupt = false;
    // Without relaxation no vertex moves, so a collapse or flip that was refused is refused again until an operation changes the
    // ring of one of the edge's quad vertices (its end points and the two opposite vertices).  Every operation stamps the half-edges
    // that have a vertex it touched in their quad (`mark`), and the collapse / flip passes skip the edges without a stamp from this or
    // the previous iteration: same order, same decisions as the full scans (checked: identical output), without re-running the
    // refused tests every iteration.  (With relaxation every vertex moves in every iteration: `all_dirty`.)
    std::vector<int> estamp;
    int cur_it = 1;
    int it_base = 0;                            // (a session's mesh lives through many calls: stamps of an earlier call must not look fresh)
    bool all_dirty = false;
    bool fresh_edge(int h) const { return all_dirty || estamp[h] >= cur_it - 1; }
    void mark(int v)
    {
        if (vhe[v] < 0 || boundary[v]) return;
        ring(v, [&](int o) {
            estamp[o] = cur_it;
            const int n = next[o];
            estamp[n] = cur_it;
            if (twin[o] >= 0) estamp[twin[o]] = cur_it;
            if (twin[n] >= 0) estamp[twin[n]] = cur_it;
        });
    }

    // squared length of every half-edge, kept current by the operations: no vertex moves between relaxations, so the split and
    // collapse passes -- most of which look at an edge only to find it neither long nor short -- read one sequential array instead
    // of two vertex records per edge
    std::vector<double> l2;

    int from(int h) const { return vert[prev[h]]; }
    double len2(int h) const { return norm2(pos[vert[h]] - pos[from(h)]); }
    void set_len(int h)
    {
        const double v = len2(h);
        l2[h] = v;
        if (twin[h] >= 0) l2[twin[h]] = v;
    }
    void refresh_lengths() { for (size_t h = 0; h < vert.size(); ++h) if (vert[h] >= 0) l2[h] = len2((int)h); }

    // outgoing half-edges of an interior vertex, counter-clockwise
    template <class F>
    void ring(int v, F f)
    {
        const int h0 = vhe[v];
        int h = h0, guard = 0;
        do {
            f(h);
            h = twin[prev[h]];
            if (++guard > 256) { corrupt = true; return; }
        } while (h != h0 && h >= 0);
    }

    bool connected(int a, int b)
    {
        bool hit = false;
        ring(a, [&](int h) { hit |= (vert[h] == b); });
        return hit;
    }

    int add_halfedge()
    {
        vert.push_back(-1); next.push_back(-1); prev.push_back(-1); twin.push_back(-1); face.push_back(-1); l2.push_back(0.0); estamp.push_back(cur_it);
        return (int)vert.size() - 1;
    }

    V3 face_normal(int a, int b, int c) const { return cross(pos[b] - pos[a], pos[c] - pos[a]); }

    int build(const float *vertices, int64_t nv, const int32_t *faces, int64_t nf)
    {
        pos.resize(nv);
        for (int64_t i = 0; i < nv; ++i) pos[i] = {vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]};
        vhe.assign(nv, -1); val.assign(nv, 0); boundary.assign(nv, 0);
        const int64_t nh = 3 * nf;
        vert.resize(nh); next.resize(nh); prev.resize(nh); twin.assign(nh, -1); face.resize(nh);
        fhe.resize(nf);
        for (int64_t f = 0; f < nf; ++f) {
            for (int k = 0; k < 3; ++k) {
                const int a = faces[3 * f + k], b = faces[3 * f + (k + 1) % 3];
                if (a < 0 || a >= nv || b < 0 || b >= nv || a == b) return NWR_ERR_BADARG;
                const int h = (int)(3 * f + k);
                vert[h] = b; next[h] = (int)(3 * f + (k + 1) % 3); prev[h] = (int)(3 * f + (k + 2) % 3); face[h] = (int)f;
                vhe[a] = h;
            }
            fhe[f] = (int)(3 * f);
        }
        if (match_twins(faces, nf, nv, twin.data()) != NWR_OK) return NWR_ERR_NONMANIFOLD;
        for (int64_t h = 0; h < nh; ++h) {
            val[from((int)h)] += 1;
            if (twin[h] < 0) { boundary[from((int)h)] = 1; boundary[vert[h]] = 1; }
        }
        // a vertex whose fan does not close after `val` steps is a non-manifold ("bow-tie") vertex: leave it alone
        for (int64_t v = 0; v < nv; ++v) {
            if (vhe[v] < 0 || boundary[v]) continue;
            int n = 0;
            ring((int)v, [&](int) { ++n; });
            if (n != val[v]) boundary[v] = 1;
        }
        l2.assign(nh, 0.0);
        estamp.assign(nh, 0);
        refresh_lengths();
        corrupt = false;
        return NWR_OK;
    }

    // split edge h (a->b) at its midpoint; both adjacent faces are cut in two
    void split(int h)
    {
        const int t = twin[h];
        const int hn = next[h], hp = prev[h], tn = next[t], tp = prev[t];
        const int a = vert[t], b = vert[h], c = vert[hn], d = vert[tn];
        const int f0 = face[h], f1 = face[t];
        const int m = (int)pos.size();
        pos.push_back((pos[a] + pos[b]) * 0.5);
        vhe.push_back(-1); val.push_back(4); boundary.push_back(0);
        const int h2 = add_halfedge(), e0 = add_halfedge(), e1 = add_halfedge();
        const int t2 = add_halfedge(), e2 = add_halfedge(), e3 = add_halfedge();
        const int f2 = (int)fhe.size(); fhe.push_back(h2);
        const int f3 = (int)fhe.size(); fhe.push_back(t2);
        // f0: h (a->m), e0 (m->c), hp (c->a)
        vert[h] = m; vert[e0] = c;
        next[h] = e0; prev[e0] = h; next[e0] = hp; prev[hp] = e0; next[hp] = h; prev[h] = hp;
        face[e0] = f0; fhe[f0] = h;
        // f2: h2 (m->b), hn (b->c), e1 (c->m)
        vert[h2] = b; vert[e1] = m;
        next[h2] = hn; prev[hn] = h2; next[hn] = e1; prev[e1] = hn; next[e1] = h2; prev[h2] = e1;
        face[h2] = f2; face[hn] = f2; face[e1] = f2;
        // f1: t (b->m), e2 (m->d), tp (d->b)
        vert[t] = m; vert[e2] = d;
        next[t] = e2; prev[e2] = t; next[e2] = tp; prev[tp] = e2; next[tp] = t; prev[t] = tp;
        face[e2] = f1; fhe[f1] = t;
        // f3: t2 (m->a), tn (a->d), e3 (d->m)
        vert[t2] = a; vert[e3] = m;
        next[t2] = tn; prev[tn] = t2; next[tn] = e3; prev[e3] = tn; next[e3] = t2; prev[t2] = e3;
        face[t2] = f3; face[tn] = f3; face[e3] = f3;
        twin[h] = t2; twin[t2] = h; twin[h2] = t; twin[t] = h2; twin[e0] = e1; twin[e1] = e0; twin[e2] = e3; twin[e3] = e2;
        vhe[m] = h2;
        val[c] += 1; val[d] += 1;
        set_len(h); set_len(h2); set_len(e0); set_len(e2);
        mark(a); mark(b); mark(c); mark(d); mark(m);
        ++n_split;
    }

    // collapse a = from(h) into b = vert[h] (b keeps its position); returns false if not allowed
    bool collapse(int h, double high2)
    {
        const int t = twin[h];
        const int hn = next[h], hp = prev[h], tn = next[t], tp = prev[t];
        const int a = vert[t], b = vert[h], c = vert[hn], d = vert[tn];
        if (boundary[a] || boundary[b] || boundary[c] || boundary[d]) return false;
        if (c == d || val[a] < 3 || val[b] < 3 || val[c] <= 3 || val[d] <= 3) return false;
        if (val[a] + val[b] - 4 > max_valence || val[a] + val[b] - 4 < 3) return false;
        // link condition: the only common neighbours of a and b are c and d
        int ra[64], na = 0, rh[64];
        bool overflow = false;
        ring(a, [&](int o) { if (na < 64) { rh[na] = o; ra[na++] = vert[o]; } else overflow = true; });
        if (overflow || corrupt || na != val[a]) return false;
        int common = 0;
        ring(b, [&](int o) { const int x = vert[o]; for (int i = 0; i < na; ++i) common += (ra[i] == x); });
        if (common != 2) return false;
        const V3 pb = pos[b];
        for (int i = 0; i < na; ++i) {
            const int x = ra[i];
            if (x == b) continue;
            if (norm2(pos[x] - pb) > high2) return false;                    // would create a long edge
        }
        // triangles around a that survive (a,x,y) -> (b,x,y) must keep their orientation and a sane shape
        for (int i = 0; i < na; ++i) {
            const int o = rh[i];
            const int x = vert[o], y = vert[next[o]];
            if (x == b || y == b) continue;
            const V3 n0 = face_normal(a, x, y), n1 = face_normal(b, x, y);
            const double d01 = dot(n0, n1);
            if (!(d01 > 0.0) || d01 * d01 < 0.04 * norm2(n0) * norm2(n1)) return false;   // flips or turns by > ~78 degrees
        }
        const int hn_t = twin[hn], hp_t = twin[hp], tn_t = twin[tn], tp_t = twin[tp];
        if (hn_t < 0 || hp_t < 0 || tn_t < 0 || tp_t < 0) return false;
        // every half-edge that pointed to a now points to b
        for (int i = 0; i < na; ++i) vert[twin[rh[i]]] = b;
        twin[hn_t] = hp_t; twin[hp_t] = hn_t;
        twin[tn_t] = tp_t; twin[tp_t] = tn_t;
        vhe[b] = tp_t;                       // b->d, alive
        if (vhe[c] == hp) vhe[c] = hn_t;     // c->b
        if (vhe[d] == tp) vhe[d] = tn_t;     // d->(a, now b)
        const int dead[6] = {h, hn, hp, t, tn, tp};
        for (int k = 0; k < 6; ++k) vert[dead[k]] = -1;
        fhe[face[h]] = -1; fhe[face[t]] = -1;
        val[b] = val[a] + val[b] - 4; val[c] -= 1; val[d] -= 1;
        val[a] = 0; vhe[a] = -1;
        for (int i = 0; i < na; ++i) if (alive(rh[i])) set_len(rh[i]);      // the edges that ended at a now end at b
        set_len(tn_t); set_len(hp_t);                                       // (d->a whose partner a->d died; the re-paired c-b edge)
        mark(b);
        for (int i = 0; i < na; ++i) if (ra[i] != b) mark(ra[i]);
        ++n_collapse;
        return true;
    }

    // replace edge a-b (h) by c-d if that brings the four degrees closer to six
    bool flip(int h)
    {
        const int t = twin[h];
        const int hn = next[h], hp = prev[h], tn = next[t], tp = prev[t];
        const int a = vert[t], b = vert[h], c = vert[hn], d = vert[tn];
        if (boundary[a] || boundary[b] || boundary[c] || boundary[d] || c == d) return false;
        if (val[a] <= 3 || val[b] <= 3 || val[c] + 1 > max_valence || val[d] + 1 > max_valence) return false;
        const int before = std::abs(val[a] - 6) + std::abs(val[b] - 6) + std::abs(val[c] - 6) + std::abs(val[d] - 6);
        const int after = std::abs(val[a] - 7) + std::abs(val[b] - 7) + std::abs(val[c] - 5) + std::abs(val[d] - 5);
        if (after >= before) return false;
        if (connected(c, d) || corrupt) return false;
        const V3 n0 = face_normal(a, b, c), n1 = face_normal(b, a, d);
        const double l0 = norm2(n0), l1 = norm2(n1);
        if (!(l0 > 0) || !(l1 > 0)) return false;
        if (dot(n0, n1) < 0.3 * std::sqrt(l0 * l1)) return false;           // do not flip across a crease
        const V3 m0 = face_normal(a, d, c), m1 = face_normal(d, b, c);
        const V3 navg = n0 * (1.0 / std::sqrt(l0)) + n1 * (1.0 / std::sqrt(l1));
        const double q0 = dot(m0, navg), q1 = dot(m1, navg);
        if (!(q0 > 0) || !(q1 > 0)) return false;                           // non-convex quad
        if (q0 * q0 < 0.04 * norm2(m0) * norm2(navg) || q1 * q1 < 0.04 * norm2(m1) * norm2(navg)) return false;
        // ... and of some area: with c or d (nearly) ON the line through the other and an end point -- the centre of a rhombus that an
        // earlier split put on its diagonal -- the new diagonal would run through that vertex, a sliver of the right orientation and no area;
        // it would be the next pass's longest edge, and its midpoint the vertex again (seen as zero-length edges beside frozen rims)
        if (strict_rim) {                        // (pieces of a partitioned mesh only: the serial algorithm's output stays what it was)
            const double amin = 0.01 * std::min(l0, l1);
            if (norm2(m0) < amin || norm2(m1) < amin) return false;
        }
        // f0 = (a->d = tn, d->c = h, c->a = hp), f1 = (d->b = tp, b->c = hn, c->d = t)
        const int f0 = face[h], f1 = face[t];
        vert[h] = c; vert[t] = d;
        next[tn] = h; prev[h] = tn; next[h] = hp; prev[hp] = h; next[hp] = tn; prev[tn] = hp;
        next[tp] = hn; prev[hn] = tp; next[hn] = t; prev[t] = hn; next[t] = tp; prev[tp] = t;
        face[tn] = f0; face[hn] = f1;
        fhe[f0] = h; fhe[f1] = t;
        if (vhe[a] == h) vhe[a] = tn;
        if (vhe[b] == t) vhe[b] = hn;
        val[a] -= 1; val[b] -= 1; val[c] += 1; val[d] += 1;
        set_len(h);
        mark(a); mark(b); mark(c); mark(d);
        ++n_flip;
        return true;
    }

    bool alive(int h) const { return vert[h] >= 0; }

    int64_t split_cap = INT64_MAX;
    bool strict_rim = false;
    void split_long_edges(double high2)
    {
        for (int pass = 0; pass < 8; ++pass) {
            const int64_t before = n_split;
            const size_t nh = vert.size();                 // edges created by this pass are looked at in the next one
            for (size_t h = 0; h < nh; ++h) {
                if (!(l2[h] > high2) || !std::isfinite(l2[h]) || !alive((int)h)) continue;     // (an edge of non-finite length would be split for ever)
                const int t = twin[h];
                if (t < 0 || (int)h > t) continue;         // each interior edge once
                if (boundary[from((int)h)] && boundary[vert[h]]) continue;
                if (strict_rim) {
                    // a PIECE of a larger mesh (remesh_partitioned): its rim is not a real boundary, and everything that touches a rim vertex
                    // waits for the pass in which that vertex is interior (the collapses and flips wait anyway).  (Splitting the spokes of a
                    // rim corner -- a rim vertex with two faces in the piece -- was seen to split the same edge over and over: coincident vertices.)
                    if (boundary[from((int)h)] || boundary[vert[h]] || boundary[vert[next[h]]] || boundary[vert[next[t]]]) continue;
                    // ... and a long edge between two faces of no area is left alone.  A needle the optimiser has drawn out -- an edge of
                    // six target lengths whose opposite corners lie ON it -- cannot be collapsed while it touches a frozen rim, and every
                    // split of it or of the edges "across" it puts one more vertex on the same line, at dyadic points: the midpoint of the
                    // quarter and three-quarter points IS the midpoint (seen at 8 10^5 vertices: 1 154 positions held by up to 16 vertices
                    // each, a few hundred edges of no length, and the whole mesh sent to the serial algorithm for it).
                    const int a = from((int)h), b = vert[h], c = vert[next[h]], d = vert[next[t]];
                    const double flat = 1e-8 * l2[h] * l2[h];
                    if (norm2(face_normal(a, b, c)) < flat || norm2(face_normal(b, a, d)) < flat) continue;
                }
                split((int)h);
            }
            if (n_split == before || n_split > split_cap) break;
        }
    }

    void collapse_short_edges(double low2, double high2)
    {
        const size_t nh = vert.size();
        for (size_t h = 0; h < nh; ++h) {
            if (l2[h] >= low2 || !fresh_edge((int)h) || !alive((int)h)) continue;
            const int t = twin[h];
            if (t < 0 || (int)h > t) continue;
            if (!collapse((int)h, high2)) collapse(t, high2);
            if (corrupt) return;
        }
    }

    void equalize_valences()
    {
        const size_t nh = vert.size();
        for (size_t h = 0; h < nh; ++h) {
            if (!fresh_edge((int)h) || !alive((int)h)) continue;
            const int t = twin[h];
            if (t < 0 || (int)h > t) continue;
            flip((int)h);
            if (corrupt) return;
        }
    }

    // tangential relaxation: every interior vertex moves by l x (tangential part of the vector to its ring centroid)
    void relax(double l, int n_relax)
    {
        const size_t nv = pos.size();
        std::vector<V3> nrm(nv), upd(nv);
        for (int it = 0; it < n_relax; ++it) {
            std::fill(nrm.begin(), nrm.end(), V3{0, 0, 0});
            for (size_t f = 0; f < fhe.size(); ++f) {
                if (fhe[f] < 0) continue;
                const int h = fhe[f];
                const int a = vert[prev[h]], b = vert[h], c = vert[next[h]];
                const V3 n = face_normal(a, b, c);             // length = 2 x area: area weighting
                nrm[a] = nrm[a] + n; nrm[b] = nrm[b] + n; nrm[c] = nrm[c] + n;
            }
            for (size_t v = 0; v < nv; ++v) {
                upd[v] = pos[v];
                if (vhe[v] < 0 || boundary[v] || val[v] < 3) continue;
                V3 g{0, 0, 0};
                int n = 0;
                ring((int)v, [&](int o) { g = g + pos[vert[o]]; ++n; });
                if (corrupt || n == 0) continue;
                const V3 d = g * (1.0 / n) - pos[v];
                const double nn = norm2(nrm[v]);
                V3 tang = d;
                if (nn > 0) tang = d - nrm[v] * (dot(d, nrm[v]) / nn);
                upd[v] = pos[v] + tang * l;
            }
            pos.swap(upd);
        }
        refresh_lengths();
    }
};

}  // namespace

NWR_EXPORT int nwr_abi_version(void) { return 4; }

// key 0: 1 (default) = meshes of 40 000 faces and more are remeshed in pieces, on all cores (remesh_partitioned); 0 = always the serial
// algorithm (what a caller wants whose OUTPUT must not depend on the library's version: the benchmark's mesh generator).  Returns the old value.
static std::atomic<int> g_partition{1};
NWR_EXPORT int nwr_configure(int key, int value)
{
    if (key != 0) return NWR_ERR_BADARG;
    return g_partition.exchange(value ? 1 : 0);
}

NWR_EXPORT void nwr_free(void *p) { std::free(p); }

NWR_EXPORT int nwr_halfedge_twins(const int32_t *faces, int64_t n_faces, int64_t n_vertices, int32_t *twin)
{
    if (!faces || !twin || n_faces < 1 || n_vertices < 1 || n_faces > (1ll << 29)) return NWR_ERR_BADARG;
    for (int64_t i = 0; i < 3 * n_faces; ++i) if (faces[i] < 0 || faces[i] >= n_vertices) return NWR_ERR_BADARG;
    try {
        return match_twins(faces, n_faces, n_vertices, twin);
    } catch (const std::bad_alloc &) {
        return NWR_ERR_NOMEM;
    } catch (const std::exception &) {          // (std::system_error of a thread that could not be started, ...: no exception crosses the C boundary)
        return NWR_ERR_NOMEM;
    }
}

// Face normals / areas, half-edge lengths and area-weighted vertex normals, arithmetic step for step what
// trimesh.TriMesh.update_geometry does in NumPy (float32 products and differences rounded individually, three-term sums left
// to right, vertex normals accumulated in float64 corner-major), so that both paths give bit-identical arrays.
NWR_EXPORT int nwr_mesh_geometry(const void *positions, int64_t pos_stride_bytes, int64_t n_vertices, const int32_t *faces, int64_t n_faces,
                                 void *face_normal_out, int64_t fn_stride, void *face_area_out, int64_t fa_stride, void *halfedge_length_out, int64_t hl_stride,
                                 void *vertex_normal_out, int64_t vn_stride)
{
    if (!positions || !faces || !face_normal_out || !face_area_out || !halfedge_length_out || n_vertices < 1 || n_faces < 1 || pos_stride_bytes < 12 ||
        fn_stride < 12 || fa_stride < 4 || hl_stride < 4 || (vertex_normal_out && vn_stride < 12))
        return NWR_ERR_BADARG;
    // (outputs go straight into the caller's records: a face's normal fn_stride bytes after the previous face's, and so on)
    char *fnb = (char *)face_normal_out, *fab = (char *)face_area_out, *hlb = (char *)halfedge_length_out, *vnb = (char *)vertex_normal_out;
    const bool vertex_normal = vertex_normal_out != nullptr;
    const char *base = (const char *)positions;
    auto P = [&](int v) { return (const float *)(base + (int64_t)v * pos_stride_bytes); };
    for (int64_t i = 0; i < 3 * n_faces; ++i) if (faces[i] < 0 || faces[i] >= n_vertices) return NWR_ERR_BADARG;
    try {
        std::vector<float> cr(vertex_normal ? 3 * (size_t)n_faces : 0);
        parallel_for(n_faces, 1 << 14, [&](int64_t f_lo, int64_t f_hi) {
        for (int64_t f = f_lo; f < f_hi; ++f) {
            const float *p0 = P(faces[3 * f]), *p1 = P(faces[3 * f + 1]), *p2 = P(faces[3 * f + 2]);
            const float a0 = p1[0] - p0[0], a1 = p1[1] - p0[1], a2 = p1[2] - p0[2];
            const float b0 = p2[0] - p0[0], b1 = p2[1] - p0[1], b2 = p2[2] - p0[2];
            volatile float m0 = a1 * b2, m1 = a2 * b1, m2 = a2 * b0, m3 = a0 * b2, m4 = a0 * b1, m5 = a1 * b0;   // no contraction
            const float c0 = m0 - m1, c1 = m2 - m3, c2 = m4 - m5;
            volatile float q0 = c0 * c0, q1 = c1 * c1, q2 = c2 * c2;
            volatile float s01 = q0 + q1;
            const float n = std::sqrt((float)(s01 + q2));
            float n0 = c0 / n, n1 = c1 / n, n2 = c2 / n;
            if (!std::isfinite(n0)) n0 = 0;
            if (!std::isfinite(n1)) n1 = 0;
            if (!std::isfinite(n2)) n2 = 0;
            float *fno = (float *)(fnb + f * fn_stride);
            fno[0] = n0; fno[1] = n1; fno[2] = n2;
            *(float *)(fab + f * fa_stride) = 0.5f * n;
            if (vertex_normal) { cr[3 * f] = c0; cr[3 * f + 1] = c1; cr[3 * f + 2] = c2; }
            for (int k = 0; k < 3; ++k) {
                const float *o = P(faces[3 * f + k]), *d = P(faces[3 * f + (k + 1) % 3]);
                const float e0 = d[0] - o[0], e1 = d[1] - o[1], e2 = d[2] - o[2];
                volatile float r0 = e0 * e0, r1 = e1 * e1, r2 = e2 * e2;
                volatile float t01 = r0 + r1;
                *(float *)(hlb + (3 * f + k) * hl_stride) = std::sqrt((float)(t01 + r2));
            }
        }
        });
        if (vertex_normal) {
            // (every thread owns a range of vertices and walks all corners in the serial order -- corner-major, faces ascending --
            // adding only what lands in its range: the sums are those of the serial loop, bit for bit, whatever the thread count)
            std::vector<double> vn(3 * (size_t)n_vertices, 0.0);
            parallel_for(n_vertices, 1 << 14, [&](int64_t v_lo, int64_t v_hi) {
                for (int corner = 0; corner < 3; ++corner)
                    for (int64_t f = 0; f < n_faces; ++f) {
                        const int v = faces[3 * f + corner];
                        if (v < v_lo || v >= v_hi) continue;
                        vn[3 * (size_t)v] += (double)cr[3 * f]; vn[3 * (size_t)v + 1] += (double)cr[3 * f + 1]; vn[3 * (size_t)v + 2] += (double)cr[3 * f + 2];
                    }
                for (int64_t v = v_lo; v < v_hi; ++v) {
                    const double x = vn[3 * v], y = vn[3 * v + 1], z = vn[3 * v + 2];
                    volatile double xx = x * x, yy = y * y, zz = z * z;
                    volatile double sxy = xx + yy;
                    const double l = std::sqrt((double)(sxy + zz));
                    double u0 = x / l, u1 = y / l, u2 = z / l;
                    if (!std::isfinite(u0)) u0 = 0;
                    if (!std::isfinite(u1)) u1 = 0;
                    if (!std::isfinite(u2)) u2 = 0;
                    float *vno = (float *)(vnb + v * vn_stride);
                    vno[0] = (float)u0; vno[1] = (float)u1; vno[2] = (float)u2;
                }
            });
        }
        return NWR_OK;
    } catch (const std::bad_alloc &) {
        return NWR_ERR_NOMEM;
    } catch (const std::exception &) {          // (std::system_error of a thread that could not be started, ...: no exception crosses the C boundary)
        return NWR_ERR_NOMEM;
    }
}

// Half-edge tables + 1-rings of an oriented triangle mesh, written straight into the caller's records (PYME-style structured
// arrays: trimesh.HALFEDGE_DTYPE / VERTEX_DTYPE): what trimesh._build_halfedges + TriMesh._build_rings define in NumPy, same
// conventions -- half-edge 3f+k runs faces[f][k] -> faces[f][(k+1)%3]; a vertex starts its ring at its lowest-numbered outgoing
// half-edge, a boundary vertex at its (highest-numbered) outgoing half-edge without a twin; the ring is walked counter-clockwise
// (cur -> twin[prev[cur]]) for at most `neighbor_size` steps.
NWR_EXPORT int nwr_build_topology(const int32_t *faces, int64_t n_faces, int64_t n_vertex_slots,
                                  void *halfedges, int64_t he_stride, int64_t off_vertex, int64_t off_face, int64_t off_twin, int64_t off_next, int64_t off_prev,
                                  int32_t *origin, void *vertices, int64_t v_stride, int64_t off_halfedge, int64_t off_valence, int64_t off_neighbors,
                                  int32_t neighbor_size)
{
    if (!faces || !halfedges || !origin || !vertices || n_faces < 1 || n_vertex_slots < 1 || n_faces > (1ll << 29) || neighbor_size < 1 || neighbor_size > 64)
        return NWR_ERR_BADARG;
    const int64_t nh = 3 * n_faces;
    for (int64_t i = 0; i < nh; ++i) if (faces[i] < 0 || faces[i] >= n_vertex_slots) return NWR_ERR_BADARG;
    try {
        std::vector<int> twin(nh);
        const int rc = match_twins(faces, n_faces, n_vertex_slots, twin.data());
        if (rc != NWR_OK) return rc;
        char *hb = (char *)halfedges, *vb = (char *)vertices;
        auto H = [&](int64_t h, int64_t off) -> int32_t & { return *(int32_t *)(hb + h * he_stride + off); };
        parallel_for(n_faces, 1 << 14, [&](int64_t f_lo, int64_t f_hi) {
            for (int64_t f = f_lo; f < f_hi; ++f)
                for (int k = 0; k < 3; ++k) {
                    const int64_t h = 3 * f + k;
                    H(h, off_vertex) = faces[3 * f + (k + 1) % 3];
                    H(h, off_face) = (int32_t)f;
                    H(h, off_twin) = twin[h];
                    H(h, off_next) = (int32_t)(3 * f + (k + 1) % 3);
                    H(h, off_prev) = (int32_t)(3 * f + (k + 2) % 3);
                    origin[h] = faces[h];
                }
        });
        std::vector<int> start(n_vertex_slots, -1);
        for (int64_t h = nh - 1; h >= 0; --h) start[faces[h]] = (int)h;                  // lowest index wins
        for (int64_t h = 0; h < nh; ++h) if (twin[h] < 0) start[faces[h]] = (int)h;       // boundary: the last one wins
        parallel_for(n_vertex_slots, 1 << 13, [&](int64_t v_lo, int64_t v_hi) {
            for (int64_t v = v_lo; v < v_hi; ++v) {
                int32_t *nb = (int32_t *)(vb + v * v_stride + off_neighbors);
                for (int s = 0; s < neighbor_size; ++s) nb[s] = -1;
                const int s0 = start[v];
                *(int32_t *)(vb + v * v_stride + off_halfedge) = s0;
                int n = 0, cur = s0;
                while (cur >= 0 && n < neighbor_size) {
                    nb[n++] = cur;
                    const int k = cur % 3, p = cur - k + (k + 2) % 3;
                    const int nx = twin[p];
                    if (nx < 0 || nx == s0) break;
                    cur = nx;
                }
                *(int32_t *)(vb + v * v_stride + off_valence) = n;
            }
        });
        return NWR_OK;
    } catch (const std::bad_alloc &) {
        return NWR_ERR_NOMEM;
    } catch (const std::exception &) {          // (std::system_error of a thread that could not be started, ...: no exception crosses the C boundary)
        return NWR_ERR_NOMEM;
    }
}

// Per-slot tables of the 1-rings (slot s of vertex v = half-edge nb[v][s], -1 padded): the vertex it points to (the table the
// optimiser caches, mesh_conj_grad.py:50-54), the vertex its NEXT half-edge points to and the area of its face (what
// c_curvature_grad reads through the half-edge records, membrane_mesh_utils.c:1099-1104).  Any output may be NULL.
NWR_EXPORT int nwr_ring_tables(const void *halfedges, int64_t he_stride, int64_t off_vertex, int64_t off_face, int64_t off_next, int64_t n_halfedges,
                               const void *vertices, int64_t v_stride, int64_t off_neighbors, int32_t neighbor_size, int64_t n_vertex_slots,
                               const void *face_area, int64_t fa_stride, int32_t *ring_vertex, int32_t *ring_next_vertex, float *ring_area)
{
    if (!halfedges || !vertices || n_halfedges < 3 || n_vertex_slots < 1 || neighbor_size < 1 || (ring_area && !face_area)) return NWR_ERR_BADARG;
    const char *hb = (const char *)halfedges, *vb = (const char *)vertices, *fb = (const char *)face_area;
    auto H = [&](int64_t h, int64_t off) { return *(const int32_t *)(hb + h * he_stride + off); };
    parallel_for(n_vertex_slots, 1 << 13, [&](int64_t v_lo, int64_t v_hi) {
        for (int64_t v = v_lo; v < v_hi; ++v) {
            const int32_t *nb = (const int32_t *)(vb + v * v_stride + off_neighbors);
            for (int s = 0; s < neighbor_size; ++s) {
                const int h = nb[s];
                const int64_t o = v * neighbor_size + s;
                if (h < 0 || h >= n_halfedges) {
                    if (ring_vertex) ring_vertex[o] = -1;
                    if (ring_next_vertex) ring_next_vertex[o] = -1;
                    if (ring_area) ring_area[o] = 0.0f;
                    continue;
                }
                if (ring_vertex) ring_vertex[o] = H(h, off_vertex);
                if (ring_next_vertex) ring_next_vertex[o] = H(H(h, off_next), off_vertex);
                if (ring_area) ring_area[o] = *(const float *)(fb + (int64_t)H(h, off_face) * fa_stride);
            }
        }
    });
    return NWR_OK;
}

// The iterations of the algorithm on a built mesh (split / collapse / flip, optional relaxation), then flips while a vertex is above the
// degree limit.  L > 0.
static int run_passes(HalfEdgeMesh &m, int n_iterations, double L, float relax_lambda, int n_relax, bool verbose, bool first_all)
{
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const double high = 4.0 / 3.0 * L, low = 4.0 / 5.0 * L;
    // How many splits the target length can explain: a face with longest edges l1 >= l2 ends up as about (l1 / high + 1)(l2 / high + 1)
    // pieces -- quadratic for a large isotropic face, linear in its length for a sliver (a star-projected tube has 39 x 1 nm faces and
    // needs half a million splits at L = 3 nm, which a bound from the MEDIAN edge took for a runaway).  Four times the sum over the
    // faces per iteration and something feeds on itself; more than 2^26 in all (a vertex flung far away: edges of 10^5 targets)
    // and the input is refused before any work is done.
    double pieces = 0.0;
    for (size_t f = 0; f < m.fhe.size(); ++f) {
        const int h = m.fhe[f];
        if (h < 0) continue;
        double e[3] = {std::sqrt(m.l2[h]), std::sqrt(m.l2[m.next[h]]), std::sqrt(m.l2[m.prev[h]])};
        std::sort(e, e + 3);
        pieces += (e[2] / high + 1.0) * (e[1] / high + 1.0);
    }
    if (!(pieces < 67108864.0)) return NWR_ERR_RUNAWAY;
    const int64_t split0 = m.n_split;
    m.split_cap = split0 + (int64_t)(4.0 * pieces * std::max(n_iterations, 1)) + 100000;
    const bool relaxing = n_relax > 0;
    for (int it = 0; it < n_iterations; ++it) {
        m.cur_it = m.it_base + it + 1;
        // (first_all: the first iteration of a call looks at every edge -- the mesh is new, or, a session, every vertex has moved since the
        // last call; not for a seeded run, whose caller has stamped the edges that are to be looked at)
        m.all_dirty = relaxing || (first_all && it == 0);
        const int64_t ops_before = m.n_split + m.n_collapse + m.n_flip;
        const auto t0 = now();
        m.split_long_edges(high * high);
        const auto t1 = now();
        m.collapse_short_edges(low * low, high * high);
        const auto t2 = now();
        m.equalize_valences();
        if (verbose) std::fprintf(stderr, "[nw_remesh] iteration %d: split %.1f ms, collapse %.1f ms, flip %.1f ms (%lld / %lld / %lld operations so far)\n", it, ms(t0, t1), ms(t1, t2),
                                  ms(t2, now()), (long long)m.n_split, (long long)m.n_collapse, (long long)m.n_flip);
        if (relaxing) m.relax(relax_lambda, n_relax);
        if (m.corrupt) return NWR_ERR_NONMANIFOLD;
        if (m.n_split > m.split_cap) return NWR_ERR_RUNAWAY;
        // a pass that changed nothing would be repeated unchanged by every later iteration (no relaxation to move vertices)
        if (!relaxing && m.n_split + m.n_collapse + m.n_flip == ops_before) break;
    }
    // splits can pile degree onto a vertex faster than one flip pass removes it: keep flipping while it helps
    m.all_dirty = true;
    for (int extra = 0; extra < 6 && n_iterations > 0; ++extra) {
        int mv = 0;
        for (size_t v = 0; v < m.pos.size(); ++v) if (m.vhe[v] >= 0) mv = std::max(mv, m.val[v]);
        if (mv <= m.max_valence) break;
        const int64_t before = m.n_flip;
        m.equalize_valences();
        if (m.corrupt) return NWR_ERR_NONMANIFOLD;
        if (m.n_flip == before) break;
    }
    m.it_base += n_iterations + 2;
    return NWR_OK;
}

// compact: faces that are alive, vertices they reference (ids compacted, relative order kept).  out_orig[v'] = the input vertex (< n_vertices_in)
// an output vertex was, or -1; slot_of_out[v'] = its slot in the mesh (a session scatters the next call's positions through it).
static void extract_mesh(HalfEdgeMesh &m, int64_t n_vertices_in, std::vector<float> &ov, std::vector<int32_t> &of, std::vector<int> *out_orig, nwr_stats *stats,
                         std::vector<int> *slot_of_out)
{
    std::vector<int> remap(m.pos.size(), -1);
    int64_t nf = 0;
    for (size_t f = 0; f < m.fhe.size(); ++f) nf += m.fhe[f] >= 0;
    of.assign(3 * (size_t)nf, 0);
    std::vector<unsigned char> used(m.pos.size(), 0);
    int64_t k = 0;
    for (size_t f = 0; f < m.fhe.size(); ++f) {
        const int h = m.fhe[f];
        if (h < 0) continue;
        const int a = m.vert[m.prev[h]], b = m.vert[h], c = m.vert[m.next[h]];
        of[3 * k] = a; of[3 * k + 1] = b; of[3 * k + 2] = c;
        used[a] = used[b] = used[c] = 1;
        ++k;
    }
    int64_t nv = 0;
    for (size_t v = 0; v < m.pos.size(); ++v) if (used[v]) remap[v] = (int)nv++;
    ov.assign(3 * (size_t)nv, 0.0f);
    if (out_orig) out_orig->assign((size_t)nv, -1);
    if (slot_of_out) slot_of_out->assign((size_t)nv, -1);
    for (size_t v = 0; v < m.pos.size(); ++v) {
        if (remap[v] < 0) continue;
        ov[3 * remap[v]] = (float)m.pos[v].x; ov[3 * remap[v] + 1] = (float)m.pos[v].y; ov[3 * remap[v] + 2] = (float)m.pos[v].z;
        if (out_orig && (int64_t)v < n_vertices_in) (*out_orig)[remap[v]] = (int)v;
        if (slot_of_out) (*slot_of_out)[remap[v]] = (int)v;
    }
    for (int64_t i = 0; i < 3 * nf; ++i) of[i] = remap[of[i]];
    if (stats) {
        stats->n_split = m.n_split; stats->n_collapse = m.n_collapse; stats->n_flip = m.n_flip;
        double sl = 0; int64_t n = 0; int mv = 0;
        for (size_t h = 0; h < m.vert.size(); ++h) if (m.alive((int)h)) { sl += std::sqrt(m.len2((int)h)); ++n; }
        for (size_t v = 0; v < m.pos.size(); ++v) if (used[v]) mv = std::max(mv, m.val[v]);
        stats->mean_edge_length = n ? sl / n : 0.0; stats->max_valence = mv; stats->reserved = 0;
    }
}

// The serial remesher on one (sub-)mesh.  seed: per-vertex flags -- only edges around flagged vertices are looked at by the collapse and
// flip passes until an operation freshens more (NULL: every edge, the plain algorithm).  out_orig[v'] = the input vertex an output vertex
// was, or -1 for a vertex a split created.  target_edge_length must be > 0 when the caller remeshes pieces of one mesh (the default,
// the mean edge of the input, would differ from piece to piece).
static int remesh_core(const float *vertices, int64_t n_vertices, const int32_t *faces, int64_t n_faces,
                       int n_iterations, float target_edge_length, float relax_lambda, int n_relax, int max_valence, const unsigned char *seed,
                       std::vector<float> &ov, std::vector<int32_t> &of, std::vector<int> *out_orig, nwr_stats *stats, double *mean_edge_in)
{
    {
        const bool verbose = std::getenv("NWR_VERBOSE") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        const auto t_start = now();
        HalfEdgeMesh m;
        m.max_valence = max_valence > 0 ? std::min(max_valence, 60) : 16;
        for (int64_t i = 0; i < 3 * n_vertices; ++i) if (!std::isfinite(vertices[i])) return NWR_ERR_BADARG;      // a non-finite vertex: its edges would be split for ever
        int rc = m.build(vertices, n_vertices, faces, n_faces);
        if (rc != NWR_OK) return rc;
        m.strict_rim = out_orig != nullptr;          // (a piece of a partitioned mesh)
        if (seed) {
            std::fill(m.estamp.begin(), m.estamp.end(), -1);
            m.cur_it = 1;
            for (int64_t v = 0; v < n_vertices; ++v) if (seed[v]) m.mark((int)v);
        }
        if (verbose) std::fprintf(stderr, "[nw_remesh] build %.1f ms\n", ms(t_start, now()));
        double L = target_edge_length, L_in = 1.0;
        {
            double sl = 0; int64_t n = 0;
            for (size_t h = 0; h < m.vert.size(); ++h) { sl += std::sqrt(m.len2((int)h)); ++n; }
            L_in = n ? sl / n : 1.0;                 // mean edge length of the input (PYME's default target)
        }
        if (mean_edge_in) *mean_edge_in = L_in;
        if (!(L > 0)) L = L_in;
        rc = run_passes(m, n_iterations, L, relax_lambda, n_relax, verbose, seed == nullptr);
        if (rc != NWR_OK) return rc;
        extract_mesh(m, n_vertices, ov, of, out_orig, stats, nullptr);
        if (verbose) std::fprintf(stderr, "[nw_remesh] total %.1f ms\n", ms(t_start, now()));
        return NWR_OK;
    }
}

// ---- the remesher over a partitioned mesh ------------------------------------------------------------------------------------
// The algorithm above is a chain of local operations in half-edge order: serial, 0.15 s for 2 10^5 vertices, most of a fit's
// block boundary.  The same operations run on pieces of the mesh at once:
//   1. the faces are put in Morton order of their centroids and cut into NWR_REGIONS runs of equal length (the cut depends on the mesh
//      alone, not on the number of threads); every run is a sub-mesh whose rim -- the edges it shares with other runs -- is an open
//      boundary, which the remesher never touches (vertices on unmatched edges are frozen): the runs are remeshed independently, on
//      as many threads as there are, and spliced back in run order;
//   2. what the frozen rims kept from being done is done in a second pass over the SEAM ZONE alone: the faces within NWR_SEAM_RINGS
//      rings of a rim vertex, again a sub-mesh with a frozen rim of its own (which lies inside the runs, where pass 1 was free).
// Every edge has been free in one of the two passes.  The result is a valid output of the same algorithm -- not the serial one's
// (the order of the operations differs), and independent of the thread count.  Without relaxation only (n_relax = 0: what the fit's
// block boundary asks for): relaxation moves every vertex in every iteration, rims included.
#define NWR_REGIONS 16
#define NWR_SEAM_RINGS 3
#define NWR_PARALLEL_MIN_FACES 40000

struct Piece {
    std::vector<int> faces_in;           // ids of the mesh's faces that make up the piece
    std::vector<int> l2g;                // local vertex -> vertex of the mesh
    std::vector<float> ov;               // the remeshed piece
    std::vector<int32_t> of;
    std::vector<int> orig;               // output vertex -> local input vertex, -1 = new
    nwr_stats st{};
    int rc = NWR_OK;
};

static void piece_run(Piece &p, const std::vector<float> &V, const std::vector<int32_t> &F, std::vector<int> &g2l_scratch, int n_iterations, float L, int max_valence,
                      const unsigned char *seed_global)
{
    // local numbering in first-seen order
    std::vector<float> lv;
    std::vector<int32_t> lf(3 * p.faces_in.size());
    std::vector<unsigned char> lseed;
    p.l2g.clear();
    for (size_t i = 0; i < p.faces_in.size(); ++i)
        for (int k = 0; k < 3; ++k) {
            const int g = F[3 * (size_t)p.faces_in[i] + k];
            int l = g2l_scratch[g];
            if (l < 0) {
                l = (int)p.l2g.size();
                g2l_scratch[g] = l;
                p.l2g.push_back(g);
                lv.push_back(V[3 * (size_t)g]); lv.push_back(V[3 * (size_t)g + 1]); lv.push_back(V[3 * (size_t)g + 2]);
                if (seed_global) lseed.push_back(seed_global[g]);
            }
            lf[3 * i + k] = l;
        }
    for (int g : p.l2g) g2l_scratch[g] = -1;          // leave the scratch map clean for the next piece of this thread
    if (p.l2g.size() < 3 || p.faces_in.empty()) { p.rc = NWR_OK; p.ov.clear(); p.of.clear(); p.orig.clear(); return; }
    p.rc = remesh_core(lv.data(), (int64_t)p.l2g.size(), lf.data(), (int64_t)p.faces_in.size(), n_iterations, L, 0.0f, 0, max_valence,
                       seed_global ? lseed.data() : nullptr, p.ov, p.of, &p.orig, &p.st, nullptr);
}

// every piece on a thread of the pool.  The global -> local vertex map is a scratch array per thread that lives across calls (all -1 between
// pieces: piece_run leaves it clean); a piece that throws (out of memory) reports it through its rc, and with it the pass.
static void run_pieces(std::vector<Piece> &pieces, const std::vector<float> &V, const std::vector<int32_t> &F, int n_iterations, float L, int max_valence,
                       const unsigned char *seed_global)
{
    const size_t nv = V.size() / 3;
    Pool::get().run((int)pieces.size(), n_threads(), [&](int i) {
        static thread_local std::vector<int> g2l;
        try {
            if (g2l.size() < nv) g2l.assign(nv, -1);
            piece_run(pieces[i], V, F, g2l, n_iterations, L, max_valence, seed_global);
        } catch (...) {
            pieces[i].rc = NWR_ERR_NOMEM;
            std::vector<int>().swap(g2l);                     // (may have been left half-written)
        }
    });
}

// the ids in [0, n) for which keep(i) holds, ascending: chunks on all threads, their lists joined in chunk order
template <class Pred>
static std::vector<int> collect_ids(int64_t n, Pred keep)
{
    // (keep() is called once per id: it may mark what it keeps.  Every chunk writes its ids at the start of its own stretch of one array;
    // the stretches are then moved together)
    const int64_t CH = 1 << 15;
    const int64_t nch = (n + CH - 1) / CH;
    std::vector<int> out((size_t)n);
    std::vector<int64_t> cnt((size_t)nch, 0);
    parallel_for(nch, 1, [&](int64_t c_lo, int64_t c_hi) {
        for (int64_t c = c_lo; c < c_hi; ++c) {
            int *dst = out.data() + c * CH;
            int64_t k = 0;
            for (int64_t i = c * CH; i < std::min(n, (c + 1) * CH); ++i) if (keep(i)) dst[k++] = (int)i;
            cnt[c] = k;
        }
    });
    int64_t tot = 0;
    for (int64_t c = 0; c < nch; ++c) {
        if (tot != c * CH && cnt[c] > 0) std::memmove(out.data() + tot, out.data() + c * CH, sizeof(int) * (size_t)cnt[c]);
        tot += cnt[c];
    }
    out.resize((size_t)tot);
    return out;
}

// replace the faces of the pieces (disjoint sets) by their remeshed versions: original vertices keep their ids, new ones are appended in
// piece order; faces: the untouched ones in their order, then the pieces' in piece order.  (Where everything goes follows from the pieces'
// sizes, so the pieces are written at once.)
static void splice(std::vector<float> &V, std::vector<int32_t> &F, std::vector<Piece> &pieces)
{
    const size_t nf = F.size() / 3;
    StageClock clk;
    std::vector<unsigned char> gone(nf, 0);
    const int np = (int)pieces.size();
    Pool::get().run(np, n_threads(), [&](int i) { for (int f : pieces[i].faces_in) gone[f] = 1; });
    size_t n_in_pieces = 0;
    for (auto &p : pieces) n_in_pieces += p.faces_in.size();
    std::vector<int> kept;
    if (n_in_pieces < nf) kept = collect_ids((int64_t)nf, [&](int64_t f) { return !gone[f]; });
    clk.mark("splice: kept faces");
    std::vector<size_t> face_at(np + 1), vert_at(np + 1);
    face_at[0] = kept.size();
    vert_at[0] = V.size() / 3;
    for (int i = 0; i < np; ++i) {
        size_t fresh = 0;
        for (int o : pieces[i].orig) fresh += o < 0;
        face_at[i + 1] = face_at[i] + pieces[i].of.size() / 3;
        vert_at[i + 1] = vert_at[i] + fresh;
    }
    clk.mark("splice: offsets");
    std::vector<int32_t> F2(3 * face_at[np]);
    V.resize(3 * vert_at[np]);
    clk.mark("splice: alloc");
    parallel_for((int64_t)kept.size(), 1 << 15, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) { const size_t f = (size_t)kept[i]; F2[3 * i] = F[3 * f]; F2[3 * i + 1] = F[3 * f + 1]; F2[3 * i + 2] = F[3 * f + 2]; }
    });
    clk.mark("splice: copy kept");
    Pool::get().run(np, n_threads(), [&](int i) {
        Piece &p = pieces[i];
        std::vector<int> o2g(p.orig.size());
        size_t at = vert_at[i];
        for (size_t v = 0; v < p.orig.size(); ++v) {
            if (p.orig[v] >= 0) o2g[v] = p.l2g[p.orig[v]];
            else {
                o2g[v] = (int)at;
                V[3 * at] = p.ov[3 * v]; V[3 * at + 1] = p.ov[3 * v + 1]; V[3 * at + 2] = p.ov[3 * v + 2];
                ++at;
            }
        }
        int32_t *dst = F2.data() + 3 * face_at[i];
        for (size_t k = 0; k < p.of.size(); ++k) dst[k] = o2g[p.of[k]];
    });
    clk.mark("splice: pieces");
    F.swap(F2);
}

static unsigned spread10(unsigned v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// What frozen rims kept from being done: the zone of faces within NWR_SEAM_RINGS rings of the `seeds` (the rim vertices of the pass before)
// is remeshed as a sub-mesh of its own -- its outer edge is an open boundary, i.e. frozen, and lies where the pass before was free; only
// edges around the seeds are looked at until an operation freshens more.
// face_run (pass 2 only): the run of pass 1 every face belongs to.  The zone of a 16-run partition is a network of strips, a third of
// the mesh: it is cut where three runs meet -- every strip (the faces that grew from the rim between runs a and b) is a piece of its own,
// the strips are remeshed at once, and what THEIR frozen ends kept from being done (patches around the junctions) is a last, small, seam
// pass.  (Morton runs of the zone do not do: a run of a strip network is a scatter of fragments with as much rim as the zone had.)
static int seam_pass(std::vector<float> &V, std::vector<int32_t> &F, const std::vector<unsigned char> &seeds, const std::vector<int> *face_run,
                     int n_iterations, float L, int max_valence, nwr_stats &tot)
{
    const bool verbose = std::getenv("NWR_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_in = now();
    const size_t nv = V.size() / 3, nf = F.size() / 3;
    StageClock clk;
    // label of a seed = the two lowest runs among its faces (the strip it belongs to); without face_run everything is strip 0
    std::vector<int> vlab(nv, -1);
    if (face_run) {
        std::vector<int> p1(nv, INT32_MAX), p2(nv, INT32_MAX);
        // (only the faces at a seed matter: found on all threads, looked at in face order)
        const std::vector<int> at_seed = collect_ids((int64_t)nf, [&](int64_t f) { return seeds[F[3 * f]] || seeds[F[3 * f + 1]] || seeds[F[3 * f + 2]]; });
        for (int f : at_seed)
            for (int k = 0; k < 3; ++k) {
                const int v = F[3 * (size_t)f + k];
                if (!seeds[v]) continue;
                const int r = (*face_run)[f];
                if (r == p1[v] || r == p2[v]) continue;
                if (r < p1[v]) { p2[v] = p1[v]; p1[v] = r; }
                else if (r < p2[v]) p2[v] = r;
            }
        for (size_t v = 0; v < nv; ++v) if (seeds[v]) vlab[v] = p1[v] * NWR_REGIONS + (p2[v] == INT32_MAX ? p1[v] : p2[v]);
    } else {
        for (size_t v = 0; v < nv; ++v) if (seeds[v]) vlab[v] = 0;
    }
    clk.mark("seam: seed labels");
    // the zone grows ring by ring; a face takes the lowest label among its labelled vertices, its other vertices take the face's
    // (the scan for a ring's faces reads the vertex labels of the ring before and writes face labels only: on all threads; the vertices
    // then take their labels in face order)
    std::vector<int> flab(nf, -1);
    for (int ring = 0; ring < NWR_SEAM_RINGS; ++ring) {
        const std::vector<int> fresh = collect_ids((int64_t)nf, [&](int64_t f) {
            if (flab[f] >= 0) return false;
            int lab = INT32_MAX;
            for (int k = 0; k < 3; ++k) { const int l = vlab[F[3 * f + k]]; if (l >= 0 && l < lab) lab = l; }
            if (lab == INT32_MAX) return false;
            flab[f] = lab;
            return true;
        });
        for (int f : fresh)
            for (int k = 0; k < 3; ++k) { int &l = vlab[F[3 * (size_t)f + k]]; if (l < 0) l = flab[f]; }
    }
    clk.mark("seam: rings");
    const int nlab = face_run ? NWR_REGIONS * NWR_REGIONS : 1;
    std::vector<int> cnt(nlab, 0);
    size_t nz = 0;
    for (size_t f = 0; f < nf; ++f) if (flab[f] >= 0) { cnt[flab[f]] += 1; ++nz; }
    if (nz == 0) return NWR_OK;
    std::vector<int> slot(nlab, -1);
    int npieces = 0;
    for (int l = 0; l < nlab; ++l) if (cnt[l] > 0) slot[l] = npieces++;
    std::vector<Piece> pieces(npieces);
    for (int l = 0; l < nlab; ++l) if (cnt[l] > 0) pieces[slot[l]].faces_in.reserve(cnt[l]);
    for (size_t f = 0; f < nf; ++f) if (flab[f] >= 0) pieces[slot[flab[f]]].faces_in.push_back((int)f);
    clk.mark("seam: strips' faces");
    // the strips' frozen ends: vertices that faces of two strips share
    std::vector<unsigned char> ends(nv, 0);
    size_t nends = 0;
    if (npieces > 1) {
        std::vector<int> vstrip(nv, -1);
        for (size_t f = 0; f < nf; ++f) {
            if (flab[f] < 0) continue;
            for (int k = 0; k < 3; ++k) {
                const int v = F[3 * f + k];
                if (vstrip[v] < 0) vstrip[v] = flab[f];
                else if (vstrip[v] != flab[f] && !ends[v]) { ends[v] = 1; ++nends; }
            }
        }
    }
    if (verbose) {
        size_t ns = 0;
        for (unsigned char c : seeds) ns += c;
        std::fprintf(stderr, "[nw_remesh] seam pass: %zu seeds, zone of %zu faces (of %zu) in %d strips, %zu vertices at their ends\n", ns, nz, nf, npieces, nends);
    }
    clk.mark("seam: ends");
    const auto t_zone = now();
    run_pieces(pieces, V, F, n_iterations, L, max_valence, seeds.data());
    for (auto &p : pieces) {
        if (p.rc != NWR_OK) return p.rc;
        tot.n_split += p.st.n_split; tot.n_collapse += p.st.n_collapse; tot.n_flip += p.st.n_flip;
    }
    const auto t_run = now();
    splice(V, F, pieces);
    if (verbose) std::fprintf(stderr, "[nw_remesh] seam pass: zone %.1f ms, strips %.1f ms, splice %.1f ms\n", ms(t_in, t_zone), ms(t_zone, t_run), ms(t_run, now()));
    if (nends == 0) return NWR_OK;
    ends.resize(V.size() / 3, 0);                             // (frozen: they kept their ids; what the strips created is no seed)
    return seam_pass(V, F, ends, nullptr, n_iterations, L, max_valence, tot);
}

static int remesh_partitioned(const float *vertices, int64_t n_vertices, const int32_t *faces, int64_t n_faces, int n_iterations, float L, int max_valence,
                              std::vector<float> &ov, std::vector<int32_t> &of, nwr_stats *stats, double *min_edge2, bool *input_bad)
{
    const bool verbose = std::getenv("NWR_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    StageClock clk;
    std::vector<float> V(vertices, vertices + 3 * n_vertices);
    std::vector<int32_t> F(faces, faces + 3 * n_faces);
    clk.mark("order: copies");
    // Morton order of the faces' centroids over the mesh's bounding cube
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t v = 0; v < n_vertices; ++v) for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], V[3 * v + k]); hi[k] = std::max(hi[k], V[3 * v + k]); }
    const double ext = std::max({(double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2], 1e-30});
    std::vector<unsigned> key((size_t)n_faces);
    parallel_for(n_faces, 1 << 14, [&](int64_t f_lo, int64_t f_hi) {
        for (int64_t f = f_lo; f < f_hi; ++f) {
            unsigned q[3];
            for (int k = 0; k < 3; ++k) {
                const double c = ((double)V[3 * (size_t)F[3 * f] + k] + V[3 * (size_t)F[3 * f + 1] + k] + V[3 * (size_t)F[3 * f + 2] + k]) / 3.0;
                q[k] = (unsigned)std::min(1023.0, std::max(0.0, (c - lo[k]) / ext * 1024.0));
            }
            key[f] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
        }
    });
    clk.mark("order: keys");
    // Run r = the faces of rank [n r / R, n (r + 1) / R) in the order of (key, face id) -- what a stable sort by key would give -- with the
    // faces of a run in the mesh's own order.  No sort is needed for that: a histogram over the keys' upper 16 bits places every bin in the
    // ranking; a bin that no run boundary cuts belongs to one run as a whole, and only the few bins a boundary cuts are ranked face by face.
    std::vector<Piece> pieces(NWR_REGIONS);
    {
        const int SH = 14, NB = 1 << 16;
        std::vector<int64_t> start(NB + 1, 0);
        for (int64_t f = 0; f < n_faces; ++f) start[(key[f] >> SH) + 1] += 1;
        for (int b = 0; b < NB; ++b) start[b + 1] += start[b];
        int64_t bound[NWR_REGIONS + 1];
        for (int r = 0; r <= NWR_REGIONS; ++r) bound[r] = n_faces * r / NWR_REGIONS;
        auto run_of_rank = [&](int64_t rank) { int r = 0; while (r + 1 < NWR_REGIONS && bound[r + 1] <= rank) ++r; return r; };
        std::vector<int> bin_run(NB, 0);                       // run of a whole bin, or -1 - (index of the cut bin)
        std::vector<int> cut_bins;
        for (int b = 0; b < NB; ++b) {
            if (start[b + 1] == start[b]) continue;
            const int r0 = run_of_rank(start[b]), r1 = run_of_rank(start[b + 1] - 1);
            if (r0 == r1) bin_run[b] = r0;
            else { bin_run[b] = -1 - (int)cut_bins.size(); cut_bins.push_back(b); }
        }
        std::vector<std::vector<std::pair<unsigned, int>>> cut(cut_bins.size());
        std::vector<int> face_run((size_t)n_faces);
        for (int64_t f = 0; f < n_faces; ++f) {
            const int br = bin_run[key[f] >> SH];
            face_run[f] = br;
            if (br < 0) cut[-1 - br].push_back({key[f], (int)f});
        }
        for (size_t c = 0; c < cut.size(); ++c) {
            std::sort(cut[c].begin(), cut[c].end());
            for (size_t i = 0; i < cut[c].size(); ++i) face_run[cut[c][i].second] = run_of_rank(start[cut_bins[c]] + (int64_t)i);
        }
        clk.mark("order: ranks");
        for (int r = 0; r < NWR_REGIONS; ++r) pieces[r].faces_in.reserve((size_t)(bound[r + 1] - bound[r]));
        for (int64_t f = 0; f < n_faces; ++f) pieces[face_run[f]].faces_in.push_back((int)f);
    }
    clk.mark("order: runs");
    // vertices on a rim: used by faces of more than one run (every run marks its vertices: the first to come claims a vertex, whoever finds
    // it claimed by another run flags it -- which run claimed it does not matter)
    std::vector<int> vrun((size_t)n_vertices, -1);
    std::vector<unsigned char> rim((size_t)n_vertices, 0);
    Pool::get().run(NWR_REGIONS, n_threads(), [&](int r) {
        for (int f : pieces[r].faces_in)
            for (int k = 0; k < 3; ++k) {
                const int v = F[3 * (size_t)f + k];
                int seen = __atomic_load_n(&vrun[v], __ATOMIC_RELAXED);
                if (seen < 0) {
                    int expect = -1;
                    seen = __atomic_compare_exchange_n(&vrun[v], &expect, r, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED) ? r : expect;
                }
                if (seen != r) __atomic_store_n(&rim[v], (unsigned char)1, __ATOMIC_RELAXED);
            }
    });
    clk.mark("order: rims");
    // An input that is no oriented 2-manifold is refused, as the serial path refuses it when it matches the twins of the whole mesh: a mesh
    // is one exactly when no directed edge a -> b occurs twice (three faces on an edge, or two with the same sense, repeat a direction).
    // Inside a run its own twin matching finds a repeat (pass 1); a repeat ACROSS runs joins two vertices that both lie on a rim -- a few per
    // cent of the edges, which are collected, sorted and compared here instead of matching 10^6 half-edges a second time.
    {
        const std::vector<int> rr = collect_ids(3 * n_faces, [&](int64_t h) { return rim[F[h]] && rim[F[h - h % 3 + (h + 1) % 3]]; });
        std::vector<uint64_t> ek(rr.size());
        for (size_t i = 0; i < rr.size(); ++i) { const int64_t h = rr[i]; ek[i] = ((uint64_t)(uint32_t)F[h] << 32) | (uint32_t)F[h - h % 3 + (h + 1) % 3]; }
        std::sort(ek.begin(), ek.end());
        if (std::adjacent_find(ek.begin(), ek.end()) != ek.end()) { if (input_bad) *input_bad = true; return NWR_ERR_NONMANIFOLD; }
    }
    clk.mark("order: rim edges");
    const auto t1 = now();
    // pass 1: the runs, independently
    const int T = std::min(n_threads(), NWR_REGIONS);
    run_pieces(pieces, V, F, n_iterations, L, max_valence, nullptr);
    nwr_stats tot{};
    for (auto &p : pieces) {
        if (p.rc != NWR_OK) return p.rc;
        tot.n_split += p.st.n_split; tot.n_collapse += p.st.n_collapse; tot.n_flip += p.st.n_flip;
    }
    clk.mark("pass 1: pieces");
    splice(V, F, pieces);
    clk.mark("pass 1: splice");
    const auto t2 = now();
    // pass 2: the seam zone = faces within NWR_SEAM_RINGS rings of a rim vertex (rim vertices kept their ids: they were frozen)
    if (!std::getenv("NWR_NO_SEAM")) {
        std::vector<unsigned char> seeds(V.size() / 3, 0);
        for (int64_t v = 0; v < n_vertices; ++v) seeds[v] = rim[v];
        std::vector<int> face_run(F.size() / 3);              // (pass 1 covered every face: the spliced faces are the runs', in run order)
        size_t o = 0;
        for (int r = 0; r < NWR_REGIONS; ++r) for (size_t i = 0; i < pieces[r].of.size() / 3; ++i) face_run[o++] = r;
        const int rc = seam_pass(V, F, seeds, &face_run, n_iterations, L, max_valence, tot);
        if (rc != NWR_OK) return rc;
    }
    const auto t3 = now();
    double shortest2 = INFINITY;
    auto finish = [&]() {
    // compact (vertices no face refers to any more: collapsed away inside a piece), relative order kept
    {
        const size_t nv = V.size() / 3;
        std::vector<int> remap(nv, -1);
        std::vector<unsigned char> used(nv, 0);
        parallel_for((int64_t)F.size(), 1 << 16, [&](int64_t lo_, int64_t hi_) { for (int64_t i = lo_; i < hi_; ++i) __atomic_store_n(&used[F[i]], (unsigned char)1, __ATOMIC_RELAXED); });
        int64_t n = 0;
        for (size_t v = 0; v < nv; ++v) if (used[v]) remap[v] = (int)n++;
        ov.resize(3 * (size_t)n);
        parallel_for((int64_t)nv, 1 << 15, [&](int64_t lo_, int64_t hi_) {
            for (int64_t v = lo_; v < hi_; ++v) if (remap[v] >= 0) { ov[3 * (size_t)remap[v]] = V[3 * v]; ov[3 * (size_t)remap[v] + 1] = V[3 * v + 1]; ov[3 * (size_t)remap[v] + 2] = V[3 * v + 2]; }
        });
        of.resize(F.size());
        parallel_for((int64_t)F.size(), 1 << 16, [&](int64_t lo_, int64_t hi_) { for (int64_t i = lo_; i < hi_; ++i) of[i] = remap[F[i]]; });
    }
    {
        // mean edge length, largest degree and shortest edge of the result: chunks of faces of a FIXED size on all threads, the chunks' sums
        // added in chunk order (the same number whatever the thread count)
        const size_t nv = ov.size() / 3, nfo = of.size() / 3;
        const size_t CH = 32768, nch = (nfo + CH - 1) / CH;
        std::vector<int> deg(nv, 0);
        std::vector<double> psum(nch, 0.0), pmin(nch, INFINITY);
        parallel_for((int64_t)nch, 1, [&](int64_t c_lo, int64_t c_hi) {
            for (int64_t c = c_lo; c < c_hi; ++c) {
                double s = 0, mn = INFINITY;
                for (size_t f = (size_t)c * CH; f < std::min(nfo, (size_t)(c + 1) * CH); ++f)
                    for (int k = 0; k < 3; ++k) {
                        const int a = of[3 * f + k], b = of[3 * f + (k + 1) % 3];
                        const double dx = (double)ov[3 * a] - ov[3 * b], dy = (double)ov[3 * a + 1] - ov[3 * b + 1], dz = (double)ov[3 * a + 2] - ov[3 * b + 2];
                        const double d2 = dx * dx + dy * dy + dz * dz;
                        s += std::sqrt(d2);
                        mn = d2 < mn || !(d2 == d2) ? d2 : mn;
                        __atomic_fetch_add(&deg[a], 1, __ATOMIC_RELAXED);
                    }
                psum[c] = s; pmin[c] = mn;
            }
        });
        double s = 0, mn = INFINITY;
        for (size_t c = 0; c < nch; ++c) { s += psum[c]; if (!(pmin[c] >= mn)) mn = pmin[c]; }
        if (min_edge2) *min_edge2 = mn;
        shortest2 = mn;
        if (stats) {
            *stats = tot;
            stats->mean_edge_length = nfo ? s / (3.0 * (double)nfo) : 0.0;
            stats->max_valence = nv ? *std::max_element(deg.begin(), deg.end()) : 0;
            stats->reserved = 0;
        }
    }
    };
    finish();
    // An edge of (nearly) no length in the result -- two vertices that frozen rims of both passes kept apart from a collapse; seen at
    // 8 10^5 vertices in two of a fit's seven calls -- used to send the WHOLE mesh to the serial algorithm (4 s there).  It is mended where it
    // is: the end points of such edges seed one more small pass (their neighbourhood as a sub-mesh of its own, in which they are interior),
    // and only if an edge of no length survives that does the caller fall back.
    const double none2 = 1e-12 * (double)L * (double)L;
    if (!(shortest2 > none2)) {
        std::vector<unsigned char> seeds2(V.size() / 3, 0);
        size_t n_short = 0;
        for (size_t f = 0; f < F.size() / 3; ++f)
            for (int k = 0; k < 3; ++k) {
                const int a = F[3 * f + k], b = F[3 * f + (k + 1) % 3];
                const double dx = (double)V[3 * (size_t)a] - V[3 * (size_t)b], dy = (double)V[3 * (size_t)a + 1] - V[3 * (size_t)b + 1], dz = (double)V[3 * (size_t)a + 2] - V[3 * (size_t)b + 2];
                if (!(dx * dx + dy * dy + dz * dz > none2)) { seeds2[a] = seeds2[b] = 1; ++n_short; }
            }
        if (verbose) std::fprintf(stderr, "[nw_remesh] %zu half-edges of no length after both passes: one more pass around them\n", n_short);
        const int rc2 = seam_pass(V, F, seeds2, nullptr, n_iterations, L, max_valence, tot);
        if (rc2 != NWR_OK) return rc2;
        finish();
    }
    if (verbose) std::fprintf(stderr, "[nw_remesh] partitioned: order + runs %.1f ms, pass 1 (%d threads) %.1f ms, seam zone %.1f ms, compaction %.1f ms\n", ms(t0, t1), T, ms(t1, t2),
                              ms(t2, t3), ms(t3, now()));
    return NWR_OK;
}

NWR_EXPORT int nwr_remesh(const float *vertices, int64_t n_vertices, const int32_t *faces, int64_t n_faces,
                          int n_iterations, float target_edge_length, float relax_lambda, int n_relax, int max_valence,
                          float **out_vertices, int64_t *out_n_vertices, int32_t **out_faces, int64_t *out_n_faces,
                          nwr_stats *stats)
{
    if (!vertices || !faces || !out_vertices || !out_n_vertices || !out_faces || !out_n_faces) return NWR_ERR_BADARG;
    if (n_vertices < 3 || n_faces < 1 || n_vertices > (1ll << 30) || n_faces > (1ll << 29) || n_iterations < 0 || n_relax < 0)
        return NWR_ERR_BADARG;
    *out_vertices = nullptr; *out_faces = nullptr; *out_n_vertices = 0; *out_n_faces = 0;
    const auto t_call = std::chrono::steady_clock::now();
    try {
        std::vector<float> ov;
        std::vector<int32_t> of;
        int rc;
        static const bool part_env = !(std::getenv("NW_REMESH_PARTITION") && std::atoi(std::getenv("NW_REMESH_PARTITION")) == 0);
        const bool part_on = part_env && g_partition.load() != 0;
        if (part_on && n_relax == 0 && n_iterations > 0 && n_faces >= NWR_PARALLEL_MIN_FACES) {
            {
                std::atomic<int> bad{0};
                parallel_for(3 * n_vertices, 1 << 16, [&](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) if (!std::isfinite(vertices[i])) bad.store(1, std::memory_order_relaxed); });
                parallel_for(n_faces, 1 << 15, [&](int64_t lo, int64_t hi) {
                    for (int64_t f = lo; f < hi; ++f) {
                        const int a = faces[3 * f], b = faces[3 * f + 1], c = faces[3 * f + 2];
                        if (a < 0 || a >= n_vertices || b < 0 || b >= n_vertices || c < 0 || c >= n_vertices || a == b || b == c || c == a) bad.store(1, std::memory_order_relaxed);
                    }
                });
                if (bad.load()) return NWR_ERR_BADARG;
            }
            float L = target_edge_length;
            if (!(L > 0)) {                          // PYME's default: the mean edge length of the input -- of the WHOLE input
                double s = 0;
                for (int64_t f = 0; f < n_faces; ++f)
                    for (int k = 0; k < 3; ++k) {
                        const int a = faces[3 * f + k], b = faces[3 * f + (k + 1) % 3];
                        const double dx = (double)vertices[3 * a] - vertices[3 * b], dy = (double)vertices[3 * a + 1] - vertices[3 * b + 1], dz = (double)vertices[3 * a + 2] - vertices[3 * b + 2];
                        s += std::sqrt(dx * dx + dy * dy + dz * dz);
                    }
                L = (float)(s / (3.0 * (double)n_faces));
            }
            // (a non-manifold input is refused like the serial path does: remesh_partitioned looks for a directed edge that occurs twice)
            if (std::getenv("NWR_VERBOSE")) std::fprintf(stderr, "[nw_remesh] input checks %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count());
            double min_edge2 = INFINITY;
            bool input_bad = false;
            rc = remesh_partitioned(vertices, n_vertices, faces, n_faces, n_iterations, L, max_valence, ov, of, stats, &min_edge2, &input_bad);
            if (input_bad) return rc;
            // safety net: an edge of (nearly) no length in the result -- never seen since the rims wait for their pass -- and the
            // serial algorithm takes over
            // ... unless the INPUT already had one (coincident vertices the optimiser produced: seen at 8 10^5 vertices, where two of a fit's
            // seven calls then spent 4 s each in the serial algorithm, which cannot mend such an edge either -- a collapse across faces of no
            // area is never admitted): the net is for what the partition does, not for what it was given
            bool input_had_one = false;
            if (rc == NWR_OK && !(min_edge2 > 1e-12 * (double)L * (double)L)) {
                const size_t CH = 32768, nch = ((size_t)n_faces + CH - 1) / CH;
                std::vector<double> pmin(nch, INFINITY);
                parallel_for((int64_t)nch, 1, [&](int64_t c_lo, int64_t c_hi) {
                    for (int64_t c = c_lo; c < c_hi; ++c) {
                        double mn = INFINITY;
                        for (size_t f = (size_t)c * CH; f < std::min((size_t)n_faces, (size_t)(c + 1) * CH); ++f)
                            for (int k = 0; k < 3; ++k) {
                                const int a = faces[3 * f + k], b = faces[3 * f + (k + 1) % 3];
                                const double dx = (double)vertices[3 * a] - vertices[3 * b], dy = (double)vertices[3 * a + 1] - vertices[3 * b + 1], dz = (double)vertices[3 * a + 2] - vertices[3 * b + 2];
                                mn = std::min(mn, dx * dx + dy * dy + dz * dz);
                            }
                        pmin[c] = mn;
                    }
                });
                double mn = INFINITY;
                for (double x : pmin) mn = std::min(mn, x);
                input_had_one = !(mn > 1e-12 * (double)L * (double)L);
                if (input_had_one && std::getenv("NWR_VERBOSE")) std::fprintf(stderr, "[nw_remesh] an edge of no length in the result -- and in the input: kept\n");
            }
            if (rc != NWR_OK || (!(min_edge2 > 1e-12 * (double)L * (double)L) && !input_had_one && !std::getenv("NWR_KEEP_SHORT"))) {      // (NWR_KEEP_SHORT: developer aid, the partitioned result as it is)
                // (also when a piece failed -- out of memory in a worker, a runaway guard: the serial algorithm on the whole mesh decides)
                if (std::getenv("NWR_VERBOSE")) std::fprintf(stderr, "[nw_remesh] the partitioned pass %s: the serial algorithm takes over\n", rc != NWR_OK ? "failed" : "left an edge of no length");
                rc = remesh_core(vertices, n_vertices, faces, n_faces, n_iterations, target_edge_length, relax_lambda, n_relax, max_valence, nullptr, ov, of, nullptr, stats, nullptr);
            }
        } else {
            rc = remesh_core(vertices, n_vertices, faces, n_faces, n_iterations, target_edge_length, relax_lambda, n_relax, max_valence, nullptr, ov, of, nullptr, stats, nullptr);
        }
        if (rc != NWR_OK) return rc;
        float *pv = (float *)std::malloc(sizeof(float) * std::max<size_t>(ov.size(), 3));
        int32_t *pf = (int32_t *)std::malloc(sizeof(int32_t) * std::max<size_t>(of.size(), 3));
        if (!pv || !pf) { std::free(pv); std::free(pf); return NWR_ERR_NOMEM; }
        std::memcpy(pv, ov.data(), sizeof(float) * ov.size());
        std::memcpy(pf, of.data(), sizeof(int32_t) * of.size());
        *out_vertices = pv; *out_n_vertices = (int64_t)(ov.size() / 3); *out_faces = pf; *out_n_faces = (int64_t)(of.size() / 3);
        return NWR_OK;
    } catch (const std::bad_alloc &) {
        return NWR_ERR_NOMEM;
    } catch (const std::exception &) {          // (std::system_error of a thread that could not be started, ...: no exception crosses the C boundary)
        return NWR_ERR_NOMEM;
    }
}
