// libnanowrap_hip.so -- host side of the C-ABI declared in include/nanowrap.h.
// MI355X (gfx950) only.  One nw_ctx = one device + one stream; all state stays resident in HBM between
// nw_set_* and nw_get; a search() block is a fixed sequence of kernel launches with no host synchronisation
// until the per-iteration logs are read back at the end.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <algorithm>
#include <chrono>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <atomic>
#include <memory>

#include "../../include/nanowrap.h"
#include "nw_kernels.h"

// stable LSD radix sort of (key, value) pairs on the device (csrc/nw_sort.hip, hipCUB); returns a hipError_t as int
int nw_sort_pairs_u32(const unsigned *key_in, unsigned *key_out, const int *val_in, int *val_out, int n, int bits, hipStream_t stream);

static_assert(sizeof(NwIterLogDev) == sizeof(nw_iter_log), "device/host log record mismatch");
static_assert(SC_COUNT <= NW_N_SCALARS, "scalar slots");

#define NW_EXPORT extern "C" __attribute__((visibility("default")))

// RCCL is bound at run time (dlopen): a process that never calls nw_comm_init does not need it, and one that has torch loaded shares
// torch's copy (same soname).  Only the header is needed to build.
#include <dlfcn.h>
#include <rccl/rccl.h>
struct NwRccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load()
    {
        if (lib) return true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) { err = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return false; }
        GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
        Send = (decltype(Send))dlsym(lib, "ncclSend");
        Recv = (decltype(Recv))dlsym(lib, "ncclRecv");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce) { err = "RCCL: a symbol is missing"; (void)dlclose(lib); lib = nullptr; return false; }
        return true;
    }
};
static NwRccl g_rccl;

namespace {

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t ensure(size_t count)
    {
        if (count <= n && p) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        hipError_t e = hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    void swap(DevBuf &o) { std::swap(p, o.p); std::swap(n, o.n); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
};

enum { ST_TOTAL = 0, ST_GRID = 1, ST_NN = 2, ST_ATTRACT = 3, ST_PRIOR = 4, ST_AS = 5, ST_UPDATE = 6, ST_FIXUP = 7, ST_COUNT = 8 };

}  // namespace

// Small persistent host thread pool for the write-back (the strided copy into the caller's vertex records is host-memory bound: one
// thread moves ~0.4 GB/s of 12-byte rows).  A job is a number of CHUNKS taken from a shared counter by whoever is awake -- the calling
// thread included -- and it ends when every chunk has been done, not when every thread has shown up: a thread the scheduler wakes late
// (the GPU boxes' hosts are shared; a woken thread can arrive 15 ms later) finds the counter exhausted and goes back to sleep, instead
// of holding the block up (round 5).  arm(): work is about to come -- the threads wake now and spin for it for a bounded time.
struct NwHostPool {
    struct Job { std::function<void(int)> fn; int n = 0; std::atomic<int> next{0}, done{0}; };
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_work;
    std::shared_ptr<Job> job;
    std::atomic<unsigned long> generation{0};
    unsigned long arm_generation = 0;
    bool stop = false;
    int n = 1;
    static void work(Job &j)
    {
        for (;;) {
            const int c = j.next.fetch_add(1, std::memory_order_relaxed);
            if (c >= j.n) break;
            j.fn(c);
            j.done.fetch_add(1, std::memory_order_release);
        }
    }
    void start(int threads, int device)
    {
        n = threads < 1 ? 1 : threads;
        for (int t = 1; t < n; ++t)
            th.emplace_back([this, device] {
                (void)hipSetDevice(device);
                unsigned long seen = 0, seen_arm = 0;
                for (;;) {
                    std::shared_ptr<Job> j;
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cv_work.wait(lk, [&] { return stop || generation.load() != seen || arm_generation != seen_arm; });
                        if (stop) return;
                        if (generation.load() == seen) {
                            // armed: spin for the job (bounded), then take it like a woken thread
                            seen_arm = arm_generation;
                            lk.unlock();
                            const auto t0 = std::chrono::steady_clock::now();
                            while (generation.load(std::memory_order_acquire) == seen) {
                                for (int k = 0; k < 32; ++k) __builtin_ia32_pause();
                                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(1500)) break;
                            }
                            lk.lock();
                            if (stop) return;
                            if (generation.load() == seen) continue;          // nothing came: back to sleep
                        }
                        seen_arm = arm_generation;
                        seen = generation.load();
                        j = job;
                    }
                    if (j) work(*j);
                }
            });
    }
    void arm()
    {
        if (n <= 1) return;
        {
            std::lock_guard<std::mutex> lk(m);
            ++arm_generation;
        }
        cv_work.notify_all();
    }
    // a job in the BACKGROUND: the pool's threads work through it while the caller goes on (a block's strided mesh records, written
    // while the next block runs on the GPU); wait_posted() lends a hand with what is left and returns when it is done
    std::shared_ptr<Job> posted;
    void post_chunks(int nchunks, const std::function<void(int)> &f)
    {
        wait_posted();
        if (nchunks <= 0) return;
        if (n <= 1) { for (int c = 0; c < nchunks; ++c) f(c); return; }
        auto j = std::make_shared<Job>();
        j->fn = f; j->n = nchunks;
        {
            std::lock_guard<std::mutex> lk(m);
            job = j;
            generation.fetch_add(1, std::memory_order_release);
        }
        cv_work.notify_all();
        posted = j;
    }
    void wait_posted()
    {
        if (!posted) return;
        std::shared_ptr<Job> j = posted;
        posted.reset();
        work(*j);
        while (j->done.load(std::memory_order_acquire) < j->n)
            for (int k = 0; k < 8; ++k) __builtin_ia32_pause();
        std::lock_guard<std::mutex> lk(m);
        if (job == j) job.reset();
    }
    // f(c) for c in [0, nchunks), each exactly once, on whichever threads are awake; returns when all have been done
    void run_chunks(int nchunks, const std::function<void(int)> &f)
    {
        wait_posted();
        if (nchunks <= 0) return;
        if (n <= 1 || nchunks == 1) { for (int c = 0; c < nchunks; ++c) f(c); return; }
        auto j = std::make_shared<Job>();
        j->fn = f; j->n = nchunks;
        {
            std::lock_guard<std::mutex> lk(m);
            job = j;
            generation.fetch_add(1, std::memory_order_release);
        }
        cv_work.notify_all();
        work(*j);
        while (j->done.load(std::memory_order_acquire) < nchunks)
            for (int k = 0; k < 8; ++k) __builtin_ia32_pause();
        {
            std::lock_guard<std::mutex> lk(m);
            if (job == j) job.reset();
        }
    }
    void run(const std::function<void(int)> &f) { run_chunks(n, f); }
    void shutdown()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &t : th) t.join();
        th.clear();
    }
};

struct nw_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // sizes
    int64_t N = 0, M = 0, F = 0;
    int NB = 0;
    // captured search() blocks (hipGraph): replayed while nothing they bake in has changed
    struct BlockGraph { hipGraphExec_t exec = nullptr; uint64_t key = 0; };
    DevBuf<int> face_orig;            // internal face id -> the caller's (faces are kept in Morton order of their centroids, nw_set_mesh)
    bool face_sorted = false;
    bool capturing = false;
    // multi-GPU: this rank's RCCL communicator (nw_comm_init).  A block of nw_search then runs its collectives itself, on the ctx's stream,
    // between the phases -- recorded into the block's hipGraph like every launch (NW_FLAG_COMM_*: which buffers go round).
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_ranks = 1;
    uint32_t comm_mode = 0;                // NW_FLAG_COMM_* of the current search
    DevBuf<unsigned char> comm_scratch;    // staging of host buffers given to nw_comm_all_reduce
    int64_t n_staged_copy_outs = 0, n_write_backs = 0;      // nw_debug, what = 2
    std::vector<uint64_t> comm_patterns;   // communication patterns (mode, sizes, peers) that have run one block outside a capture (nw_search)
    int comm_capture_failures = 0;         // blocks with collectives whose recording failed (RCCL calls that a stream capture does not take): after two, such blocks are launched directly
    bool direct_out = false;         // nw_search: the last update of the block writes its result into the pinned staging buffer itself
    BlockGraph graphs[8];
    int graph_next = 0;
    uint64_t grid_generation = 0;
    int item_points = 64;
    bool have_points = false, have_mesh = false;

    // localizations as given (caller order) -- kept so the grid can be rebuilt when the cell size changes
    DevBuf<float> pts_in, sinv_in, w_in;
    bool sinv_array = false;
    float sinv_scalar = 1.0f, w_scalar = 1.0f;
    int w_mode = NW_WEIGHTS_FROM_SIGMA_INV;
    bool w_array = false;             // effective weights are a (3N,) array
    float pmin[3] = {0, 0, 0}, pmax[3] = {0, 0, 0};
    DevBuf<double> wsum;

    // localizations in Morton order (sorted once per nw_set_points; the order never changes afterwards)
    DevBuf<float4> pts;
    DevBuf<int> perm;                 // sorted slot -> caller index
    DevBuf<unsigned> mkey;            // sorted Morton keys (work items are cut from them)
    float proj_unit = 1.0f;
    float morton_unit = 1.0f;         // edge of one Morton quantum (cloud bounding cube / 1024)
    DevBuf<float> sinv, wnorm;
    DevBuf<unsigned char> mask;

    // grid
    NwGrid grid{};
    bool grid_valid = false;
    double last_mean_dist = -1.0, spacing = 0.0, est_mean_dist = -1.0;
    double sigma_eff = -1.0;          // mean localization precision (1 / mean sigma_inv), <= 0 if unknown
    double scene_ext = 1.0;           // extent of localizations + mesh at the last grid build
    double quantum_override = 0.0;    // > 0: nw_accumulator_quantum fixed it (multi-GPU: every rank must use the same)
    double acc_quantum = 1.0;         // fixed-point quantum of the LDS scatter accumulators (k_attract): 2^-36 of the cloud extent
    double extent_hint = 0.0;         // > 0: extent of the WHOLE mesh (a rank of a sharded mesh; nw_refresh_normals)
    double local_quantum = 1.0;       // what this ctx would choose for its own localizations and weights (nw_search_begin computes it every block)
    double cell_tune = 1.0;           // multiplier on the cell-size rule (tune_grid)
    bool tuned = false;
    int blocks_done = 0;              // completed search() calls since the localizations were set
    double force_h = 0.0;             // > 0: build_grid uses exactly this cell (nw_tune_grid probes)
    DevBuf<int> ccount, cstart, scan_tmp;
    DevBuf<int> ctile;              // sums of the cell histogram per scan tile, kept by k_face_centroids, zero between iterations
    DevBuf<float> data_in, dat;       // search(data != points): the residual's target in the caller's / the sorted order (nw_set_data)
    bool have_data = false;
    DevBuf<NwItem> items;             // work list of the NN query: runs of <= 64 Morton-consecutive localizations
    DevBuf<unsigned> item_cost;       // measured duration of every item in the last query (until the list has been ordered by it)
    bool handoff_off = false;         // NW_ERR_HANDOFF was raised once: the attraction step stays a launch of its own
    DevBuf<int> query_serial;         // number of grid builds so far (k_face_centroids): the value a finished item's word takes
    DevBuf<int> item_done;            // per work item: the number of the last iteration whose query has left it (the attraction workgroups of the same launch poll it)
    bool items_by_cost = false;       // the list has been ordered longest-first (once per work list)
    bool item_cost_valid = false;     // a warm query has filled item_cost for the current list
    int nitems = 0;
    int item_level = -1;              // Morton level (block edge = morton_unit * 2^level) the items were cut at
    // scratch of the block-boundary / diagnostic entry points (nw_curvature, nw_lfunc): kept, so that a call per block does not
    // allocate and free tens of megabytes
    DevBuf<int> aux_i;
    DevBuf<float> aux_f, aux_f2, aux_f3;
    DevBuf<double> aux_d;
    DevBuf<unsigned long long> nn_stats;   // developer counters of the NN query (nw_debug, what = 0); null unless enabled
    DevBuf<unsigned> proj_key;        // projection keys of the last completed query (second sort, see k_projection_keys)
    DevBuf<int> proj_idx;
    bool proj_ready = false, proj_sorted = false;
    bool face_warm = false;           // `face` holds the previous iteration's nearest faces of THIS topology (warm start)

    // mesh
    DevBuf<float> pos, meshpos, nrm;
    DevBuf<int> nbr, nbr_t, faces;
    DevBuf<unsigned char> valid, owned;
    bool have_valid = false, have_owned = false;
    // sharded mesh ('halo' mode): the boundary vertices this rank holds, the dense exchange buffers over the global boundary list
    DevBuf<int> hb_local, hb_slot, hb_slot2local, hb_gv;
    DevBuf<long long> halo_acc;       // (n_boundary, 4) accumulator rows
    DevBuf<float> halo_rows;          // (n_boundary, 3) position / normal rows (owner-only non-zero)
    DevBuf<float> halo_full;          // (M_global, 3) owners' rows of the whole mesh (one all-reduce per block)
    DevBuf<float> halo_ref;           // (M_global, 3) the whole mesh when the shares were cut (nw_halo_set_reference)
    DevBuf<float> halo_d0;            // (N,) caller order: this rank's nearest distances when the shares were cut (per-localization halos), or unused
    bool have_halo_d0 = false;
    int64_t halo_d0_n = 0;
    DevBuf<float> halo_stats;         // {largest nearest distance, quantum, max drift^2, 0}: one MAX all-reduce per block
    void *pin_full = nullptr;         // pinned: the whole mesh (3 M_global floats) + the 4 statistics, staged by the tail of a sharded nw_search block
    size_t pin_full_bytes = 0;
    bool full_staged = false;         // pin_full holds the result of the last block (until the next search begins)
    bool have_halo_ref = false;
    int64_t hb_n = 0, hb_nslot = 0, M_global = 0;
    bool have_boundary = false;
    bool pos_unpack_pending = false;  // the owners' new boundary positions are in halo_rows (all-reduced by the caller), not yet taken
    // owner-wise exchange of the boundary rows (nw_set_boundary with peers): rows = (peer, vertex) pairs, peer after peer.  `ghost` rows:
    // this rank's copies of vertices the peer owns; `owned` rows: vertices this rank owns that the peer holds a copy of.  A copy's partial
    // accumulator row goes to the owner, the owner's sum and, after the update, its new position come back: three neighbour exchanges per
    // iteration instead of two all-reduces over the dense global boundary list.
    bool have_peers = false;
    std::vector<int> px_rank;
    std::vector<int64_t> px_goff, px_ooff;        // [n_peers + 1]
    DevBuf<int> px_ghost, px_owned;               // local vertex ids, by peer
    DevBuf<long long> px_send, px_recv;           // max(ghost rows, owned rows) x 4 int64 (the float32 rows of positions / normals use the front)
    int64_t px_ng = 0, px_no = 0;
    int maxdeg = 0;
    DevBuf<int> d_small;              // small int scratch (maxdeg, flags)

    // per-iteration work arrays
    DevBuf<float4> cent_tmp, cent;
    DevBuf<int> fcell, frank, face, vidx, ambig_list, ambig_count;
    DevBuf<float> dist, w, res, S, fdef, pi;
    DevBuf<float4> ring4;             // ring half of the curvature prior, per vertex: {1-ring position sums (f32), number of neighbours} ...
    DevBuf<double> ring_a;            // ... and the normal-consistency sum: written beside the query (k_nn_wave's appended workgroups), read by k_prior_directions
    DevBuf<long long> vacc;           // (M, 4) fixed-point accumulator {A^T res, sum w}: exact, order-independent sums
    DevBuf<double> scalars;           // [NW_N_SCALARS][NW_SPARTS] sums of the current iteration, NW_SPARTS ordered parts per slot (k_reduce_scalars)
    DevBuf<float> wv;                 // per-vertex weights of the 'wfunc' regulariser (NW_FLAG_WFUNC)
    DevBuf<double> part_a, part_p, part_s;   // per-workgroup partial sums of k_attract / k_prior_directions / k_subspace_point_sums
    bool vacc_dirty = true;           // something other than the iteration has written the scatter accumulator since it was last zeroed
    int attract_rows = 0;             // rows of part_a the last attraction wrote (its workgroups)
    double w_quantum = 1.0;           // fixed-point quantum of the {w} column
    double w_bound = 1.0;             // largest |weight| after normalisation (bounds |res| together with the cloud extent)
    DevBuf<NwDevState> state;
    DevBuf<NwIterLogDev> logs;
    DevBuf<float> mm;                 // min/max scratch
    DevBuf<float> tmp_f;              // read-back / operator scratch
    DevBuf<float> tmp_f2;

    // search bookkeeping
    int global_iter = 0;              // iteration counter across search() calls (stop condition indexing)
    int search_iters = 0, search_done = 0;
    uint32_t search_flags = 0;
    float lam0 = 0.0f;
    bool in_search = false;
    bool begin_ops_pending = false;
    bool last_direct_out = false;     // the previous block's direct_out (nw_optimize_layout pre-records the next block with it)
    bool searched = false;

    void *wb_rows = nullptr;          // strided write-back target registered with nw_set_write_back
    int64_t wb_stride = 0;
    void *pin_log = nullptr;          // pinned staging for the per-iteration logs + device state
    int *pin_flag = nullptr;          // pinned: the word k_block_done writes last (its device counter's value); the waiting host thread spins on it
    DevBuf<int> done_count;           // that counter
    int done_expected = 0;            // value the flag takes when the block in flight has ended
    bool done_launched = false;       // a k_block_done is queued behind the block in flight (nw_search with a staged result)
    double block_us_ema = 0.0;        // how long blocks like the last one take from launch to flag (the copy threads are woken shortly before)
    uint64_t block_us_key = 0;
    std::chrono::steady_clock::time_point block_t0;
    int sliced_S = 0, sliced_base = 0;      // a large result on its way to the staging buffer in slices (k_copy_slice): how many, flag value before them
    int64_t sliced_rows = 0;
    size_t pin_log_bytes = 0;
    void *pin = nullptr;              // pinned staging for the write-back: the half of pin_base in use (two halves: with NW_FLAG_ROWS_ASYNC the copy
    size_t pin_bytes = 0;             // threads still read the last block's half while the next block's last kernel fills the other)
    void *pin_base = nullptr;
    int pin_sel = 0;
    bool rows_async = false;          // this block: the strided vertex records are filled in the background (NW_FLAG_ROWS_ASYNC)
    NwHostPool *pool = nullptr;       // host threads of the write-back (created on first use)
    std::vector<hipEvent_t> wb_events;
    std::vector<unsigned char> valid_host;

    // profiling
    int profiling = 0;                // 0 off, 1 = events around the NN query only (dominant kernel), 2 = every stage
    std::vector<hipEvent_t> events;
    size_t ev_used = 0;
    double stage_ms[ST_COUNT] = {0};
    int64_t stage_launches[ST_COUNT] = {0};
};

namespace {

int fail(nw_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    return code;
}

#define NW_HIP(call)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(ctx, NW_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                \
    } while (0)

#define NW_TRY(expr)                                                                                         \
    do {                                                                                                     \
        int r_ = (expr);                                                                                     \
        if (r_ != NW_OK) return r_;                                                                          \
    } while (0)

inline int nblk(int64_t n, int b = NW_BLOCK) { return (int)((n + b - 1) / b); }

// tile_sums: the raw tile sums are already there (the histogram's producer kept them, k_face_centroids) -- no first pass
int scan_exclusive(nw_ctx *ctx, int *in, int n, int *out, bool zero_input = false, int *tile_sums = nullptr)
{
    const int nb = (n + NW_SCAN_TILE - 1) / NW_SCAN_TILE;
    const bool two_pass = nb <= 4096;          // every workgroup of the last pass adds the earlier tile sums itself
    if (tile_sums && two_pass) {
        hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(NW_BLOCK), 0, ctx->stream, in, n, tile_sums, out, zero_input ? 1 : 0, 1);
        NW_HIP(hipGetLastError());
        return NW_OK;
    }
    NW_HIP(ctx->scan_tmp.ensure((size_t)nb + 1));
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(nb), dim3(NW_BLOCK), 0, ctx->stream, in, n, ctx->scan_tmp.p);
    if (!two_pass) hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(1024), 0, ctx->stream, ctx->scan_tmp.p, nb);
    hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(NW_BLOCK), 0, ctx->stream, in, n, ctx->scan_tmp.p, out, zero_input ? 1 : 0, two_pass ? 1 : 0);
    NW_HIP(hipGetLastError());
    return NW_OK;
}

float dec_ord(int v)   // inverse of the monotone float->int map used by k_minmax3
{
    unsigned u = v >= 0 ? (unsigned)v : (0x80000000u - (unsigned)v);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int enc_ord(float f)
{
    f = f + 0.0f;
    unsigned u;
    memcpy(&u, &f, 4);
    return f >= 0 ? (int)u : (int)(0x80000000u - u);
}

int minmax3(nw_ctx *ctx, const float *xyz, int64_t n, float lo[3], float hi[3], bool *nonfinite)
{
    NW_HIP(ctx->mm.ensure(8));
    int init[8];
    for (int k = 0; k < 3; ++k) { init[k] = enc_ord(INFINITY); init[3 + k] = enc_ord(-INFINITY); }
    init[6] = 0; init[7] = 0;
    NW_HIP(hipMemcpyAsync(ctx->mm.p, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    const int blocks = (int)std::min<int64_t>(2048, (n + NW_BLOCK - 1) / NW_BLOCK);
    hipLaunchKernelGGL(k_minmax3, dim3(blocks), dim3(NW_BLOCK), 0, ctx->stream, xyz, n, ctx->mm.p, (int *)(ctx->mm.p + 6));
    NW_HIP(hipGetLastError());
    int out[8];
    NW_HIP(hipMemcpyAsync(out, ctx->mm.p, sizeof(out), hipMemcpyDeviceToHost, ctx->stream));
    NW_HIP(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 3; ++k) { lo[k] = dec_ord(out[k]); hi[k] = dec_ord(out[3 + k]); }
    if (nonfinite) *nonfinite = out[6] != 0;
    return NW_OK;
}

// Desired fine-cell edge.  Measured on MI355X (tools/cell_sweep.py, DESIGN.md "cell size"): the cost of the staged query is
// a sharp function of the cell size, and its minimum follows the centroid spacing (how many candidates a stage stages),
// the localization precision sigma (how thick the slab of localizations around the surface is, i.e. how well the 256-lane
// workgroups fill) and, weakly, the localizations-per-face ratio; it hardly moves with the point->surface distance until
// that distance exceeds the cell.  Scale-free fit over the four BASELINE configurations, 0.25x / 4x point densities and
// sigma = 4 / 10 / 25 nm:
//     h = 2.9 * spacing * (sigma / spacing)^0.3 * (N / 2.5 F)^-0.11,   floor 0.5 x mean distance (far starts)
// with sigma / spacing clamped to [1, 16] and taken as 4 when sigma is unknown (scalar sigma_inv), times ctx->cell_tune,
// the factor the autotuner (tune_grid) finds at the start of a cloud's second block.
double desired_cell(const nw_ctx *ctx, double mean_dist, double spacing)
{
    const double ratio = std::max((double)ctx->N, 1.0) / (2.5 * std::max((double)ctx->F, 1.0));
    const double sp = std::max(spacing, 1e-30);
    const double sr = ctx->sigma_eff > 0 ? std::min(std::max(ctx->sigma_eff / sp, 1.0), 16.0) : 4.0;
    double h = 2.9 * sp * std::pow(sr, 0.3) * std::pow(ratio, -0.11);
    const char *e = getenv("NW_CELL_FACTOR");             // developer knob: multiplies the rule
    if (e && atof(e) > 0) h *= atof(e);
    return std::max(h, 0.5 * mean_dist) * ctx->cell_tune;      // far starts: cells of at least half the mean distance; then the tuner's factor
}

// ---- work list of the NN query ---------------------------------------------------------------------------
// Items = runs of <= 64 consecutive localizations of the Morton-sorted list that stay inside one aligned Morton block of edge
// morton_unit * 2^level (about four cells): a dense block is cut into equal runs, a sparse one is a single under-filled wave.
// per-item words of the in-launch hand-off query -> attraction step (zero = no iteration yet), and rows of partial sums for as many
// attraction workgroups as the query has workgroups
int ensure_item_done(nw_ctx *ctx)
{
    const size_t want = (size_t)std::max(ctx->nitems, 1) + 256;
    if (!ctx->query_serial.p) {
        NW_HIP(ctx->query_serial.ensure(1));
        NW_HIP(hipMemsetAsync(ctx->query_serial.p, 0, sizeof(int), ctx->stream));
    }
    if (ctx->item_done.n < want) {
        NW_HIP(ctx->item_done.ensure(2 * want));
        NW_HIP(hipMemsetAsync(ctx->item_done.p, 0, 2 * want * sizeof(int), ctx->stream));
    }
    const size_t rows = (size_t)(ctx->nitems + 8 * 16 + 8);             // >= the query's workgroups whatever their size (64-thread workgroups: one per item; launch_query pads to 128)
    if (ctx->part_a.n < 5 * rows) NW_HIP(ctx->part_a.ensure(5 * rows));
    return NW_OK;
}

int build_items(nw_ctx *ctx, int level)
{
    const int64_t N = ctx->N;
    // localizations per wave: 64 when that still gives every SIMD several waves; a small cloud is cut finer (more, lighter waves: the
    // launch then ends with its slowest wave, and a wave's candidate set shrinks with its patch)
    int ipts = (N >= 64 * 9216) ? NW_ITEM_POINTS : NW_ITEM_POINTS / 2;       // 9216 = 1.5 waves for each of the 256 x 4 x 6 wave slots
    if (const char *e = getenv("NW_ITEM_PTS")) ipts = std::max(8, std::min(NW_ITEM_POINTS, atoi(e)));
    if (level == ctx->item_level && ctx->nitems > 0 && ipts == ctx->item_points) return NW_OK;
    ctx->item_points = ipts;
    DevBuf<int> head, hscan, bstart, icount, istart;
    NW_HIP(head.ensure(N)); NW_HIP(hscan.ensure(N + 1));
    hipLaunchKernelGGL(k_block_heads, dim3(nblk(N)), dim3(NW_BLOCK), 0, ctx->stream, ctx->mkey.p, (int)N, 3 * level, head.p);
    NW_TRY(scan_exclusive(ctx, head.p, (int)N, hscan.p));
    int nblocks = 0;
    NW_HIP(hipMemcpyAsync(&nblocks, hscan.p + N, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    NW_HIP(hipStreamSynchronize(ctx->stream));
    NW_HIP(bstart.ensure((size_t)nblocks + 1)); NW_HIP(icount.ensure((size_t)nblocks)); NW_HIP(istart.ensure((size_t)nblocks + 1));
    hipLaunchKernelGGL(k_block_starts, dim3(nblk(N)), dim3(NW_BLOCK), 0, ctx->stream, head.p, hscan.p, (int)N, bstart.p);
    hipLaunchKernelGGL(k_block_item_counts, dim3(nblk(nblocks)), dim3(NW_BLOCK), 0, ctx->stream, bstart.p, nblocks, icount.p, ctx->item_points);
    NW_TRY(scan_exclusive(ctx, icount.p, nblocks, istart.p));
    int nitems = 0;
    NW_HIP(hipMemcpyAsync(&nitems, istart.p + nblocks, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    NW_HIP(hipStreamSynchronize(ctx->stream));
    NW_HIP(ctx->items.ensure((size_t)nitems));
    hipLaunchKernelGGL(k_block_fill_items, dim3(nblk(nblocks)), dim3(NW_BLOCK), 0, ctx->stream, bstart.p, istart.p, nblocks, ctx->items.p, ctx->item_points);
    NW_HIP(hipGetLastError());
    NW_HIP(hipStreamSynchronize(ctx->stream));          // the temporaries die with this scope
    ctx->nitems = nitems;
    ctx->item_level = level;
    NW_HIP(ctx->item_cost.ensure((size_t)2 * nitems));        // (second half: start times, developer aid NW_ITEM_TIMES)
    NW_HIP(hipMemsetAsync(ctx->item_cost.p, 0, (size_t)2 * nitems * sizeof(unsigned), ctx->stream));
    NW_TRY(ensure_item_done(ctx));
    ctx->items_by_cost = false; ctx->item_cost_valid = false;
    if (getenv("NW_VERBOSE"))
        fprintf(stderr, "[nanowrap] work list: Morton level %d (block %.2f), %d blocks, %d items (%.1f localizations per wave)\n", level,
                ctx->morton_unit * (float)(1 << level), nblocks, nitems, (double)N / std::max(nitems, 1));
    return NW_OK;
}

// ---- grid construction ---------------------------------------------------------------------------------
// Cell size: desired_cell() (or ctx->force_h while nw_tune_grid probes); dims capped at 2^25 cells.  Only the CENTROID lattice
// and the work list depend on it; the localizations keep their Morton order.
int build_grid(nw_ctx *ctx, double mean_dist)
{
    const int64_t N = ctx->N, F = ctx->F;
    // mesh bbox + area
    float mlo[3], mhi[3];
    bool bad = false;
    NW_TRY(minmax3(ctx, ctx->meshpos.p, ctx->M, mlo, mhi, &bad));
    if (bad) return fail(ctx, NW_ERR_NONFINITE, "non-finite vertex coordinate");
    NW_HIP(ctx->wsum.ensure(4));
    NW_HIP(hipMemsetAsync(ctx->wsum.p + 1, 0, 2 * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(k_mesh_area, dim3(std::min(1024, nblk(F))), dim3(NW_BLOCK), 0, ctx->stream, ctx->meshpos.p, ctx->faces.p, (int)F, ctx->wsum.p + 1);
    const int nsample = (int)std::min<int64_t>(N, 256);
    if (mean_dist <= 0.0)
        hipLaunchKernelGGL(k_sample_nn, dim3(nsample), dim3(NW_BLOCK), 0, ctx->stream, ctx->pts_in.p, N, nsample, ctx->meshpos.p, ctx->faces.p, (int)F, ctx->wsum.p + 2);
    NW_HIP(hipGetLastError());
    double h2[2];
    NW_HIP(hipMemcpyAsync(h2, ctx->wsum.p + 1, sizeof(h2), hipMemcpyDeviceToHost, ctx->stream));
    NW_HIP(hipStreamSynchronize(ctx->stream));
    const double area = h2[0];
    if (mean_dist <= 0.0) mean_dist = h2[1] / nsample;
    double spacing = std::sqrt(std::max(area, 1e-30) / (double)F);
    float lo[3], hi[3];
    double ext = 0;
    for (int k = 0; k < 3; ++k) ext = std::max(ext, (double)std::max(mhi[k], ctx->pmax[k]) - (double)std::min(mlo[k], ctx->pmin[k]));
    if (!(ext > 0)) ext = 1.0;
    ctx->scene_ext = ext;                                // extent of everything (bounds the residuals: quantum of the scatter)
    ctx->spacing = spacing;
    double h = ctx->force_h > 0 ? ctx->force_h : desired_cell(ctx, mean_dist, spacing);
    if (!(h > 0) || !std::isfinite(h)) h = ext / 16;
    ctx->est_mean_dist = mean_dist;
    // The grid bins CENTROIDS: its box is the mesh's, widened to the localizations only where they are near (what the vertices can
    // reach while the grid is kept).  A localization far away -- background, a fiducial -- lies outside the grid and is handled by the
    // clamped cell coordinates; it must not decide the cell size (one such point used to blow the box, and with it the cells, up).
    double gext = 0;
    const double reach = 4.0 * h + 3.0 * std::max(mean_dist, 0.0);
    for (int k = 0; k < 3; ++k) {
        lo[k] = std::min(mlo[k], std::max(ctx->pmin[k], (float)(mlo[k] - reach)));
        hi[k] = std::max(mhi[k], std::min(ctx->pmax[k], (float)(mhi[k] + reach)));
        gext = std::max(gext, (double)hi[k] - lo[k]);
    }
    if (!(gext > 0)) gext = 1.0;
    h = std::max(h, gext / 1024.0);                      // at most ~1024 cells per axis
    const char *env_h = getenv("NW_CELL_SIZE");
    if (env_h && atof(env_h) > 0 && !(ctx->force_h > 0)) h = atof(env_h);
    NwGrid g;
    for (;;) {
        const double margin = 0.5 * h;
        g.ox = (float)(lo[0] - margin); g.oy = (float)(lo[1] - margin); g.oz = (float)(lo[2] - margin);
        g.gx = std::max(1, (int)std::ceil((hi[0] + margin - g.ox) / h));
        g.gy = std::max(1, (int)std::ceil((hi[1] + margin - g.oy) / h));
        g.gz = std::max(1, (int)std::ceil((hi[2] + margin - g.oz) / h));
        if ((double)g.gx * g.gy * g.gz <= (double)(1 << 25)) break;
        h *= 1.26;
    }
    g.h = (float)h;
    g.inv_h = 1.0f / g.h;
    g.ncell = g.gx * g.gy * g.gz;
    const double maxc = std::max({std::fabs((double)g.ox), std::fabs((double)g.oy), std::fabs((double)g.oz),
                                  std::fabs(g.ox + g.gx * h), std::fabs(g.oy + g.gy * h), std::fabs(g.oz + g.gz * h)});
    g.eps = (float)(1e-3 * h + 2e-6 * maxc);
    ctx->grid = g;
    ctx->grid_generation += 1;

    const size_t nc = (size_t)g.ncell;
    NW_HIP(ctx->ccount.ensure(nc));
    NW_HIP(ctx->cstart.ensure(nc + 1));
    NW_HIP(ctx->scan_tmp.ensure(nc / NW_SCAN_TILE + 2));      // (no allocation inside a captured block)
    NW_HIP(hipMemsetAsync(ctx->ccount.p, 0, nc * sizeof(int), ctx->stream));
    NW_HIP(ctx->ctile.ensure(nc / NW_SCAN_TILE + 2));
    NW_HIP(hipMemsetAsync(ctx->ctile.p, 0, (nc / NW_SCAN_TILE + 2) * sizeof(int), ctx->stream));
    // work list: Morton blocks of about NW_ITEM_BLOCK_CELLS cells per edge
    const double block_cells = getenv("NW_ITEM_BLOCK_CELLS") ? atof(getenv("NW_ITEM_BLOCK_CELLS")) : 4.0;
    int level = (int)std::lround(std::log2(std::max(block_cells * h / (double)ctx->morton_unit, 1.0)));
    level = std::min(std::max(level, 0), 10);
    NW_TRY(build_items(ctx, level));
    ctx->grid_valid = true;
    if (getenv("NW_VERBOSE"))
        fprintf(stderr, "[nanowrap] grid %dx%dx%d h=%.3f (mean_dist %.3f, spacing %.3f) items=%d\n", g.gx, g.gy, g.gz, g.h, mean_dist, spacing, ctx->nitems);
    return NW_OK;
}

int ensure_grid(nw_ctx *ctx)
{
    if (ctx->grid_valid && ctx->last_mean_dist > 0) {
        // keep the grid (and the cell-sorted localizations) unless the wanted cell size drifted far from the current one
        // (re-sorting N points costs about half an iteration)
        const double want = desired_cell(ctx, ctx->last_mean_dist, ctx->spacing);
        static const double band = getenv("NW_GRID_BAND") ? atof(getenv("NW_GRID_BAND")) : 1.15;
        if (want < band * ctx->grid.h && want > ctx->grid.h / band) return NW_OK;
        if (getenv("NW_CELL_SIZE")) return NW_OK;
    } else if (ctx->grid_valid) {
        return NW_OK;
    }
    return build_grid(ctx, ctx->last_mean_dist);
}

hipEvent_t next_event(nw_ctx *ctx)
{
    if (ctx->ev_used == ctx->events.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        ctx->events.push_back(e);
    }
    hipEvent_t e = ctx->events[ctx->ev_used++];
    (void)hipEventRecord(e, ctx->stream);
    return e;
}

struct StageMarks { std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> spans; };
thread_local StageMarks g_marks;

struct StageScope {
    nw_ctx *c; int stage; hipEvent_t a;
    bool on;
    StageScope(nw_ctx *ctx, int s, bool sampled_iteration = true) : c(ctx), stage(s), a(nullptr),
        on(!ctx->capturing && (ctx->profiling == 2 || ((ctx->profiling == 1 || (ctx->profiling == 4 && sampled_iteration)) && s == ST_NN))) { if (on) a = next_event(c); }
    ~StageScope() { if (on) { hipEvent_t b = next_event(c); g_marks.spans.push_back({stage, {a, b}}); c->stage_launches[stage] += 1; } }
};

inline int attract_blocks(const nw_ctx *ctx) { const int nb = (int)((ctx->N + NW_BLOCK * NW_ATTRACT_PPT - 1) / (NW_BLOCK * NW_ATTRACT_PPT)); return 8 * ((nb + 7) / 8); }      // k_attract (XCD-remapped grid)
inline int subspace_blocks(const nw_ctx *ctx) { const int nb = (int)((ctx->N + NW_BLOCK * NW_SUBSPACE_PPT - 1) / (NW_BLOCK * NW_SUBSPACE_PPT)); return 8 * ((nb + 7) / 8); }
inline int prior_blocks(const nw_ctx *ctx) { return std::min(nblk(ctx->M), 512); }

int alloc_work(nw_ctx *ctx)
{
    const int64_t N = ctx->N, M = ctx->M, F = ctx->F;
    NW_HIP(ctx->cent_tmp.ensure(F));
    NW_HIP(ctx->cent.ensure(F));
    NW_HIP(ctx->fcell.ensure(F));
    NW_HIP(ctx->frank.ensure(F));
    NW_HIP(ctx->face.ensure(N));
    NW_HIP(ctx->ambig_list.ensure(N));
    NW_HIP(ctx->ambig_count.ensure(4));
    NW_HIP(ctx->dist.ensure(N));
    NW_HIP(ctx->vidx.ensure(3 * N));
    NW_HIP(ctx->w.ensure(3 * N));
    NW_HIP(ctx->res.ensure(3 * N));
    if (ctx->vacc.n < (size_t)(4 * M)) ctx->vacc_dirty = true;
    NW_HIP(ctx->vacc.ensure(4 * M));
    NW_HIP(ctx->S.ensure(9 * M));
    NW_HIP(ctx->fdef.ensure(3 * M));
    NW_HIP(ctx->pi.ensure(M));
    NW_HIP(ctx->ring4.ensure(M));
    NW_HIP(ctx->ring_a.ensure(M));
    NW_HIP(ctx->scalars.ensure(NW_N_SCALARS * NW_SPARTS));
    NW_HIP(ctx->part_a.ensure((size_t)5 * attract_blocks(ctx)));
    NW_HIP(ctx->part_s.ensure((size_t)9 * subspace_blocks(ctx)));
    NW_HIP(ctx->part_p.ensure((size_t)14 * prior_blocks(ctx)));
    NW_HIP(ctx->wv.ensure(ctx->M));
    return NW_OK;
}

// (Folding k_reduce_scalars into its producers with tickets was built twice -- rounds 3 and 5 -- and is slower both times: tools/experiments/r05_notes.md)

}  // namespace

// =============================================================================================================
NW_EXPORT int nw_abi_version(void) { return NW_ABI_VERSION; }
NW_EXPORT int nw_info(int what)
{
    switch (what) {
    case NW_INFO_POINT_SCALARS: return SC_NPOINT;
    case NW_INFO_SCALARS: return SC_MAXD;                     // the SUMMED slots (what ranks all-reduce); the max-distance slot behind them stays local
    case NW_INFO_SCALAR_STRIDE: return NW_SPARTS;
    default: return NW_ERR_BADARG;
    }
}

NW_EXPORT int nw_create(int device, nw_ctx **out)
{
    if (!out) return NW_ERR_BADARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return NW_ERR_HIP;
    if (device < 0 || device >= ndev) return NW_ERR_BADARG;
    if (hipSetDevice(device) != hipSuccess) return NW_ERR_HIP;
    nw_ctx *ctx = new nw_ctx();
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return NW_ERR_HIP; }
    ctx->own_stream = true;
    if (ctx->state.ensure(1) != hipSuccess) { delete ctx; return NW_ERR_NOMEM; }
    NwDevState st{};
    st.stop_at = 0x7fffffff;
    (void)hipMemcpy(ctx->state.p, &st, sizeof(st), hipMemcpyHostToDevice);
    *out = ctx;
    return NW_OK;
}

NW_EXPORT void nw_destroy(nw_ctx *ctx)
{
    if (ctx && ctx->comm) (void)nw_comm_init(ctx, nullptr, 0, 0, 0);
    if (ctx && ctx->pin_full) { (void)hipHostFree(ctx->pin_full); ctx->pin_full = nullptr; }
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->pts_in.release(); ctx->sinv_in.release(); ctx->w_in.release(); ctx->wsum.release();
    ctx->pts.release(); ctx->perm.release(); ctx->mkey.release(); ctx->proj_key.release(); ctx->proj_idx.release(); ctx->sinv.release(); ctx->wnorm.release(); ctx->mask.release();
    ctx->ccount.release(); ctx->cstart.release(); ctx->scan_tmp.release(); ctx->ctile.release(); ctx->items.release(); ctx->nn_stats.release(); ctx->aux_i.release(); ctx->aux_f.release(); ctx->aux_f2.release(); ctx->aux_f3.release(); ctx->aux_d.release();
    ctx->pos.release(); ctx->meshpos.release(); ctx->nrm.release(); ctx->nbr.release(); ctx->nbr_t.release(); ctx->faces.release();
    ctx->valid.release(); ctx->owned.release(); ctx->d_small.release();
    ctx->px_ghost.release(); ctx->px_owned.release(); ctx->px_send.release(); ctx->px_recv.release(); ctx->hb_local.release(); ctx->hb_slot.release(); ctx->hb_slot2local.release(); ctx->hb_gv.release(); ctx->halo_acc.release(); ctx->halo_rows.release(); ctx->halo_full.release(); ctx->halo_ref.release(); ctx->halo_stats.release();
    ctx->ambig_list.release(); ctx->ambig_count.release(); ctx->cent_tmp.release(); ctx->cent.release(); ctx->fcell.release(); ctx->frank.release(); ctx->face.release(); ctx->vidx.release();
    ctx->dist.release(); ctx->w.release(); ctx->res.release(); ctx->vacc.release(); ctx->S.release(); ctx->fdef.release(); ctx->pi.release();
    ctx->scalars.release(); ctx->part_a.release(); ctx->part_p.release(); ctx->part_s.release(); ctx->wv.release(); ctx->state.release(); ctx->logs.release(); ctx->mm.release(); ctx->tmp_f.release(); ctx->tmp_f2.release();
    for (auto &gph : ctx->graphs) { if (gph.exec) (void)hipGraphExecDestroy(gph.exec); }
    if (ctx->pool) { ctx->pool->wait_posted(); ctx->pool->shutdown(); delete ctx->pool; }
    if (ctx->pin_base) (void)hipHostFree(ctx->pin_base);
    if (ctx->pin_log) (void)hipHostFree(ctx->pin_log);
    if (ctx->pin_flag) (void)hipHostFree(ctx->pin_flag);
    for (hipEvent_t e : ctx->wb_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

NW_EXPORT const char *nw_last_error(nw_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

NW_EXPORT int nw_set_stream(nw_ctx *ctx, void *hip_stream)
{
    if (!ctx) return NW_ERR_BADARG;
    NW_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (hip_stream) { ctx->stream = (hipStream_t)hip_stream; ctx->own_stream = false; }
    else { NW_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->own_stream = true; }
    return NW_OK;
}

NW_EXPORT int nw_synchronize(nw_ctx *ctx)
{
    if (!ctx) return NW_ERR_BADARG;
    NW_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->pool) ctx->pool->wait_posted();            // (NW_FLAG_ROWS_ASYNC: the vertex records of the last block)
    return NW_OK;
}

NW_EXPORT int nw_set_points(nw_ctx *ctx, const float *xyz, int64_t n_points, const float *sigma_inv, float sigma_inv_scalar,
                            int weights_mode, const float *weights, float weights_scalar)
{
    if (!ctx) return NW_ERR_BADARG;
    if (!xyz || n_points <= 0 || n_points > 0x7fffffff / 4) return fail(ctx, NW_ERR_BADARG, "nw_set_points: bad points array/size");
    if (weights_mode < 0 || weights_mode > 3) return fail(ctx, NW_ERR_BADARG, "nw_set_points: bad weights_mode");
    if ((weights_mode == NW_WEIGHTS_ARRAY || weights_mode == NW_WEIGHTS_PRENORMALIZED) && !weights) return fail(ctx, NW_ERR_BADARG, "nw_set_points: weights array missing");
    NW_HIP(hipSetDevice(ctx->device));
    const int64_t N = n_points;
    ctx->N = N;
    NW_HIP(ctx->pts_in.ensure(3 * N));
    NW_HIP(hipMemcpyAsync(ctx->pts_in.p, xyz, 3 * N * sizeof(float), hipMemcpyDefault, ctx->stream));
    ctx->sinv_array = sigma_inv != nullptr;
    ctx->sinv_scalar = sigma_inv_scalar;
    if (sigma_inv) {
        NW_HIP(ctx->sinv_in.ensure(3 * N));
        NW_HIP(hipMemcpyAsync(ctx->sinv_in.p, sigma_inv, 3 * N * sizeof(float), hipMemcpyDefault, ctx->stream));
    }
    ctx->w_mode = weights_mode;
    if (weights_mode == NW_WEIGHTS_ARRAY || weights_mode == NW_WEIGHTS_PRENORMALIZED) {
        NW_HIP(ctx->w_in.ensure(3 * N));
        NW_HIP(hipMemcpyAsync(ctx->w_in.p, weights, 3 * N * sizeof(float), hipMemcpyDefault, ctx->stream));
        ctx->w_array = true;
    } else if (weights_mode == NW_WEIGHTS_SCALAR) {
        ctx->w_array = false;
        ctx->w_scalar = weights_scalar;
    } else {
        ctx->w_array = ctx->sinv_array;       // weights = sigma_inv
        ctx->w_scalar = sigma_inv_scalar;
    }
    NW_HIP(ctx->wsum.ensure(4));
    NW_HIP(hipMemsetAsync(ctx->wsum.p, 0, 4 * sizeof(double), ctx->stream));
    if (ctx->w_array) {
        const float *src = (weights_mode == NW_WEIGHTS_ARRAY || weights_mode == NW_WEIGHTS_PRENORMALIZED) ? ctx->w_in.p : ctx->sinv_in.p;
        hipLaunchKernelGGL(k_sum_f64, dim3(std::min(1024, nblk(3 * N))), dim3(NW_BLOCK), 0, ctx->stream, src, 3 * N, ctx->wsum.p);
        NW_HIP(hipGetLastError());
    }
    // largest |weight| after normalisation: together with the scene extent it bounds |res| (quantum of the fixed-point scatter)
    ctx->w_bound = std::fabs((double)ctx->w_scalar);
    if (ctx->w_array) {
        const float *src = (weights_mode == NW_WEIGHTS_ARRAY || weights_mode == NW_WEIGHTS_PRENORMALIZED) ? ctx->w_in.p : ctx->sinv_in.p;
        NW_HIP(ctx->d_small.ensure(8));
        NW_HIP(hipMemsetAsync(ctx->d_small.p, 0, 8 * sizeof(int), ctx->stream));
        hipLaunchKernelGGL(k_absmax_f32, dim3(std::min(1024, nblk(3 * N))), dim3(NW_BLOCK), 0, ctx->stream, src, 3 * N, ctx->d_small.p);
        NW_HIP(hipGetLastError());
        int mbits = 0;
        double wsum_h = 0;
        NW_HIP(hipMemcpyAsync(&mbits, ctx->d_small.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        NW_HIP(hipMemcpyAsync(&wsum_h, ctx->wsum.p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        NW_HIP(hipStreamSynchronize(ctx->stream));
        float wmax; memcpy(&wmax, &mbits, 4);
        const double mean = weights_mode == NW_WEIGHTS_PRENORMALIZED ? 1.0 : wsum_h / (3.0 * (double)N);
        ctx->w_bound = (mean > 0 && std::isfinite(mean)) ? (double)wmax / mean : (double)wmax;
    }
    if (!(ctx->w_bound > 0) || !std::isfinite(ctx->w_bound)) ctx->w_bound = 1.0;
    // localization precision for the cell-size rule: 1 / mean(sigma_inv) when sigma_inv is an array (a scalar is ambiguous:
    // the reference's own driver passes a scalar sigma UN-inverted, _membrane_mesh.pyx:1460-1461)
    ctx->sigma_eff = -1.0;
    if (sigma_inv) {
        hipLaunchKernelGGL(k_sum_f64, dim3(std::min(1024, nblk(3 * N))), dim3(NW_BLOCK), 0, ctx->stream, ctx->sinv_in.p, 3 * N, ctx->wsum.p + 3);
        NW_HIP(hipGetLastError());
        double ssum = 0;
        NW_HIP(hipMemcpyAsync(&ssum, ctx->wsum.p + 3, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        NW_HIP(hipStreamSynchronize(ctx->stream));
        if (std::isfinite(ssum) && ssum > 0) ctx->sigma_eff = 3.0 * (double)N / ssum;
    }
    bool bad = false;
    NW_TRY(minmax3(ctx, ctx->pts_in.p, N, ctx->pmin, ctx->pmax, &bad));
    if (bad) { ctx->have_points = false; return fail(ctx, NW_ERR_NONFINITE, "non-finite localization coordinate"); }
    double ext = 0;
    for (int k = 0; k < 3; ++k) ext = std::max(ext, (double)ctx->pmax[k] - (double)ctx->pmin[k]);
    if (!(ext > 0) || !std::isfinite(ext)) ext = 1.0;
    // Morton order, once: 30-bit code of the position inside the cloud's bounding cube, stable radix sort (deterministic order).
    // A few localizations far from the rest (a fiducial, hot pixels) must not stretch that cube -- the structure would fall into a
    // handful of its 1024^3 quanta and a wave's 64 localizations would be scattered all over it until the second sort: the cube is
    // cut to mean +- 4 sd per axis of the bulk (three trimming passes) when that is less than half the extent; what lies outside gets
    // the keys of the cube's faces.  An ordinary cloud keeps its bounding cube.
    float mlo3[3] = {ctx->pmin[0], ctx->pmin[1], ctx->pmin[2]};
    {
        float blo[3] = {ctx->pmin[0], ctx->pmin[1], ctx->pmin[2]}, bhi[3] = {ctx->pmax[0], ctx->pmax[1], ctx->pmax[2]};
        DevBuf<unsigned long long> mom;
        NW_HIP(mom.ensure(7));
        const double q = ext / 65535.0;
        bool ok = true;
        for (int pass = 0; pass < 3 && ok; ++pass) {
            unsigned long long m[7];
            NW_HIP(hipMemsetAsync(mom.p, 0, sizeof(m), ctx->stream));
            hipLaunchKernelGGL(k_box_moments, dim3((unsigned)std::min<int64_t>(1024, nblk(N))), dim3(NW_BLOCK), 0, ctx->stream, ctx->pts_in.p, N, ctx->pmin[0], ctx->pmin[1],
                               ctx->pmin[2], (float)(1.0 / q), blo[0], blo[1], blo[2], bhi[0], bhi[1], bhi[2], mom.p);
            NW_HIP(hipGetLastError());
            NW_HIP(hipMemcpyAsync(m, mom.p, sizeof(m), hipMemcpyDeviceToHost, ctx->stream));
            NW_HIP(hipStreamSynchronize(ctx->stream));
            if (m[0] < 2) { ok = false; break; }
            for (int k = 0; k < 3; ++k) {
                const double mean = (double)m[1 + k] / (double)m[0];
                const double sd = std::sqrt(std::max((double)m[4 + k] / (double)m[0] - mean * mean, 0.0));
                blo[k] = std::max(ctx->pmin[k], (float)(ctx->pmin[k] + (mean - 4.0 * sd - 1.0) * q));
                bhi[k] = std::min(ctx->pmax[k], (float)(ctx->pmin[k] + (mean + 4.0 * sd + 2.0) * q));
            }
        }
        double bext = 0;
        for (int k = 0; k < 3; ++k) bext = std::max(bext, (double)bhi[k] - (double)blo[k]);
        if (ok && bext > 0 && bext < 0.5 * ext) {
            ext = bext;
            for (int k = 0; k < 3; ++k) mlo3[k] = blo[k];
            if (getenv("NW_VERBOSE")) fprintf(stderr, "[nanowrap] Morton cube cut to the bulk of the cloud: edge %.1f\n", bext);
        }
    }
    {
        ctx->morton_unit = (float)(ext / 1024.0);
        DevBuf<unsigned> key_in;
        DevBuf<int> idx_in;
        NW_HIP(key_in.ensure(N)); NW_HIP(idx_in.ensure(N));
        NW_HIP(ctx->mkey.ensure(N)); NW_HIP(ctx->perm.ensure(N));
        hipLaunchKernelGGL(k_morton_keys, dim3(nblk(N)), dim3(NW_BLOCK), 0, ctx->stream, ctx->pts_in.p, (int)N, mlo3[0], mlo3[1], mlo3[2],
                           1.0f / ctx->morton_unit, key_in.p, idx_in.p);
        NW_HIP(hipGetLastError());
        const int se = nw_sort_pairs_u32(key_in.p, ctx->mkey.p, idx_in.p, ctx->perm.p, (int)N, 30, ctx->stream);
        if (se != 0) return fail(ctx, NW_ERR_HIP, std::string("radix sort of the Morton keys: ") + hipGetErrorString((hipError_t)se));
        NW_HIP(ctx->pts.ensure(N));
        NW_HIP(ctx->mask.ensure(N));
        if (ctx->sinv_array) NW_HIP(ctx->sinv.ensure(3 * N));
        if (ctx->w_array) NW_HIP(ctx->wnorm.ensure(3 * N));
        const float *w_src = ctx->w_array ? ((ctx->w_mode == NW_WEIGHTS_ARRAY || ctx->w_mode == NW_WEIGHTS_PRENORMALIZED) ? ctx->w_in.p : ctx->sinv_in.p) : nullptr;
        hipLaunchKernelGGL(k_point_gather, dim3(nblk(N)), dim3(NW_BLOCK), 0, ctx->stream, (int)N, ctx->pts_in.p, ctx->perm.p,
                           ctx->sinv_array ? ctx->sinv_in.p : nullptr, w_src, ctx->wsum.p, ctx->w_array ? (ctx->w_mode == NW_WEIGHTS_PRENORMALIZED ? 2 : 1) : 0,
                           ctx->pts.p, ctx->sinv_array ? ctx->sinv.p : nullptr, ctx->w_array ? ctx->wnorm.p : nullptr, ctx->mask.p);
        NW_HIP(hipGetLastError());
        NW_HIP(hipStreamSynchronize(ctx->stream));      // key_in / idx_in die with this scope
    }
    ctx->have_points = true;
    ctx->have_data = false;                  // (a residual target given with nw_set_data belonged to the previous localizations)
    ctx->grid_valid = false;
    ctx->item_level = -1;
    ctx->nitems = 0;
    ctx->face_warm = false;
    if (getenv("NW_VERBOSE") && atoi(getenv("NW_VERBOSE")) >= 2) fprintf(stderr, "[nanowrap] warm start dropped (%s)\n", __func__);
    ctx->proj_ready = false; ctx->proj_sorted = false;
    ctx->tuned = false; ctx->cell_tune = 1.0; ctx->blocks_done = 0;
    ctx->last_mean_dist = -1.0;
    ctx->searched = false;
    return NW_OK;
}

// search(data, ...) with `data` other than the localizations of nw_set_points (mesh_conj_grad.py:150, :164, :180-181, :222): the residual
// is taken against `data` ((N,3) float32, the caller's point order) while the weight matrix keeps coming from the localizations.
// NULL: back to the localizations (what nw_set_points leaves).
static int gather_data(nw_ctx *ctx)
{
    const int64_t N = ctx->N;
    if (ctx->have_data) NW_HIP(ctx->dat.ensure(3 * N));
    hipLaunchKernelGGL(k_data_gather, dim3(nblk(N)), dim3(NW_BLOCK), 0, ctx->stream, (int)N, ctx->perm.p, ctx->have_data ? ctx->data_in.p : nullptr, ctx->pts.p,
                       ctx->have_data ? ctx->dat.p : nullptr, ctx->w_array ? nullptr : ctx->mask.p);
    NW_HIP(hipGetLastError());
    return NW_OK;
}

NW_EXPORT int nw_set_data(nw_ctx *ctx, const float *data)
{
    if (!ctx) return NW_ERR_BADARG;
    if (!ctx->have_points) return fail(ctx, NW_ERR_BADARG, "nw_set_data: set the localizations first");
    if (ctx->in_search) return fail(ctx, NW_ERR_BADARG, "nw_set_data inside a search");
    NW_HIP(hipSetDevice(ctx->device));
    if (!data && !ctx->have_data) return NW_OK;
    ctx->have_data = data != nullptr;
    if (data) {
        NW_HIP(ctx->data_in.ensure(3 * ctx->N));
        NW_HIP(hipMemcpyAsync(ctx->data_in.p, data, 3 * ctx->N * sizeof(float), hipMemcpyDefault, ctx->stream));
    }
    NW_TRY(gather_data(ctx));
    NW_HIP(hipStreamSynchronize(ctx->stream));
    return NW_OK;
}

// The faces in Morton order of their centroids (k_face_morton_keys), once per nw_set_mesh: measured on the C3 start mesh (surface-nets
// scan order) -13 us per iteration (centroid binning 29 -> 24, query 131 -> 126), on a mesh with shuffled ids -33 us (binning 50 -> 25).
// Face ids never leave the library except through nw_get(NW_ARR_FACE), which translates; the query's tie rule (lowest face id among
// equidistant centroids) keeps comparing the CALLER's ids (nw_id_less), so results do not depend on the internal order.  Faces that
// name a vertex outside the mesh keep their place in line (the ring-table pass has reported them, or the first query will).
static int sort_faces(nw_ctx *ctx)
{
    static const bool on = !(getenv("NW_FACE_ORDER") && atoi(getenv("NW_FACE_ORDER")) == 0);
    ctx->face_sorted = false;
    if (!on || ctx->F < 2) return NW_OK;
    const int64_t F = ctx->F, M = ctx->M;
    float lo[3], hi[3];
    bool bad = false;
    NW_TRY(minmax3(ctx, ctx->pos.p, M, lo, hi, &bad));
    double ext = 0;
    for (int k = 0; k < 3; ++k) ext = std::max(ext, (double)hi[k] - (double)lo[k]);
    if (bad || !(ext > 0) || !std::isfinite(ext)) return NW_OK;      // (non-finite positions: the caller's order is as good as any)
    DevBuf<unsigned> key_in, key_out;
    DevBuf<int> idx_in, faces_in;
    NW_HIP(key_in.ensure(F)); NW_HIP(key_out.ensure(F)); NW_HIP(idx_in.ensure(F)); NW_HIP(faces_in.ensure(3 * F)); NW_HIP(ctx->face_orig.ensure(F));
    NW_HIP(hipMemcpyAsync(faces_in.p, ctx->faces.p, 3 * F * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_face_morton_keys, dim3(nblk(F)), dim3(NW_BLOCK), 0, ctx->stream, ctx->pos.p, faces_in.p, (int)F, (int)M, lo[0], lo[1], lo[2], (float)(1024.0 / ext),
                       key_in.p, idx_in.p);
    NW_HIP(hipGetLastError());
    const int se = nw_sort_pairs_u32(key_in.p, key_out.p, idx_in.p, ctx->face_orig.p, (int)F, 30, ctx->stream);       // stable: equal keys keep the caller's order
    if (se != 0) return fail(ctx, NW_ERR_HIP, std::string("radix sort of the face keys: ") + hipGetErrorString((hipError_t)se));
    hipLaunchKernelGGL(k_face_regather, dim3(nblk(F)), dim3(NW_BLOCK), 0, ctx->stream, (int)F, ctx->face_orig.p, faces_in.p, ctx->faces.p);
    NW_HIP(hipGetLastError());
    NW_HIP(hipStreamSynchronize(ctx->stream));          // (the scratch buffers die with this scope)
    ctx->face_sorted = true;
    return NW_OK;
}

NW_EXPORT int nw_set_mesh(nw_ctx *ctx, const float *pos, const float *nrm, const int32_t *nbr, const uint8_t *valid, const int32_t *faces,
                          int64_t n_vertices, int64_t n_faces, int n_nbr)
{
    if (!ctx) return NW_ERR_BADARG;
    if (!pos || !faces || n_vertices <= 0 || n_faces <= 0 || n_nbr <= 0 || n_nbr > 64 || n_vertices > 0x7fffffff / 16 || n_faces > 0x7fffffff / 4)
        return fail(ctx, NW_ERR_BADARG, "nw_set_mesh: bad array/size");
    NW_HIP(hipSetDevice(ctx->device));
    const int64_t M = n_vertices, F = n_faces;
    const bool topo_change = (M != ctx->M) || (F != ctx->F);
    ctx->full_staged = false;                             // (a staged whole-mesh result belonged to the previous mesh)
    ctx->M = M; ctx->F = F; ctx->NB = n_nbr;
    NW_HIP(ctx->pos.ensure(3 * M));
    NW_HIP(ctx->meshpos.ensure(3 * M));
    NW_HIP(ctx->nrm.ensure(3 * M));
    NW_HIP(ctx->nbr.ensure((size_t)M * n_nbr));
    NW_HIP(ctx->nbr_t.ensure((size_t)M * n_nbr));
    NW_HIP(ctx->faces.ensure(3 * F));
    NW_HIP(ctx->d_small.ensure(8));
    NW_HIP(hipMemcpyAsync(ctx->pos.p, pos, 3 * M * sizeof(float), hipMemcpyDefault, ctx->stream));
    NW_HIP(hipMemcpyAsync(ctx->meshpos.p, ctx->pos.p, 3 * M * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));      // (the same rows: one trip over PCIe is enough)
    NW_HIP(hipMemcpyAsync(ctx->faces.p, faces, 3 * F * sizeof(int), hipMemcpyDefault, ctx->stream));       // (caller's order; re-ordered below)
    ctx->face_sorted = false;
    ctx->have_valid = valid != nullptr;
    if (ctx->pool) ctx->pool->wait_posted();              // (a block's vertex records still being written read the valid flags)
    if (valid) {
        NW_HIP(ctx->valid.ensure(M));
        NW_HIP(hipMemcpyAsync(ctx->valid.p, valid, M, hipMemcpyDefault, ctx->stream));
        ctx->valid_host.assign(valid, valid + M);          // the write-back masks the caller's vertex records with it
    }
    if (nbr) {
        NW_HIP(hipMemcpyAsync(ctx->nbr.p, nbr, (size_t)M * n_nbr * sizeof(int), hipMemcpyDefault, ctx->stream));
    } else {
        // 1-ring table (and, unless given, the valid flags) from the faces array on the device
        DevBuf<int> cnt;
        DevBuf<int4> pairs;
        NW_HIP(cnt.ensure(M)); NW_HIP(pairs.ensure((size_t)M * n_nbr));
        NW_HIP(ctx->valid.ensure(M));
        NW_HIP(hipMemsetAsync(cnt.p, 0, M * sizeof(int), ctx->stream));
        NW_HIP(hipMemsetAsync(ctx->d_small.p, 0, 8 * sizeof(int), ctx->stream));
        hipLaunchKernelGGL(k_ring_collect, dim3(nblk(3 * F)), dim3(NW_BLOCK), 0, ctx->stream, ctx->faces.p, (int)F, (int)M, n_nbr, cnt.p, pairs.p, ctx->d_small.p, (const int *)nullptr);
        DevBuf<unsigned char> vtmp;
        NW_HIP(vtmp.ensure(M));
        hipLaunchKernelGGL(k_ring_order, dim3(nblk(M)), dim3(NW_BLOCK), 0, ctx->stream, (int)M, n_nbr, cnt.p, pairs.p, ctx->nbr.p, valid ? vtmp.p : ctx->valid.p, (int *)nullptr, (int *)nullptr);
        NW_HIP(hipGetLastError());
        int err[2] = {0, 0};
        NW_HIP(hipMemcpyAsync(err, ctx->d_small.p, sizeof(err), hipMemcpyDeviceToHost, ctx->stream));
        if (!valid) {
            ctx->valid_host.resize(M);
            NW_HIP(hipMemcpyAsync(ctx->valid_host.data(), ctx->valid.p, M, hipMemcpyDeviceToHost, ctx->stream));
            ctx->have_valid = true;
        }
        NW_HIP(hipStreamSynchronize(ctx->stream));
        cnt.release(); pairs.release(); vtmp.release();
        if (err[1]) { ctx->have_mesh = false; return fail(ctx, NW_ERR_BADARG, "nw_set_mesh: a face refers to a vertex outside [0, n_vertices)"); }
        if (err[0]) { ctx->have_mesh = false; return fail(ctx, NW_ERR_BADARG, "nw_set_mesh: a vertex has more neighbours than the table holds (n_nbr)"); }
    }
    NW_HIP(hipMemsetAsync(ctx->d_small.p, 0, 8 * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_nbr_transpose, dim3(nblk(M)), dim3(NW_BLOCK), 0, ctx->stream, ctx->nbr.p, (int)M, n_nbr, ctx->nbr_t.p, ctx->d_small.p);
    NW_HIP(hipGetLastError());
    NW_HIP(hipMemcpyAsync(&ctx->maxdeg, ctx->d_small.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    NW_HIP(hipStreamSynchronize(ctx->stream));
    ctx->have_mesh = true;
    NW_TRY(sort_faces(ctx));
    if (nrm) NW_HIP(hipMemcpyAsync(ctx->nrm.p, nrm, 3 * M * sizeof(float), hipMemcpyDefault, ctx->stream));
    else NW_TRY(nw_refresh_normals(ctx, nullptr, 0.0));         // area-weighted vertex normals from positions + faces on the device
    if (topo_change) { ctx->grid_valid = false; }
    ctx->face_warm = false;                               // face ids of another topology are no starting guess
    if (getenv("NW_VERBOSE") && atoi(getenv("NW_VERBOSE")) >= 2) fprintf(stderr, "[nanowrap] warm start dropped (%s)\n", __func__);
    ctx->have_owned = false;
    ctx->have_boundary = false; ctx->have_peers = false; ctx->pos_unpack_pending = false;      // (a sharding belongs to the mesh it was made for)
    // a new mesh object = a new optimiser in the reference (_membrane_mesh.pyx:1510): history restarts
    NwDevState st{};
    st.stop_at = 0x7fffffff;
    NW_HIP(hipMemcpyAsync(ctx->state.p, &st, sizeof(st), hipMemcpyHostToDevice, ctx->stream));
    NW_HIP(hipStreamSynchronize(ctx->stream));
    ctx->global_iter = 0;
    ctx->searched = false;
    return NW_OK;
}

static int nw_set_owned(nw_ctx *ctx, const uint8_t *owned)
{
    if (!ctx || !ctx->have_mesh) return fail(ctx, NW_ERR_BADARG, "nw_set_boundary: mesh not set");
    ctx->have_owned = owned != nullptr;
    if (owned) {
        NW_HIP(ctx->owned.ensure(ctx->M));
        NW_HIP(hipMemcpyAsync(ctx->owned.p, owned, ctx->M, hipMemcpyDefault, ctx->stream));
        NW_HIP(hipStreamSynchronize(ctx->stream));
    }
    return NW_OK;
}

// Sharded mesh ('halo' mode): the boundary vertices this rank holds (b_local[k] = local vertex, b_slot[k] = its row in the global
// boundary list of n_slots rows), ownership flags, and the global id of every local vertex (gv, M_global) for the per-block gather.
// From then on the split-phase iteration packs / unpacks the exchange buffers itself:
//   nw_iter_attract    ... -> NW_ARR_HALO_ACC  = this rank's partial accumulator rows      (caller: all-reduce SUM, int64)
//   nw_iter_directions takes the reduced rows first
//   nw_iter_update     ... -> NW_ARR_HALO_ROWS = the new positions of the boundary vertices this rank OWNS, zero elsewhere (all-reduce SUM)
//   the next nw_iter_attract / nw_search_end takes the owners' rows (positions and mesh positions) first.
// n_local == 0 with n_slots == 0 is a valid sharding (no shared vertex); b_local == NULL with n_slots < 0 clears it.
NW_EXPORT int nw_set_boundary(nw_ctx *ctx, const int32_t *b_local, const int32_t *b_slot, int64_t n_local, int64_t n_slots, const uint8_t *owned,
                              const int32_t *gv, int64_t n_global, int32_t n_peers, const int32_t *peer_rank, const int64_t *ghost_off,
                              const int32_t *ghost_local, const int64_t *owned_off, const int32_t *owned_local)
{
    if (!ctx || !ctx->have_mesh) return fail(ctx, NW_ERR_BADARG, "nw_set_boundary: mesh not set");
    if (ctx->in_search) return fail(ctx, NW_ERR_BADARG, "nw_set_boundary inside a search");
    ctx->full_staged = false;
    NW_HIP(hipSetDevice(ctx->device));
    if (n_slots < 0) { ctx->have_boundary = false; ctx->have_peers = false; ctx->pos_unpack_pending = false; return nw_set_owned(ctx, nullptr); }
    if (n_local < 0 || n_local > ctx->M || n_local > n_slots || (n_local > 0 && (!b_local || !b_slot)) || !owned || !gv || n_global <= 0 || n_global > 0x7fffffff / 4)
        return fail(ctx, NW_ERR_BADARG, "nw_set_boundary: bad array/size");
    // validate on the host (a bad index would fault a kernel): local ids inside the mesh, slots inside the list, each slot at most once
    {
        std::vector<int32_t> hl, hs, hg;
        auto fetch = [&](const int32_t *p, int64_t n, std::vector<int32_t> &dst) -> bool {
            dst.resize((size_t)n);
            return n == 0 || hipMemcpy(dst.data(), p, (size_t)n * sizeof(int32_t), hipMemcpyDefault) == hipSuccess;
        };
        if (!fetch(b_local, n_local, hl) || !fetch(b_slot, n_local, hs) || !fetch(gv, ctx->M, hg)) return fail(ctx, NW_ERR_HIP, "nw_set_boundary: cannot read the index arrays");
        std::vector<unsigned char> seen((size_t)n_slots, 0);
        for (int64_t k = 0; k < n_local; ++k) {
            if (hl[k] < 0 || hl[k] >= ctx->M || hs[k] < 0 || hs[k] >= n_slots || seen[hs[k]]) return fail(ctx, NW_ERR_BADARG, "nw_set_boundary: index out of range or slot listed twice");
            seen[hs[k]] = 1;
        }
        for (int64_t v = 0; v < ctx->M; ++v)
            if (hg[v] < 0 || hg[v] >= n_global) return fail(ctx, NW_ERR_BADARG, "nw_set_boundary: global vertex id out of range");
        std::vector<int32_t> s2l((size_t)std::max<int64_t>(n_slots, 1), -1);
        for (int64_t k = 0; k < n_local; ++k) s2l[hs[k]] = hl[k];
        NW_HIP(ctx->hb_local.ensure((size_t)std::max<int64_t>(n_local, 1)));
        NW_HIP(ctx->hb_slot.ensure((size_t)std::max<int64_t>(n_local, 1)));
        NW_HIP(ctx->hb_slot2local.ensure((size_t)std::max<int64_t>(n_slots, 1)));
        NW_HIP(ctx->hb_gv.ensure((size_t)ctx->M));
        if (n_local > 0) {
            NW_HIP(hipMemcpyAsync(ctx->hb_local.p, hl.data(), (size_t)n_local * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            NW_HIP(hipMemcpyAsync(ctx->hb_slot.p, hs.data(), (size_t)n_local * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        }
        NW_HIP(hipMemcpyAsync(ctx->hb_slot2local.p, s2l.data(), s2l.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        NW_HIP(hipMemcpyAsync(ctx->hb_gv.p, hg.data(), (size_t)ctx->M * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        NW_HIP(hipStreamSynchronize(ctx->stream));          // the staging vectors die with this scope
    }
    NW_HIP(ctx->halo_acc.ensure((size_t)4 * std::max<int64_t>(n_slots, 1)));
    NW_HIP(ctx->halo_rows.ensure((size_t)3 * std::max<int64_t>(n_slots, 1)));
    NW_HIP(ctx->halo_full.ensure((size_t)3 * n_global));
    NW_HIP(ctx->halo_ref.ensure((size_t)3 * n_global));
    NW_HIP(ctx->halo_stats.ensure(4));
    ctx->have_halo_ref = false; ctx->have_halo_d0 = false;
    NW_HIP(hipMemsetAsync(ctx->halo_acc.p, 0, (size_t)4 * std::max<int64_t>(n_slots, 1) * sizeof(long long), ctx->stream));
    NW_HIP(hipMemsetAsync(ctx->halo_rows.p, 0, (size_t)3 * std::max<int64_t>(n_slots, 1) * sizeof(float), ctx->stream));
    ctx->hb_n = n_local; ctx->hb_nslot = n_slots; ctx->M_global = n_global;
    NW_TRY(nw_set_owned(ctx, owned));
    // owner-wise exchange: the (peer, vertex) rows, validated on the host like the dense list (a bad index would fault a kernel; a count
    // that differs from the peer's would hang the exchange -- the peers' lists come from the same all-reduced holder masks, parallel.py)
    ctx->have_peers = false;
    if (n_peers >= 0 && peer_rank && ghost_off && owned_off) {
        if (n_peers > 4096) return fail(ctx, NW_ERR_BADARG, "nw_set_boundary: too many peers");
        const int64_t ng = ghost_off[n_peers], no = owned_off[n_peers];
        if (ghost_off[0] != 0 || owned_off[0] != 0 || ng < 0 || no < 0 || ng > ctx->M || no > (int64_t)n_peers * ctx->M || (ng > 0 && !ghost_local) || (no > 0 && !owned_local))
            return fail(ctx, NW_ERR_BADARG, "nw_set_boundary: bad peer offsets");
        std::vector<uint8_t> ho((size_t)ctx->M);
        if (hipMemcpy(ho.data(), owned, (size_t)ctx->M, hipMemcpyDefault) != hipSuccess) return fail(ctx, NW_ERR_HIP, "nw_set_boundary: cannot read `owned`");
        for (int p = 0; p < n_peers; ++p) {
            if (ghost_off[p + 1] < ghost_off[p] || owned_off[p + 1] < owned_off[p] || peer_rank[p] < 0) return fail(ctx, NW_ERR_BADARG, "nw_set_boundary: bad peer offsets");
            for (int q = 0; q < p; ++q) if (peer_rank[q] == peer_rank[p]) return fail(ctx, NW_ERR_BADARG, "nw_set_boundary: a peer listed twice");
        }
        std::vector<uint8_t> seen((size_t)ctx->M, 0);
        for (int64_t k = 0; k < ng; ++k) {
            const int l = ghost_local[k];
            if (l < 0 || l >= ctx->M || ho[l] || seen[l]) return fail(ctx, NW_ERR_BADARG, "nw_set_boundary: a ghost row that is out of range, owned here, or listed twice");
            seen[l] = 1;
        }
        for (int64_t k = 0; k < no; ++k) {
            const int l = owned_local[k];
            if (l < 0 || l >= ctx->M || !ho[l]) return fail(ctx, NW_ERR_BADARG, "nw_set_boundary: an owned row that is out of range or not owned here");
        }
        NW_HIP(ctx->px_ghost.ensure((size_t)std::max<int64_t>(ng, 1)));
        NW_HIP(ctx->px_owned.ensure((size_t)std::max<int64_t>(no, 1)));
        if (ng > 0) NW_HIP(hipMemcpy(ctx->px_ghost.p, ghost_local, (size_t)ng * sizeof(int), hipMemcpyHostToDevice));
        if (no > 0) NW_HIP(hipMemcpy(ctx->px_owned.p, owned_local, (size_t)no * sizeof(int), hipMemcpyHostToDevice));
        const size_t rows = (size_t)std::max<int64_t>(std::max(ng, no), 1);
        NW_HIP(ctx->px_send.ensure(4 * rows));
        NW_HIP(ctx->px_recv.ensure(4 * rows));
        NW_HIP(hipMemset(ctx->px_send.p, 0, 4 * rows * sizeof(long long)));
        NW_HIP(hipMemset(ctx->px_recv.p, 0, 4 * rows * sizeof(long long)));
        ctx->px_rank.assign(peer_rank, peer_rank + n_peers);
        ctx->px_goff.assign(ghost_off, ghost_off + n_peers + 1);
        ctx->px_ooff.assign(owned_off, owned_off + n_peers + 1);
        ctx->px_ng = ng; ctx->px_no = no;
        ctx->have_peers = true;
    }
    ctx->have_boundary = true;
    ctx->pos_unpack_pending = false;
    return NW_OK;
}

// owner-wise exchange, the steps around the transfers (the iteration phases call them; the transfers are the caller's -- split-phase -- or
// px_exchange's -- nw_search with a communicator):
//   accumulator  0: the copies' partial rows (4 int64) -> send                          [exchange: ghost rows out, owned rows in]
//                1: received partial rows added to the owners' rows; the owners' sums, as the 4 float32 the kernels convert them
//                   to, -> send                                                        [exchange: owned rows out, ghost rows in]
//                2: the owners' sums taken by the copies (stored as the integers that convert to exactly those floats)
//   positions / normals  0: the owners' rows -> send          [exchange: owned rows out, ghost rows in]     1: taken by the copies
static int px_stage(nw_ctx *ctx, int what, int stage)
{
    const int ng = (int)ctx->px_ng, no = (int)ctx->px_no;
    if (what == NW_ARR_VACC) {
        if (stage == 0) { if (ng > 0) hipLaunchKernelGGL(k_halo_pack_acc, dim3(nblk(ng)), dim3(NW_BLOCK), 0, ctx->stream, ng, ctx->px_ghost.p, ctx->vacc.p, ctx->px_send.p); }
        else if (stage == 1) {
            if (no > 0) {
                hipLaunchKernelGGL(k_px_add_acc, dim3(nblk(no)), dim3(NW_BLOCK), 0, ctx->stream, no, ctx->px_owned.p, ctx->px_recv.p, ctx->vacc.p);
                hipLaunchKernelGGL(k_px_pack_acc_f32, dim3(nblk(no)), dim3(NW_BLOCK), 0, ctx->stream, no, ctx->px_owned.p, ctx->vacc.p, ctx->acc_quantum, ctx->w_quantum,
                                   (float4 *)ctx->px_send.p);
            }
        } else if (ng > 0) hipLaunchKernelGGL(k_px_take_acc_f32, dim3(nblk(ng)), dim3(NW_BLOCK), 0, ctx->stream, ng, ctx->px_ghost.p, (const float4 *)ctx->px_recv.p,
                                              1.0 / ctx->acc_quantum, 1.0 / ctx->w_quantum, ctx->vacc.p);
    } else {
        float *rows = what == NW_ARR_NRM ? ctx->nrm.p : ctx->pos.p;
        if (stage == 0) { if (no > 0) hipLaunchKernelGGL(k_px_pack_rows, dim3(nblk(no)), dim3(NW_BLOCK), 0, ctx->stream, no, ctx->px_owned.p, rows, (float *)ctx->px_send.p); }
        else if (ng > 0) hipLaunchKernelGGL(k_px_take_rows, dim3(nblk(ng)), dim3(NW_BLOCK), 0, ctx->stream, ng, ctx->px_ghost.p, (const float *)ctx->px_recv.p, rows,
                                            what == NW_ARR_NRM ? (float *)nullptr : ctx->meshpos.p);
    }
    NW_HIP(hipGetLastError());
    return NW_OK;
}

// the exchange buffers by hand (the iteration phases call these themselves; a caller needs them for the vertex normals after
// nw_refresh_normals): what = NW_ARR_VACC -> NW_ARR_HALO_ACC; NW_ARR_POS / NW_ARR_NRM -> NW_ARR_HALO_ROWS (owner-only rows)
static int halo_pack(nw_ctx *ctx, int what)
{
    if (ctx->have_peers) return px_stage(ctx, what, 0);
    const int ns = (int)ctx->hb_nslot;
    if (ns <= 0) return NW_OK;
    if (what == NW_ARR_VACC) hipLaunchKernelGGL(k_halo_pack_acc, dim3(nblk(ns)), dim3(NW_BLOCK), 0, ctx->stream, ns, ctx->hb_slot2local.p, ctx->vacc.p, ctx->halo_acc.p);
    else hipLaunchKernelGGL(k_halo_pack_rows, dim3(nblk(ns)), dim3(NW_BLOCK), 0, ctx->stream, ns, ctx->hb_slot2local.p, ctx->owned.p, what == NW_ARR_NRM ? ctx->nrm.p : ctx->pos.p, ctx->halo_rows.p);
    NW_HIP(hipGetLastError());
    return NW_OK;
}

static int halo_unpack(nw_ctx *ctx, int what)
{
    if (ctx->have_peers) return px_stage(ctx, what, what == NW_ARR_VACC ? 2 : 1);
    const int n = (int)ctx->hb_n;
    if (n <= 0) return NW_OK;
    if (what == NW_ARR_VACC) hipLaunchKernelGGL(k_halo_unpack_acc, dim3(nblk(n)), dim3(NW_BLOCK), 0, ctx->stream, n, ctx->hb_local.p, ctx->hb_slot.p, ctx->halo_acc.p, ctx->vacc.p);
    else if (what == NW_ARR_NRM) hipLaunchKernelGGL(k_halo_unpack_rows, dim3(nblk(n)), dim3(NW_BLOCK), 0, ctx->stream, n, ctx->hb_local.p, ctx->hb_slot.p, ctx->halo_rows.p, ctx->nrm.p, (float *)nullptr);
    else hipLaunchKernelGGL(k_halo_unpack_rows, dim3(nblk(n)), dim3(NW_BLOCK), 0, ctx->stream, n, ctx->hb_local.p, ctx->hb_slot.p, ctx->halo_rows.p, ctx->pos.p, ctx->meshpos.p);
    NW_HIP(hipGetLastError());
    return NW_OK;
}

NW_EXPORT int nw_halo_rows(nw_ctx *ctx, int what, int unpack)
{
    if (!ctx || !ctx->have_boundary) return fail(ctx, NW_ERR_BADARG, "nw_halo_rows: no boundary set (nw_set_boundary)");
    if (what != NW_ARR_VACC && what != NW_ARR_POS && what != NW_ARR_NRM) return fail(ctx, NW_ERR_BADARG, "nw_halo_rows: accumulator, positions or normals");
    if (what == NW_ARR_VACC && !ctx->vacc.p) return fail(ctx, NW_ERR_BADARG, "nw_halo_rows: no accumulator yet");
    if (ctx->have_peers) {                                  // `unpack` = the step (px_stage)
        if (unpack < 0 || unpack > (what == NW_ARR_VACC ? 2 : 1)) return fail(ctx, NW_ERR_BADARG, "nw_halo_rows: no such step of the owner-wise exchange");
        if (what == NW_ARR_POS && unpack == 1) ctx->pos_unpack_pending = false;
        return px_stage(ctx, what, unpack);
    }
    if (!unpack) return halo_pack(ctx, what);
    if (what == NW_ARR_POS) ctx->pos_unpack_pending = false;
    return halo_unpack(ctx, what);
}

// where the WHOLE mesh was when the shares were cut ((M_global,3) float32, host or device): the drift budget of the halo is measured from it
NW_EXPORT int nw_halo_set_reference(nw_ctx *ctx, const float *full, const float *d0, int64_t n_d0)
{
    if (!ctx || !ctx->have_boundary || !full) return fail(ctx, NW_ERR_BADARG, "nw_halo_set_reference: no boundary set (nw_set_boundary)");
    ctx->full_staged = false;
    NW_HIP(hipMemcpyAsync(ctx->halo_ref.p, full, (size_t)3 * ctx->M_global * sizeof(float), hipMemcpyDefault, ctx->stream));
    ctx->have_halo_d0 = false; ctx->halo_d0_n = 0;
    if (d0) {
        if (n_d0 < 1) return fail(ctx, NW_ERR_BADARG, "nw_halo_set_reference: d0 needs its length (the rank's localizations)");
        NW_HIP(ctx->halo_d0.ensure((size_t)n_d0));
        NW_HIP(hipMemcpyAsync(ctx->halo_d0.p, d0, (size_t)n_d0 * sizeof(float), hipMemcpyDefault, ctx->stream));
        ctx->have_halo_d0 = true; ctx->halo_d0_n = n_d0;      // (the localizations may be uploaded later: nw_search_begin checks the length)
    }
    NW_HIP(hipStreamSynchronize(ctx->stream));
    ctx->have_halo_ref = true;
    return NW_OK;
}

// NW_ARR_HALO_STATS <- {max_dist, this rank's accumulator quantum, max |NW_ARR_HALO_FULL - reference|^2, 0} (float32): what the ranks
// agree on at the end of a block with ONE MAX all-reduce (exactness of the sharded query, drift of the mesh, quantum of the next block)
NW_EXPORT int nw_halo_block_stats(nw_ctx *ctx, double max_dist)
{
    if (!ctx || !ctx->have_boundary || !ctx->have_halo_ref) return fail(ctx, NW_ERR_BADARG, "nw_halo_block_stats: boundary and reference first (nw_set_boundary, nw_halo_set_reference)");
    if (ctx->in_search) return fail(ctx, NW_ERR_BADARG, "nw_halo_block_stats inside a search");
    ctx->full_staged = false;                             // (the statistics asked for by hand replace the staged ones of the last block)
    NW_HIP(hipMemsetAsync(ctx->halo_stats.p, 0, 4 * sizeof(float), ctx->stream));
    const int blocks = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (ctx->M_global + NW_BLOCK - 1) / NW_BLOCK));
    hipLaunchKernelGGL(k_halo_block_stats, dim3(blocks), dim3(NW_BLOCK), 0, ctx->stream, ctx->M_global, ctx->halo_full.p, ctx->halo_ref.p, (float)max_dist,
                       (float)ctx->local_quantum, ctx->halo_stats.p, (const NwIterLogDev *)nullptr, 0, (float *)nullptr);
    NW_HIP(hipGetLastError());
    return NW_OK;
}

// NW_ARR_HALO_FULL <- the rows of NW_ARR_POS / NW_ARR_NRM of the vertices this rank owns at their global ids, zero elsewhere: one
// all-reduce(sum) over the ranks then holds the whole mesh on every rank (once per block)
NW_EXPORT int nw_halo_gather_owned(nw_ctx *ctx, int what)
{
    if (!ctx || !ctx->have_boundary) return fail(ctx, NW_ERR_BADARG, "nw_halo_gather_owned: no boundary set (nw_set_boundary)");
    if (what != NW_ARR_POS && what != NW_ARR_NRM) return fail(ctx, NW_ERR_BADARG, "nw_halo_gather_owned: positions or normals");
    if (ctx->in_search) return fail(ctx, NW_ERR_BADARG, "nw_halo_gather_owned inside a search");
    NW_HIP(hipMemsetAsync(ctx->halo_full.p, 0, (size_t)3 * ctx->M_global * sizeof(float), ctx->stream));
    hipLaunchKernelGGL(k_halo_gather_owned, dim3(nblk(ctx->M)), dim3(NW_BLOCK), 0, ctx->stream, (int)ctx->M, ctx->hb_gv.p, ctx->owned.p, what == NW_ARR_NRM ? ctx->nrm.p : ctx->pos.p, ctx->halo_full.p);
    NW_HIP(hipGetLastError());
    return NW_OK;
}

NW_EXPORT int nw_refresh_normals(nw_ctx *ctx, float *nrm_out, double whole_mesh_extent)
{
    if (ctx) ctx->extent_hint = (whole_mesh_extent > 0 && std::isfinite(whole_mesh_extent)) ? whole_mesh_extent : 0.0;
    if (!ctx || !ctx->have_mesh) return fail(ctx, NW_ERR_BADARG, "nw_refresh_normals: mesh not set");
    NW_HIP(hipSetDevice(ctx->device));
    if (ctx->in_search) return fail(ctx, NW_ERR_BADARG, "nw_refresh_normals inside a search");
    // the scatter accumulator of the iteration doubles as the normals' accumulator (every block zeroes it before it starts)
    NW_HIP(ctx->vacc.ensure(4 * ctx->M));
    ctx->vacc_dirty = true;                              // (the next block zeroes it again before its first scatter)
    NW_HIP(hipMemsetAsync(ctx->vacc.p, 0, 3 * ctx->M * sizeof(long long), ctx->stream));
    double ext = 1e-30;
    if (ctx->extent_hint > 0) ext = ctx->extent_hint;          // a sharded mesh: the WHOLE mesh's extent, so that every rank uses the same quantum
    else {
        float mlo[3], mhi[3];
        bool bad = false;
        NW_TRY(minmax3(ctx, ctx->meshpos.p, ctx->M, mlo, mhi, &bad));
        if (bad) return fail(ctx, NW_ERR_NONFINITE, "non-finite vertex coordinate");
        for (int k = 0; k < 3; ++k) ext = std::max(ext, (double)mhi[k] - (double)mlo[k]);
    }
    const double q = std::ldexp(1.0, (int)std::ceil(std::log2(ext * ext)) - 44);      // |cross product| <= extent^2; <= 2^5 terms per vertex
    hipLaunchKernelGGL(k_normals_scatter, dim3(nblk(ctx->F)), dim3(NW_BLOCK), 0, ctx->stream, (int)ctx->F, ctx->faces.p, ctx->meshpos.p, ctx->vacc.p, 1.0 / q);
    hipLaunchKernelGGL(k_normals_finish, dim3(nblk(ctx->M)), dim3(NW_BLOCK), 0, ctx->stream, (int)ctx->M, ctx->vacc.p, ctx->nrm.p);
    NW_HIP(hipGetLastError());
    if (nrm_out) NW_HIP(hipMemcpyAsync(nrm_out, ctx->nrm.p, 3 * ctx->M * sizeof(float), hipMemcpyDefault, ctx->stream));
    NW_HIP(hipStreamSynchronize(ctx->stream));
    return NW_OK;
}

// Extent of the whole mesh for a rank that holds a share of it: nw_refresh_normals derives the quantum of its fixed-point normal sums
// from the mesh's bounding box, and ranks whose shares have different boxes would round the same vertex's sum differently.  ext <= 0:
// back to the local box.
NW_EXPORT int nw_reset_history(nw_ctx *ctx)
{
    if (!ctx || !ctx->have_mesh) return fail(ctx, NW_ERR_BADARG, "nw_reset_history: mesh not set");
    if (ctx->in_search) return fail(ctx, NW_ERR_BADARG, "nw_reset_history inside a search");
    NwDevState st{};
    st.stop_at = 0x7fffffff;
    NW_HIP(hipMemcpyAsync(ctx->state.p, &st, sizeof(st), hipMemcpyHostToDevice, ctx->stream));
    NW_HIP(hipStreamSynchronize(ctx->stream));
    ctx->global_iter = 0;
    return NW_OK;
}

// Second sort of the localizations, by the foot point on the surface (k_projection_keys): once per cloud, at the start of the
// first block after a completed query.  Everything per-localization that survives a block boundary moves along; the cached
// weight matrix / residual of the previous block (cg.w, cg.res, Afunc/Ahfunc) are recomputed by the block that starts now.
static int resort_by_projection(nw_ctx *ctx)
{
    const int64_t N = ctx->N;
    DevBuf<int> order;
    DevBuf<float4> pts2;
    DevBuf<int> perm2, face2;
    DevBuf<float> sinv2, wnorm2;
    DevBuf<unsigned char> mask2;
    NW_HIP(order.ensure(N)); NW_HIP(pts2.ensure(N)); NW_HIP(perm2.ensure(N)); NW_HIP(face2.ensure(N)); NW_HIP(mask2.ensure(N));
    if (ctx->sinv_array) NW_HIP(sinv2.ensure(3 * N));
    if (ctx->w_array) NW_HIP(wnorm2.ensure(3 * N));
    const int se = nw_sort_pairs_u32(ctx->proj_key.p, ctx->mkey.p, ctx->proj_idx.p, order.p, (int)N, 30, ctx->stream);
    if (se != 0) return fail(ctx, NW_ERR_HIP, std::string("radix sort of the projection keys: ") + hipGetErrorString((hipError_t)se));
    hipLaunchKernelGGL(k_point_regather, dim3(nblk(N)), dim3(NW_BLOCK), 0, ctx->stream, (int)N, order.p, ctx->pts.p, ctx->perm.p,
                       ctx->sinv_array ? ctx->sinv.p : nullptr, ctx->w_array ? ctx->wnorm.p : nullptr, ctx->mask.p, ctx->face.p,
                       pts2.p, perm2.p, sinv2.p, wnorm2.p, mask2.p, face2.p);
    NW_HIP(hipGetLastError());
    NW_HIP(hipStreamSynchronize(ctx->stream));
    ctx->pts.swap(pts2); ctx->perm.swap(perm2); ctx->mask.swap(mask2); ctx->face.swap(face2);
    if (ctx->sinv_array) ctx->sinv.swap(sinv2);
    if (ctx->w_array) ctx->wnorm.swap(wnorm2);
    ctx->morton_unit = ctx->proj_unit;
    ctx->item_level = -1; ctx->nitems = 0;
    ctx->grid_valid = false;                 // the work list is cut again (build_grid)
    ctx->proj_sorted = true; ctx->proj_ready = false;
    ctx->searched = false;                   // cached per-localization results are in the old order
    ctx->proj_key.release(); ctx->proj_idx.release();
    if (ctx->have_data) NW_TRY(gather_data(ctx));
    if (getenv("NW_VERBOSE")) fprintf(stderr, "[nanowrap] localizations re-sorted by their foot point on the surface\n");
    return NW_OK;
}

enum { QP_GRID = 1, QP_NN = 2, QP_FIXUP = 4, QP_ATTRACT = 8, QP_ALL = 15 };     // parts of the first half of an iteration
static int launch_query(nw_ctx *ctx, int it, int parts = QP_GRID | QP_NN | QP_FIXUP, bool with_ring = false, bool with_attract = false, bool *attract_rode = nullptr);
static NwAttractArgs attract_args(const nw_ctx *ctx);
static nw_ctx::BlockGraph *block_graph(nw_ctx *ctx, int num_iters, bool head = false);
// Cell-size tuner, once per localization cloud.  The query is exact for every cell size, and its cost depends on more than the rule
// (desired_cell) can see -- a 200k-localization tube leaves half of the GPU's wave slots empty and prefers smaller cells (fewer
// candidates per wave) than the 1M-localization vesicle, for which the cost is flat between 9 and 13 nm -- so the query is simply
// timed on a few cells around the rule's (x0.65 / x0.8 / x1 / x1.25; a probe = new cell table + work list + two warm queries) and the
// fastest is kept as a multiplier on the rule for all later meshes of this cloud.  Runs from nw_optimize_layout (a caller can keep it
// out of a timed region), else at the start of the third block; needs a warm query (the first block ran).  NW_AUTOTUNE=0 disables it.
static int tune_grid(nw_ctx *ctx)
{
    if (ctx->tuned) return NW_OK;
    const char *at = getenv("NW_AUTOTUNE");
    if ((at && atoi(at) == 0) || getenv("NW_CELL_SIZE") || getenv("NW_CELL_FACTOR") || ctx->N < 20000 || !ctx->face_warm) {
        if (ctx->face_warm || (at && atoi(at) == 0)) ctx->tuned = true;
        return NW_OK;
    }
    ctx->tuned = true;
    const int verbose = getenv("NW_VERBOSE") != nullptr;
    const int it = 0;
    hipLaunchKernelGGL(k_set_iter_base, dim3(1), dim3(1), 0, ctx->stream, ctx->state.p, ctx->global_iter);
    const int prof = ctx->profiling;
    ctx->profiling = 0;
    NW_TRY(alloc_work(ctx));
    hipEvent_t e0, e1;
    NW_HIP(hipEventCreate(&e0)); NW_HIP(hipEventCreate(&e1));
    const double md = ctx->last_mean_dist > 0 ? ctx->last_mean_dist : ctx->est_mean_dist;
    if (!ctx->grid_valid) { const int r = build_grid(ctx, md); if (r != NW_OK) return r; }
    const double h_rule = desired_cell(ctx, md, ctx->spacing) / ctx->cell_tune;        // what the rule gives for the current state
    const double factors[4] = {1.0, 0.8, 0.65, 1.25};
    double best_f = 1.0, best_t = 1e30, t_rule = 1e30;
    int rc = NW_OK;
    for (int k = 0; k < 4 && rc == NW_OK; ++k) {
        ctx->force_h = h_rule * factors[k];
        rc = build_grid(ctx, md);
        ctx->force_h = 0.0;
        if (rc != NW_OK) break;
        float t = 0, tmin = 1e30f;
        for (int rep = 0; rep < 3 && rc == NW_OK; ++rep) {                     // the first repetition warms the caches of the new table
            if (hipEventRecord(e0, ctx->stream) != hipSuccess) { rc = NW_ERR_HIP; break; }
            rc = launch_query(ctx, it);
            if (rc != NW_OK) break;
            if (hipEventRecord(e1, ctx->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess) rc = NW_ERR_HIP;
            if (rep > 0) tmin = std::min(tmin, t);
        }
        if (verbose) fprintf(stderr, "[nanowrap] autotune: cell %.3f -> query %.4f ms\n", ctx->grid.h, tmin);
        if (k == 0) t_rule = tmin;
        // leave the rule only for a clear gain (the timings of neighbouring cells differ by a few per cent of noise)
        if (rc == NW_OK && tmin < best_t && (k == 0 || tmin < 0.95 * t_rule)) { best_t = tmin; best_f = factors[k]; }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    ctx->profiling = prof;
    if (rc != NW_OK) return rc;
    ctx->cell_tune = best_f;
    ctx->force_h = h_rule * best_f;
    rc = build_grid(ctx, md);
    ctx->force_h = 0.0;
    if (verbose) fprintf(stderr, "[nanowrap] autotune: rule %.3f -> cell %.3f (x%.2f)\n", h_rule, ctx->grid.h, ctx->cell_tune);
    return rc;
}

// Heavy-first order of the work list (once per list, from the durations the last warm query measured): a launch ends with its
// slowest waves, and a cloud of a few hundred thousand localizations is a single round of waves -- the heavy ones must not start last.
static int order_items_by_cost(nw_ctx *ctx)
{
    static const bool on = !(getenv("NW_ITEM_ORDER") && atoi(getenv("NW_ITEM_ORDER")) == 0);
    if (!on || ctx->items_by_cost || !ctx->item_cost_valid || ctx->nitems < 2) return NW_OK;
    const int n = ctx->nitems;
    // a few thousand items: done on the host
    std::vector<NwItem> items(n);
    std::vector<unsigned> cost(n);
    NW_HIP(hipMemcpyAsync(items.data(), ctx->items.p, (size_t)n * sizeof(NwItem), hipMemcpyDeviceToHost, ctx->stream));
    NW_HIP(hipMemcpyAsync(cost.data(), ctx->item_cost.p, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    NW_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<unsigned> tmp(cost);
    std::nth_element(tmp.begin(), tmp.begin() + n / 2, tmp.end());
    const double median = std::max(1.0, (double)tmp[n / 2]);
    // While the list is shorter than two rounds of waves the launch ends on its heaviest items: those are cut into pieces of about the
    // median cost (a piece repeats the item's walk over fewer localizations -- more work in all, shorter waves).
    static const bool split_on = !(getenv("NW_ITEM_SPLIT") && atoi(getenv("NW_ITEM_SPLIT")) == 0);
    const bool split = split_on && n < 2 * 256 * 4 * 6;
    struct Piece { NwItem it; double est; };
    std::vector<Piece> out;
    out.reserve(2 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        int k = 1;
        if (split) k = std::max(1, std::min((int)std::lround(cost[i] / (1.25 * median)), items[i].n / 4));
        const int per = (items[i].n + k - 1) / k;
        for (int p = 0; p < items[i].n; p += per) {
            Piece pc;
            pc.it.p0 = items[i].p0 + p; pc.it.n = std::min(per, items[i].n - p);
            pc.est = (double)cost[i] / k;
            out.push_back(pc);
        }
    }
    // heavy pieces first -- but in two classes only, each in the list's own (spatial) order: neighbours in the list share centroid
    // cells, and the XCD mapping of the launch keeps them on one L2 (fully sorted by cost the query fetched 148 MB instead of 64)
    auto mid = std::stable_partition(out.begin(), out.end(), [&](const Piece &a) { return a.est > 1.5 * median; });
    // ... and the LIGHT ones last: a launch is a few rounds of waves, each wave lives for a quarter of it, and the launch ends one wave's
    // duration after its last wave started -- measured: every wave has started after two thirds of the launch, the rest is waves running
    // out on a GPU that is emptying.  With the lightest third of the list at its end that last stretch is a light wave's, not an
    // average one's.  (Same rule: a class keeps its spatial order.)
    static const double light_share = getenv("NW_ITEM_LIGHT") ? atof(getenv("NW_ITEM_LIGHT")) : 0.35;
    if (light_share > 0 && out.end() - mid > 8) {
        std::vector<double> est;
        for (auto it = mid; it != out.end(); ++it) est.push_back(it->est);
        const size_t kth = (size_t)std::min<double>((double)est.size() - 1, light_share * (double)out.size());
        std::nth_element(est.begin(), est.begin() + kth, est.end());
        const double cut = est[kth];
        std::stable_partition(mid, out.end(), [&](const Piece &a) { return a.est >= cut; });
    }
    const int m = (int)out.size();
    std::vector<NwItem> flat(m);
    for (int i = 0; i < m; ++i) flat[i] = out[i].it;
    NW_HIP(ctx->items.ensure((size_t)m));
    NW_HIP(hipMemcpyAsync(ctx->items.p, flat.data(), (size_t)m * sizeof(NwItem), hipMemcpyHostToDevice, ctx->stream));
    NW_HIP(ctx->item_cost.ensure((size_t)2 * m));
    NW_HIP(hipStreamSynchronize(ctx->stream));
    ctx->nitems = m;
    NW_TRY(ensure_item_done(ctx));
    ctx->items_by_cost = true;
    if (getenv("NW_VERBOSE")) fprintf(stderr, "[nanowrap] work list: heavy items first%s: %d -> %d items\n", split ? ", heavy items cut" : "", n, m);
    return NW_OK;
}

// One-off set-up that would otherwise run at the start of the next block (the projection re-sort of the localizations, the
// work list cut from it, the cell-size tuner and the capture of the block's hipGraph): lets a caller (bench.py) take it out of a
// timed region.  No-op when there is nothing to do.
NW_EXPORT int nw_optimize_layout(nw_ctx *ctx)
{
    if (!ctx) return NW_ERR_BADARG;
    if (ctx->in_search) return fail(ctx, NW_ERR_BADARG, "nw_optimize_layout inside a search");
    if (!ctx->have_points || !ctx->have_mesh) return NW_OK;
    NW_HIP(hipSetDevice(ctx->device));
    if (ctx->proj_ready && !ctx->proj_sorted && !getenv("NW_NO_PROJ_SORT")) NW_TRY(resort_by_projection(ctx));
    NW_TRY(alloc_work(ctx));
    NW_TRY(ensure_grid(ctx));
    NW_TRY(tune_grid(ctx));
    if (!ctx->items_by_cost && ctx->face_warm) {
        if (!ctx->item_cost_valid) {            // (the tuner ends on a fresh work list: one query to time its items)
            hipLaunchKernelGGL(k_set_iter_base, dim3(1), dim3(1), 0, ctx->stream, ctx->state.p, ctx->global_iter);
            const int prof = ctx->profiling;
            ctx->profiling = 0;
            const int rq = launch_query(ctx, 0);
            ctx->profiling = prof;
            if (rq != NW_OK) return rq;
        }
        NW_TRY(order_items_by_cost(ctx));
    }
    // pre-record the next block as a hipGraph, assuming it repeats the last one (iterations, lambda, flags): the capture
    // (a fraction of a millisecond) then does not fall into the caller's next block either.  A different next block just captures again.
    if (ctx->searched && ctx->search_iters > 0 && ctx->face_warm) {
        const float lam = ctx->lam0;
        NW_TRY(nw_search_begin(ctx, &lam, 1, ctx->search_iters, ctx->search_flags));
        ctx->direct_out = ctx->last_direct_out;             // (part of what the graph bakes in: where the last update writes the result)
        (void)block_graph(ctx, ctx->search_iters, ctx->profiling == 4);      // (level 4: everything before the block's last iteration)
        ctx->direct_out = false;
        ctx->in_search = false;
        ctx->begin_ops_pending = false;
    }
    return NW_OK;
}

// ---- the iteration -------------------------------------------------------------------------------------------
static int ensure_staging(nw_ctx *ctx);
NW_EXPORT int nw_search_begin(nw_ctx *ctx, const float *lams, int n_lams, int num_iters, uint32_t flags)
{
    if (!ctx) return NW_ERR_BADARG;
    if (!ctx->have_points || !ctx->have_mesh) return fail(ctx, NW_ERR_BADARG, "nw_search: set points and mesh first");
    if (!lams || n_lams < 1 || num_iters < 0) return fail(ctx, NW_ERR_BADARG, "nw_search: need at least one lambda and num_iters >= 0");
    if (ctx->in_search) return fail(ctx, NW_ERR_BADARG, "nw_search_begin: previous search not ended");
    NW_HIP(hipSetDevice(ctx->device));
    if (ctx->have_boundary && ctx->have_halo_d0 && ctx->halo_d0_n != ctx->N)
        return fail(ctx, NW_ERR_BADARG, "nw_search: the distances given to nw_halo_set_reference are not this rank's localizations' (length)");
    NW_TRY(alloc_work(ctx));
    if (ctx->proj_ready && !ctx->proj_sorted && num_iters > 0 && !getenv("NW_NO_PROJ_SORT")) NW_TRY(resort_by_projection(ctx));
    NW_TRY(ensure_grid(ctx));
    if (ctx->blocks_done >= 2 && num_iters > 0) NW_TRY(tune_grid(ctx));      // (nw_optimize_layout does it earlier if the caller asks)
    if (ctx->blocks_done >= 3 && num_iters > 0) NW_TRY(order_items_by_cost(ctx));
    // fixed-point quanta of the scatter (k_attract): 2^-36 of a bound on |w res| <= largest weight x scene extent; 2^-40 for sum w
    ctx->local_quantum = std::ldexp(1.0, (int)std::ceil(std::log2(std::max(ctx->scene_ext * ctx->w_bound, 1e-300))) - 36);
    ctx->acc_quantum = ctx->quantum_override > 0 ? ctx->quantum_override : ctx->local_quantum;
    ctx->w_quantum = std::ldexp(1.0, -40);
    ctx->lam0 = lams[0];
    ctx->full_staged = false;
    ctx->comm_mode = flags & (NW_FLAG_COMM_TILES | NW_FLAG_COMM_REPLICATED | NW_FLAG_COMM_HALO);
    ctx->search_flags = flags & ~(NW_FLAG_RESULT_TO_HOST | NW_FLAG_COMM_TILES | NW_FLAG_COMM_REPLICATED | NW_FLAG_COMM_HALO | NW_FLAG_ROWS_ASYNC);
    ctx->rows_async = (flags & NW_FLAG_ROWS_ASYNC) != 0 && ctx->wb_rows != nullptr;
    ctx->direct_out = false;
    if ((flags & NW_FLAG_RESULT_TO_HOST) && num_iters > 0 && 3 * ctx->M * sizeof(float) <= (4u << 20) && !(getenv("NW_DIRECT_OUT") && atoi(getenv("NW_DIRECT_OUT")) == 0)) {
        NW_TRY(ensure_staging(ctx));                // (larger results: the sliced copy of nw_search_end is the faster one)
        ctx->direct_out = true;
    }
    ctx->search_iters = num_iters;
    ctx->search_done = 0;
    NW_HIP(ctx->logs.ensure((size_t)std::max(num_iters, 1)));
    hipLaunchKernelGGL(k_set_iter_base, dim3(1), dim3(1), 0, ctx->stream, ctx->state.p, ctx->global_iter);
    if (ctx->profiling) { ctx->ev_used = 0; g_marks.spans.clear(); for (int k = 0; k < ST_COUNT; ++k) { ctx->stage_ms[k] = 0; ctx->stage_launches[k] = 0; } }
    ctx->in_search = true;
    ctx->begin_ops_pending = true;         // the stream operations of the block's start are issued with its first iteration (or captured)
    return NW_OK;
}

// the attraction step's arguments (pointers, weights mode, quanta of this block)
static NwAttractArgs attract_args(const nw_ctx *ctx)
{
    NwAttractArgs a;
    memset(&a, 0, sizeof(a));
    a.F = (int)ctx->F; a.sinv_scalar = ctx->sinv_scalar; a.w_scalar = ctx->w_scalar;
    a.cent_by_face = ctx->cent_tmp.p; a.faces = ctx->faces.p; a.pos = ctx->pos.p;
    a.sinv = ctx->sinv_array ? ctx->sinv.p : nullptr; a.wnorm = ctx->w_array ? ctx->wnorm.p : nullptr; a.dat = ctx->have_data ? ctx->dat.p : nullptr;
    a.mask = ctx->mask.p; a.dist = ctx->dist.p; a.wout = ctx->w.p; a.res = ctx->res.p; a.vidx = ctx->vidx.p; a.vacc = ctx->vacc.p; a.part = ctx->part_a.p;
    a.inv_q = 1.0 / ctx->acc_quantum; a.inv_qw = 1.0 / ctx->w_quantum;
    a.d0 = (ctx->have_boundary && ctx->have_halo_d0) ? ctx->halo_d0.p : nullptr; a.perm = ctx->perm.p;
    return a;
}

// stream operations at the start of a block (capturable: no host synchronisation)
static int enqueue_begin_ops(nw_ctx *ctx)
{
    ctx->begin_ops_pending = false;
    const int n = std::max(ctx->search_iters, 1);
    // one launch: f restarts from the mesh positions, logs cleared, and the reference's zero-initialised arrays zeroed where the block
    // would not overwrite them anyway (k_block_begin)
    const int flag = (ctx->search_iters <= 0 ? 1 : 0) | ((ctx->search_flags & NW_FLAG_NO_LAST_STEP) ? 2 : 0);
    const int64_t work = std::max<int64_t>(9 * ctx->M, flag & 1 ? 3 * ctx->N : 0);
    const int blocks = (int)std::min<int64_t>(4096, std::max<int64_t>(1, (work + NW_BLOCK - 1) / NW_BLOCK));
    hipLaunchKernelGGL(k_block_begin, dim3(blocks), dim3(NW_BLOCK), 0, ctx->stream, (int)ctx->M, (int64_t)(3 * ctx->N), ctx->pos.p, ctx->meshpos.p, ctx->S.p, ctx->res.p,
                       ctx->vacc.p, (unsigned *)ctx->logs.p, (int)((size_t)n * sizeof(NwIterLogDev) / 4), ctx->state.p, flag, ctx->vacc_dirty ? 1 : 0);
    NW_HIP(hipGetLastError());
    if (!ctx->capturing) ctx->vacc_dirty = false;          // (a captured launch bakes the flag in: the graph's key carries it)
    return NW_OK;
}

// centroid binning + exact nearest-face query + float64 fix-up of the ambiguous points, for the current positions
// (`parts`: 1 grid build, 2 the query kernel, 4 fix-up -- a block captured for sampled profiling launches part 2 of its first
// iteration outside the graphs, between two events)
// the query kernel resolves its ambiguous localizations itself (default); NW_FUSE_FIXUP=0 brings back the separate fix-up launch
static bool fuse_fixup() { static const bool on = !(getenv("NW_FUSE_FIXUP") && atoi(getenv("NW_FUSE_FIXUP")) == 0); return on; }

// the ring half of the curvature prior rides in the query launch (workgroups appended to its grid: they run in its drain); NW_RING_IN_NN=0
// (developer knob) makes it a launch of its own in front of k_prior_directions
static bool ring_in_nn() { static const bool on = !(getenv("NW_RING_IN_NN") && atoi(getenv("NW_RING_IN_NN")) == 0); return on; }
static NwRingArgs ring_args(const nw_ctx *ctx)
{
    NwRingArgs r;
    r.M = (int)ctx->M; r.maxdeg = ctx->maxdeg; r.nbr_t = ctx->nbr_t.p; r.meshpos = ctx->meshpos.p; r.nrm = ctx->nrm.p; r.ring4 = ctx->ring4.p; r.ring_a = ctx->ring_a.p;
    return r;
}

// the attraction step rides in the query launch too (one workgroup per workgroup of the query, appended behind the ring workgroups: they fill
// the launch's drain); NW_ATTRACT_IN_NN=0 (developer knob): k_attract as a launch of its own behind the query
static bool attract_in_nn() { static const bool on = !(getenv("NW_ATTRACT_IN_NN") && atoi(getenv("NW_ATTRACT_IN_NN")) == 0); return on; }
static int launch_query(nw_ctx *ctx, int it, int parts, bool with_ring, bool with_attract, bool *attract_rode)
{
    if (attract_rode) *attract_rode = false;
    const int64_t F = ctx->F;
    const NwGrid g = ctx->grid;
    if (parts & QP_GRID) {
        StageScope s(ctx, ST_GRID);
        static const bool tile_fuse = !(getenv("NW_TILE_FUSE") && atoi(getenv("NW_TILE_FUSE")) == 0);      // developer knob: 0 = the scan's own first pass
        int *tiles = tile_fuse ? ctx->ctile.p : nullptr;
        hipLaunchKernelGGL(k_face_centroids, dim3(nblk(F)), dim3(NW_BLOCK), 0, ctx->stream, g, ctx->pos.p, ctx->faces.p, (int)F,
                           ctx->cent_tmp.p, ctx->fcell.p, ctx->frank.p, ctx->ccount.p, tiles, ctx->ambig_count.p, ctx->state.p, it, ctx->query_serial.p);
        NW_TRY(scan_exclusive(ctx, ctx->ccount.p, g.ncell, ctx->cstart.p, true, tiles));      // also re-zeroes the histogram
        hipLaunchKernelGGL(k_centroid_scatter, dim3(nblk(F)), dim3(NW_BLOCK), 0, ctx->stream, (int)F, ctx->cent_tmp.p, ctx->fcell.p, ctx->frank.p, ctx->cstart.p,
                           ctx->cent.p, ctx->state.p, it, ctx->ctile.p, (g.ncell + NW_SCAN_TILE - 1) / NW_SCAN_TILE);
    }
    if (parts & QP_NN) {
        StageScope s(ctx, ST_NN, it == ctx->search_iters - 1);      // level 4 samples the block's LAST iteration (the one launched from the host)
        static const int nn_map = getenv("NW_NN_MAP") ? (atoi(getenv("NW_NN_MAP")) == 0 ? 0 : (atoi(getenv("NW_NN_MAP")) == 1 ? 2 : 4)) : 4;   // 0 slabs, 1 round-robin, 2 interleaved runs (default)
        static const bool no_outliers = getenv("NW_NO_OUTLIERS") != nullptr;      // developer knob
        static const bool item_times = getenv("NW_ITEM_TIMES") != nullptr;        // developer aid: nw_debug_items also returns when every item started
        static const float outl_f = getenv("NW_OUTL_F") ? (float)atof(getenv("NW_OUTL_F")) : 6.25f;          // developer knobs: which lanes a wave sets aside (nw_nn.h "Outliers")
        static const int outl_max = getenv("NW_OUTL_MAX") ? atoi(getenv("NW_OUTL_MAX")) : NW_OUTLIERS;
        static const int tb = getenv("NW_NN_BLOCK") ? std::max(64, std::min(256, atoi(getenv("NW_NN_BLOCK")) & ~63)) : 128;
        const int wpb = tb / 64, nb = (ctx->nitems + wpb - 1) / wpb;   // one wave = one work item
        const int nbp = nn_map == 4 ? (8 * NW_XCD_RUN) * ((nb + 8 * NW_XCD_RUN - 1) / (8 * NW_XCD_RUN)) : 8 * ((nb + 7) / 8);
        NwRingArgs R = ring_args(ctx);
        const bool ring_here = with_ring && ring_in_nn();
        if (!ring_here) R.M = 0;
        const int nbq = nbp;                                          // workgroups of the query proper; behind them one thread per vertex for the ring half
        const int nbr = ring_here ? 8 * (int)(((ctx->M + tb - 1) / tb + 7) / 8) : 0;      // (a multiple of 8: the attraction workgroups behind them keep the query's XCD mapping)
        const bool attract_here = with_attract && attract_in_nn() && fuse_fixup() && ctx->item_done.p != nullptr && tb <= 128 && !ctx->handoff_off;
        const int nbt = nbq + nbr + (attract_here ? nbq : 0);
        const NwAttractArgs AA = attract_args(ctx);
        int *done_p = attract_here ? ctx->item_done.p : nullptr;
        if (attract_here) { ctx->attract_rows = nbq; if (attract_rode) *attract_rode = true; }
        if (ctx->nn_stats.p) {
            hipLaunchKernelGGL(k_nn_wave<true>, dim3(nbt), dim3(tb), 0, ctx->stream, g, ctx->items.p, ctx->nitems, ctx->pts.p, ctx->cstart.p, ctx->cent.p,
                           ctx->cent_tmp.p, (int)F, ctx->face.p, (ctx->face_warm ? 1 : 0) | nn_map | (fuse_fixup() ? 8 : 0) | (no_outliers ? 16 : 0) | (item_times ? 64 : 0), ctx->ambig_list.p, ctx->ambig_count.p,
                           ctx->state.p, it, ctx->nn_stats.p, (ctx->items_by_cost && !item_times) ? nullptr : ctx->item_cost.p, ctx->face_sorted ? ctx->face_orig.p : nullptr, outl_f, outl_max, R, nbq, nbr, AA, done_p, ctx->query_serial.p);
        } else {
            hipLaunchKernelGGL(k_nn_wave<false>, dim3(nbt), dim3(tb), 0, ctx->stream, g, ctx->items.p, ctx->nitems, ctx->pts.p, ctx->cstart.p, ctx->cent.p,
                           ctx->cent_tmp.p, (int)F, ctx->face.p, (ctx->face_warm ? 1 : 0) | nn_map | (fuse_fixup() ? 8 : 0) | (no_outliers ? 16 : 0) | (item_times ? 64 : 0), ctx->ambig_list.p, ctx->ambig_count.p,
                           ctx->state.p, it, ctx->nn_stats.p, (ctx->items_by_cost && !item_times) ? nullptr : ctx->item_cost.p, ctx->face_sorted ? ctx->face_orig.p : nullptr, outl_f, outl_max, R, nbq, nbr, AA, done_p, ctx->query_serial.p);
        }
        if (ctx->face_warm && !ctx->items_by_cost) ctx->item_cost_valid = true;      // (a cold query's costs say little about the warm ones)
        ctx->face_warm = true;
    }
    if ((parts & QP_FIXUP) && !fuse_fixup()) {
        StageScope s(ctx, ST_FIXUP);
        static const int fb = getenv("NW_FIXUP_BLOCKS") ? std::max(64, atoi(getenv("NW_FIXUP_BLOCKS"))) : 2048;      // one wave per ambiguous localization: ~7000 of them at 10^6, 4 waves per workgroup
        hipLaunchKernelGGL(k_nn_fixup, dim3(fb), dim3(NW_BLOCK), 0, ctx->stream, g, ctx->ambig_list.p, ctx->ambig_count.p, ctx->pts.p, ctx->cstart.p, ctx->cent.p,
                           ctx->cent_tmp.p, ctx->face.p, ctx->state.p, it, ctx->face_sorted ? ctx->face_orig.p : nullptr);
    }
    NW_HIP(hipGetLastError());
    return NW_OK;
}

static int iter_attract_parts(nw_ctx *ctx, int parts)
{
    if (ctx->begin_ops_pending) NW_TRY(enqueue_begin_ops(ctx));
    const int it = ctx->search_done;
    const int64_t N = ctx->N;
    // (only behind a WARM query: a cold one -- the first of a topology -- has waves that walk the whole grid for a far localization and
    // take milliseconds; its attraction step stays a launch of its own)
    const bool attract_rides = (parts & QP_NN) && (parts & QP_ATTRACT) && attract_in_nn() && fuse_fixup() && ctx->item_done.p != nullptr && ctx->face_warm && !ctx->handoff_off;
    bool rode = false;
    if (parts & (QP_GRID | QP_NN | QP_FIXUP)) NW_TRY(launch_query(ctx, it, parts & (QP_GRID | QP_NN | QP_FIXUP), true, attract_rides, &rode));
    if ((parts & QP_ATTRACT) && !rode) {
        StageScope s(ctx, ST_ATTRACT);
        hipLaunchKernelGGL(k_attract, dim3(attract_blocks(ctx)), dim3(NW_BLOCK), 0, ctx->stream, (int)N, ctx->pts.p, ctx->face.p, attract_args(ctx), ctx->state.p, it);
        ctx->attract_rows = attract_blocks(ctx);
    }
    NW_HIP(hipGetLastError());
    return NW_OK;
}

NW_EXPORT int nw_iter_attract(nw_ctx *ctx)
{
    if (!ctx || !ctx->in_search) return fail(ctx, NW_ERR_BADARG, "nw_iter_attract outside a search");
    if (ctx->have_boundary && ctx->pos_unpack_pending) { ctx->pos_unpack_pending = false; NW_TRY(halo_unpack(ctx, NW_ARR_POS)); }      // the owners' rows of the previous update
    NW_TRY(iter_attract_parts(ctx, QP_ALL));
    if (ctx->have_boundary) NW_TRY(halo_pack(ctx, NW_ARR_VACC));
    return NW_OK;
}

NW_EXPORT int nw_iter_directions(nw_ctx *ctx)
{
    if (!ctx || !ctx->in_search) return fail(ctx, NW_ERR_BADARG, "nw_iter_directions outside a search");
    if (ctx->have_boundary) NW_TRY(halo_unpack(ctx, NW_ARR_VACC));      // the boundary rows summed over the ranks
    const int it = ctx->search_done;
    const int n_search = (ctx->search_done == 0 || (ctx->search_flags & NW_FLAG_NO_LAST_STEP)) ? 2 : 3;
    {
        StageScope s(ctx, ST_PRIOR);
        const bool wfunc = (ctx->search_flags & NW_FLAG_WFUNC) != 0;
        if (wfunc)
            hipLaunchKernelGGL(k_vertex_area_weights, dim3(nblk(ctx->M)), dim3(NW_BLOCK), 0, ctx->stream, (int)ctx->M, ctx->NB, ctx->nbr.p, ctx->pos.p, ctx->wv.p, ctx->state.p, it);
        if (!ring_in_nn()) hipLaunchKernelGGL(k_prior_ring, dim3(nblk(ctx->M)), dim3(NW_BLOCK), 0, ctx->stream, ring_args(ctx), ctx->state.p, it);
        hipLaunchKernelGGL(k_prior_directions, dim3(prior_blocks(ctx)), dim3(NW_BLOCK), 0, ctx->stream, (int)ctx->M, ctx->ring4.p, ctx->ring_a.p, ctx->pos.p,
                           ctx->meshpos.p, ctx->nrm.p, ctx->vacc.p, ctx->S.p, ctx->fdef.p, ctx->pi.p, ctx->part_p.p, ctx->state.p, it, n_search,
                           ctx->acc_quantum, ctx->w_quantum, wfunc ? ctx->wv.p : nullptr, ctx->have_owned ? ctx->owned.p : nullptr);
    }
    {
        StageScope s(ctx, ST_AS);
        hipLaunchKernelGGL(k_subspace_point_sums, dim3(subspace_blocks(ctx)), dim3(NW_BLOCK), 0, ctx->stream, (int)ctx->N, (int)ctx->M, ctx->vidx.p, ctx->w.p, ctx->res.p,
                           ctx->mask.p, ctx->S.p, ctx->part_s.p, ctx->state.p, it, n_search);
        // the 28 sums of this iteration, added in a fixed order (deterministic); multi-GPU runs all-reduce them after this call
        hipLaunchKernelGGL(k_reduce_scalars, dim3(3 * NW_SPARTS), dim3(NW_BLOCK), 0, ctx->stream, ctx->part_a.p, ctx->attract_rows, ctx->part_p.p, prior_blocks(ctx),
                           ctx->part_s.p, subspace_blocks(ctx), ctx->scalars.p, ctx->state.p, it);
    }
    NW_HIP(hipGetLastError());
    return NW_OK;
}

NW_EXPORT int nw_iter_update(nw_ctx *ctx)
{
    if (!ctx || !ctx->in_search) return fail(ctx, NW_ERR_BADARG, "nw_iter_update outside a search");
    if (ctx->search_done >= ctx->search_iters) return fail(ctx, NW_ERR_BADARG, "nw_iter_update: more iterations than announced");
    const int it = ctx->search_done;
    const int n_search = (ctx->search_done == 0 || (ctx->search_flags & NW_FLAG_NO_LAST_STEP)) ? 2 : 3;
    {
        StageScope s(ctx, ST_UPDATE);
        hipLaunchKernelGGL(k_solve_update, dim3(std::min(nblk(ctx->M), 512)), dim3(NW_BLOCK), 0, ctx->stream, (int)ctx->M, ctx->lam0, n_search, ctx->search_flags,
                           ctx->have_valid ? ctx->valid.p : nullptr, ctx->pos.p, ctx->meshpos.p, ctx->S.p, ctx->vacc.p, ctx->scalars.p, ctx->state.p,
                           ctx->logs.p + ctx->search_done, it, (ctx->direct_out && it == ctx->search_iters - 1) ? (float *)ctx->pin : nullptr);
    }
    NW_HIP(hipGetLastError());
    if (ctx->have_boundary) { NW_TRY(halo_pack(ctx, NW_ARR_POS)); ctx->pos_unpack_pending = true; }
    ctx->global_iter += 1;
    ctx->search_done += 1;
    return NW_OK;
}

static int write_back_impl(nw_ctx *ctx, float *contiguous, void *rows, int64_t row_stride_bytes);
static int ensure_staging(nw_ctx *ctx);
static void copy_out_staged(nw_ctx *ctx, float *contiguous, void *rows, int64_t row_stride_bytes);
static bool copy_out_chunks(nw_ctx *ctx, float *contiguous, void *rows, int64_t row_stride_bytes, int64_t slice_rows, int flag_base);

// pinned landing zone of a block's logs (n records) + the device state behind them
static int ensure_pin_log(nw_ctx *ctx, int n_records)
{
    const size_t log_bytes = (size_t)std::max(n_records, 1) * sizeof(nw_iter_log);
    if (ctx->pin_log_bytes < log_bytes + sizeof(NwDevState)) {
        if (ctx->pin_log) (void)hipHostFree(ctx->pin_log);
        ctx->pin_log = nullptr; ctx->pin_log_bytes = 0;
        NW_HIP(hipHostMalloc(&ctx->pin_log, 2 * log_bytes + sizeof(NwDevState), hipHostMallocDefault));
        ctx->pin_log_bytes = 2 * log_bytes + sizeof(NwDevState);
    }
    return NW_OK;
}

// The end of a block as a word in pinned memory (k_block_done): queued behind the block's last kernel by nw_search when the result is
// staged by that kernel.  NW_SPIN_WAIT=0 (developer knob): the blocking hipStreamSynchronize of round 4.
static bool spin_wait_on() { static const bool on = !(getenv("NW_SPIN_WAIT") && atoi(getenv("NW_SPIN_WAIT")) == 0); return on; }
static int enqueue_block_done(nw_ctx *ctx)
{
    ctx->done_launched = false;
    if (!spin_wait_on() || ctx->search_done <= 0) return NW_OK;
    NW_TRY(ensure_pin_log(ctx, ctx->search_iters));
    if (!ctx->pin_flag) {
        NW_HIP(hipHostMalloc((void **)&ctx->pin_flag, 64, hipHostMallocDefault));
        ctx->pin_flag[0] = 0;
        NW_HIP(ctx->done_count.ensure(1));
        NW_HIP(hipMemsetAsync(ctx->done_count.p, 0, sizeof(int), ctx->stream));
        ctx->done_expected = 0;
    }
    NwDevState *stp = (NwDevState *)((char *)ctx->pin_log + ctx->pin_log_bytes - sizeof(NwDevState));
    hipLaunchKernelGGL(k_block_done, dim3(1), dim3(NW_BLOCK), 0, ctx->stream, (const unsigned *)ctx->logs.p, (int)((size_t)ctx->search_done * sizeof(nw_iter_log) / 4),
                       ctx->state.p, (unsigned *)ctx->pin_log, (unsigned *)stp, ctx->done_count.p, ctx->pin_flag);
    NW_HIP(hipGetLastError());
    ctx->done_expected += 1;
    ctx->done_launched = true;
    return NW_OK;
}

// a result above the direct-output limit: slices of the estimate into the staging buffer behind the block's last kernel, each launch
// announcing the one before it (k_copy_slice), k_block_done behind the last
static int enqueue_sliced_result(nw_ctx *ctx)
{
    ctx->sliced_S = 0;
    if (!spin_wait_on() || ctx->search_done <= 0) return NW_OK;
    static const int64_t rows_per_slice = getenv("NW_SLICE_ROWS") ? std::max<int64_t>(4096, atoll(getenv("NW_SLICE_ROWS")) & ~(int64_t)3) : 65536;
    NW_TRY(ensure_staging(ctx));
    NW_TRY(enqueue_block_done(ctx));            // (allocates flag and counter; its launch is re-issued behind the slices below)
    if (!ctx->done_launched) return NW_OK;
    // that launch went first: it only copied logs that are complete anyway and took one flag value; the slices follow, then the real end
    const int base = ctx->done_expected;
    const int S = (int)((ctx->M + rows_per_slice - 1) / rows_per_slice);
    for (int sI = 0; sI < S; ++sI) {
        const int64_t v0 = (int64_t)sI * rows_per_slice, v1 = std::min<int64_t>(ctx->M, v0 + rows_per_slice);
        const int64_t nf = 3 * (v1 - v0);
        const int blocks = (int)std::min<int64_t>(512, std::max<int64_t>(1, (nf / 4 + NW_BLOCK - 1) / NW_BLOCK));
        hipLaunchKernelGGL(k_copy_slice, dim3(blocks), dim3(NW_BLOCK), 0, ctx->stream, ctx->pos.p + 3 * v0, (float *)ctx->pin + 3 * v0, nf, ctx->done_count.p, ctx->pin_flag);
    }
    NW_HIP(hipGetLastError());
    ctx->done_expected += S;
    NW_TRY(enqueue_block_done(ctx));
    ctx->sliced_S = S; ctx->sliced_base = base; ctx->sliced_rows = rows_per_slice;
    return NW_OK;
}

// wait for the flag (true), or give up (false: the caller synchronises the stream the ordinary way).  The calling thread spins on the
// pinned word; once the block is within 250 us of its expected end the copy threads are woken so that they spin too (NwHostPool::arm).
static bool wait_block_done(nw_ctx *ctx, int want)
{
    if (!ctx->done_launched) return false;
    volatile int *flag = ctx->pin_flag;
    const auto t0 = ctx->block_t0;
    const double expect_us = ctx->block_us_ema;
    bool armed = false;
    long spins = 0;
    for (;;) {
        if (*flag - want >= 0) break;
        for (int k = 0; k < 8; ++k) __builtin_ia32_pause();
        if ((++spins & 63) == 0) {
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (!armed && ctx->pool && expect_us > 0 && us > expect_us - 250.0) { ctx->pool->arm(); armed = true; }
            if (us > 2000.0 + 20.0 * expect_us) {                      // far beyond anything a healthy block takes: let the runtime say what happened
                if (hipStreamQuery(ctx->stream) != hipErrorNotReady) { (void)hipGetLastError(); return *flag - want >= 0; }
                if (us > 30e6) return false;
            }
        }
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    ctx->block_us_ema = ctx->block_us_ema > 0 ? 0.7 * ctx->block_us_ema + 0.3 * us : us;
    return true;
}

NW_EXPORT int nw_search_end(nw_ctx *ctx, float *pos_out, nw_iter_log *log, int *loopcount)
{
    if (!ctx || !ctx->in_search) return fail(ctx, NW_ERR_BADARG, "nw_search_end outside a search");
    if (ctx->begin_ops_pending) NW_TRY(enqueue_begin_ops(ctx));
    if (ctx->have_boundary && ctx->pos_unpack_pending) { ctx->pos_unpack_pending = false; NW_TRY(halo_unpack(ctx, NW_ARR_POS)); }
    ctx->in_search = false;
    // logs + device state land in PINNED memory: a device-to-host copy into pageable memory blocks the host until everything queued
    // before it has run, which would serialise "wait for the kernels", the two small copies and the position slices below
    NW_TRY(ensure_pin_log(ctx, std::max(ctx->search_done, ctx->search_iters)));
    nw_iter_log *host = (nw_iter_log *)ctx->pin_log;
    NwDevState *stp = (NwDevState *)((char *)ctx->pin_log + ctx->pin_log_bytes - sizeof(NwDevState));
    const bool by_flag = ctx->done_launched;                 // k_block_done brings logs and state along, and says when
    if (!by_flag) {
        if (ctx->search_done > 0)
            NW_HIP(hipMemcpyAsync(host, ctx->logs.p, (size_t)ctx->search_done * sizeof(nw_iter_log), hipMemcpyDeviceToHost, ctx->stream));
        NW_HIP(hipMemcpyAsync(stp, ctx->state.p, sizeof(NwDevState), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (pos_out) {
        // the positions come back in slices that host threads copy out (contiguous result + the registered strided vertex records)
        // while the later slices are still in flight; a device pointer gets a plain copy
        hipPointerAttribute_t attr;
        const bool on_device = hipPointerGetAttributes(&attr, pos_out) == hipSuccess && attr.type == hipMemoryTypeDevice;
        (void)hipGetLastError();
        if (on_device) NW_HIP(hipMemcpyAsync(pos_out, ctx->pos.p, 3 * ctx->M * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        else if (ctx->direct_out) {
            // the block's last update wrote the result into the pinned staging buffer itself: wait for the stream, copy out.  (If the
            // device-side stop condition ended the block early that kernel did not run: the ordinary copy below takes over.)
            const auto tw0 = std::chrono::steady_clock::now();
            if (!(by_flag && wait_block_done(ctx, ctx->done_expected))) {
                if (by_flag) {          // (the flag did not come: the copies it would have made, the ordinary way)
                    if (ctx->search_done > 0) NW_HIP(hipMemcpyAsync(host, ctx->logs.p, (size_t)ctx->search_done * sizeof(nw_iter_log), hipMemcpyDeviceToHost, ctx->stream));
                    NW_HIP(hipMemcpyAsync(stp, ctx->state.p, sizeof(NwDevState), hipMemcpyDeviceToHost, ctx->stream));
                }
                NW_HIP(hipStreamSynchronize(ctx->stream));
            }
            const bool last_ran = ctx->search_done > 0 && ctx->search_done == ctx->search_iters && host[ctx->search_done - 1].executed;
            const auto tcp0 = std::chrono::steady_clock::now();
            if (last_ran) {
                copy_out_staged(ctx, pos_out, ctx->wb_rows, ctx->wb_stride);
                if (getenv("NW_VERBOSE") && atoi(getenv("NW_VERBOSE")) >= 3)
                    fprintf(stderr, "[nanowrap] search_end: copy-out of the staged result alone %ld us\n", (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tcp0).count());
            } else {
                NW_TRY(write_back_impl(ctx, pos_out, ctx->wb_rows, ctx->wb_stride));
            }
            if (getenv("NW_VERBOSE") && atoi(getenv("NW_VERBOSE")) >= 3)
                fprintf(stderr, "[nanowrap] search_end: wait + copy-out of the staged result %ld us\n", (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tw0).count());
        } else if (by_flag && ctx->sliced_S > 0 && wait_block_done(ctx, ctx->sliced_base + 1)) {
            // the block's kernels have run and the slices are landing in the staging buffer: the copy threads take them as they come
            const auto tw0 = std::chrono::steady_clock::now();
            ctx->n_write_backs += 1;
            if (!copy_out_chunks(ctx, pos_out, ctx->wb_rows, ctx->wb_stride, ctx->sliced_rows, ctx->sliced_base)) {
                NW_HIP(hipStreamSynchronize(ctx->stream));           // (says what happened to the device, if anything)
                return fail(ctx, NW_ERR_HIP, "the block's result did not arrive in the staging buffer");
            }
            if (getenv("NW_VERBOSE") && atoi(getenv("NW_VERBOSE")) >= 3)
                fprintf(stderr, "[nanowrap] search_end: %d slices through the flag word, copy-out %ld us after the last kernel\n", ctx->sliced_S, (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tw0).count());
        } else {
            const auto tw0 = std::chrono::steady_clock::now();
            if (by_flag) {          // (the flag did not come: the copies it would have made, the ordinary way)
                if (ctx->search_done > 0) NW_HIP(hipMemcpyAsync(host, ctx->logs.p, (size_t)ctx->search_done * sizeof(nw_iter_log), hipMemcpyDeviceToHost, ctx->stream));
                NW_HIP(hipMemcpyAsync(stp, ctx->state.p, sizeof(NwDevState), hipMemcpyDeviceToHost, ctx->stream));
            }
            NW_TRY(write_back_impl(ctx, pos_out, ctx->wb_rows, ctx->wb_stride));
            if (getenv("NW_VERBOSE") && atoi(getenv("NW_VERBOSE")) >= 3)
                fprintf(stderr, "[nanowrap] search_end: wait + sliced write-back %ld us\n", (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tw0).count());
        }
    }
    const auto ts0 = std::chrono::steady_clock::now();
    ctx->direct_out = false;
    ctx->done_launched = false; ctx->sliced_S = 0;
    NW_HIP(hipStreamSynchronize(ctx->stream));
    if (getenv("NW_VERBOSE") && atoi(getenv("NW_VERBOSE")) >= 3)
        fprintf(stderr, "[nanowrap] search_end: final synchronize %ld us\n", (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - ts0).count());
    const NwDevState st = *stp;
    int executed = 0;
    for (int i = 0; i < ctx->search_done; ++i) executed += host[i].executed ? 1 : 0;
    if (loopcount) *loopcount = executed;
    if (log) for (int i = 0; i < ctx->search_done; ++i) log[i] = host[i];
    if (executed > 0) ctx->last_mean_dist = host[executed - 1].mean_dist;
    ctx->searched = ctx->searched || executed > 0;
    if (executed > 0) ctx->blocks_done += 1;
    if (st.grid_escape) ctx->grid_valid = false;            // the mesh left the cell grid during this block: lay a new one before the next
    if (executed > 0 && !ctx->proj_sorted && !ctx->proj_ready && ctx->face_warm) {
        // foot points of this block's last query -> keys of the second sort (applied at the start of the next block)
        const NwGrid &g = ctx->grid;
        const double ext = std::max({(double)g.gx, (double)g.gy, (double)g.gz}) * g.h;
        ctx->proj_unit = (float)(ext / 1024.0);
        NW_HIP(ctx->proj_key.ensure(ctx->N)); NW_HIP(ctx->proj_idx.ensure(ctx->N));
        hipLaunchKernelGGL(k_projection_keys, dim3(nblk(ctx->N)), dim3(NW_BLOCK), 0, ctx->stream, (int)ctx->N, ctx->face.p, (int)ctx->F, ctx->cent_tmp.p, ctx->pts.p,
                           g.ox, g.oy, g.oz, 1.0f / ctx->proj_unit, ctx->proj_key.p, ctx->proj_idx.p);
        NW_HIP(hipGetLastError());
        NW_HIP(hipStreamSynchronize(ctx->stream));
        ctx->proj_ready = true;
    }
    if (ctx->profiling) {
        for (auto &sp : g_marks.spans) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, sp.second.first, sp.second.second) == hipSuccess) { ctx->stage_ms[sp.first] += ms; ctx->stage_ms[ST_TOTAL] += ms; }
        }
        g_marks.spans.clear();
    }
    if (st.status != 0) {
        // clear the sticky device status so the ctx stays usable; report it once
        const int code = st.status;
        NwDevState st2 = st; st2.status = 0;
        st2.stop_at = 0x7fffffff;                           // (the failure cancelled the rest of its block, not the searches that follow)
        (void)hipMemcpy(ctx->state.p, &st2, sizeof(st2), hipMemcpyHostToDevice);
        if (code == NW_ERR_NAN) return fail(ctx, NW_ERR_NAN, "NaN detected in weight matrix / A f / A^T r (reference asserts at mesh_conj_grad.py:514,548,580)");
        if (code == NW_ERR_SINGULAR) return fail(ctx, NW_ERR_SINGULAR, "singular subspace normal equations (numpy.linalg.solve would raise LinAlgError)");
        if (code == NW_ERR_INTERNAL) return fail(ctx, NW_ERR_INTERNAL, "internal error: the nearest-face query produced an invalid face id");
        if (code == NW_ERR_HANDOFF) {
            ctx->handoff_off = true;                          // (from now on k_attract is a launch of its own behind the query)
            for (auto &gph : ctx->graphs) if (gph.exec) { (void)hipGraphExecDestroy(gph.exec); gph.exec = nullptr; }
            return fail(ctx, NW_ERR_HANDOFF, "the attraction step inside the query launch gave up waiting for its work item (200 ms); the ctx now runs the two as separate launches: run the block again");
        }
        return fail(ctx, code, "device-side error");
    }
    return NW_OK;
}

// everything a captured block bakes into its launches
static uint64_t block_graph_key(const nw_ctx *ctx)
{
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h ^= v; h *= 1099511628211ull; };
    auto mixp = [&](const void *p) { mix((uint64_t)(uintptr_t)p); };
    mix((uint64_t)ctx->search_iters); mix(ctx->search_flags); mix(ctx->face_warm ? 1 : 0); mix(ctx->grid_generation); mix((uint64_t)ctx->nitems);
    mix((uint64_t)ctx->N); mix((uint64_t)ctx->M); mix((uint64_t)ctx->F); mix((uint64_t)ctx->NB); mix((uint64_t)ctx->maxdeg);
    uint32_t lb; memcpy(&lb, &ctx->lam0, 4); mix(lb);
    uint64_t qb; memcpy(&qb, &ctx->acc_quantum, 8); mix(qb);
    uint32_t sb; memcpy(&sb, &ctx->sinv_scalar, 4); mix(sb); memcpy(&sb, &ctx->w_scalar, 4); mix(sb);
    mix((ctx->sinv_array ? 1 : 0) | (ctx->w_array ? 2 : 0) | (ctx->have_valid ? 4 : 0) | (ctx->have_owned ? 8 : 0) | (ctx->nn_stats.p ? 16 : 0) | (ctx->profiling == 4 ? 256 : 0) | (ctx->direct_out ? 64 : 0) | (ctx->have_data ? 128 : 0) | (ctx->have_boundary ? 512 : 0) | (ctx->vacc_dirty ? 1024 : 0) | (ctx->handoff_off ? 2048 : 0));
    mixp(ctx->direct_out ? ctx->pin : nullptr);
    mixp(ctx->have_data ? ctx->dat.p : nullptr);
    const void *ptrs[] = {ctx->pts.p, ctx->sinv.p, ctx->wnorm.p, ctx->mask.p, ctx->items.p, ctx->ccount.p, ctx->cstart.p, ctx->scan_tmp.p, ctx->ctile.p, ctx->pos.p, ctx->meshpos.p, ctx->nrm.p,
                          ctx->nbr.p, ctx->nbr_t.p, ctx->faces.p, ctx->valid.p, ctx->owned.p, ctx->cent_tmp.p, ctx->cent.p, ctx->fcell.p, ctx->frank.p, ctx->face.p, ctx->vidx.p,
                          ctx->ambig_list.p, ctx->ambig_count.p, ctx->dist.p, ctx->w.p, ctx->res.p, ctx->S.p, ctx->fdef.p, ctx->pi.p, ctx->vacc.p, ctx->scalars.p, ctx->part_a.p,
                          ctx->part_p.p, ctx->part_s.p, ctx->wv.p, ctx->state.p, ctx->logs.p, ctx->nn_stats.p, ctx->ring4.p, ctx->ring_a.p, ctx->item_done.p, ctx->query_serial.p,
                          ctx->hb_local.p, ctx->hb_slot.p, ctx->hb_slot2local.p, ctx->halo_acc.p, ctx->halo_rows.p, ctx->face_sorted ? ctx->face_orig.p : nullptr,
                          ctx->have_peers ? ctx->px_ghost.p : nullptr, ctx->have_peers ? ctx->px_owned.p : nullptr, ctx->have_peers ? ctx->px_send.p : nullptr,
                          ctx->have_peers ? ctx->px_recv.p : nullptr};
    if (ctx->have_peers) { mix((uint64_t)ctx->px_ng); mix((uint64_t)ctx->px_no); for (size_t p = 0; p < ctx->px_rank.size(); ++p) { mix((uint64_t)ctx->px_rank[p]); mix((uint64_t)ctx->px_goff[p + 1]); mix((uint64_t)ctx->px_ooff[p + 1]); } }
    mix((uint64_t)ctx->hb_n); mix((uint64_t)ctx->hb_nslot); mix((uint64_t)(uintptr_t)ctx->stream);
    mix((uint64_t)(uintptr_t)ctx->comm); mix((uint64_t)ctx->comm_mode); mixp((ctx->have_boundary && ctx->have_halo_d0) ? ctx->halo_d0.p : nullptr);
    for (const void *p : ptrs) mixp(p);
    return h;
}

// The captured form of the block the ctx is about to run (nw_search_begin done): found among the cached ones or captured now
// (the launches are recorded, not run).  nullptr: not eligible, or the capture failed -> the caller launches directly.
// Profiling level 4 replays everything BEFORE the block's last iteration from a graph (`head`) and launches that last iteration from
// the host, its query kernel between two events: the host enqueues it while the graph is still running, so the GPU never waits for a
// launch (with the live iteration FIRST, as until round 3, the block started launch-bound: 0.03 ms per step on a box with a slow host).
static int run_iteration(nw_ctx *ctx);
static nw_ctx::BlockGraph *block_graph(nw_ctx *ctx, int num_iters, bool head)
{
    static const bool graphs_on = !(getenv("NW_GRAPH") && atoi(getenv("NW_GRAPH")) == 0);
    if (!(graphs_on && ctx->own_stream && (ctx->profiling == 0 || (head && ctx->profiling == 4)) && num_iters > (head ? 1 : 0))) return nullptr;
    if (ctx->search_done != 0) return nullptr;
    const uint64_t key = block_graph_key(ctx);
    for (auto &gph : ctx->graphs) if (gph.exec && gph.key == key) return &gph;
    // (developer knob NW_GRAPH_DEFER=1: record a key only when it comes up the second time -- tools/experiments/r05_notes.md section 9)
    static const bool defer_on = getenv("NW_GRAPH_DEFER") && atoi(getenv("NW_GRAPH_DEFER")) != 0;
    if (defer_on) {
        static std::vector<uint64_t> seen_keys;
        bool seen = false;
        for (uint64_t k : seen_keys) seen = seen || k == key;
        if (!seen) { seen_keys.push_back(key); return nullptr; }
    }
    if (getenv("NW_VERBOSE")) fprintf(stderr, "[nanowrap] capturing a block of %d (%s, key %016llx, warm %d, grid generation %llu)\n", num_iters,
                                      head ? "all but its last iteration" : "one graph",
                                      (unsigned long long)key, ctx->face_warm ? 1 : 0, (unsigned long long)ctx->grid_generation);
    const bool warm0 = ctx->face_warm, pend0 = ctx->pos_unpack_pending;
    const int last = head ? num_iters - 1 : num_iters;
    hipGraphExec_t ea = nullptr;
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
        ctx->capturing = true;
        int r = enqueue_begin_ops(ctx);
        for (int i = 0; i < last && r == NW_OK; ++i) r = run_iteration(ctx);
        ctx->capturing = false;
        hipGraph_t graph = nullptr;
        const hipError_t ce = hipStreamEndCapture(ctx->stream, &graph);
        if (graph && getenv("NW_VERBOSE")) {          // what the recording holds (with a communicator: what RCCL put into it)
            size_t nn = 0;
            if (hipGraphGetNodes(graph, nullptr, &nn) == hipSuccess && nn > 0) {
                std::vector<hipGraphNode_t> nodes(nn);
                int cnt[16] = {0};
                if (hipGraphGetNodes(graph, nodes.data(), &nn) == hipSuccess)
                    for (size_t i = 0; i < nn; ++i) { hipGraphNodeType ty; if (hipGraphNodeGetType(nodes[i], &ty) == hipSuccess && (int)ty >= 0 && (int)ty < 16) cnt[(int)ty] += 1; }
                fprintf(stderr, "[nanowrap] recorded graph: %zu nodes -- kernel %d, memcpy %d, memset %d, host %d, child graph %d, empty %d, event wait %d, event record %d\n", nn,
                        cnt[hipGraphNodeTypeKernel], cnt[hipGraphNodeTypeMemcpy], cnt[hipGraphNodeTypeMemset], cnt[hipGraphNodeTypeHost], cnt[hipGraphNodeTypeGraph],
                        cnt[hipGraphNodeTypeEmpty], cnt[hipGraphNodeTypeWaitEvent], cnt[hipGraphNodeTypeEventRecord]);
            }
            (void)hipGetLastError();
        }
        if (r == NW_OK && ce == hipSuccess && graph && hipGraphInstantiate(&ea, graph, nullptr, nullptr, 0) != hipSuccess) ea = nullptr;
        if (graph) (void)hipGraphDestroy(graph);
    }
    (void)hipGetLastError();
    // rewind the host-side bookkeeping the recorded calls advanced
    ctx->global_iter -= ctx->search_done; ctx->search_done = 0; ctx->face_warm = warm0;
    ctx->pos_unpack_pending = pend0;
    ctx->begin_ops_pending = true;          // recorded, not run
    if (!ea) return nullptr;
    nw_ctx::BlockGraph &dst = ctx->graphs[ctx->graph_next];
    ctx->graph_next = (ctx->graph_next + 1) % 8;
    if (dst.exec) (void)hipGraphExecDestroy(dst.exec);
    dst.exec = ea; dst.key = key;
    return &dst;
}

// ---- multi-GPU: the collectives of a block, issued by the library on its own stream -----------------------------------------------
static int comm_all_reduce_dev(nw_ctx *ctx, void *buf, size_t count, ncclDataType_t dt, ncclRedOp_t op)
{
    if (!ctx->comm) return fail(ctx, NW_ERR_BADARG, "no communicator (nw_comm_init)");
    if (count == 0) return NW_OK;
    const ncclResult_t r = g_rccl.AllReduce(buf, buf, count, dt, op, ctx->comm, ctx->stream);
    if (r != ncclSuccess) return fail(ctx, NW_ERR_HIP, std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
    return NW_OK;
}

// one neighbour exchange of the owner-wise scheme: every peer's segment of px_send goes out, its segment of px_recv comes in (grouped
// ncclSend / ncclRecv: one launch).  send_owned: the owned rows go out and the ghost rows come in (else the other way round).
static int px_exchange(nw_ctx *ctx, bool send_owned, int elems_per_row, ncclDataType_t dt, size_t elem_bytes)
{
    if (!ctx->comm) return fail(ctx, NW_ERR_BADARG, "no communicator (nw_comm_init)");
    if (ctx->px_rank.empty()) return NW_OK;
    if (!g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd) return fail(ctx, NW_ERR_HIP, "RCCL: ncclSend / ncclRecv / ncclGroup* missing");
    const std::vector<int64_t> &so = send_owned ? ctx->px_ooff : ctx->px_goff, &ro = send_owned ? ctx->px_goff : ctx->px_ooff;
    ncclResult_t r = g_rccl.GroupStart();
    for (size_t p = 0; p < ctx->px_rank.size() && r == ncclSuccess; ++p) {
        const int peer = ctx->px_rank[p];
        if (peer >= ctx->comm_ranks || peer == ctx->comm_rank) { (void)g_rccl.GroupEnd(); return fail(ctx, NW_ERR_BADARG, "owner-wise exchange: a peer outside the communicator"); }
        const size_t ns = (size_t)(so[p + 1] - so[p]) * elems_per_row, nr = (size_t)(ro[p + 1] - ro[p]) * elems_per_row;
        if (ns > 0) r = g_rccl.Send((const char *)ctx->px_send.p + (size_t)so[p] * elems_per_row * elem_bytes, ns, dt, peer, ctx->comm, ctx->stream);
        if (nr > 0 && r == ncclSuccess) r = g_rccl.Recv((char *)ctx->px_recv.p + (size_t)ro[p] * elems_per_row * elem_bytes, nr, dt, peer, ctx->comm, ctx->stream);
    }
    const ncclResult_t e = g_rccl.GroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) return fail(ctx, NW_ERR_HIP, std::string("ncclSend/ncclRecv: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
    return NW_OK;
}

// what goes round between the phases of one iteration (ch_shrinkwrap_amd/parallel.py run_search is the same protocol
// over an external process group):
//   after the attraction step   'replicated': the whole per-vertex accumulator (M x 4 int64);  'halo': its boundary rows (n_slots x 4 int64)
//   after the directions        the normal-equation sums ('replicated': the point-side ones only) -- every rank then solves the same system
//   after the update            'halo': the owners' new positions of the boundary vertices (n_slots x 3 float32, owner-only non-zero rows)
static int comm_after_attract(nw_ctx *ctx)
{
    if (!ctx->comm || !ctx->comm_mode) return NW_OK;
    if (ctx->comm_mode & NW_FLAG_COMM_REPLICATED) return comm_all_reduce_dev(ctx, ctx->vacc.p, (size_t)4 * ctx->M, ncclInt64, ncclSum);
    if ((ctx->comm_mode & NW_FLAG_COMM_HALO) && ctx->have_boundary && ctx->have_peers) {
        // the copies' partial rows to the owners, the owners' sums back (the attraction step packed, the directions step takes)
        NW_TRY(px_exchange(ctx, false, 4, ncclInt64, 8));
        NW_TRY(px_stage(ctx, NW_ARR_VACC, 1));
        return px_exchange(ctx, true, 4, ncclFloat, 4);          // (the sums as the four float32 they are converted to: 16 B a row)
    }
    if ((ctx->comm_mode & NW_FLAG_COMM_HALO) && ctx->have_boundary && ctx->hb_nslot > 0) return comm_all_reduce_dev(ctx, ctx->halo_acc.p, (size_t)4 * ctx->hb_nslot, ncclInt64, ncclSum);
    return NW_OK;
}
static int comm_after_directions(nw_ctx *ctx)
{
    if (!ctx->comm || !ctx->comm_mode) return NW_OK;
    const size_t n = (size_t)((ctx->comm_mode & NW_FLAG_COMM_REPLICATED) ? SC_NPOINT : SC_MAXD) * NW_SPARTS;
    return comm_all_reduce_dev(ctx, ctx->scalars.p, n, ncclDouble, ncclSum);
}
static int comm_after_update(nw_ctx *ctx)
{
    if (!ctx->comm || !ctx->comm_mode) return NW_OK;
    if ((ctx->comm_mode & NW_FLAG_COMM_HALO) && ctx->have_boundary && ctx->have_peers) return px_exchange(ctx, true, 3, ncclFloat, 4);
    if ((ctx->comm_mode & NW_FLAG_COMM_HALO) && ctx->have_boundary && ctx->hb_nslot > 0) return comm_all_reduce_dev(ctx, ctx->halo_rows.p, (size_t)3 * ctx->hb_nslot, ncclFloat, ncclSum);
    return NW_OK;
}

// one iteration of a block, with its collectives (none without a communicator)
static int run_iteration(nw_ctx *ctx)
{
    int r = nw_iter_attract(ctx);
    if (r == NW_OK) r = comm_after_attract(ctx);
    if (r == NW_OK) r = nw_iter_directions(ctx);
    if (r == NW_OK) r = comm_after_directions(ctx);
    if (r == NW_OK) r = nw_iter_update(ctx);
    if (r == NW_OK) r = comm_after_update(ctx);
    return r;
}

NW_EXPORT int nw_comm_unique_id(uint8_t *out, int64_t nbytes)
{
    if (!out || nbytes < (int64_t)sizeof(ncclUniqueId)) return NW_ERR_BADARG;
    if (!g_rccl.load()) return NW_ERR_HIP;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return NW_ERR_HIP;
    memcpy(out, &id, sizeof(id));
    return NW_OK;
}

NW_EXPORT int nw_comm_init(nw_ctx *ctx, const uint8_t *unique_id, int64_t nbytes, int rank, int nranks)
{
    if (!ctx) return NW_ERR_BADARG;
    NW_HIP(hipSetDevice(ctx->device));
    if (ctx->comm) {                                           // (nranks <= 0: just leave)
        NW_HIP(hipStreamSynchronize(ctx->stream));
        for (auto &gph : ctx->graphs) if (gph.exec) { (void)hipGraphExecDestroy(gph.exec); gph.exec = nullptr; }      // recorded blocks hold the communicator's kernels
        (void)g_rccl.CommDestroy(ctx->comm);
        ctx->comm = nullptr; ctx->comm_rank = 0; ctx->comm_ranks = 1;
        ctx->comm_patterns.clear();
    }
    if (nranks <= 0) return NW_OK;
    if (!unique_id || nbytes < (int64_t)sizeof(ncclUniqueId) || rank < 0 || rank >= nranks) return fail(ctx, NW_ERR_BADARG, "nw_comm_init: unique id (nw_comm_unique_id of rank 0), 0 <= rank < nranks");
    // blocks are recorded with their collectives: no buffer registration at capture time (it would be a hand-shake between ranks that
    // record and ranks that replay; the library's buffers are ordinary device memory).  Only a default, and only seen by an RCCL that has
    // not read its parameters yet -- a process that initialises torch's process group first sets it there (bench.py does)
    (void)setenv("NCCL_GRAPH_REGISTER", "0", 0);
    if (!g_rccl.load()) return fail(ctx, NW_ERR_HIP, g_rccl.err);
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    const ncclResult_t r = g_rccl.CommInitRank(&ctx->comm, nranks, id, rank);
    if (r != ncclSuccess) { ctx->comm = nullptr; return fail(ctx, NW_ERR_HIP, std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error")); }
    ctx->comm_rank = rank; ctx->comm_ranks = nranks;
    return NW_OK;
}

NW_EXPORT int nw_comm_all_reduce(nw_ctx *ctx, void *buf, int64_t count, int dtype, int op)
{
    static const ncclDataType_t dts[4] = {ncclFloat, ncclDouble, ncclInt64, ncclInt32};
    static const size_t sz[4] = {4, 8, 8, 4};
    if (ctx && !buf && (op == 2 || op == 3) && dtype >= 0 && dtype <= 3 && count > 0 && count <= 4) {
        // the neighbour exchange of a sharded mesh's peers outside a block (the owners' normals after a refresh): `count` elements per row
        // of NW_ARR_PEER_SEND go out, NW_ARR_PEER_RECV comes in; op 2 = the owned rows out / the ghost rows in, 3 = the other way round
        if (!ctx->comm) return fail(ctx, NW_ERR_BADARG, "nw_comm_all_reduce: no communicator (nw_comm_init)");
        if (!ctx->have_boundary || !ctx->have_peers) return fail(ctx, NW_ERR_BADARG, "nw_comm_all_reduce: no peers (nw_set_boundary)");
        if (ctx->in_search) return fail(ctx, NW_ERR_BADARG, "neighbour exchange by hand inside a search");
        NW_HIP(hipSetDevice(ctx->device));
        return px_exchange(ctx, op == 2, (int)count, dts[dtype], sz[dtype]);
    }
    if (!ctx || !buf || count < 0 || dtype < 0 || dtype > 3 || op < 0 || op > 1) return fail(ctx, NW_ERR_BADARG, "nw_comm_all_reduce: buffer, count >= 0, dtype 0..3 (f32, f64, i64, i32), op 0 (sum) / 1 (max)");
    if (!ctx->comm) return fail(ctx, NW_ERR_BADARG, "nw_comm_all_reduce: no communicator (nw_comm_init)");
    if (ctx->in_search && ctx->capturing) return fail(ctx, NW_ERR_BADARG, "nw_comm_all_reduce while a block is being recorded");
    NW_HIP(hipSetDevice(ctx->device));
    hipPointerAttribute_t attr;
    const bool on_device = hipPointerGetAttributes(&attr, buf) == hipSuccess && attr.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    if (on_device) return comm_all_reduce_dev(ctx, buf, (size_t)count, dts[dtype], op ? ncclMax : ncclSum);
    // a host buffer (set-up: weight means, quanta, counts): staged through device memory, blocking
    const size_t bytes = (size_t)count * sz[dtype];
    NW_HIP(ctx->comm_scratch.ensure(std::max<size_t>(bytes, 256)));
    NW_HIP(hipMemcpyAsync(ctx->comm_scratch.p, buf, bytes, hipMemcpyHostToDevice, ctx->stream));
    NW_TRY(comm_all_reduce_dev(ctx, ctx->comm_scratch.p, (size_t)count, dts[dtype], op ? ncclMax : ncclSum));
    NW_HIP(hipMemcpyAsync(buf, ctx->comm_scratch.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    NW_HIP(hipStreamSynchronize(ctx->stream));
    return NW_OK;
}

NW_EXPORT int nw_search(nw_ctx *ctx, const float *lams, int n_lams, int num_iters, uint32_t flags, float *pos_out, nw_iter_log *log, int *loopcount)
{
    const bool verbose = getenv("NW_VERBOSE") != nullptr && atoi(getenv("NW_VERBOSE")) >= 2;
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t cmode = flags & (NW_FLAG_COMM_TILES | NW_FLAG_COMM_REPLICATED | NW_FLAG_COMM_HALO);
    if (ctx && cmode && !ctx->comm) return fail(ctx, NW_ERR_BADARG, "nw_search: NW_FLAG_COMM_* without a communicator (nw_comm_init)");
    if (ctx && ctx->have_boundary && (ctx->hb_nslot > 0 || (ctx->have_peers && ctx->px_ng + ctx->px_no > 0)) && !(cmode & NW_FLAG_COMM_HALO))
        return fail(ctx, NW_ERR_BADARG, "nw_search on a sharded mesh: the boundary rows must go round between the phases (NW_FLAG_COMM_HALO with a communicator, or nw_search_begin / nw_iter_* / nw_search_end with the caller's collectives)");
    NW_TRY(nw_search_begin(ctx, lams, n_lams, num_iters, flags));
    const auto t1 = std::chrono::steady_clock::now();
    // A block is a fixed launch sequence (begin ops + num_iters x 10 launches) with block-relative arguments: captured once as a
    // hipGraph and replayed for later blocks while nothing it bakes in has changed (sizes, buffers, grid, flags, lambda, quantum,
    // warm/cold start).  Host-side enqueue drops from ~3.5 us per launch to one graph launch: what small meshes are bound by.
    // Not with per-launch profiling (levels 1, 2): events recorded by graph nodes do not give elapsed times on ROCm 7.2 (they read 0).
    // a host result comes back straight from the block's last update kernel (pinned staging buffer, then host threads copy it out)
    static const bool direct_on = !(getenv("NW_DIRECT_OUT") && atoi(getenv("NW_DIRECT_OUT")) == 0);
    ctx->direct_out = false;
    bool pos_on_device = false;
    if (pos_out && num_iters > 0) {
        hipPointerAttribute_t attr;
        pos_on_device = hipPointerGetAttributes(&attr, pos_out) == hipSuccess && attr.type == hipMemoryTypeDevice;
        (void)hipGetLastError();
        // (above ~4 MB the sliced copy, which overlaps the transfer with the host-side copy-out, is the faster one: measured at 810k vertices)
        if (!pos_on_device) {
            NW_TRY(ensure_staging(ctx));
            if (ctx->rows_async) {          // the copy threads may still be reading the half the last block was staged in: this block takes the other
                ctx->pin_sel ^= 1;
                ctx->pin = (char *)ctx->pin_base + (ctx->pin_sel ? ctx->pin_bytes : 0);
            }
        }
        if (direct_on && !pos_on_device && 3 * ctx->M * sizeof(float) <= (4u << 20)) ctx->direct_out = true;
    }
    ctx->last_direct_out = ctx->direct_out;
    bool replayed = false;
    static const bool trace_blocks = getenv("NW_VERBOSE") != nullptr && atoi(getenv("NW_VERBOSE")) >= 3;
    static hipEvent_t tb0 = nullptr, tb1 = nullptr;
    if (trace_blocks) { if (!tb0) { (void)hipEventCreate(&tb0); (void)hipEventCreate(&tb1); } (void)hipEventRecord(tb0, ctx->stream); }
    // level 4: the graph holds everything before the last iteration, which is launched below (its query kernel between two events)
    const bool head = ctx->profiling == 4 && num_iters > 1;
    // With more than one rank the FIRST block of a communication pattern (mode, sizes, set of peers) is launched directly: RCCL sets up
    // channels and peer connections when an operation first needs them -- host-side hand-shakes and allocations that have no place inside a
    // stream capture.  Every later block of the pattern is recorded / replayed.  (The pattern is the same on all ranks of a block except for
    // the peer lists, which change together when new shares are cut.)
    bool eager_first = false;
    if (ctx->comm && ctx->comm_ranks > 1 && cmode) {
        uint64_t pat = 1469598103934665603ull;
        auto mixp = [&](uint64_t v) { pat ^= v; pat *= 1099511628211ull; };
        mixp(cmode); mixp((uint64_t)(uintptr_t)ctx->comm); mixp((uint64_t)ctx->M);
        mixp((uint64_t)((cmode & NW_FLAG_COMM_HALO) ? (ctx->have_peers ? 2 : 1) : 0)); mixp((uint64_t)ctx->hb_nslot);
        if ((cmode & NW_FLAG_COMM_HALO) && ctx->have_peers) for (int r : ctx->px_rank) mixp((uint64_t)r + 1);
        if (std::find(ctx->comm_patterns.begin(), ctx->comm_patterns.end(), pat) == ctx->comm_patterns.end()) {
            eager_first = true;
            if (ctx->comm_patterns.size() >= 64) ctx->comm_patterns.clear();
            ctx->comm_patterns.push_back(pat);
        }
    }
    // NW_GRAPH_COMM=0: blocks with collectives are never recorded (every launch and every RCCL call issued directly, as in the first block of a
    // pattern) -- the fall-back if RCCL's kernels as graph nodes misbehave on some node; single-GPU blocks are not affected
    static const bool graph_comm = !(getenv("NW_GRAPH_COMM") && atoi(getenv("NW_GRAPH_COMM")) == 0);
    if (!graph_comm && ctx->comm && ctx->comm_ranks > 1 && cmode) eager_first = true;
    // ... and the same by itself if recording such a block has failed twice (a capture that RCCL's calls invalidate is not an error of the
    // block: it runs directly; but trying again before every block would cost a failed capture each time)
    const bool comm_block = ctx->comm && ctx->comm_ranks > 1 && cmode;
    if (comm_block && ctx->comm_capture_failures >= 2) eager_first = true;
    nw_ctx::BlockGraph *slot = eager_first ? nullptr : block_graph(ctx, num_iters, head);
    static const bool graphs_wanted = !(getenv("NW_GRAPH") && atoi(getenv("NW_GRAPH")) == 0);
    if (graphs_wanted && comm_block && !eager_first && !slot && ctx->own_stream && ctx->search_done == 0 && (ctx->profiling == 0 || (head && ctx->profiling == 4)) && num_iters > (head ? 1 : 0)) {
        if (++ctx->comm_capture_failures == 2)
            fprintf(stderr, "[nanowrap] a block with RCCL calls could not be recorded as a hipGraph (twice): such blocks are launched directly from now on (as with NW_GRAPH_COMM=0)\n");
    }
    {
        // (how long blocks like this one take, launch to end: the copy threads are woken shortly before -- wait_block_done)
        const uint64_t bk = ((uint64_t)num_iters << 48) ^ ((uint64_t)ctx->N << 20) ^ (uint64_t)ctx->M ^ ((uint64_t)ctx->profiling << 60);
        if (bk != ctx->block_us_key) { ctx->block_us_key = bk; ctx->block_us_ema = 0.0; }
        ctx->block_t0 = std::chrono::steady_clock::now();
    }
    if (slot) {
        if (hipGraphLaunch(slot->exec, ctx->stream) == hipSuccess) {
            const int n_done = head ? num_iters - 1 : num_iters;
            ctx->begin_ops_pending = false;
            ctx->vacc_dirty = false;                       // (a graph recorded while the accumulator was dirty zeroes it: the key carries the flag)
            ctx->global_iter += n_done; ctx->search_done = n_done; ctx->face_warm = true;
            ctx->pos_unpack_pending = ctx->have_boundary;      // (the recording ends with the owners' rows of its last update packed and all-reduced)
            replayed = true;
        } else (void)hipGetLastError();
    }
    for (int i = ctx->search_done; i < num_iters; ++i) {
        const int r = run_iteration(ctx);
        if (r != NW_OK) { ctx->in_search = false; return r; }
    }
    if ((cmode & NW_FLAG_COMM_HALO) && ctx->have_boundary && ctx->have_halo_ref && num_iters > 0) {
        // (a failure in here must leave the ctx usable -- the caller's recovery, HaloExceeded / new shares, reuses it: the search is over either way)
        auto tail = [&]() -> int {
        // The tail of a sharded block, enqueued behind its last iteration (no host round trip in between): the owners' rows of the whole
        // mesh go round, then the three numbers the ranks agree on per block (exactness of the sharded query, drift, next quantum), and
        // both land in pinned host memory -- nw_search_end's one synchronisation covers them (nw_host_copy_rows with src = NULL and
        // nw_get(NW_ARR_HALO_STATS) then read the staged copies).
        if (ctx->pos_unpack_pending) { ctx->pos_unpack_pending = false; NW_TRY(halo_unpack(ctx, NW_ARR_POS)); }
        const size_t fb = (size_t)3 * ctx->M_global * sizeof(float);
        if (ctx->pin_full_bytes < fb + 16) {
            if (ctx->pin_full) (void)hipHostFree(ctx->pin_full);
            ctx->pin_full = nullptr; ctx->pin_full_bytes = 0;
            NW_HIP(hipHostMalloc(&ctx->pin_full, fb + 16, hipHostMallocDefault));
            ctx->pin_full_bytes = fb + 16;
        }
        NW_HIP(hipMemsetAsync(ctx->halo_full.p, 0, fb, ctx->stream));
        hipLaunchKernelGGL(k_halo_gather_owned, dim3(nblk(ctx->M)), dim3(NW_BLOCK), 0, ctx->stream, (int)ctx->M, ctx->hb_gv.p, ctx->owned.p, ctx->pos.p, ctx->halo_full.p);
        NW_TRY(comm_all_reduce_dev(ctx, ctx->halo_full.p, (size_t)3 * ctx->M_global, ncclFloat, ncclSum));
        NW_HIP(hipMemsetAsync(ctx->halo_stats.p, 0, 4 * sizeof(float), ctx->stream));
        const int sblocks = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (ctx->M_global + NW_BLOCK - 1) / NW_BLOCK));
        static const bool tail_direct = !(getenv("NW_HALO_TAIL_DIRECT") && atoi(getenv("NW_HALO_TAIL_DIRECT")) == 0);      // developer knob: 0 = a device-to-host copy behind the kernel
        hipLaunchKernelGGL(k_halo_block_stats, dim3(sblocks), dim3(NW_BLOCK), 0, ctx->stream, ctx->M_global, ctx->halo_full.p, ctx->halo_ref.p, 0.0f,
                           (float)ctx->local_quantum, ctx->halo_stats.p, (const NwIterLogDev *)ctx->logs.p, num_iters, tail_direct ? (float *)ctx->pin_full : (float *)nullptr);
        NW_HIP(hipGetLastError());
        NW_TRY(comm_all_reduce_dev(ctx, ctx->halo_stats.p, 4, ncclFloat, ncclMax));
        if (!tail_direct) NW_HIP(hipMemcpyAsync(ctx->pin_full, ctx->halo_full.p, fb, hipMemcpyDeviceToHost, ctx->stream));
        NW_HIP(hipMemcpyAsync((char *)ctx->pin_full + fb, ctx->halo_stats.p, 16, hipMemcpyDeviceToHost, ctx->stream));
        ctx->full_staged = true;
        return NW_OK;
        };
        const int rt = tail();
        if (rt != NW_OK) { ctx->in_search = false; ctx->full_staged = false; ctx->begin_ops_pending = false; return rt; }
    }
    ctx->sliced_S = 0;
    if (ctx->direct_out) {
        const int r = enqueue_block_done(ctx);                  // the block's end as a word in pinned memory (the host spins on it)
        if (r != NW_OK) { ctx->in_search = false; return r; }
    } else if (pos_out && num_iters > 0 && !pos_on_device) {
        const int r = enqueue_sliced_result(ctx);               // a large result: slices into the staging buffer, each announced by the next launch
        if (r != NW_OK) { ctx->in_search = false; return r; }
    }
    const auto t2 = std::chrono::steady_clock::now();
    if (trace_blocks) (void)hipEventRecord(tb1, ctx->stream);
    const int rc = nw_search_end(ctx, pos_out, log, loopcount);
    if (trace_blocks) { float ms = 0; (void)hipEventElapsedTime(&ms, tb0, tb1); fprintf(stderr, "[nanowrap] block device time (first to last launch) %.3f ms\n", ms); }
    if (verbose) {
        const auto t3 = std::chrono::steady_clock::now();
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        fprintf(stderr, "[nanowrap] search(%d): begin %ld us, enqueue %ld us (%s), end(sync+D2H) %ld us\n", num_iters, us(t0, t1), us(t1, t2), replayed ? "graph" : "direct", us(t2, t3));
    }
    return rc;
}

// ---- operators & read-back -------------------------------------------------------------------------------------
NW_EXPORT int nw_apply_A(nw_ctx *ctx, const float *x, float *y)
{
    if (!ctx || !x || !y) return NW_ERR_BADARG;
    if (!ctx->searched) return fail(ctx, NW_ERR_BADARG, "nw_apply_A: no weight matrix yet (run a search iteration first)");
    NW_HIP(ctx->tmp_f.ensure(3 * ctx->M));
    NW_HIP(ctx->tmp_f2.ensure(3 * ctx->N));
    NW_HIP(hipMemcpyAsync(ctx->tmp_f.p, x, 3 * ctx->M * sizeof(float), hipMemcpyDefault, ctx->stream));
    hipLaunchKernelGGL(k_apply_A, dim3(nblk(ctx->N)), dim3(NW_BLOCK), 0, ctx->stream, (int)ctx->N, ctx->perm.p, ctx->vidx.p, ctx->w.p, ctx->tmp_f.p, ctx->tmp_f2.p);
    NW_HIP(hipGetLastError());
    NW_HIP(hipMemcpyAsync(y, ctx->tmp_f2.p, 3 * ctx->N * sizeof(float), hipMemcpyDefault, ctx->stream));
    NW_HIP(hipStreamSynchronize(ctx->stream));
    return NW_OK;
}

NW_EXPORT int nw_apply_At(nw_ctx *ctx, const float *r, float *z)
{
    if (!ctx || !r || !z) return NW_ERR_BADARG;
    if (!ctx->searched) return fail(ctx, NW_ERR_BADARG, "nw_apply_At: no weight matrix yet (run a search iteration first)");
    NW_HIP(ctx->tmp_f.ensure(3 * ctx->M));
    NW_HIP(ctx->tmp_f2.ensure(3 * ctx->N));
    NW_HIP(hipMemcpyAsync(ctx->tmp_f2.p, r, 3 * ctx->N * sizeof(float), hipMemcpyDefault, ctx->stream));
    NW_HIP(hipMemsetAsync(ctx->tmp_f.p, 0, 3 * ctx->M * sizeof(float), ctx->stream));
    hipLaunchKernelGGL(k_apply_At, dim3(nblk(ctx->N)), dim3(NW_BLOCK), 0, ctx->stream, (int)ctx->N, ctx->perm.p, ctx->vidx.p, ctx->w.p, ctx->tmp_f2.p, ctx->tmp_f.p);
    NW_HIP(hipGetLastError());
    NW_HIP(hipMemcpyAsync(z, ctx->tmp_f.p, 3 * ctx->M * sizeof(float), hipMemcpyDefault, ctx->stream));
    NW_HIP(hipStreamSynchronize(ctx->stream));
    return NW_OK;
}

NW_EXPORT int nw_device_ptr(nw_ctx *ctx, int what, void **ptr, int64_t *nbytes)
{
    if (!ctx || !ptr) return NW_ERR_BADARG;
    void *p = nullptr;
    int64_t nb = 0;
    switch (what) {
    case NW_ARR_S: p = ctx->S.p; nb = 9 * ctx->M * 4; break;
    case NW_ARR_POS: p = ctx->pos.p; nb = 3 * ctx->M * 4; break;
    case NW_ARR_MESHPOS: p = ctx->meshpos.p; nb = 3 * ctx->M * 4; break;
    case NW_ARR_FDEF: p = ctx->fdef.p; nb = 3 * ctx->M * 4; break;
    case NW_ARR_PI: p = ctx->pi.p; nb = ctx->M * 4; break;
    case NW_ARR_VACC: p = ctx->vacc.p; nb = 4 * ctx->M * 8; break;
    case NW_ARR_NBR: p = ctx->nbr.p; nb = (int64_t)ctx->NB * ctx->M * 4; break;
    case NW_ARR_NRM: p = ctx->nrm.p; nb = 3 * ctx->M * 4; break;
    case NW_ARR_VALID: p = ctx->have_valid ? ctx->valid.p : nullptr; nb = ctx->M; break;
    case NW_ARR_SCALARS: p = ctx->scalars.p; nb = (int64_t)NW_N_SCALARS * NW_SPARTS * 8; break;
    case NW_ARR_HALO_ACC: p = ctx->have_boundary ? ctx->halo_acc.p : nullptr; nb = 4 * ctx->hb_nslot * 8; break;
    case NW_ARR_HALO_ROWS: p = ctx->have_boundary ? ctx->halo_rows.p : nullptr; nb = 3 * ctx->hb_nslot * 4; break;
    case NW_ARR_HALO_FULL: p = ctx->have_boundary ? ctx->halo_full.p : nullptr; nb = 3 * ctx->M_global * 4; break;
    case NW_ARR_HALO_STATS: p = ctx->have_boundary ? ctx->halo_stats.p : nullptr; nb = 4 * 4; break;
    case NW_ARR_PEER_SEND: p = (ctx->have_boundary && ctx->have_peers) ? ctx->px_send.p : nullptr; nb = 4 * std::max(ctx->px_ng, ctx->px_no) * 8; break;
    case NW_ARR_PEER_RECV: p = (ctx->have_boundary && ctx->have_peers) ? ctx->px_recv.p : nullptr; nb = 4 * std::max(ctx->px_ng, ctx->px_no) * 8; break;
    default: return fail(ctx, NW_ERR_BADARG, "nw_device_ptr: array is not device-addressable in caller order");
    }
    if (!p) return fail(ctx, NW_ERR_BADARG, "nw_device_ptr: array not allocated yet");
    *ptr = p;
    if (nbytes) *nbytes = nb;
    return NW_OK;
}

NW_EXPORT int nw_get(nw_ctx *ctx, int what, void *dst, int64_t nbytes)
{
    if (!ctx || !dst) return NW_ERR_BADARG;
    const int64_t N = ctx->N;
    const void *src = nullptr;
    int64_t need = 0;
    int width = 0;                       // >0: per-point array that must be un-permuted
    if (what == NW_ARR_HALO_STATS && ctx->full_staged && nbytes >= 16) {      // the staged copy of a sharded block's tail: no device round trip
        memcpy(dst, (const char *)ctx->pin_full + (size_t)3 * ctx->M_global * sizeof(float), 16);
        return NW_OK;
    }
    switch (what) {
    case NW_ARR_RES: src = ctx->res.p; width = 3; break;
    case NW_ARR_VIDX: src = ctx->vidx.p; width = 3; break;
    case NW_ARR_W: src = ctx->w.p; width = 3; break;
    case NW_ARR_DIST: src = ctx->dist.p; width = 1; break;
    case NW_ARR_FACE: src = ctx->face.p; width = 1; break;
    default: {
        void *p; int64_t nb;
        NW_TRY(nw_device_ptr(ctx, what, &p, &nb));
        src = p; need = nb;
    }
    }
    if (width > 0) {
        if (!src || !ctx->perm.p) return fail(ctx, NW_ERR_BADARG, "nw_get: array not available yet");
        need = (int64_t)width * N * 4;
        if (nbytes < need) return fail(ctx, NW_ERR_BADARG, "nw_get: destination too small");
        NW_HIP(ctx->tmp_f2.ensure(3 * N));
        hipLaunchKernelGGL(k_unpermute, dim3(nblk((int64_t)width * N)), dim3(NW_BLOCK), 0, ctx->stream, (int)N, width, ctx->perm.p, (const uint32_t *)src, (uint32_t *)ctx->tmp_f2.p);
        if (what == NW_ARR_FACE && ctx->face_sorted)         // the library's internal face order -> the caller's ids
            hipLaunchKernelGGL(k_translate_ids, dim3(nblk(N)), dim3(NW_BLOCK), 0, ctx->stream, (int)N, (int *)ctx->tmp_f2.p, ctx->face_orig.p, (int)ctx->F);
        NW_HIP(hipGetLastError());
        NW_HIP(hipMemcpyAsync(dst, ctx->tmp_f2.p, need, hipMemcpyDefault, ctx->stream));
    } else {
        if (nbytes < need) return fail(ctx, NW_ERR_BADARG, "nw_get: destination too small");
        NW_HIP(hipMemcpyAsync(dst, src, need, hipMemcpyDefault, ctx->stream));
    }
    NW_HIP(hipStreamSynchronize(ctx->stream));
    return NW_OK;
}

// pinned staging buffer for a block's result (3 M floats) and the host threads that copy it out
static int ensure_staging(nw_ctx *ctx)
{
    const size_t bytes = (((size_t)3 * ctx->M * sizeof(float)) + 4095) & ~(size_t)4095;
    if (ctx->pin_bytes < bytes) {
        if (ctx->pool) ctx->pool->wait_posted();
        if (ctx->pin_base) (void)hipHostFree(ctx->pin_base);
        ctx->pin = nullptr; ctx->pin_base = nullptr; ctx->pin_bytes = 0;
        NW_HIP(hipHostMalloc(&ctx->pin_base, 2 * bytes, hipHostMallocDefault));
        ctx->pin_bytes = bytes;
        ctx->pin_sel = 0;
        ctx->pin = ctx->pin_base;
    }
    if (!ctx->pool) {
        int T = 8;
        if (const char *e = getenv("NW_HOST_THREADS")) T = atoi(e);
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && T > hw) T = hw;
        ctx->pool = new NwHostPool();
        ctx->pool->start(T, ctx->device);
        ctx->wb_events.resize(ctx->pool->n);
        for (auto &e : ctx->wb_events) NW_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    return NW_OK;
}

// the staging buffer already holds the result (written by the block's last kernel, stream synchronised): copy it out with the host threads
static int64_t rows_per_copy_thread() { static const int64_t v = getenv("NW_HOST_ROWS_PER_THREAD") ? std::max<int64_t>(1000, atoll(getenv("NW_HOST_ROWS_PER_THREAD"))) : 25000; return v; }      // developer knob
static void copy_out_staged(nw_ctx *ctx, float *contiguous, void *rows, int64_t row_stride_bytes)
{
    ctx->n_staged_copy_outs += 1;
    (void)copy_out_chunks(ctx, contiguous, rows, row_stride_bytes, 0, 0);
}

// The staged result -> the caller's arrays, in chunks of 8192 rows taken from a counter by the copy threads and the calling thread
// (NwHostPool::run_chunks).  slice_rows > 0: the staging buffer is being filled slice by slice (k_copy_slice launches behind the block's
// last kernel); the flag word reads flag_base + 1 + (slices complete), and a chunk waits for its slice.
static bool copy_out_chunks(nw_ctx *ctx, float *contiguous, void *rows, int64_t row_stride_bytes, int64_t slice_rows, int flag_base)
{
    const int64_t M = ctx->M;
    const float *stage = (const float *)ctx->pin;
    const bool masked = rows && ctx->have_valid;
    const unsigned char *vstage = masked ? ctx->valid_host.data() : nullptr;
    const int64_t chunk = slice_rows > 0 ? std::max<int64_t>(4096, slice_rows / 2) : 8192;
    const int nchunks = (int)((M + chunk - 1) / chunk);
    volatile int *flag = ctx->pin_flag;
    // NW_FLAG_ROWS_ASYNC: the contiguous result now, the strided records behind the caller's back (the pool's threads; nw_synchronize, the
    // next block's copy-out and everything that touches the staging buffer or the records' description wait for them)
    const bool defer_rows = ctx->rows_async && rows && ctx->pool && ctx->pool->n > 1;
    std::atomic<bool> failed(false);
    auto work = [&](int c) {
        const int64_t v0 = (int64_t)c * chunk, v1 = std::min<int64_t>(M, v0 + chunk);
        if (slice_rows > 0) {
            const int need = flag_base + 1 + (int)((v1 - 1) / slice_rows) + 1;      // the chunk's last slice complete
            long spins = 0;
            const auto t0 = std::chrono::steady_clock::now();
            while (*flag - need < 0 && !failed.load(std::memory_order_relaxed)) {
                for (int k = 0; k < 8; ++k) __builtin_ia32_pause();
                // (a slice is tens of microseconds of PCIe: seconds of silence mean the device is not going to answer)
                if ((++spins & 4095) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) failed = true;
            }
            if (failed.load()) return;
        }
        if (contiguous) memcpy(contiguous + 3 * v0, stage + 3 * v0, (size_t)(v1 - v0) * 12);
        if (rows && !defer_rows) {
            char *dst = (char *)rows;
            for (int64_t v = v0; v < v1; ++v)
                if (!masked || vstage[v]) memcpy(dst + v * row_stride_bytes, stage + 3 * v, 12);
        }
    };
    if (!ctx->pool) { for (int c = 0; c < nchunks; ++c) work(c); return !failed.load(); }
    if (contiguous || !defer_rows || slice_rows > 0) ctx->pool->run_chunks(nchunks, work);
    if (failed.load()) return false;
    if (defer_rows) {
        char *dst = (char *)rows;
        const int64_t rchunk = 8192;
        ctx->pool->post_chunks((int)((M + rchunk - 1) / rchunk), [=](int c) {           // (by value: the job outlives this frame)
            const int64_t v0 = (int64_t)c * rchunk, v1 = std::min<int64_t>(M, v0 + rchunk);
            for (int64_t v = v0; v < v1; ++v)
                if (!masked || vstage[v]) memcpy(dst + v * row_stride_bytes, stage + 3 * v, 12);
        });
    }
    return true;
}

static int write_back_impl(nw_ctx *ctx, float *contiguous, void *rows, int64_t row_stride_bytes)
{
    if (!ctx || !ctx->have_mesh) return fail(ctx, NW_ERR_BADARG, "nw_write_back: mesh not set");
    if (rows && row_stride_bytes < 12) return fail(ctx, NW_ERR_BADARG, "nw_write_back: bad stride");
    ctx->n_write_backs += 1;
    const int64_t M = ctx->M;
    NW_TRY(ensure_staging(ctx));
    float *stage = (float *)ctx->pin;
    const bool masked = rows && ctx->have_valid;
    const unsigned char *vstage = masked ? ctx->valid_host.data() : nullptr;
    // The positions come back in T slices; slice t is copied out (contiguous result + strided vertex records) by host
    // thread t as soon as ITS part of the device-to-host transfer has landed, while the later slices are still in flight.
    static const int64_t rows_per_thread = getenv("NW_WB_ROWS_PER_THREAD") ? std::max(1000, atoi(getenv("NW_WB_ROWS_PER_THREAD"))) : 50000;   // measured: more, smaller slices lose to thread wake-up latency
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(ctx->pool->n, M / rows_per_thread));
    std::vector<int64_t> cut(T + 1);
    for (int t = 0; t <= T; ++t) cut[t] = (t == T) ? M : ((M * t / T) & ~(int64_t)3);      // slice starts on 16-byte boundaries (4 vertices = 48 B)
    for (int t = 0; t < T; ++t) {
        NW_HIP(hipMemcpyAsync(stage + 3 * cut[t], ctx->pos.p + 3 * cut[t], (size_t)(cut[t + 1] - cut[t]) * 12, hipMemcpyDeviceToHost, ctx->stream));
        NW_HIP(hipEventRecord(ctx->wb_events[t], ctx->stream));
    }
    std::atomic<int> failed(0);
    auto work = [&](int t) {
        if (t >= T) return;
        if (hipEventSynchronize(ctx->wb_events[t]) != hipSuccess) { failed = 1; return; }
        const int64_t v0 = cut[t], v1 = cut[t + 1];
        if (contiguous) memcpy(contiguous + 3 * v0, stage + 3 * v0, (size_t)(v1 - v0) * 12);
        if (rows) {
            char *dst = (char *)rows;
            for (int64_t v = v0; v < v1; ++v)
                if (!masked || vstage[v]) memcpy(dst + v * row_stride_bytes, stage + 3 * v, 12);
        }
    };
    if (T == 1) work(0);
    else ctx->pool->run(work);
    if (failed.load()) return fail(ctx, NW_ERR_HIP, "nw_write_back: device-to-host transfer failed");
    return NW_OK;
}

NW_EXPORT int nw_write_back(nw_ctx *ctx, float *contiguous, void *rows, int64_t row_stride_bytes)
{
    return write_back_impl(ctx, contiguous, rows, row_stride_bytes);
}

// The host half of that write-back on its own, with the library's copy threads: `src` (n_rows x 3 float32 on the HOST, e.g. the pinned
// buffer a sharded run's whole-mesh all-reduce landed in) -> `contiguous` (may be NULL) and / or the strided vertex records `rows`
// (only where valid[v] != 0 if `valid` is given).  No device work.
NW_EXPORT int nw_host_copy_rows(nw_ctx *ctx, const float *src, int64_t n_rows, float *contiguous, void *rows, int64_t row_stride_bytes, const uint8_t *valid)
{
    if (!ctx || n_rows < 0) return NW_ERR_BADARG;
    if (!src) {          // the whole mesh a sharded nw_search block left in pinned memory (its tail; nw_search_end has synchronised)
        if (!ctx->full_staged || n_rows != ctx->M_global || ctx->pin_full_bytes < (size_t)12 * (size_t)n_rows + 16) return fail(ctx, NW_ERR_BADARG, "nw_host_copy_rows: no staged result of a sharded block (nw_search with NW_FLAG_COMM_HALO)");
        src = (const float *)ctx->pin_full;
    }
    if (rows && row_stride_bytes < 12) return fail(ctx, NW_ERR_BADARG, "nw_host_copy_rows: bad stride");
    if (!ctx->pool) {
        int T = 8;
        if (const char *e = getenv("NW_HOST_THREADS")) T = atoi(e);
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && T > hw) T = hw;
        ctx->pool = new NwHostPool();
        ctx->pool->start(T, ctx->device);
        ctx->wb_events.resize(ctx->pool->n);
        for (auto &e : ctx->wb_events) NW_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(ctx->pool->n, n_rows / rows_per_copy_thread()));
    auto work = [&](int t) {
        if (t >= T) return;
        const int64_t v0 = n_rows * t / T, v1 = n_rows * (t + 1) / T;
        if (contiguous) memcpy(contiguous + 3 * v0, src + 3 * v0, (size_t)(v1 - v0) * 12);
        if (rows) {
            char *dst = (char *)rows;
            for (int64_t v = v0; v < v1; ++v)
                if (!valid || valid[v]) memcpy(dst + v * row_stride_bytes, src + 3 * v, 12);
        }
    };
    static const bool verbose3 = getenv("NW_VERBOSE") != nullptr && atoi(getenv("NW_VERBOSE")) >= 3;
    const auto tc0 = std::chrono::steady_clock::now();
    if (T == 1) work(0);
    else ctx->pool->run(work);
    if (verbose3) fprintf(stderr, "[nanowrap] host_copy_rows: %lld rows on %d threads, %ld us\n", (long long)n_rows, T,
                          (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tc0).count());
    return NW_OK;
}

// Strided target that nw_search / nw_search_end fill together with `pos_out` (mesh._vertices['position'] rows of the caller's
// vertex records, valid vertices only: mesh_conj_grad.py:288-289).  rows = NULL switches it off.  The pointer must stay valid
// until it is replaced.
NW_EXPORT int nw_set_write_back(nw_ctx *ctx, void *rows, int64_t row_stride_bytes)
{
    if (!ctx) return NW_ERR_BADARG;
    if (rows && row_stride_bytes < 12) return fail(ctx, NW_ERR_BADARG, "nw_set_write_back: bad stride");
    ctx->wb_rows = rows;
    ctx->wb_stride = rows ? row_stride_bytes : 0;
    return NW_OK;
}

NW_EXPORT int nw_lfunc(nw_ctx *ctx, int kind, const float *x, const float *f0, float *out)
{
    if (!ctx || !x || !out || kind < 0 || kind > 4) return NW_ERR_BADARG;
    if (!ctx->have_mesh) return fail(ctx, NW_ERR_BADARG, "nw_lfunc: mesh not set");
    if ((kind == 2 || kind == 3) && !f0) return fail(ctx, NW_ERR_BADARG, "nw_lfunc: f0 required");
    const int64_t M = ctx->M;
    DevBuf<float> &dx = ctx->aux_f, &df = ctx->aux_f2, &dout = ctx->aux_f3;
    NW_HIP(dx.ensure(3 * M)); NW_HIP(df.ensure(3 * M)); NW_HIP(dout.ensure(3 * M));
    int rc = NW_OK;
    do {
        if (hipMemcpyAsync(dx.p, x, 3 * M * 4, hipMemcpyDefault, ctx->stream) != hipSuccess) { rc = NW_ERR_HIP; break; }
        if (f0 && hipMemcpyAsync(df.p, f0, 3 * M * 4, hipMemcpyDefault, ctx->stream) != hipSuccess) { rc = NW_ERR_HIP; break; }
        // outputs accumulate into the caller's array, as in the reference (kind 4 overwrites the rows it visits)
        if (hipMemcpyAsync(dout.p, out, 3 * M * 4, hipMemcpyDefault, ctx->stream) != hipSuccess) { rc = NW_ERR_HIP; break; }
        if (kind == 1) {
            // gather form (one thread per vertex) when the neighbour table is symmetric -- checked by the kernel itself -- else the serial walk
            if (ctx->aux_i.ensure(1) != hipSuccess) { rc = NW_ERR_NOMEM; break; }
            if (hipMemsetAsync(ctx->aux_i.p, 0, sizeof(int), ctx->stream) != hipSuccess) { rc = NW_ERR_HIP; break; }
            hipLaunchKernelGGL(k_lfunc_lh_gather, dim3(nblk(M)), dim3(NW_BLOCK), 0, ctx->stream, (int)M, ctx->NB, ctx->nbr.p, dx.p, dout.p, df.p, ctx->aux_i.p);
            int asym = 1;
            if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&asym, ctx->aux_i.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = NW_ERR_HIP; break; }
            if (asym || getenv("NW_LH_SERIAL")) hipLaunchKernelGGL(k_lfunc_lh_serial, dim3(1), dim3(64), 0, ctx->stream, (int)M, ctx->NB, ctx->nbr.p, dx.p, dout.p);
            else if (hipMemcpyAsync(dout.p, df.p, 3 * M * 4, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { rc = NW_ERR_HIP; break; }
        }
        else if (kind == 3) hipLaunchKernelGGL(k_lfunc_lhw, dim3(nblk(M)), dim3(NW_BLOCK), 0, ctx->stream, (int)M, ctx->NB, ctx->nbr.p, dx.p, df.p, dout.p);
        else hipLaunchKernelGGL(k_lfunc_gather, dim3(nblk(M)), dim3(NW_BLOCK), 0, ctx->stream, kind, (int)M, ctx->NB, ctx->nbr.p, dx.p, df.p, dout.p);
        if (hipGetLastError() != hipSuccess) { rc = NW_ERR_HIP; break; }
        if (hipMemcpyAsync(out, dout.p, 3 * M * 4, hipMemcpyDefault, ctx->stream) != hipSuccess) { rc = NW_ERR_HIP; break; }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = NW_ERR_HIP; break; }
    } while (0);
    if (rc != NW_OK) return fail(ctx, rc, "nw_lfunc: HIP failure");
    return NW_OK;
}

NW_EXPORT int nw_curvature(nw_ctx *ctx, const int32_t *nbr_next, const float *nbr_area, const double *jitter, float kc, float kg, float c0, float dN,
                           float *k0, float *k1, float *e0, float *e1, float *H, float *K, float *dH, float *dK, float *E, float *pE,
                           float *dE_neighbors, float *dEdN)
{
    if (!ctx || ((nbr_next == nullptr) != (nbr_area == nullptr))) return NW_ERR_BADARG;
    if (!ctx->have_mesh) return fail(ctx, NW_ERR_BADARG, "nw_curvature: mesh not set");
    NW_HIP(hipSetDevice(ctx->device));
    const int64_t M = ctx->M;
    const int NB = ctx->NB;
    DevBuf<int> &d_next = ctx->aux_i;
    DevBuf<float> &d_area = ctx->aux_f, &d_out = ctx->aux_f2;
    DevBuf<double> &d_jit = ctx->aux_d;
    int rc = NW_OK;
    std::string msg;
    do {
        if (d_next.ensure((size_t)M * NB) != hipSuccess || d_area.ensure((size_t)M * NB) != hipSuccess || d_out.ensure((size_t)18 * M) != hipSuccess) { rc = NW_ERR_NOMEM; break; }
        if (nbr_next) {
            if (hipMemcpyAsync(d_next.p, nbr_next, (size_t)M * NB * 4, hipMemcpyDefault, ctx->stream) != hipSuccess) { rc = NW_ERR_HIP; break; }
            if (hipMemcpyAsync(d_area.p, nbr_area, (size_t)M * NB * 4, hipMemcpyDefault, ctx->stream) != hipSuccess) { rc = NW_ERR_HIP; break; }
        } else {
            // Both tables from the faces and positions on the device (round 5; SURVEY 8 f1: the block boundary without host table builders): the
            // ring walk of nw_set_mesh once more, this time keeping per slot the face's third corner and the face; then the faces' areas.  The
            // ring order found here is the one of NW_ARR_NBR (same pairs, same start rule with the caller's half-edge ids), so the slots line up.
            DevBuf<int> cnt, slot_face, ring_chk;
            DevBuf<int4> pairs;
            if (cnt.ensure(M) != hipSuccess || pairs.ensure((size_t)M * NB) != hipSuccess || slot_face.ensure((size_t)M * NB) != hipSuccess || ring_chk.ensure((size_t)M * NB) != hipSuccess ||
                ctx->d_small.ensure(8) != hipSuccess) { rc = NW_ERR_NOMEM; break; }
            if (hipMemsetAsync(cnt.p, 0, M * sizeof(int), ctx->stream) != hipSuccess || hipMemsetAsync(ctx->d_small.p, 0, 8 * sizeof(int), ctx->stream) != hipSuccess) { rc = NW_ERR_HIP; break; }
            hipLaunchKernelGGL(k_ring_collect, dim3(nblk(3 * ctx->F)), dim3(NW_BLOCK), 0, ctx->stream, ctx->faces.p, (int)ctx->F, (int)M, NB, cnt.p, pairs.p, ctx->d_small.p,
                               ctx->face_sorted ? ctx->face_orig.p : (const int *)nullptr);
            hipLaunchKernelGGL(k_ring_order, dim3(nblk(M)), dim3(NW_BLOCK), 0, ctx->stream, (int)M, NB, cnt.p, pairs.p, ring_chk.p, (unsigned char *)nullptr, d_next.p, slot_face.p);
            hipLaunchKernelGGL(k_slot_area, dim3(nblk((int64_t)M * NB)), dim3(NW_BLOCK), 0, ctx->stream, (int64_t)M * NB, slot_face.p, ctx->faces.p, ctx->meshpos.p, d_area.p);
            // the slots must be those of the table the kernel walks (NW_ARR_NBR): a table the caller gave nw_set_mesh in another order cannot be paired
            hipLaunchKernelGGL(k_count_mismatch, dim3(nblk((int64_t)M * NB)), dim3(NW_BLOCK), 0, ctx->stream, (int64_t)M * NB, ring_chk.p, ctx->nbr.p, ctx->d_small.p + 2);
            int bad_slots = 0;
            if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&bad_slots, ctx->d_small.p + 2, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = NW_ERR_HIP; break; }      // (the scratch buffers die with this scope)
            if (bad_slots != 0) return fail(ctx, NW_ERR_BADARG, "nw_curvature: the 1-ring table given to nw_set_mesh does not follow the library's ring order -- pass nbr_next and nbr_area in that table's order");
        }
        if (jitter) {
            if (d_jit.ensure((size_t)3 * M) != hipSuccess) { rc = NW_ERR_NOMEM; break; }
            if (hipMemcpyAsync(d_jit.p, jitter, (size_t)3 * M * 8, hipMemcpyDefault, ctx->stream) != hipSuccess) { rc = NW_ERR_HIP; break; }
        }
        float *o = d_out.p;
        float *p_k0 = o, *p_k1 = o + M, *p_H = o + 2 * M, *p_K = o + 3 * M, *p_dH = o + 4 * M, *p_dK = o + 5 * M, *p_E = o + 6 * M, *p_pE = o + 7 * M,
              *p_dEn = o + 8 * M, *p_e0 = o + 9 * M, *p_e1 = o + 12 * M, *p_dEdN = o + 15 * M;
        // unused (or stochastically skipped, membrane_mesh_utils.c:962) vertices keep what the caller's k0 / k1 / e0 / e1 hold: the reference
        // does not touch those rows
        if (ctx->have_valid) {
            struct { const float *src; float *dst; int64_t n; } ins[4] = {{k0, p_k0, M}, {k1, p_k1, M}, {e0, p_e0, 3 * M}, {e1, p_e1, 3 * M}};
            for (auto &t : ins)
                if (t.src && hipMemcpyAsync(t.dst, t.src, (size_t)t.n * 4, hipMemcpyDefault, ctx->stream) != hipSuccess) { rc = NW_ERR_HIP; break; }
            if (rc != NW_OK) break;
        }
        hipLaunchKernelGGL(k_curvature, dim3(nblk(M, 128)), dim3(128), 0, ctx->stream, (int)M, NB, ctx->meshpos.p, ctx->nrm.p,
                           ctx->have_valid ? ctx->valid.p : nullptr, ctx->nbr.p, d_next.p, d_area.p, jitter ? d_jit.p : nullptr, dN, kc, kg, c0,
                           p_k0, p_k1, p_e0, p_e1, p_H, p_K, p_dH, p_dK, p_E, p_pE, p_dEn, p_dEdN);
        if (hipGetLastError() != hipSuccess) { rc = NW_ERR_HIP; break; }
        struct { float *dst; float *src; int64_t n; } outs[12] = {{k0, p_k0, M}, {k1, p_k1, M}, {H, p_H, M}, {K, p_K, M}, {dH, p_dH, M}, {dK, p_dK, M},
                                                                   {E, p_E, M}, {pE, p_pE, M}, {dE_neighbors, p_dEn, M}, {e0, p_e0, 3 * M}, {e1, p_e1, 3 * M}, {dEdN, p_dEdN, 3 * M}};
        for (auto &t : outs)
            if (t.dst && hipMemcpyAsync(t.dst, t.src, (size_t)t.n * 4, hipMemcpyDefault, ctx->stream) != hipSuccess) { rc = NW_ERR_HIP; break; }
        if (rc != NW_OK) break;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = NW_ERR_HIP; break; }
    } while (0);
    if (rc != NW_OK) return fail(ctx, rc, "nw_curvature: HIP failure");
    return NW_OK;
}

// developer aid: the work list of the NN query and the duration the last query measured for every item (s_memtime ticks / 16;
// zeros once the list has been ordered).  out_items: int32[2 * cap] = {first localization (sorted order), count}; returns the number
// of items through *n.
static int debug_items(nw_ctx *ctx, int32_t *out_items, uint32_t *out_cost, int cap, int *n)
{
    if (!ctx || !n) return NW_ERR_BADARG;
    *n = ctx->nitems;
    const int m = std::min(cap, ctx->nitems);
    NW_HIP(hipStreamSynchronize(ctx->stream));
    if (out_items && m > 0) NW_HIP(hipMemcpy(out_items, ctx->items.p, (size_t)m * sizeof(NwItem), hipMemcpyDeviceToHost));
    if (out_cost && m > 0) NW_HIP(hipMemcpy(out_cost, ctx->item_cost.p, (size_t)m * sizeof(unsigned), hipMemcpyDeviceToHost));
    // with NW_ITEM_TIMES set, out_cost must hold 2 * cap entries: the second `cap` receive the items' start times (s_memtime >> 4)
    if (out_cost && m > 0 && getenv("NW_ITEM_TIMES")) NW_HIP(hipMemcpy(out_cost + cap, ctx->item_cost.p + ctx->nitems, (size_t)m * sizeof(unsigned), hipMemcpyDeviceToHost));
    return NW_OK;
}

// developer aid (not part of the drop-in surface): counters of the NN query accumulated since the last call.
// out[0..7] = candidates evaluated (x64 lanes each), non-empty rows listed, rows visited, cells tested, cells visited, box rows,
// rounds, small runs; out[8] = work items.  The first call switches the counting on.
static int debug_nn_stats(nw_ctx *ctx, int64_t *out)
{
    if (!ctx || !out) return NW_ERR_BADARG;
    if (!ctx->nn_stats.p) {
        NW_HIP(ctx->nn_stats.ensure((size_t)NWS_COUNT * NWS_COPIES));
        NW_HIP(hipMemset(ctx->nn_stats.p, 0, (size_t)NWS_COUNT * NWS_COPIES * sizeof(unsigned long long)));
        for (int k = 0; k <= NWS_COUNT; ++k) out[k] = 0;
        return NW_OK;
    }
    std::vector<unsigned long long> h((size_t)NWS_COUNT * NWS_COPIES);
    NW_HIP(hipStreamSynchronize(ctx->stream));
    NW_HIP(hipMemcpy(h.data(), ctx->nn_stats.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    NW_HIP(hipMemset(ctx->nn_stats.p, 0, h.size() * sizeof(unsigned long long)));
    for (int k = 0; k < NWS_COUNT; ++k) {
        unsigned long long v = 0;
        for (int c = 0; c < NWS_COPIES; ++c) v = (k == NWS_T_WAVE_MAX) ? std::max(v, h[(size_t)c * NWS_COUNT + k]) : v + h[(size_t)c * NWS_COUNT + k];
        out[k] = (int64_t)v;
    }
    out[NWS_COUNT] = ctx->nitems;
    return NW_OK;
}

// developer aids behind one entry point (not part of the drop-in surface): what = 0: counters of the NN query since the last call
// (a = int64[NWS_COUNT + 1]; the first call switches the counting on); what = 1: the query's work list (a = int32 {p0, n}[cap], b =
// uint32 cost[cap] (2 cap with NW_ITEM_TIMES), *n = items in the list)
NW_EXPORT int nw_debug(nw_ctx *ctx, int what, void *a, void *b, int cap, int *n)
{
    if (what == 0) return debug_nn_stats(ctx, (int64_t *)a);
    if (what == 1) return debug_items(ctx, (int32_t *)a, (uint32_t *)b, cap, n);
    if (what == 2 && ctx && a) {          // how a block's result reached the host so far: {staged copy-outs, sliced write-backs} (int64[2])
        ((int64_t *)a)[0] = ctx->n_staged_copy_outs; ((int64_t *)a)[1] = ctx->n_write_backs;
        return NW_OK;
    }
    if (what == 3 && ctx) {               // cap != 0: the attraction step as a launch of its own from now on (per-stage timings); 0: back inside the query launch
        if (ctx->in_search) return fail(ctx, NW_ERR_BADARG, "nw_debug(3) inside a search");
        ctx->handoff_off = cap != 0;
        return NW_OK;
    }
    return NW_ERR_BADARG;
}

// Quantum of the fixed-point A^T accumulator (NW_ARR_VACC = int64 counts of this quantum).  *q > 0 on entry fixes it for all later
// searches (ranks that all-reduce NW_ARR_VACC must agree on it: take the MAX of their LOCAL values); *q < 0 drops such an override;
// *q == 0 only asks.  On return *q = the quantum this ctx chooses by itself for its current localizations, weights and scene extent
// (computed by every nw_search_begin -- NOT the override: ranks all-reduce this value again for every block, so a scene that grows is
// followed); the quantum in use is the override if there is one, else that local value.
// May be called between nw_search_begin and the first nw_iter_attract (then it applies at once).
NW_EXPORT int nw_accumulator_quantum(nw_ctx *ctx, double *q)
{
    if (!ctx || !q) return NW_ERR_BADARG;
    if (*q != 0 && ctx->in_search && ctx->search_done > 0) return fail(ctx, NW_ERR_BADARG, "nw_accumulator_quantum: cannot change the quantum between the iterations of a search");
    if (*q > 0 && std::isfinite(*q)) {
        int e = 0;
        const double m = std::frexp(*q, &e);
        ctx->quantum_override = std::ldexp(1.0, m == 0.5 ? e - 1 : e);          // a power of two (scaling must be exact)
        if (ctx->in_search) ctx->acc_quantum = ctx->quantum_override;           // between nw_search_begin and the first iteration: applies at once
    } else if (*q < 0) {
        ctx->quantum_override = 0.0;
        if (ctx->in_search) ctx->acc_quantum = ctx->local_quantum;
    }
    if (!ctx->in_search && ctx->have_points && ctx->have_mesh) {
        // asked before the block (a rank of a multi-GPU run agrees on the quantum first, then calls nw_search): what nw_search_begin would choose
        NW_HIP(hipSetDevice(ctx->device));
        NW_TRY(alloc_work(ctx));
        NW_TRY(ensure_grid(ctx));
        ctx->local_quantum = std::ldexp(1.0, (int)std::ceil(std::log2(std::max(ctx->scene_ext * ctx->w_bound, 1e-300))) - 36);
    }
    *q = ctx->local_quantum;
    return NW_OK;
}

NW_EXPORT int nw_set_profiling(nw_ctx *ctx, int enable)
{
    if (!ctx) return NW_ERR_BADARG;
    // (level 3, a block replayed as two half graphs around a directly launched query, was removed in ABI 3: see include/nanowrap.h)
    if (enable == 3) return fail(ctx, NW_ERR_BADARG, "nw_set_profiling: level 3 (sampled, two half-block graphs) was removed; use 4");
    ctx->profiling = enable < 0 ? 0 : (enable > 4 ? 4 : enable);
    return NW_OK;
}

NW_EXPORT int nw_stage_ms(nw_ctx *ctx, int stage, double *ms, int64_t *launches)
{
    if (!ctx || stage < 0 || stage >= ST_COUNT) return NW_ERR_BADARG;
    if (ms) *ms = ctx->stage_ms[stage];
    if (launches) *launches = ctx->stage_launches[stage];
    return NW_OK;
}
