/*
 * nanowrap.h -- C-ABI of libnanowrap_hip.so: the MI355X (gfx950) implementation of the NanoWrap inner loop.
 *
 * Plain C, plain pointers and sizes, no Python.h / torch types: loadable with ctypes/cffi/dlopen.
 * Every entry point names the reference interface it replaces (paths relative to /root/reference/).
 *
 * Conventions
 *   - every call returns an int status: NW_OK (0) or a negative nw_status; nw_last_error(ctx) gives text;
 *     nothing is thrown across the ABI;
 *   - the caller owns all host buffers, the library owns all device buffers and copies on nw_set_xxx and nw_get;
 *     input pointers may be host OR device pointers (copied with hipMemcpyDefault);
 *   - one nw_ctx = one device + one HIP stream; a ctx is not thread-safe, distinct ctxs are independent;
 *   - all floating-point data is float32, indices are int32, row-major, C-contiguous (the reference's
 *     contract is "contiguous ndarray, dtypes unchecked": ch_shrinkwrap/conj_grad_utils.c:130-149).
 */
#ifndef NANOWRAP_H_
#define NANOWRAP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NW_ABI_VERSION 6

typedef struct nw_ctx nw_ctx;

typedef enum nw_status {
    NW_OK = 0,
    NW_ERR_BADARG = -1,      /* NULL pointer, non-positive size, call order (mesh/points not set) */
    NW_ERR_HIP = -2,         /* a HIP runtime call failed; text in nw_last_error */
    NW_ERR_NAN = -3,         /* NaN detected where the reference asserts: mesh_conj_grad.py:514,548,580 (also: a residual so large that the
                                fixed-point A^T accumulators would overflow, > 2^27 cloud extents) */
    NW_ERR_SINGULAR = -4,    /* subspace normal equations singular: numpy.linalg.solve raises LinAlgError, conj_grad.py:219 */
    NW_ERR_NONFINITE = -5,   /* non-finite localization or vertex coordinate (cKDTree cannot index it) */
    NW_ERR_NOMEM = -6,
    NW_ERR_INTERNAL = -7,    /* invariant violated inside the library (reported instead of risking a GPU fault) */
    NW_ERR_REMOTE = -8,      /* multi-GPU: another rank raised a status in this iteration (its own code is in that rank's log); this rank stopped with it */
    NW_ERR_HANDOFF = -9,     /* the attraction step that rides in the query launch gave up waiting for its work item's nearest faces (200 ms; it never
                                uses an old face): the block is cancelled and the ctx runs the two as separate launches from then on -- run the block again */
    NW_ERR_NONMANIFOLD = -10 /* nw_remesh_device: the mesh is not an oriented 2-manifold (a directed edge occurs twice: three faces on an edge, or two of one sense) */
} nw_status;

/* how the residual weights are given -- mirrors `search(..., weights=None, sigma_inv=1.0)`,
 * ch_shrinkwrap/mesh_conj_grad.py:150-164 */
typedef enum nw_weights_mode {
    NW_WEIGHTS_FROM_SIGMA_INV = 0,   /* weights=None  -> weights = sigma_inv (array or scalar)            */
    NW_WEIGHTS_SCALAR = 1,           /* scalar weights: mask = isfinite(data), no normalisation          */
    NW_WEIGHTS_ARRAY = 2,            /* (3N,) weights: mask = weights > 0, weights /= weights.mean()      */
    NW_WEIGHTS_PRENORMALIZED = 3     /* (3N,) weights already divided by the GLOBAL mean (multi-GPU: the mean runs over all ranks) */
} nw_weights_mode;

/* nw_search flags -- keyword arguments of search(), mesh_conj_grad.py:150 */
#define NW_FLAG_POSITIVITY   1u      /* pos=True        (mesh_conj_grad.py:277-278) */
#define NW_FLAG_NO_LAST_STEP 2u      /* last_step=False (mesh_conj_grad.py:281-283) */
#define NW_FLAG_WFUNC        4u      /* Lfuncs = Lhfuncs = ["wfunc"] (mesh_conj_grad.py:36-39, 724-735) instead of ["I"]: prefs = w (f - fdef),
                                        S1 = -w prefs, LS_k = w S_k with w = vertex_area_weights(f) (conj_grad_utils.c:500-548).  The other
                                        names the reference offers (Lfunc, Lfunc2..4) fail upstream in the first iteration: they hand the
                                        float64 `f - _ncc()` to float32 C code (conj_grad_utils.c:286-302 reads it blindly) */
#define NW_FLAG_COMM_TILES      16u   /* nw_search with a communicator (nw_comm_init): which buffers go round between the phases, see there */
#define NW_FLAG_COMM_REPLICATED 32u
#define NW_FLAG_COMM_HALO       64u
#define NW_FLAG_RESULT_TO_HOST 8u    /* nw_search_begin only: nw_search_end will be given a HOST pos_out -- the block's last nw_iter_update then writes the
                                        result into the pinned staging buffer itself, as nw_search arranges on its own (the write-back, mesh_conj_grad.py:288-289) */
#define NW_FLAG_ROWS_ASYNC 128u    /* nw_search with a strided target registered (nw_set_write_back): the call returns when `pos_out` is complete; the vertex
                                      records are filled by the library's host threads while the caller goes on (typically: while the next block runs on
                                      the GPU -- the 12-byte writes into 120-byte records are memory-bound host work, 95 us per block at 200 000 vertices).
                                      nw_synchronize waits for them; so do the next block's copy-out, nw_set_mesh and nw_destroy.  The records must stay
                                      allocated until then, and nobody may read them before.  (The reference updates the mesh inside every iteration,
                                      mesh_conj_grad.py:288-290; without this flag the records are complete when nw_search returns.) */

/* per-iteration record; the reference keeps these as Python lists / attributes:
 * tests, ress, prefs (mesh_conj_grad.py:269-271), cpred, wpreds (:274, conj_grad.py:223-225) */
typedef struct nw_iter_log {
    double test;            /* 1 - |S0.S1| / (|S0||S1|)                                                */
    double res_norm;        /* || res ||_2 after distance de-weighting                                  */
    double prefs_norm;      /* || f - fdef ||_2                                                         */
    double cpred;           /* predicted data misfit (with the reference's aliasing quirk)              */
    double wpred;           /* predicted regulariser misfit                                             */
    double c[3];            /* subspace coefficients                                                    */
    double H[9];            /* regularised normal matrix actually solved (float32 values), row-major 3x3 */
    double G[3];
    double mean_dist;       /* mean point -> nearest-centroid distance (lower bound of the grid cell size) */
    double max_dist;        /* largest point -> nearest-centroid distance (a sharded mesh checks it against its halo radius) */
    int32_t n_search;       /* 2 in the first iteration of a search() call, 3 afterwards               */
    int32_t nn_max_ring;    /* largest ring any point needed in the exact NN query                      */
    int32_t status;         /* NW_OK or the nw_status raised in this iteration                          */
    int32_t executed;       /* 0 if the stop condition (mesh_conj_grad.py:1009-1016) cancelled it        */
} nw_iter_log;

/* what nw_get / nw_device_ptr can return */
typedef enum nw_array {
    NW_ARR_S = 0,           /* (3M, 3) f32  search directions            -- `cg.S`,  _membrane_mesh.pyx:1569-1599 */
    NW_ARR_RES = 1,         /* (3N,)   f32  weighted residual, point order of nw_set_points -- `cg.res`, :1621  */
    NW_ARR_VIDX = 2,        /* (N, 3)  i32  vertices of the nearest face -- `cg.w[0]`, mesh_conj_grad.py:488       */
    NW_ARR_W = 3,           /* (N, 3)  f32  normalised inverse-distance weights -- `cg.w[1]`, :503-510            */
    NW_ARR_DIST = 4,        /* (N,)    f32  distance to the nearest face centroid -- `cg.d[:,0]`, :483            */
    NW_ARR_FACE = 5,        /* (N,)    i32  nearest face id                                                       */
    NW_ARR_POS = 6,         /* (M, 3)  f32  current estimate `cg.f`                                              */
    NW_ARR_FDEF = 7,        /* (M, 3)  f32  curvature prior of the last iteration (float32 copy), :770-820        */
    NW_ARR_PI = 8,          /* (M,)    f32  point influence ||A^T 1||, _membrane_mesh.pyx:1625-1634              */
    NW_ARR_MESHPOS = 9,     /* (M, 3)  f32  mesh._vertices['position'] as written back at :289                   */
    NW_ARR_VACC = 10,       /* (M, 4)  i64  device-only: per-vertex fixed-point accumulator {A^T res, sum w} in units of nw_accumulator_quantum
                               (xyz) / 2^-40 (w): exact, order-independent sums (multi-GPU all-reduce of integers) */
    NW_ARR_SCALARS = 11,    /* f64 device-only: normal-equation partial sums of the current iteration (multi-GPU all-reduce), see nw_info */
    NW_ARR_NBR = 12,        /* (M, NB) i32  1-ring vertex ids, -1 padded (as given to, or built by, nw_set_mesh)          */
    NW_ARR_NRM = 13,        /* (M, 3)  f32  vertex normals in use (nw_set_mesh / nw_refresh_normals)     */
    NW_ARR_VALID = 14,      /* (M,)    u8   valid flags (only when given to, or built by, nw_set_mesh)                   */
    NW_ARR_HALO_ACC = 15,   /* (n_slots, 4) i64  device-only, sharded mesh: this rank's accumulator rows of the boundary vertices, one row per
                               entry of the GLOBAL boundary list (zero where the rank does not hold the vertex): all-reduce(sum) between
                               nw_iter_attract and nw_iter_directions (nw_set_boundary) */
    NW_ARR_HALO_ROWS = 16,  /* (n_slots, 3) f32  device-only: rows of the boundary vertices this rank OWNS (new positions after nw_iter_update;
                               normals after nw_halo_rows(NW_ARR_NRM, 0)), zero elsewhere: all-reduce(sum) = the owner's row on every rank */
    NW_ARR_HALO_FULL = 17,  /* (M_global, 3) f32 device-only: nw_halo_gather_owned -- the owners' rows of the whole mesh */
    NW_ARR_HALO_STATS = 18, /* (4,) f32 device-only: nw_halo_block_stats -- {largest nearest distance, accumulator quantum, max drift^2, 0}: all-reduce(MAX) */
    NW_ARR_PEER_SEND = 19,  /* device-only, nw_set_boundary with peers: the rows going out in the next neighbour exchange, peer after peer ((rows, 4) i64 for */
    NW_ARR_PEER_RECV = 20   /*   partial accumulator rows, (rows, 4) f32 for the sums going back, (rows, 3) f32 for positions / normals, all at the front of the same buffer), and the rows that came in */
} nw_array;

#define NW_N_SCALARS 32

/* ---- lifetime ------------------------------------------------------------------------------------------ */
int nw_abi_version(void);
int nw_create(int device, nw_ctx **out);          /* replaces ShrinkwrapMeshConjGrad.__init__, mesh_conj_grad.py:33-65 */
void nw_destroy(nw_ctx *ctx);
const char *nw_last_error(nw_ctx *ctx);
/* run on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = own stream */
int nw_set_stream(nw_ctx *ctx, void *hip_stream);
int nw_synchronize(nw_ctx *ctx);                  /* the ctx's stream has drained AND the host threads have finished (NW_FLAG_ROWS_ASYNC) */

/* ---- inputs -------------------------------------------------------------------------------------------- */
/* localizations + residual weighting; replaces the `points` setter (mesh_conj_grad.py:127-130, without the
 * unused point kd-tree) and the weight handling at :156-164.  sigma_inv: (3N,) or NULL (then sigma_inv_scalar);
 * weights: (3N,) for NW_WEIGHTS_ARRAY / NW_WEIGHTS_PRENORMALIZED else NULL. */
int nw_set_points(nw_ctx *ctx, const float *xyz, int64_t n_points,
                  const float *sigma_inv, float sigma_inv_scalar,
                  int weights_mode, const float *weights, float weights_scalar);
/* `data` of search(data, lams, ...) when it is NOT the localizations the optimiser was built with (mesh_conj_grad.py:150): the
 * residual targets `data` ((N,3) float32, the order of nw_set_points; :180-181, :222) and, with scalar weights, the mask is
 * isfinite(data) (:164), while the weight matrix keeps coming from the localizations (:222 -> :433).  NULL: the localizations
 * themselves (the state nw_set_points leaves; what every upstream caller passes, _membrane_mesh.pyx:1516). */
int nw_set_data(nw_ctx *ctx, const float *data);

/* mesh arrays the optimiser reads: positions `mesh._vertices['position']` (M,3), block-stale vertex normals
 * `mesh.vertex_normals` (M,3), 1-ring VERTEX ids (M,NB), -1 padded (= mesh._halfedges['vertex'][mesh._vertices
 * ['neighbors']], mesh_conj_grad.py:50-54), valid = (mesh._vertices['halfedge'] != -1) (M bytes or NULL = all
 * valid, :44), faces (F,3) (:47).
 * Block-boundary refresh on the device (what _membrane_mesh.pyx:1524-1527 asks PYME for): nbr == NULL builds the 1-ring
 * table from `faces` (counter-clockwise fans starting at the vertex's lowest-numbered outgoing half-edge 3f+k, at the
 * twin-less one for a boundary vertex; open fans list their outgoing half-edges) and, if valid is NULL too, marks the
 * vertices without faces invalid; nrm == NULL computes area-weighted vertex normals from pos and faces.  A vertex with
 * more than n_nbr neighbours is an error (the reference's table has 20 slots, membrane_mesh_utils.h:29). */
int nw_set_mesh(nw_ctx *ctx, const float *pos, const float *nrm, const int32_t *nbr, const uint8_t *valid,
                const int32_t *faces, int64_t n_vertices, int64_t n_faces, int n_nbr);
/* The refresh between blocks with unchanged topology (_membrane_mesh.pyx:1524-1527) done ON the device: area-weighted vertex normals
 * recomputed from the current device-resident positions (no upload); nrm_out (M,3) receives a copy unless NULL.  (Positions or normals
 * edited on the host go through nw_set_mesh: the same topology costs an upload, nothing else.)
 * whole_mesh_extent > 0: the largest bounding-box edge of the WHOLE mesh, for a rank that holds a share of it -- the quantum of the
 * fixed-point normal sums is then taken from it instead of from the share's own box, so every holder of a vertex rounds its sum the same
 * way and the normals are bit-identical to a single-process run; <= 0: this mesh's own box. */
int nw_refresh_normals(nw_ctx *ctx, float *nrm_out, double whole_mesh_extent);
/* Start a new optimiser on the resident mesh + localizations: what constructing a new ShrinkwrapMeshConjGrad per block does
 * to the logs and the stop-condition history (_membrane_mesh.pyx:1510, conj_grad.py:35-39), without re-uploading anything. */
int nw_reset_history(nw_ctx *ctx);

/* ---- the hot path -------------------------------------------------------------------------------------- */
/* `num_iters` iterations of ShrinkwrapMeshConjGrad.search (mesh_conj_grad.py:150-292), device resident.
 * lams: regularisation weights (only lams[0] is live with Lfuncs=['I'], :38,184); pos_out (M,3) or NULL;
 * log: num_iters records or NULL; loopcount: iterations actually executed (stop condition) or NULL. */
int nw_search(nw_ctx *ctx, const float *lams, int n_lams, int num_iters, uint32_t flags,
              float *pos_out, nw_iter_log *log, int *loopcount);

/* split-phase form of ONE iteration for multi-GPU runs (points sharded across ranks, mesh replicated):
 *   nw_iter_attract  : grid build, exact NN, weights, residual, A^T scatter -> NW_ARR_VACC partial sums
 *   -- all-reduce(sum) NW_ARR_VACC over ranks --
 *   nw_iter_directions: curvature prior, S0/S1, vertex dot products, A.S_k and point dot products -> NW_ARR_SCALARS
 *   -- all-reduce(sum) the first nw_info(NW_INFO_POINT_SCALARS) slots of NW_ARR_SCALARS over ranks --
 *   nw_iter_update   : <=3x3 solve, f += S c, write-back, log record
 * nw_search_begin/_end bracket the iterations of one search() call. */
int nw_search_begin(nw_ctx *ctx, const float *lams, int n_lams, int num_iters, uint32_t flags);
int nw_iter_attract(nw_ctx *ctx);
int nw_iter_directions(nw_ctx *ctx);
int nw_iter_update(nw_ctx *ctx);
int nw_search_end(nw_ctx *ctx, float *pos_out, nw_iter_log *log, int *loopcount);
/* NW_ARR_SCALARS layout: slot k occupies nw_info(NW_INFO_SCALAR_STRIDE) consecutive doubles (its ordered partial sums; the consumer adds
 * them in order, so the result is deterministic); all-reduce the first NW_INFO_POINT_SCALARS x stride doubles (a shared mesh: the point-side
 * sums and the status slot) or NW_INFO_SCALARS x stride (disjoint tiles, one global subspace solve) between nw_iter_directions and
 * nw_iter_update. */
#define NW_INFO_POINT_SCALARS 0
#define NW_INFO_SCALARS 1
#define NW_INFO_SCALAR_STRIDE 2
int nw_info(int what);

/* ---- operators & state read-back ------------------------------------------------------------------------- */
/* y = A x   (Afunc, mesh_conj_grad.py:518-551)  x: (3M,) -> y: (3N,), with the cached weight matrix */
int nw_apply_A(nw_ctx *ctx, const float *x, float *y);
/* z = A^T r (Ahfunc, mesh_conj_grad.py:553-588 + c_shrinkwrap_ah_helper, conj_grad_utils.c:123-167) */
int nw_apply_At(nw_ctx *ctx, const float *r, float *z);
int nw_get(nw_ctx *ctx, int what, void *dst, int64_t nbytes);
/* The write-back of search(), mesh_conj_grad.py:288-289: copies the current estimate to `contiguous` ((M,3) float32, may be
 * NULL) and/or into a strided host array of vertex records -- `rows` points at the first record's position field, consecutive
 * records are `row_stride_bytes` apart (120 for PYME's vertex_t) -- touching only the valid vertices (halfedge != -1). */
int nw_write_back(nw_ctx *ctx, float *contiguous, void *rows, int64_t row_stride_bytes);
/* the host half of that write-back alone (mesh_conj_grad.py:288-289 for a result that is already in host memory -- a sharded run's
 * whole-mesh all-reduce lands in a pinned buffer): src (n_rows,3) float32 HOST -> `contiguous` and / or the strided records `rows`
 * (only where valid[v] != 0 if `valid` is given), copied by the library's host threads.  No device work. */
int nw_host_copy_rows(nw_ctx *ctx, const float *src, int64_t n_rows, float *contiguous, void *rows, int64_t row_stride_bytes, const uint8_t *valid);
/* ---- multi-GPU: RCCL inside the library ------------------------------------------------------------------------------------------
 * (No reference counterpart: the reference is one process; SURVEY.md section 8e designs the exchange.)  One process per GPU, one nw_ctx
 * per process.  Rank 0 asks for a unique id (128 bytes) and hands it to the other ranks by whatever host channel the application has;
 * every rank then joins with nw_comm_init(ctx, id, 128, rank, nranks) -- collective, blocking.  nranks <= 0 leaves the communicator
 * (nw_destroy does too).  From then on nw_search with ONE of the NW_FLAG_COMM_* flags runs a block's collectives itself: ncclAllReduce
 * on the ctx's own stream between the phases, recorded into the block's hipGraph like every kernel launch --
 *   NW_FLAG_COMM_TILES       disjoint tiles (every rank owns its localizations AND the mesh components inside its tile): the normal-equation
 *                            sums only (nw_info(NW_INFO_SCALARS) x nw_info(NW_INFO_SCALAR_STRIDE) doubles): one global <=3x3 solve, conj_grad.py:202-219;
 *   NW_FLAG_COMM_REPLICATED  mesh replicated, localizations sharded: the per-vertex accumulator (M x 4 int64) and the point-side sums;
 *   NW_FLAG_COMM_HALO        one mesh sharded with nw_set_boundary: the accumulator's boundary rows, the sums, the owners' new boundary rows.
 * With more than one rank the first block of a communication pattern (mode, sizes, peers) is launched directly -- RCCL connects channels and
 * peers at first use, which cannot happen inside a stream capture -- and the blocks after it are recorded / replayed.
 * Every rank must call nw_search with the same flags, num_iters and lams.  A status raised on one rank (NaN, ...) travels with the sums:
 * the other ranks stop in the same iteration with NW_ERR_REMOTE.
 * nw_comm_all_reduce: the set-up and block-boundary collectives of such a run (weight means, quanta, the whole mesh of a sharded run) on the
 * same communicator and stream.  buf: device pointer (in place, asynchronous on the ctx's stream) or host pointer (staged, blocking);
 * dtype 0 float32, 1 float64, 2 int64, 3 int32; op 0 sum, 1 max.  buf = NULL with op 2 / 3: one neighbour exchange of a sharded mesh's peers
 * outside a block (nw_set_boundary with peers; the owners' normals after nw_refresh_normals): `count` (1..4) elements of `dtype` per row of
 * NW_ARR_PEER_SEND go out and NW_ARR_PEER_RECV comes in -- op 2: owned rows out, ghost rows in; op 3: ghost rows out, owned rows in. */
int nw_comm_unique_id(uint8_t *out, int64_t nbytes);
int nw_comm_init(nw_ctx *ctx, const uint8_t *unique_id, int64_t nbytes, int rank, int nranks);
int nw_comm_all_reduce(nw_ctx *ctx, void *buf, int64_t count, int dtype, int op);
/* Sharded mesh, complete form (SURVEY.md section 8e "all-reduce on the boundary-vertex rows only"; no reference counterpart -- the reference is
 * one process, conj_grad.py:202-219 solves ONE system for the whole mesh, which the all-reduced sums reproduce).  b_local[n_local] = local ids
 * of the vertices this rank holds that are held by another rank as well; b_slot[n_local] = their rows in the global boundary list of n_slots
 * entries (the same list on every rank); owned[M] = 1 for the vertices this rank owns, 0 for its copies of vertices another rank owns (the
 * vertex-side normal-equation sums then run over the owned vertices only: the all-reduce counts every vertex once; NULL = all owned); gv[M] = global id of every local vertex, n_global = vertices of the whole
 * mesh.  After this call the split-phase iteration fills / takes the exchange buffers itself and the caller only all-reduces them:
 *   nw_iter_attract -> all-reduce NW_ARR_HALO_ACC (int64 sum) -> nw_iter_directions -> all-reduce NW_ARR_SCALARS -> nw_iter_update ->
 *   all-reduce NW_ARR_HALO_ROWS (f32 sum of owner-only rows); the next nw_iter_attract / nw_search_end takes the owners' positions.
 * nw_search runs such a mesh only with a communicator and NW_FLAG_COMM_HALO.  n_slots < 0 clears the sharding; nw_set_mesh clears it too. */
int nw_set_boundary(nw_ctx *ctx, const int32_t *b_local, const int32_t *b_slot, int64_t n_local, int64_t n_slots, const uint8_t *owned,
                    const int32_t *gv, int64_t n_global, int32_t n_peers, const int32_t *peer_rank, const int64_t *ghost_off,
                    const int32_t *ghost_local, const int64_t *owned_off, const int32_t *owned_local);
/* Owner-wise exchange (ABI 5; peer_rank = NULL: the dense list above goes round).  The ranks this one shares vertices with, and for
 * peer p: ghost_local[ghost_off[p] .. ghost_off[p+1]) = local ids of this rank's COPIES of vertices peer p owns, owned_local[owned_off[p] ..
 * owned_off[p+1]) = local ids of the vertices this rank OWNS that peer p holds a copy of -- both in ascending global id, so that this rank's
 * ghost segment for p lists the same vertices in the same order as p's owned segment for this rank.  Per iteration, instead of the two
 * all-reduces over the dense list:  copies' partial accumulator rows -> owners (added);  owners' sums -> copies (as the four float32
 * the kernels convert a row to: the quanta are powers of two, the copy stores the integer that converts to the same float);  after the
 * update the owners' new positions -> copies.  Three neighbour exchanges of (rows x 32 B), (rows x 16 B), (rows x 12 B) between ranks
 * that share vertices, nothing to anyone else.  nw_search (communicator + NW_FLAG_COMM_HALO) runs them itself: grouped ncclSend / ncclRecv on the
 * ctx's stream, recorded in the block's hipGraph.  Split-phase callers move NW_ARR_PEER_SEND -> the peers' NW_ARR_PEER_RECV themselves:
 *   nw_iter_attract [ghost segments out, owned segments in] nw_halo_rows(NW_ARR_VACC, 1) [owned out, ghost in] nw_iter_directions
 *   ... nw_iter_update [owned out, ghost in]; the next nw_iter_attract / nw_search_end takes the owners' positions. */
/* the exchange buffers by hand: what = NW_ARR_VACC (-> / <- NW_ARR_HALO_ACC), NW_ARR_POS or NW_ARR_NRM (owner-only rows -> / <- NW_ARR_HALO_ROWS;
 * positions are taken into NW_ARR_POS and NW_ARR_MESHPOS); unpack = 0 fills the buffer from this rank's rows, 1 takes the (all-reduced) buffer.
 * With peers `unpack` is the step of the owner-wise exchange: accumulator 0 = copies' rows (4 int64) -> NW_ARR_PEER_SEND, 1 = NW_ARR_PEER_RECV added
 * to the owners' rows and their sums (4 float32) -> NW_ARR_PEER_SEND, 2 = NW_ARR_PEER_RECV taken by the copies; positions / normals 0 = owners' rows ->
 * NW_ARR_PEER_SEND, 1 = NW_ARR_PEER_RECV taken by the copies.
 * Needed by a caller only for the vertex normals after nw_refresh_normals (a rank does not hold every face of the vertices at the rim of its
 * share: the owner's normal is the mesh's, _membrane_mesh.pyx:1524-1527). */
int nw_halo_rows(nw_ctx *ctx, int what, int unpack);
/* NW_ARR_HALO_FULL <- the rows (what = NW_ARR_POS or NW_ARR_NRM) of the vertices this rank owns at their global ids, zero elsewhere: one
 * all-reduce(sum) per BLOCK gives every rank the whole mesh (the positions search() returns, mesh_conj_grad.py:288-292) */
int nw_halo_gather_owned(nw_ctx *ctx, int what);
/* end of a block of a sharded mesh (no reference counterpart): nw_halo_set_reference = where the WHOLE mesh was when the shares were cut
 * ((M_global,3) float32, host or device) and, optionally, d0 (n_d0 = N,) = this rank's nearest distances at that moment, caller order -- shares
 * cut with PER-LOCALIZATION halos (every face within d0_i + margin of localization i is held): the logs' max_dist of such a rank is then the
 * largest GROWTH of a nearest distance, max (d_i - d0_i), and the sharded query is exact while that growth + the mesh's drift stay within
 * the margin (d0 = NULL: one halo radius for all, max_dist = the largest nearest distance).  nw_set_boundary / nw_set_mesh drop both.
 * nw_halo_block_stats fills NW_ARR_HALO_STATS from NW_ARR_HALO_FULL (after its all-reduce), that
 * reference, the block's largest nearest distance (the caller has it from the iteration logs) and this rank's accumulator quantum -- the
 * three numbers the ranks agree on with ONE all-reduce(MAX) per block: is the sharded query still exact, how far has the mesh drifted,
 * which quantum do the integer accumulators of the next block share. */
int nw_halo_set_reference(nw_ctx *ctx, const float *full, const float *d0, int64_t n_d0);
int nw_halo_block_stats(nw_ctx *ctx, double max_dist);

/* registers the strided vertex records (mesh._vertices['position'] rows, `row_stride_bytes` apart) that nw_search / nw_search_end
 * fill together with `pos_out` at the end of every search (valid vertices only, mesh_conj_grad.py:288-289); NULL switches it off */
int nw_set_write_back(nw_ctx *ctx, void *rows, int64_t row_stride_bytes);
/* raw device pointer + byte size of a device-resident array (NW_ARR_VACC, NW_ARR_SCALARS, NW_ARR_POS, ...) */
int nw_device_ptr(nw_ctx *ctx, int what, void **ptr, int64_t *nbytes);

/* alternate regularisers (default off in the reference; selectable through Lfuncs, mesh_conj_grad.py:36-39):
 * kind 0: c_shrinkwrap_l_func   conj_grad_utils.c:249-306      d = L x        (umbrella / N)
 *      1: c_shrinkwrap_lh_func  :308-368                        d = L^H x      (order-dependent in-place /N: evaluated per target vertex in the
 *                                                                            reference's visiting order; serial walk if the table is not symmetric)
 *      2: c_shrinkwrap_lw_func  :370-497                        area-normalised, metric from f0
 *      3: c_shrinkwrap_lhw_func :585-710                        transpose of 2
 *      4: vertex_area_weights   :551-582                        1/sqrt(sum|e|^2+1) replicated x3 (x = positions)
 * x, f0, out: (3M,) float32 host or device; f0 may be NULL for kinds 0,1,4. Uses the neighbour table of nw_set_mesh. */
int nw_lfunc(nw_ctx *ctx, int kind, const float *x, const float *f0, float *out);

/* Block-boundary kernel: per-vertex curvature tensor and Canham-Helfrich energy, replaces c_curvature_grad
 * (membrane_mesh_utils.c:915-1250, reached through MembraneMesh.curvature_grad_c, _membrane_mesh.pyx:323-347), with
 * skip_prob folded into the valid flags of nw_set_mesh: a vertex the reference would skip (:962, `r2() < skip_prob`; never > 0 on the
 * live path) is an unused slot for this call -- its H, K, dH, dK, dE_neighbors, E, pE, dEdN read 0 and its k0, k1, e0, e1 rows keep
 * what the caller's arrays held.  Works on the ctx's CURRENT device-resident mesh positions, the
 * normals of nw_set_mesh / nw_refresh_normals and the neighbour table of nw_set_mesh (walk stops at the first -1, as the
 * reference's does).  nbr_next (M,NB) i32 = halfedges[halfedges[neighbors[j]].next].vertex, nbr_area (M,NB) f32 =
 * faces[halfedges[neighbors[j]].face].area; BOTH NULL (round 5): the library builds the two tables itself from its faces and current
 * positions (the ring walk of nw_set_mesh keeping the face's third corner and the face; areas 0.5 |(p1 - p0) x (p2 - p0)| in float32, every
 * operation rounded on its own: bit for bit what trimesh.TriMesh / nwr_mesh_geometry hold) -- valid when the 1-ring table follows the
 * library's ring order (built by nw_set_mesh(nbr = NULL), or by the host substrate, whose order is the same); any other order is refused.  jitter: (M,3) float64 in [0,1) standing for the reference's rand() stream
 * (:1017), or NULL for a deterministic hash of (vertex, axis).  Outputs (host or device): k0,k1,H,K,dH,dK,E,pE,
 * dE_neighbors (M) f32; e0,e1,dEdN (M,3) f32; any output pointer may be NULL. */
int nw_curvature(nw_ctx *ctx, const int32_t *nbr_next, const float *nbr_area, const double *jitter,
                 float kc, float kg, float c0, float dN,
                 float *k0, float *k1, float *e0, float *e1, float *H, float *K, float *dH, float *dK,
                 float *E, float *pE, float *dE_neighbors, float *dEdN);

/* device timing of the last nw_search (ms), split by stage; for bench.py's roofline object.
 * stage: 0 total, 1 grid build, 2 NN query, 3 attraction (weights/residual/scatter), 4 prior+directions,
 * 5 A.S + dots, 6 solve+update, 7 float64 NN fix-up (separate launch only with NW_FUSE_FIXUP=0).  nw_set_profiling level: 0 off (nw_search replays
 * each block as a hipGraph); 1 = HIP events around every NN query launch (the dominant kernel; each event pair costs a few microseconds of
 * stream serialisation); 2 = around every stage -- levels 1 and 2 launch every kernel from the host, because events inside graph nodes read
 * 0 on ROCm 7.2; 4 = only the NN query of each block's LAST iteration is bracketed: everything before it is one replayed hipGraph, the last
 * iteration is launched from the host while that graph is still running (the GPU never waits for a launch), and the event pairs come once per
 * block (what bench.py times at).
 * Level 3 (ABI 2: the block as two half graphs around a directly launched query) is gone and returns NW_ERR_BADARG: in a process that
 * had also loaded PyTorch about one block in twenty waited 5-6 ms behind its last kernel, the traces kept from then do not contain such
 * a block, and level 4 gives the same one sample per block with a single graph launch per block. */
int nw_set_profiling(nw_ctx *ctx, int enable);
int nw_stage_ms(nw_ctx *ctx, int stage, double *ms, int64_t *launches);

/* One-off set-up the library would otherwise do at the start of the next block -- after the first completed block the
 * localizations are re-sorted once by their foot point on the surface (a second radix sort + regather, ~1-2 ms at 1M points) --
 * done NOW, so that a caller can keep it out of a timed region.  No reference counterpart (speed only); no-op if nothing is due. */
int nw_optimize_layout(nw_ctx *ctx);

/* quantum of NW_ARR_VACC's xyz columns (a power of two).  *q > 0 on entry fixes it for all later searches -- ranks that all-reduce
 * NW_ARR_VACC must agree on it (all-reduce MAX of their own LOCAL values); *q < 0 drops the override, *q == 0 only asks.  On return *q is
 * the quantum this ctx would choose by itself (recomputed by every nw_search_begin from its localizations, weights and scene extent), NOT
 * the override: ranks all-reduce it again for every block. */
int nw_accumulator_quantum(nw_ctx *ctx, double *q);

/* developer aids, no reference counterpart (one entry point):
 * what = 0: counters of the exact nearest-face query accumulated since the previous call (the first call switches the counting on; the
 *   counting variant of the kernel is a few per cent slower).  a = int64 out[17]: candidate evaluations per wave summed, non-empty rows listed,
 *   rows visited, cells tested, cells visited, rows of the boxes, rounds, slowest wave, the waves' time (s_memtime ticks / 16) in the
 *   candidate stream / in all / before the walk / after it, then what a lane's OWN ball reaches of the cells its wave tested: cells summed
 *   over the lanes, the wave's largest lane summed over the waves, and the same two for the candidates in those cells; out[16] = work
 *   items.  b, cap, n unused.
 * what = 1: the work list of the query -- a = int32 {first localization in sorted order, count}[cap], b = uint32 cost[cap] (duration in
 *   s_memtime ticks / 16 the last query measured for each item; zeros once the list has been ordered: heavy first, light last), *n = items in
 *   the list.  With NW_ITEM_TIMES set in the environment the timing stays on, durations are in 10 ns ticks of the clock all XCDs share, and b
 *   must hold 2 * cap entries: the second `cap` receive when each item started -- the launch's time line (tools/nn_costs.py).
 * what = 2: a = int64 out[2]: how often a block's result has reached the host so far as a copy-out of the staging buffer the last update
 *   kernel wrote (the fast path of results up to 4 MB), and how often as a sliced device-to-host write-back (larger results, or a block
 *   the device-side stop condition ended early) -- a block must take ONE of the two.
 * what = 3: cap != 0 makes the attraction step a launch of its own from now on (k_attract behind the query), cap = 0 puts it back into the
 *   query launch (workgroups appended to k_nn_wave's grid: the default).  Results are bit-identical either way; bench.py takes its per-stage
 *   timings with the steps apart. */
int nw_debug(nw_ctx *ctx, int what, void *a, void *b, int cap, int *n);

/* ---- the block-boundary remesher on the device (ABI 6) ---------------------------------------------------------------------------
 * Replaces the call `self.remesh(5, target_length, 0.5, n_relax=0)` of the reference's outer loop (_membrane_mesh.pyx:1546; schedule of
 * target_length :1443-1455, :1544), i.e. PYME's TriangleMesh.remesh -- not in the reference tree: parity unpinned, SURVEY.md 8c.  The same
 * published algorithm and the same admission tests as the host remesher (include/nw_remesh.h: nwr_remesh), operation by operation, but
 * as rounds of independent operations (csrc/nw_remesh_dev.hip): per iteration split every edge longer than 4/3 L, collapse every edge
 * shorter than 4/5 L that may be collapsed, flip where that brings the degrees closer to six, each until nothing is left to do, then n_relax
 * steps of tangential relaxation with step relax_lambda (the block boundary's call has n_relax = 0; PYME's default, which remove_necks uses at
 * _membrane_mesh.pyx:1219, is l = 0.5, n_relax = 10).  A valid result of the algorithm (closed stays closed, genus kept, boundary and bow-tie vertices never touched), the same
 * arrays on every run -- not the host remesher's arrays: the order of the operations differs.
 * vertices float[3 n_vertices], faces int32[3 n_faces] in HOST memory; target_edge_length > 0; max_valence <= 0 -> 16.  Outputs are
 * allocated by the library (release with nw_host_free): vertices no face refers to are dropped; the vertices are numbered in Morton order of
 * their positions (what gathers by vertex id want in the next block), the faces keep their relative order.
 * NW_ERR_BADARG: sizes, an index out of range, a face that names a vertex twice, a non-finite vertex, or lengths that call for more than
 * 2^26 faces; NW_ERR_NONMANIFOLD: a directed edge occurs twice. */
typedef struct nw_remesh_stats {
    int64_t n_split, n_collapse, n_flip; /* operations performed */
    double mean_edge_length;             /* of the result */
    int32_t max_valence;                 /* of the result */
    int32_t rounds_split, rounds_collapse, rounds_flip; /* rounds of independent operations over all iterations */
} nw_remesh_stats;
int nw_remesh_device(int device, const float *vertices, int64_t n_vertices, const int32_t *faces, int64_t n_faces, int n_iterations,
                     float target_edge_length, float relax_lambda, int n_relax, int max_valence, float **out_vertices, int64_t *out_n_vertices,
                     int32_t **out_faces, int64_t *out_n_faces, nw_remesh_stats *stats /* may be NULL */);
void nw_host_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* NANOWRAP_H_ */
