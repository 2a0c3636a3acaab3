/*
 * nw_remesh.h -- C-ABI of the block-boundary isotropic remesher (host code, libnw_remesh.so).
 *
 * Replaces, for meshes without PYME, the call
 *     self.remesh(5, target_length, 0.5, n_relax=0)          /root/reference/ch_shrinkwrap/_membrane_mesh.pyx:1546
 * (schedule of target_length: :1443-1455, :1544; also :1219 inside remove_necks).  The reference inherits `remesh`
 * from PYME's TriangleMesh, which is NOT in the reference tree (third-party arithmetic, parity
 * unpinned).  What is implemented here is the published algorithm that method follows -- Botsch & Kobbelt, "A remeshing
 * approach to multiresolution modeling", SGP 2004: per iteration (1) split every edge longer than 4/3 L at its midpoint,
 * (2) collapse every edge shorter than 4/5 L unless that creates an edge longer than 4/3 L, breaks the manifold (link
 * condition), flips a triangle or exceeds the neighbour-table width, (3) flip edges where that brings the vertex degrees
 * closer to six, (4) n_relax steps of tangential relaxation with step l.  The interface mirrors the Python signature
 * remesh(n, target_edge_length, l, n_relax).
 *
 * Conventions: plain pointers and sizes, int status (0 ok, negative = error), outputs are allocated by the library and
 * released with nwr_free; closed or open oriented 2-manifold triangle meshes (boundary edges are left untouched);
 * thread safe (no global state).
 */
#ifndef NW_REMESH_H
#define NW_REMESH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NWR_OK 0
#define NWR_ERR_BADARG (-1)      /* bad sizes / indices, or a non-finite vertex coordinate */
#define NWR_ERR_NONMANIFOLD (-2) /* an edge is used by more than two faces, or twice in the same direction */
#define NWR_ERR_NOMEM (-3)
#define NWR_ERR_RUNAWAY (-4)     /* far more splits than the target length can explain (> 4 x the pieces the faces' own edge lengths call for, or > 2^26 of those: a vertex flung far away): degenerate input */

typedef struct nwr_stats {
    int64_t n_split, n_collapse, n_flip; /* operations performed over all iterations */
    double mean_edge_length;             /* of the result */
    int32_t max_valence;                 /* of the result */
    int32_t reserved;
} nwr_stats;

int nwr_abi_version(void);
/* key 0: 1 (default) = meshes of 40 000 faces and more are remeshed in pieces on all cores when n_relax == 0 (16 Morton runs of the faces
 * with frozen rims, then the seam zone: a valid result of the same algorithm, independent of the number of threads -- NW_REMESH_THREADS --
 * but not the serial algorithm's arrays); 0 = always the serial algorithm.  Returns the previous value. */
int nwr_configure(int key, int value);

/* vertices: float[3*n_vertices]; faces: int32[3*n_faces] (counter-clockwise).  target_edge_length < 0 -> the mean edge
 * length of the input (PYME's default).  max_valence: collapses/flips never raise a vertex degree above it (the
 * optimiser's neighbour table holds 20 slots, membrane_mesh_utils.h:29); <= 0 -> 16.
 * Outputs: *out_vertices float[3 * *out_n_vertices], *out_faces int32[3 * *out_n_faces] (vertices referenced by no face
 * are dropped; ids are compacted, relative order kept). */
int nwr_remesh(const float *vertices, int64_t n_vertices, const int32_t *faces, int64_t n_faces,
               int n_iterations, float target_edge_length, float relax_lambda, int n_relax, int max_valence,
               float **out_vertices, int64_t *out_n_vertices, int32_t **out_faces, int64_t *out_n_faces,
               nwr_stats *stats /* may be NULL */);

void nwr_free(void *p);

/* Half-edge pairing of an oriented triangle mesh in linear time: half-edge 3f+k runs from faces[f][k] to faces[f][(k+1)%3];
 * twin[3f+k] = the half-edge running the other way along the same edge, -1 on a boundary.  This is the table a half-edge
 * substrate (PYME's TriangleMesh upstream, trimesh.TriMesh here) builds first after every topology change.
 * NWR_ERR_NONMANIFOLD if an edge is used twice in the same direction or by more than two faces. */
int nwr_halfedge_twins(const int32_t *faces, int64_t n_faces, int64_t n_vertices, int32_t *twin /* int32[3 * n_faces] */);

/* Block-boundary geometry refresh on the host (what the reference asks PYME for at _membrane_mesh.pyx:1524-1527, here the
 * substrate's TriMesh.update_geometry): unit face normals (F,3), face areas (F,), half-edge lengths (3F,: half-edge 3f+k runs
 * from faces[f][k] to faces[f][(k+1)%3]) and, if vertex_normal is not NULL, area-weighted unit vertex normals (V,3).
 * `positions` points at the first vertex's xyz (float32), consecutive vertices pos_stride_bytes apart (120 for PYME's vertex_t,
 * 12 for a packed array).  The outputs are float32 fields written in place: element i of an output lies i * its stride (bytes) after
 * the pointer -- 12 / 4 / 4 / 12 for packed arrays, the record size for a field of PYME's face_t / halfedge_t / vertex_t records (ABI 4:
 * no staging copies).  Bit-identical to the NumPy definition it replaces, for any number of host threads. */
int nwr_mesh_geometry(const void *positions, int64_t pos_stride_bytes, int64_t n_vertices, const int32_t *faces, int64_t n_faces,
                      void *face_normal, int64_t fn_stride_bytes, void *face_area, int64_t fa_stride_bytes,
                      void *halfedge_length, int64_t hl_stride_bytes, void *vertex_normal /* may be NULL */, int64_t vn_stride_bytes);

/* Half-edge tables and 1-rings written straight into the caller's records (structured arrays in PYME's layout: half-edge records
 * he_stride bytes apart with int32 fields at the given byte offsets, vertex records likewise): what a half-edge substrate
 * rebuilds after every topology change (trimesh._build_halfedges + TriMesh._build_rings, same conventions).  origin[3F]
 * receives the origin vertex of every half-edge.  NWR_ERR_NONMANIFOLD as for nwr_halfedge_twins. */
int nwr_build_topology(const int32_t *faces, int64_t n_faces, int64_t n_vertex_slots,
                       void *halfedges, int64_t he_stride, int64_t off_vertex, int64_t off_face, int64_t off_twin, int64_t off_next, int64_t off_prev,
                       int32_t *origin, void *vertices, int64_t v_stride, int64_t off_halfedge, int64_t off_valence, int64_t off_neighbors,
                       int32_t neighbor_size);

/* Per-slot tables of the 1-rings: vertex each ring half-edge points to (the optimiser's neighbour table,
 * /root/reference/ch_shrinkwrap/mesh_conj_grad.py:50-54), vertex its next half-edge points to and area of its face
 * (read by c_curvature_grad, membrane_mesh_utils.c:1099-1104); -1 / 0 in empty slots.  Any output may be NULL. */
int nwr_ring_tables(const void *halfedges, int64_t he_stride, int64_t off_vertex, int64_t off_face, int64_t off_next, int64_t n_halfedges,
                    const void *vertices, int64_t v_stride, int64_t off_neighbors, int32_t neighbor_size, int64_t n_vertex_slots,
                    const void *face_area, int64_t fa_stride, int32_t *ring_vertex, int32_t *ring_next_vertex, float *ring_area);

#ifdef __cplusplus
}
#endif
#endif
