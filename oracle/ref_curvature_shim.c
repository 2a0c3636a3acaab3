/*
 * TEST INFRASTRUCTURE (build container only).  The reference's curvature kernel `c_curvature_grad` is a `static`
 * function of /root/reference/ch_shrinkwrap/membrane_mesh_utils.c that upstream reaches only through the Cython class
 * (_membrane_mesh.pyx:48,323-347), which cannot be built without PYME.  This translation unit compiles the reference
 * file FROM WHERE IT LIES (textual #include, exactly what the .pyx does with `cdef extern from "membrane_mesh_utils.c"`)
 * and exports one plain-C entry point over the reference's own struct layouts (membrane_mesh_utils.h:31-65), so that
 * golden vectors can be produced and the oracle restatement checked.  Output only into oracle/_ref/.
 */
#include <stdlib.h>
#include "membrane_mesh_utils.c"

__attribute__((visibility("default")))
void ref_c_curvature_grad(void *vertices, void *faces, void *halfedges, float dN, float skip_prob, int n_vertices,
                          float *k_0, float *k_1, float *e_0, float *e_1, float *H, float *K, float *dH, float *dK, float *E, float *pE,
                          float *dE_neighbors, float kc, float kg, float c0, void *dEdN, unsigned seed)
{
    srand(seed);
    c_curvature_grad(vertices, faces, (halfedge_t *)halfedges, dN, skip_prob, n_vertices, k_0, k_1, e_0, e_1, H, K, dH, dK, E, pE,
                     dE_neighbors, kc, kg, c0, (points_t *)dEdN);
}
