/*
 * TEST INFRASTRUCTURE -- CPU oracle, not part of the product path.
 *
 * Plain-C restatement (no Python.h, plain pointers) of the native helpers on the NanoWrap hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Each function cites the reference lines whose arithmetic (order of operations, float width) it follows.
 * Parity of this file against the compiled reference (oracle/_ref/conj_grad_utils*.so, built in place from
 * /root/reference by oracle/Makefile) is checked by tests/test_oracle_vs_reference.py in the build container.
 */
#include <stdint.h>
#include <stdlib.h>
#include <math.h>

#define NWO_API __attribute__((visibility("default")))

/* z = A^T r.  Follows /root/reference/ch_shrinkwrap/conj_grad_utils.c:153-162: serial loop over points j,
 * face corners i, dimensions k; out[v_idx[j,i], k] += w[j,i] * r[j,k]; float32 accumulate in point order.
 * `out` is accumulated into (caller zeroes it, mesh_conj_grad.py:562). */
NWO_API void nwo_scatter_At(const int32_t *v_idx, const float *w, const float *r, int64_t n_points, float *out)
{
    for (int64_t j = 0; j < n_points; ++j)
        for (int i = 0; i < 3; ++i) {
            float *o = out + 3 * (int64_t)v_idx[3 * j + i];
            const float wi = w[3 * j + i];
            for (int k = 0; k < 3; ++k) {
                /* volatile-free but contraction must stay off (Makefile: -ffp-contract=off) so that the
                 * product is rounded before the add, as gcc -O2 on x86-64 does for the reference */
                float prod = wi * r[3 * j + k];
                o[k] += prod;
            }
        }
}

/* Umbrella operator.  conj_grad_utils.c:286-302: per vertex i (skipped if first neighbour == -1), per dim j:
 * d[i,j] += sum_k (f[n_k,j] - f[i,j]) accumulated directly into the output, then divided by the count. */
NWO_API void nwo_lfunc(const float *f, const int32_t *nbr, int n_verts, int n_n, float *d)
{
    for (int i = 0; i < n_verts; ++i) {
        if (nbr[(int64_t)i * n_n] == -1) continue;
        for (int j = 0; j < 3; ++j) {
            int N = 0;
            for (int k = 0; k < n_n; ++k) {
                int32_t n = nbr[(int64_t)i * n_n + k];
                if (n == -1) break;
                d[i * 3 + j] += (f[n * 3 + j] - f[i * 3 + j]);
                N += 1;
            }
            d[i * 3 + j] /= N;
        }
    }
}

/* Transposed umbrella operator.  conj_grad_utils.c:344-364.  Note the order dependence kept from the
 * reference: after vertex i has scattered into its neighbours, each touched neighbour entry is divided by
 * N *in place*, so earlier contributions to that entry get divided again by later vertices. */
NWO_API void nwo_lhfunc(const float *f, const int32_t *nbr, int n_verts, int n_n, float *d)
{
    for (int i = 0; i < n_verts; ++i) {
        if (nbr[(int64_t)i * n_n] == -1) continue;
        for (int j = 0; j < 3; ++j) {
            int N = 0;
            for (int k = 0; k < n_n; ++k) {
                int32_t n = nbr[(int64_t)i * n_n + k];
                if (n == -1) break;
                d[n * 3 + j] += (f[i * 3 + j] - f[n * 3 + j]);
                N += 1;
            }
            for (int k = 0; k < N; ++k) {
                int32_t n = nbr[(int64_t)i * n_n + k];
                d[n * 3 + j] /= N;
            }
        }
    }
}

/* Area-normalised umbrella operator.  conj_grad_utils.c:412-491: metric = sum of squared edge lengths of the
 * ORIGINAL surface `f0` (the reference passes it through the "w" slot); the Gaussian-curvature angle sum
 * computed at :448-471 is dead (commented out of :485) and is not restated. */
NWO_API void nwo_lwfunc(const float *f, const float *f0, const int32_t *nbr, int n_verts, int n_n, float *d)
{
    for (int i = 0; i < n_verts; ++i) {
        if (nbr[(int64_t)i * n_n] == -1) continue;
        float w = 0;
        int N = 0;
        for (int k = 0; k < n_n; ++k) {
            int32_t n = nbr[(int64_t)i * n_n + k];
            if (n == -1) break;
            float d2 = 0;
            for (int j = 0; j < 3; ++j) {
                float dd = (f0[n * 3 + j] - f0[i * 3 + j]);
                d2 += dd * dd;
            }
            w += d2;
            N += 1;
        }
        if (w > 0) {
            for (int k = 0; k < N; ++k) {
                int32_t n = nbr[(int64_t)i * n_n + k];
                for (int j = 0; j < 3; ++j)
                    d[i * 3 + j] += (f[n * 3 + j] - f[i * 3 + j]) / sqrtf(w);
            }
        }
    }
}

/* Transpose of the above.  conj_grad_utils.c:628-697 (scatter form: d[n] += (f[i]-f[n])/sqrt(w_i)). */
NWO_API void nwo_lhwfunc(const float *f, const float *f0, const int32_t *nbr, int n_verts, int n_n, float *d)
{
    for (int i = 0; i < n_verts; ++i) {
        if (nbr[(int64_t)i * n_n] == -1) continue;
        float w = 0;
        int N = 0;
        for (int k = 0; k < n_n; ++k) {
            int32_t n = nbr[(int64_t)i * n_n + k];
            if (n == -1) break;
            float d2 = 0;
            for (int j = 0; j < 3; ++j) {
                float dd = (f0[i * 3 + j] - f0[n * 3 + j]);
                d2 += dd * dd;
            }
            w += d2;
            N += 1;
        }
        if (w > 0) {
            for (int k = 0; k < N; ++k) {
                int32_t n = nbr[(int64_t)i * n_n + k];
                for (int j = 0; j < 3; ++j)
                    d[n * 3 + j] += (f[i * 3 + j] - f[n * 3 + j]) / sqrtf(w);
            }
        }
    }
}

/* conj_grad_utils.c:500-548: w_i = 1/sqrt(sum_k |x_nk - x_i|^2 + 1), replicated over the 3 dims; 0 if the
 * sum is 0; entries of skipped vertices are left untouched. */
NWO_API void nwo_vertex_area_weights(const float *x, const int32_t *nbr, int n_verts, int n_n, float *out)
{
    for (int i = 0; i < n_verts; ++i) {
        if (nbr[(int64_t)i * n_n] == -1) continue;
        float w = 0;
        for (int k = 0; k < n_n; ++k) {
            int32_t n = nbr[(int64_t)i * n_n + k];
            if (n == -1) break;
            float d2 = 0;
            for (int j = 0; j < 3; ++j) {
                float dd = (x[n * 3 + j] - x[i * 3 + j]);
                d2 += dd * dd;
            }
            w += d2;
        }
        if (w > 0) w = 1.0 / sqrtf(w + 1);
        else w = 0;
        out[i * 3] = w; out[i * 3 + 1] = w; out[i * 3 + 2] = w;
    }
}

/* Exact 1-NN of every point over the face centroids, float64 distances, lowest face id on ties.
 * Stands in for scipy.spatial.cKDTree(face_centers).query(points, k=1) (mesh_conj_grad.py:451-454) in
 * small test cases: cKDTree promotes both inputs to float64 and returns the Euclidean distance. O(N*F). */
NWO_API void nwo_nearest_centroid(const float *cent, int64_t n_faces, const float *pts, int64_t n_points,
                                  int32_t *idx, double *dist)
{
    for (int64_t i = 0; i < n_points; ++i) {
        const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
        double best = INFINITY;
        int32_t bi = -1;
        for (int64_t c = 0; c < n_faces; ++c) {
            const double dx = px - (double)cent[3 * c], dy = py - (double)cent[3 * c + 1], dz = pz - (double)cent[3 * c + 2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < best) { best = d2; bi = (int32_t)c; }
        }
        idx[i] = bi;
        dist[i] = sqrt(best);
    }
}
