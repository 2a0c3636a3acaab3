/*
 * TEST INFRASTRUCTURE -- CPU oracle, not part of the product path.
 *
 * Plain-C restatement (no Python.h, plain pointers) of the native helpers on the NanoWrap hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Each function cites the reference lines whose arithmetic (order of operations, float width) it follows.
 * Parity of this file against the compiled reference (oracle/_ref/conj_grad_utils*.so, built in place from
 * /root/reference by oracle/Makefile) is checked by tests/test_oracle_golden.py (test_live_reference_*) in the build container.
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

/* ------------------------------------------------------------------------------------------------------------------
 * Per-vertex curvature / Canham-Helfrich energy: restatement of c_curvature_grad,
 * /root/reference/ch_shrinkwrap/membrane_mesh_utils.c:915-1250 (+ Householder/Givens eigen-solve :618-720, closed-form
 * 2x2 pseudo-inverse :841-890, float/double helper widths :32-431), over FLAT arrays instead of the half-edge structs:
 *   pos, nrm   (M,3) f32          vertex_t.position / .normal
 *   valid      (M)   u8           vertex_t.halfedge != -1
 *   nbr        (M,NB) i32         halfedges[vertex_t.neighbors[j]].vertex, -1 padded (walk stops at the first -1, :989)
 *   nbr_next   (M,NB) i32         halfedges[halfedges[neighbors[j]].next].vertex   (:1102-1104)
 *   nbr_area   (M,NB) f32         faces[halfedges[neighbors[j]].face].area         (:1099)
 *   jitter     (M,3) f64 in [0,1) or NULL.  The reference draws three rand() numbers per processed vertex (:1017);
 *              pass the same sequence to reproduce it, NULL = the deterministic hash the HIP kernel uses.
 * Quirks kept on purpose: the un-braced `if` at :1002-1005 (jitter_width = min over ALL neighbour distances);
 * dv_hat / dv_1_hat keep the previous neighbour's value when the edge is degenerate (:1059-1062); the off-diagonal
 * terms of the projection matrices pass through float (:233,240-242); float accumulators where the reference's are.
 * Outputs: k0,k1,H,K,dH,dK,E,pE,dE_nb (M) f32; e0,e1,dEdN (M,3) f32.
 * ---------------------------------------------------------------------------------------------------------------- */
#define NWO_EPS 1e-15
#define NWO_KBT 0.0257
#define NWO_NB_MAX 20

static double nwo_norm3(const double *p) { double n = 0.0; for (int i = 0; i < 3; ++i) n += p[i] * p[i]; return sqrt(n); }
static float nwo_fnorm3f(const float *p) { float n = 0.0; for (int i = 0; i < 3; ++i) n += p[i] * p[i]; return sqrt(n); }
static double nwo_safe_div(double x, double y) { if (((y < 0) ? -y : y) < NWO_EPS) return 0.0; return x / y; }
static double nwo_fddot(const float *a, const double *b) { double c = 0.0; for (int i = 0; i < 3; ++i) c += ((double)a[i]) * b[i]; return c; }
static void nwo_proj_matrix(const float *v, double *m, double coef)
{
    const double v0 = v[0], v1 = v[1], v2 = v[2];
    const float xy = -1.0 * coef * v0 * v1, xz = -1.0 * coef * v0 * v2, yz = -1.0 * coef * v1 * v2;   /* float on purpose */
    m[0] = 1.0 - coef * v0 * v0; m[1] = xy; m[2] = xz;
    m[3] = xy; m[4] = 1.0 - coef * v1 * v1; m[5] = yz;
    m[6] = xz; m[7] = yz; m[8] = 1.0 - coef * v2 * v2;
}
static void nwo_matmul(const double *a, const double *b, double *c, int m, int n, int p)
{
    for (int i = 0; i < m; ++i) for (int j = 0; j < p; ++j) { c[i * p + j] = 0.0; for (int k = 0; k < n; ++k) c[i * p + j] += a[i * n + k] * b[k * p + j]; }
}
static void nwo_eig_givens(const double *Mvi, const float *Nvi, double *l1, double *l2, double *v1, double *v2)
{
    float e1[3] = {1, 0, 0}, sub[3], add[3], W[3];
    for (int i = 0; i < 3; ++i) { sub[i] = e1[i] - Nvi[i]; add[i] = e1[i] + Nvi[i]; }
    const float ns = nwo_fnorm3f(sub), na = nwo_fnorm3f(add);
    if (ns > na) for (int i = 0; i < 3; ++i) W[i] = sub[i] / ns; else for (int i = 0; i < 3; ++i) W[i] = add[i] / na;
    double Q[9], QT[9], QM[9], QMQ[9];
    nwo_proj_matrix(W, Q, 2.0);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) QT[j * 3 + i] = Q[i * 3 + j];
    nwo_matmul(Q, Mvi, QM, 3, 3, 3);
    nwo_matmul(QM, QT, QMQ, 3, 3, 3);
    const double tau = nwo_safe_div(QMQ[8] - QMQ[4], 2.0 * QMQ[5]);
    const double t = ((tau < 0) ? -1 : 1) / (fabs(tau) + sqrt(1 + tau * tau));
    *l1 = QMQ[4] - t * QMQ[5];
    *l2 = QMQ[8] + t * QMQ[5];
    const double c = 1.0 / sqrt(1 + t * t), s = t * c;
    double a[3], b[3];
    for (int i = 0; i < 3; ++i) { a[i] = c * QT[3 * i + 1] - s * QT[3 * i + 2]; b[i] = s * QT[3 * i + 1] + c * QT[3 * i + 2]; }
    if (*l1 > *l2) { for (int i = 0; i < 3; ++i) { v1[i] = a[i]; v2[i] = b[i]; } }
    else { const double tmp = *l1; *l1 = *l2; *l2 = tmp; for (int i = 0; i < 3; ++i) { v2[i] = a[i]; v1[i] = b[i]; } }
}
static void nwo_pinv2(const double *A, double *Ai)
{
    const double a = A[0], b = A[1], c = A[2], d = A[3];
    const double a2 = a * a, b2 = b * b, c2 = c * c, d2 = d * d;
    const double a2b2 = a2 + b2, c2d2 = c2 + d2, dif = a2b2 - c2d2, tacbd = 2 * (a * c + b * d);
    const double theta = 0.5 * atan2(2 * (a * b + c * d), a2 + c2 - b2 - d2), phi = 0.5 * atan2(tacbd, dif);
    const double ct = cos(theta), cp = cos(phi), st = sin(theta), sp = sin(phi);
    const double ctcp = ct * cp, ctsp = ct * sp, stcp = st * cp, stsp = st * sp;
    const double sign0 = ((ctcp * a + ctsp * c + stcp * b + stsp * d) < 0) ? -1 : 1;
    const double sign1 = ((stsp * a - stcp * c - ctsp * b + ctcp * d) < 0) ? -1 : 1;
    const double ss = a2b2 + c2d2, sd = sqrt(dif * dif + tacbd * tacbd);
    const double sig0 = sqrt((ss + sd) / 2.0), sssd = ss - sd;
    const double sig1 = (sssd > 0) ? sqrt(sssd / 2.0) : 0.0;
    const double thresh = (1e-8) * 0.5 * sqrt(5.0) * sig0;
    const double si0 = (sig0 < thresh) ? 0.0 : (1.0 / sig0), si1 = (sig1 < thresh) ? 0.0 : (1.0 / sig1);
    const double s0 = sign0 * si0, s1 = sign1 * si1;
    Ai[0] = ctcp * s0 + stsp * s1; Ai[1] = ctsp * s0 - stcp * s1; Ai[2] = stcp * s0 - ctsp * s1; Ai[3] = stsp * s0 + ctcp * s1;
}
static double nwo_hash_unit(uint32_t v, uint32_t k)
{   /* counter-based stand-in for rand(): identical in the HIP kernel (nw_jitter_unit) */
    uint32_t x = v * 0x9E3779B1u + k * 0x85EBCA77u + 0x165667B1u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (double)(x >> 8) / 16777216.0;
}

NWO_API void nwo_curvature_grad(const float *pos, const float *nrm, const uint8_t *valid, const int32_t *nbr, const int32_t *nbr_next,
                                const float *nbr_area, const double *jitter, int n_vertices, int NB, float dN, float kc, float kg, float c0,
                                float *k_0, float *k_1, float *e_0, float *e_1, float *H, float *K, float *dH, float *dK, float *E, float *pE,
                                float *dE_nb, float *dEdN)
{
    for (int i = 0; i < n_vertices; ++i) {
        if (!valid[i]) {
            H[i] = K[i] = dH[i] = dK[i] = dE_nb[i] = E[i] = pE[i] = 0.0f;
            dEdN[3 * i] = dEdN[3 * i + 1] = dEdN[3 * i + 2] = 0.0f;
            continue;
        }
        const float *vi = pos + 3 * i, *Nvi = nrm + 3 * i;
        const int32_t *row = nbr + (int64_t)i * NB;
        float cen[3] = {0, 0, 0};
        double r_sum = 0.0, jw = 10000000000000000.0, dv[3];
        int n = 0;
        while (n < NB && row[n] != -1) {
            const float *vj = pos + 3 * row[n];
            for (int k = 0; k < 3; ++k) { cen[k] += vj[k]; dv[k] = (double)vj[k] - (double)vi[k]; }
            const double dn = nwo_norm3(dv);
            if (dn > NWO_EPS) r_sum += 1.0 / dn;
            if (dn < jw) jw = dn;                      /* un-braced in the reference: runs for every neighbour */
            ++n;
        }
        for (int k = 0; k < 3; ++k) cen[k] /= n;
        for (int k = 0; k < 3; ++k) cen[k] += jw * ((jitter ? jitter[3 * i + k] : nwo_hash_unit((uint32_t)i, (uint32_t)k)) - 0.5);
        float vivj[3];
        for (int k = 0; k < 3; ++k) vivj[k] = cen[k] - vi[k];
        const float vivj_norm = nwo_fnorm3f(vivj);
        if (vivj_norm > 0.0) for (int k = 0; k < 3; ++k) vivj[k] /= vivj_norm; else for (int k = 0; k < 3; ++k) vivj[k] = 0.0;
        double NvidN[3], viN[3], p[9], Mvi[9];
        for (int k = 0; k < 3; ++k) { NvidN[k] = ((double)vivj[k]) * ((double)dN); viN[k] = ((double)vi[k]) - NvidN[k]; }
        nwo_proj_matrix(Nvi, p, 1.0);
        for (int k = 0; k < 9; ++k) Mvi[k] = 0.0;
        double dareas = 0.0, areas = 0.0, dv_hat[3] = {0, 0, 0}, dv1_hat[3] = {0, 0, 0};
        dE_nb[i] = 0.0f;
        for (int j = 0; j < n; ++j) {
            const float *vj = pos + 3 * row[j], *Nvj = nrm + 3 * row[j];
            double dv1[3], ndv[3], T[3], Tij[3];
            for (int k = 0; k < 3; ++k) { dv[k] = (double)vj[k] - (double)vi[k]; dv1[k] = dv[k] - NvidN[k]; }
            const double dn = nwo_norm3(dv), dn1 = nwo_norm3(dv1);
            if (dn > NWO_EPS) for (int k = 0; k < 3; ++k) dv_hat[k] = dv[k] / dn;
            if (dn1 > NWO_EPS) for (int k = 0; k < 3; ++k) dv1_hat[k] = dv1[k] / dn1;
            for (int k = 0; k < 3; ++k) ndv[k] = dv[k] * -1.0;
            for (int r = 0; r < 3; ++r) T[r] = p[3 * r] * ndv[0] + p[3 * r + 1] * ndv[1] + p[3 * r + 2] * ndv[2];
            const double Tn = nwo_norm3(T);
            if (Tn > NWO_EPS) for (int k = 0; k < 3; ++k) Tij[k] = T[k] / Tn; else for (int k = 0; k < 3; ++k) Tij[k] = 0.0;
            double q, Ni_diff, Nj_diff, Nj1_diff;
            q = nwo_fddot(Nvi, dv_hat); q *= q; Ni_diff = (q > 1.0) ? sqrt(2.0) : sqrt(2.0 - 2.0 * sqrt(1.0 - q));
            q = nwo_fddot(Nvj, dv_hat); q *= q; Nj_diff = (q > 1.0) ? sqrt(2.0) : sqrt(2.0 - 2.0 * sqrt(1.0 - q));
            q = nwo_fddot(Nvj, dv1_hat); q *= q; Nj1_diff = (q > 1.0) ? sqrt(2.0) : sqrt(2.0 - 2.0 * sqrt(1.0 - q));
            const double kj = nwo_safe_div(2.0 * Nj_diff, dn), kj1 = nwo_safe_div(2.0 * Nj1_diff, dn1);
            const double w = nwo_safe_div(nwo_safe_div(1.0, dn), r_sum);
            const double k = nwo_safe_div(2.0 * ((nwo_fddot(Nvi, ndv) < 0) ? -1 : 1) * Ni_diff, dn);
            const double Aj = nbr_area[(int64_t)i * NB + j];
            const float *vn = pos + 3 * nbr_next[(int64_t)i * NB + j];
            double dvn[3], cr[3];
            for (int kk = 0; kk < 3; ++kk) dvn[kk] = ((double)vn[kk]) - viN[kk];
            cr[0] = dv1[1] * dvn[2] - dv1[2] * dvn[1]; cr[1] = dv1[2] * dvn[0] - dv1[0] * dvn[2]; cr[2] = dv1[0] * dvn[1] - dv1[1] * dvn[0];
            const double dAj = 0.5 * nwo_norm3(cr);
            dareas += dAj;
            areas += Aj;
            const double a0 = 2.0 * kj - ((double)c0), a1 = 2.0 * kj1 - ((double)c0);
            dE_nb[i] += ((float)((Aj * w * 0.5 * ((double)kc) * (a0 * a0) - dAj * w * 0.5 * ((double)kc) * (a1 * a1)))) / dN;
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Mvi[3 * r + c] += (Tij[r] * Tij[c]) * (w * k);
        }
        double l1, l2, v1[3], v2[3];
        nwo_eig_givens(Mvi, Nvi, &l1, &l2, v1, v2);
        if (isnan(l1)) { k_0[i] = 0.0; k_1[i] = 0.0; for (int k = 0; k < 3; ++k) v1[k] = v2[k] = 0.0; }
        else { k_0[i] = 3.0 * l1 - l2; k_1[i] = 3.0 * l2 - l1; }
        for (int k = 0; k < 3; ++k) { e_0[3 * i + k] = v1[k]; e_1[3 * i + k] = v2[k]; }
        H[i] = (float)(0.5 * (k_0[i] + k_1[i]));
        K[i] = (float)(k_0[i] * k_1[i]);
        double A[2 * NWO_NB_MAX], At[2 * NWO_NB_MAX], AtA[4], AtAinv[4], AtAinvAt[2 * NWO_NB_MAX], b[NWO_NB_MAX], k_p[2];
        for (int j = 0; j < 2 * NWO_NB_MAX; ++j) A[j] = At[j] = AtAinvAt[j] = 0.0;
        for (int j = 0; j < NWO_NB_MAX; ++j) b[j] = 0.0;
        for (int j = 0; j < n; ++j) {
            const float *vj = pos + 3 * row[j];
            for (int k = 0; k < 3; ++k) dv[k] = (double)vj[k] - (double)vi[k];
            const double t0 = dv[0] * v1[0] + dv[1] * v1[1] + dv[2] * v1[2], t1 = dv[0] * v2[0] + dv[1] * v2[1] + dv[2] * v2[2];
            A[2 * j] = t0 * t0; A[2 * j + 1] = t1 * t1;
            b[j] = A[2 * j] * k_0[i] + A[2 * j + 1] * k_1[i] - (double)dN;
        }
        for (int r = 0; r < NWO_NB_MAX; ++r) { At[r] = A[2 * r]; At[NWO_NB_MAX + r] = A[2 * r + 1]; }
        nwo_matmul(At, A, AtA, 2, NWO_NB_MAX, 2);
        nwo_pinv2(AtA, AtAinv);
        nwo_matmul(AtAinv, At, AtAinvAt, 2, 2, NWO_NB_MAX);
        nwo_matmul(AtAinvAt, b, k_p, 2, NWO_NB_MAX, 1);
        dH[i] = (float)(0.5 * (k_p[0] + k_p[1]));
        dK[i] = (float)(k_p[0] * k_p[1]);
        const double hh = 2.0 * ((double)H[i]) - ((double)c0);
        E[i] = (float)(areas * ((0.5 * ((double)kc) * (hh * hh) + ((double)kg) * ((double)K[i]))));
        pE[i] = (float)(exp(-(1.0 / NWO_KBT) * ((double)E[i])));
        const double dh = 2.0 * ((double)dH[i]) - ((double)c0);
        const double dEdN_H = (dareas * ((0.5 * ((double)kc) * (dh * dh) + ((double)kg) * ((double)dK[i]))));
        const double dsum = ((double)E[i] - dEdN_H) / ((double)dN) + ((double)dE_nb[i]);
        const double lo = -0.5 * ((double)vivj_norm), hi = 0.5 * ((double)vivj_norm);
        const float dEdNs = -1.0 * ((float)((dsum > hi) ? hi : ((dsum < lo) ? lo : dsum))) * (1.0 - pE[i]);
        for (int k = 0; k < 3; ++k) dEdN[3 * i + k] = dEdNs * vivj[k];
    }
}
