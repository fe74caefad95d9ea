// mcx_host_linalg.hpp -- the initial factor on the host (MCMC_calculate_R at MCMC_init.F90:109): dpotf2 / dpotri / the pinned Jacobi SVD, operation for operation the device's.
// Part of the ONE translation unit mcx_api.hip (included there, in this order: mcx_host_engine, mcx_host_linalg, mcx_host_launch, mcx_host_adapt, mcx_host_pooled, mcx_host_callbacks); not a stand-alone header.

// dpotf2('U') + scaling on the host for the shared initial factor: same operation sequence as the
// device's calculate_R (MCMC_calculate_R at MCMC_init.F90:109).  cm: col-major d*d, Rp: packed upper.
static inline int h_rowstart(int i, int d) { return i * d - (i * (i - 1)) / 2; }
static inline int h_pidx(int i, int j, int d) { return h_rowstart(i, d) + (j - i); }

static int host_initial_R(int d, const std::vector<double> &cm, std::vector<double> &Rp, std::vector<double> &Cp)
{
    int P = d * (d + 1) / 2;
    std::vector<double> A(P);
    for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) A[h_pidx(i, j, d)] = cm[(size_t)i + (size_t)j * d];
    Cp = A;
    for (int j = 0; j < d; ++j) {
        double dot = 0.0;
        for (int i = 0; i < j; ++i) dot = std::fma(A[h_pidx(i, j, d)], A[h_pidx(i, j, d)], dot);
        double ajj = A[h_pidx(j, j, d)] - dot;
        if (!(ajj > 0.0)) return j + 1;
        double rj = std::sqrt(ajj);
        A[h_pidx(j, j, d)] = rj;
        double rinv = 1.0 / rj;
        for (int k = j + 1; k < d; ++k) {
            double t = 0.0;
            for (int i = 0; i < j; ++i) t = std::fma(A[h_pidx(i, k, d)], A[h_pidx(i, j, d)], t);
            A[h_pidx(j, k, d)] = (A[h_pidx(j, k, d)] - t) * rinv;
        }
    }
    double sq = std::sqrt((double)d);
    Rp.resize(P);
    for (int e = 0; e < P; ++e) Rp[e] = A[e] * 2.4 / sq;
    return 0;
}

// The pinned dgesvd('A','N') of a symmetric PSD matrix (one-sided Jacobi), same operation sequence as the device's
// symsvd_dev; used for the shared initial factor.  G, V column-major n*n.
static inline double h_tree8(const double *p) { return ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])); }
static void host_symsvd(int n, std::vector<double> &G, std::vector<double> &V, std::vector<double> &sv)
{
    V.assign((size_t)n * n, 0.0); sv.assign(n, 0.0);
    for (int j = 0; j < n; ++j) V[(size_t)j * n + j] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double *gp = &G[(size_t)p * n], *gq = &G[(size_t)q * n];
                // the routine's dot products: eight partial fma chains by row index mod 8, added pairwise (oracle/mcx_svd.h)
                double pa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pb[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int k = 0; k < n; ++k) { const int j = k & 7; pa[j] = std::fma(gp[k], gp[k], pa[j]); pb[j] = std::fma(gq[k], gq[k],
                    pb[j]); pg[j] = std::fma(gp[k], gq[k], pg[j]); }
                const double alpha = h_tree8(pa), beta = h_tree8(pb), gamma = h_tree8(pg);
                if (gamma == 0.0) continue;
                if (std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta)) continue;
                rotated = true;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = std::copysign(1.0, zeta) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < n; ++k) { double a = gp[k], b = gq[k]; gp[k] = c * a - sn * b; gq[k] = sn * a + c * b; }
                double *vp = &V[(size_t)p * n], *vq = &V[(size_t)q * n];
                for (int k = 0; k < n; ++k) { double a = vp[k], b = vq[k]; vp[k] = c * a - sn * b; vq[k] = sn * a + c * b; }
            }
        if (!rotated) break;
    }
    for (int j = 0; j < n; ++j) {
        double pa[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < n; ++k) pa[k & 7] = std::fma(G[(size_t)j * n + k], G[(size_t)j * n + k], pa[k & 7]);
        sv[j] = std::sqrt(h_tree8(pa));
    }
    for (int i = 0; i < n - 1; ++i) {
        int m = i;
        for (int j = i + 1; j < n; ++j) if (sv[j] > sv[m]) m = j;
        if (m != i) { std::swap(sv[i], sv[m]); for (int k = 0; k < n; ++k) std::swap(V[(size_t)i * n + k], V[(size_t)m * n + k]); }
    }
}

// MCMC_calculate_R, SVD branches, for the shared initial covariance (MCMC_init.F90:109): returns 0 or an error code.
// Rfull: column-major d*d factor (U for scam, U sqrt(s) 2.4/sqrt(d) otherwise); std: sqrt(s) (scam)
static int host_initial_svd(int d, const std::vector<double> &cm, double condmax, bool scam,
                            std::vector<double> &Rfull, std::vector<double> &std, std::vector<double> *floored_cm = nullptr,
                                bool scaled = true)
{
    std::vector<double> G((size_t)d * d), V, sv;
    for (int j = 0; j < d; ++j) for (int i = 0; i < d;
        ++i) G[(size_t)j * d + i] = (i <= j) ? cm[(size_t)i + (size_t)j * d] : cm[(size_t)j + (size_t)i * d];
    host_symsvd(d, G, V, sv);
    if (sv[0] == 0.0) return d;
    const double tol = sv[0] / condmax;
    bool floored = false;
    if (sv[d - 1] <= tol) { floored = true; for (int i = 0; i < d; ++i) if (sv[i] < tol) sv[i] = tol; }
    Rfull.resize((size_t)d * d); std.assign(d, 0.0);
    if (scam) {
        Rfull = V;
        for (int i = 0; i < d; ++i) std[i] = std::sqrt(sv[i]);
    } else {
        const double sqd = std::sqrt((double)d);
        // R0 = U diag(sqrt(s))
        for (int i = 0; i < d; ++i) { double sq = std::sqrt(sv[i]); for (int k = 0; k < d;
            ++k) V[(size_t)i * d + k] = sq * V[(size_t)i * d + k]; }
        if (floored && floored_cm) {                    // covtor_svd info = -1: cmat = matmul(R0, transpose(R0)), matutils.F90:441-446
            floored_cm->assign((size_t)d * d, 0.0);
            for (int j = 0; j < d; ++j)
                for (int i = 0; i <= j; ++i) {
                    double acc = 0.0;
                    for (int k = 0; k < d; ++k) acc = std::fma(V[(size_t)k * d + i], V[(size_t)k * d + j], acc);
                    (*floored_cm)[(size_t)i + (size_t)j * d] = acc;
                }
        }
        for (size_t e = 0; e < (size_t)d * d; ++e) Rfull[e] = scaled ? V[e] * 2.4 / sqd : V[e];
    }
    return 0;
}

// dpotri('U') on the packed factor (dtrti2 + dlauu2), same operation sequence as the device's potri_packed
static int host_potri(int d, std::vector<double> &A)
{
    for (int j = 0; j < d; ++j) if (A[h_pidx(j, j, d)] == 0.0) return j + 1;
    std::vector<double> x(d);
    for (int j = 0; j < d; ++j) {
        double ajj = 1.0 / A[h_pidx(j, j, d)];
        A[h_pidx(j, j, d)] = ajj; ajj = -ajj;
        for (int i = 0; i < j; ++i) x[i] = A[h_pidx(i, j, d)];
        for (int jj = 0; jj < j; ++jj) {
            double temp = x[jj];
            if (temp != 0.0) {
                for (int i = 0; i < jj; ++i) x[i] = std::fma(temp, A[h_pidx(i, jj, d)], x[i]);
                x[jj] = temp * A[h_pidx(jj, jj, d)];
            }
        }
        for (int i = 0; i < j; ++i) A[h_pidx(i, j, d)] = ajj * x[i];
    }
    for (int i = 0; i < d; ++i) {
        double aii = A[h_pidx(i, i, d)];
        if (i < d - 1) {
            double dot = 0.0;
            for (int k = i; k < d; ++k) dot = std::fma(A[h_pidx(i, k, d)], A[h_pidx(i, k, d)], dot);
            A[h_pidx(i, i, d)] = dot;
            for (int r = 0; r < i; ++r) x[r] = aii * A[h_pidx(r, i, d)];
            for (int k = i + 1; k < d; ++k) {
                double temp = A[h_pidx(i, k, d)];
                if (temp != 0.0) for (int r = 0; r < i; ++r) x[r] = std::fma(temp, A[h_pidx(r, k, d)], x[r]);
            }
            for (int r = 0; r < i; ++r) A[h_pidx(r, i, d)] = x[r];
        } else {
            for (int r = 0; r <= i; ++r) A[h_pidx(r, i, d)] = aii * A[h_pidx(r, i, d)];
        }
    }
    return 0;
}
