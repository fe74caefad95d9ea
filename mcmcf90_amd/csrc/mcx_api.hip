// mcx_api.hip -- host side of libmcmcx.so: the C ABI of include/mcmcx.h.
//
// Owns the device state of N chains, cuts the iteration range into launches of the
// step kernel separated by the adaptation ticks of MCMC_adapt (MCMC_adapt.F90:40-46),
// and decodes the device history back into the reference's chain/sschain/s2chain
// form.  No CPU fallback: every numerical step of the sampler runs in the HIP kernels
// of mcx_kernels.hpp; without a GPU every entry point that needs one fails.
#include <hip/hip_runtime.h>
#include <csignal>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include "../../include/mcmcx.h"
#include "../../include/mcmcx_target.h"
#include "mcx_kernels.hpp"
#include "mcx_group.hpp"
#include "mcx_group_ram.hpp"

using namespace mcx;
static_assert(MCMCX_HE_INB == HE_INB && MCMCX_HE_PRI == HE_PRI && MCMCX_HE_SS == HE_SS && MCMCX_HX_STAGE2 == HX_STAGE2
    && MCMCX_HX_CRIT == HX_CRIT,
              "include/mcmcx_target.h and mcx_kernels.hpp disagree about the phase-state slots");

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(-100, std::string(#call) + ": " + hipGetErrorString(e_));                 \
    } while (0)

// The test switches (tests, tools): each FORCES one of two kernel forms that the engine also chooses by itself for some configuration -- so
// that the parity tests can run both forms on the same small problem.  None selects a form the engine would never take (those live in
// tools/variants, outside this library).  Read from the environment ONCE per engine, at mcmcx_create and again at mcmcx_init -- never at
// launch time.  -1 = not set.
struct mcx_switches {
    int pooled_mfma_dr_min = -1, pooled_scalar = -1, dr_big = -1, scam_pooled_16 = -1, scam_fast_lanes = -1, scam_waves = -1,
        svd_lane = -1, cov_batch_rows = -1, ram_wide = -1, pooled_waves = -1, cols_phased = -1, host_mapped = -1, host_fuse = -1;
    static int get(const char *name) { const char *e = getenv(name); return e ? atoi(e) : -1; }
    void read()
    {
        pooled_mfma_dr_min = get("MCMCX_POOLED_MFMA_DR_MIN"); pooled_scalar = get("MCMCX_POOLED_SCALAR"); dr_big = get("MCMCX_DR_BIG");
        scam_pooled_16 = get("MCMCX_SCAM_POOLED_16"); scam_fast_lanes = get("MCMCX_SCAM_FAST_LANES");
        scam_waves = get("MCMCX_SCAM_WAVES"); svd_lane = get("MCMCX_SVD_LANE"); cov_batch_rows = get("MCMCX_COV_BATCH_ROWS");
        ram_wide = get("MCMCX_RAM_WIDE"); pooled_waves = get("MCMCX_POOLED_WAVES");
        cols_phased = get("MCMCX_COLS_PHASED"); host_mapped = get("MCMCX_HOST_MAPPED"); host_fuse = get("MCMCX_HOST_FUSE");
    }
};
struct mcmcx_engine {
    mcmcx_config cfg;
    mcx_switches sw;
    int d = 0, P = 0, ntiles = 0, nlanes = 0;
    int dodr = 0, usesvd = 0;
    bool inited = false;
    int simuind = 0;
    const char *last_kernel = "";       // name of the sampling kernel launch_step / launch_scam chose last (mcmcx_last_kernel)
    std::string launch_err;             // set by a launcher that found no kernel for the configuration: the run fails with it
    // host copies of the problem
    std::vector<double> par0, cmat0;                 // cmat0 col-major d*d
    double sigma2 = 1.0; int nobs = 1; bool sigma2ok = false;
    int ny = 1; std::vector<double> sigma2v; std::vector<int> nobsv;      // nycol columns (host callbacks only when > 1)
    int tkind = -1, tncols = 1; std::vector<double> tmu, tlam, tx, ty, tlo, thi, tpmu, tpsig; double tb = 0.1;
    bool has_lo = false, has_hi = false, has_pri = false;
    mcmcx_ssfun_t h_ss = nullptr; mcmcx_ssfun_er_t h_ss_er = nullptr; mcmcx_priorfun_t h_pri = nullptr; mcmcx_checkbounds_t h_cb = nullptr;
        void *h_user = nullptr;
    mcmcx_ssfun_batch_t h_ss_batch = nullptr; int h_threads = 1;      // batched form of the user's ssfunction (opt-in)
    int mod_max_ny = 8;
    hipModule_t mod = nullptr; hipFunction_t mod_fn = nullptr; void *d_moddata = nullptr;   // user target module (include/mcmcx_target.h)
    std::vector<double> h_bth, h_bss; std::vector<int> h_bidx;
    // host side of the callback path: page-locked, so that the candidates come back and the results go out as asynchronous copies
    // on the engine's stream with ONE synchronisation per stage (pageable buffers cost a staged, blocking copy each way)
    struct Pinned {
        double *p = nullptr; size_t cap = 0, n = 0;
        int resize(size_t m) {
            n = m;
            if (m <= cap) return 0;
            if (p) (void)hipHostFree(p);
            p = nullptr; cap = 0;
            if (hipHostMalloc((void **)&p, m * sizeof(double), hipHostMallocDefault) != hipSuccess) return -1;
            cap = m; return 0;
        }
        double &operator[](size_t i) { return p[i]; }
        double *data() { return p; }
        size_t size() const { return n; }
        void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = n = 0; }
    } h_cand, h_ev, h_hx;
    // pooled mode
    int pool_status = 0; double pool_alpha = 0.0;       // pooled RAM: skipped ticks, mean acceptance of the last tick
    int pooled = 0; double pool_W = 0.0; std::vector<double> pool_mean, pool_C, pool_R;   // packed upper, row-major
    std::vector<double> pool_U, pool_std;             // pooled SCAM: the shared rotation (column-major) and qcovstd
    // pooled AM with condmax > 0: covtor_svd's full factor U sqrt(s) 2.4/sqrt(d), column-major
    std::vector<double> pool_Rf;
    // pooled mode with delayed rejection: R / drscale (packed, or full with condmax > 0) and dpotri(R) (packed)
    std::vector<double> pool_R2, pool_iC;
    double *d_sharedR2 = nullptr, *d_sharediC = nullptr;
    double *d_sharedU = nullptr;                      // [U col-major | pad | U row-major | pad | std]
    // host callbacks: the next iteration's proposal already ran in the previous iteration's last launch
    bool p0_done = false;
    // a host-callback iteration failed half way: the chains' stream positions are undefined, later runs are refused
    bool failed = false;
    // host callbacks with few chains: the exchange vectors live in page-locked host memory the device reads and writes directly (no copies
    // between the phases)
    bool host_mapped = false, cs_mapped = false;
    std::vector<void *> hallocs;
    // pooled SCAM above npar 240 (the tile kernels' LDS vector does not fit): the shared rotation copied to every chain, the per-chain
    // kernels run
    bool scam_replicated = false;
    // pooled AM on the matrix cores: dense R, M[s*d + o] = R(s,o), zero below the diagonal and in the pad rows
    double *d_sharedRT = nullptr;
    double *d_sharedR2T = nullptr, *d_sharediCd = nullptr;   // ... with delayed rejection: R2 in the same form, iC dense and symmetric
    double *d_sharedR = nullptr; mcmcx_exchange_t xfn = nullptr; void *xuser = nullptr; double *xbuf = nullptr;
    struct mcmcx_comm *comm = nullptr;                // the node's communicator (mcx_comm.hpp); nullptr = this GPU alone
    // [nranks][len + 1] per-rank moment vectors (+ the rank's stop flag), [len + 1] their tree sum
    double *d_gather = nullptr, *d_pooled = nullptr;
    double h_flag = 0.0;                              // this rank's stop flag of the exchange being enqueued (1 = a caught signal)
    bool run_entered = false;                         // mcmcx_run is past its argument checks (a failure from here on may strand peers)
    // the summed stop flag of the tick just applied was non-zero: every rank leaves after this tick
    bool stop_seen = false;
    double S02eff = 0.0;
    // device
    hipStream_t stream = nullptr; bool own_stream = false;
    EngineDev E{};
    std::vector<void *> allocs;
    double *d_ramscale = nullptr, *d_moments = nullptr;
    // blocked SVD of the adaptation (large npar)
    double *d_Gc = nullptr, *d_Vc = nullptr, *d_svc = nullptr; uint8_t *d_need = nullptr, *d_state = nullptr; int *d_anyrot = nullptr;
    int wcap = 0;
    bool tile_factor = false;           // the adaptation's Cholesky branch through tile_factor_kernel (npar <= 64)
    // method = 'ram' on group_ram_kernel (mcx_group_ram.hpp): npar rounded up to its instantiation, 0 = not
    int ram_group_d4 = 0;
    // lane-group step kernel (mcx_group.hpp): npar rounded up to four when it is the one to launch; accept bytes of a launch
    int group_d4 = 0, group_drm = 0, group_gw = 16; bool group_check_due = true; int *d_gflag = nullptr; uint8_t *d_accb = nullptr;
    // MCMC_run1: the caller evaluates; exchange vectors of run1_kernel
    bool external = false; double *d_r1 = nullptr; std::vector<double> h_r1;
    // timing of the step kernel
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double ms_total = 0.0; long long launches = 0, steps = 0;
};

struct DevBufs {                                       // hipFree on every exit path
    std::vector<void *> p;
    template <typename T> hipError_t alloc(T **q, size_t bytes) { void *v = nullptr; hipError_t e = hipMalloc(&v, bytes);
        if (e == hipSuccess) p.push_back(v); *q = (T *)v; return e; }
    ~DevBufs() { for (void *v : p) (void)hipFree(v); }
};

template <typename T>
static int dev_alloc(mcmcx_engine *h, T **p, size_t n, bool zero = true)
{
    void *q = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(-101, "hipMalloc of " + std::to_string(bytes) + " bytes: " + hipGetErrorString(e));
    if (zero) { e = hipMemsetAsync(q, 0, bytes, h->stream); if (e != hipSuccess) return fail(-101, hipGetErrorString(e)); }
    h->allocs.push_back(q);
    *p = (T *)q;
    return 0;
}

// page-locked host memory mapped into the device's address space (the same pointer on both sides)
template <typename T>
static int host_alloc(mcmcx_engine *h, T **p, size_t n)
{
    void *q = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    hipError_t e = hipHostMalloc(&q, bytes, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) return fail(-101, "hipHostMalloc of " + std::to_string(bytes) + " bytes: " + hipGetErrorString(e));
    memset(q, 0, bytes);
    h->hallocs.push_back(q);
    *p = (T *)q;
    return 0;
}

template <typename T>
static int dev_upload(mcmcx_engine *h, const T **p, const std::vector<T> &v)
{
    T *q = nullptr;
    int rc = dev_alloc(h, &q, v.size(), false);
    if (rc) return rc;
    if (!v.empty()) {
        hipError_t e = hipMemcpy(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(-102, hipGetErrorString(e));
    }
    *p = q;
    return 0;
}

// No limit on npar like the reference (MCMC_init.F90:81-102 allocates whatever the namelist says) -- beyond int-sized packed indices.  Up
// to 256 every kernel family applies; above, the forms that keep an npar-vector per lane in LDS give way to global scratch where the 160
// KiB end (delayed rejection > 160, the adaptation's work vector > 320, the pooled-moment kernel from 316 on), the blocked SVD to the
// lane-per-chain SVD (> 256), the matrix-core pooled kernels to the lane kernels (their own LDS tests): slower, never refused. response
// columns (mcmc.F90:30-33: whatever mcmcnycol.dat says): every per-column array is sized at mcmcx_init
static const int MCX_MAX_NYCOL = 4096;
// P = npar (npar + 1) / 2 = 8 390 656 at the cap, 64 P = 537 M and (2 npar + P) 64 = 538 M: every int-typed index expression of the device
// code (pidx, rowstart, e * 64 + lane) stays below 2**31 with a factor of four to spare -- at 8192, the cap up to round 5, 64 P is 2.147e9
// > INT_MAX and only the size_t casts of every current use site kept it correct.  Larger problems are refused loudly at
// mcmcx_create.
static const int MCX_MAX_NPAR = 4096;

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

// LDS of pooled_mfma_kernel: the tile's vector [d4][64] (+ the products [16 nt][64] when they need more than one pass)
// and the partial ss chains [4 nt][64]
static size_t pooled_mfma_lds(int d)
{
    const size_t d4 = (size_t)((d + 3) & ~3), nt = (size_t)((d + 15) / 16);
    // single pass: products and ss chains reuse the vector's rows
    const size_t rows = (nt <= 4) ? std::max(d4, 4 * nt) : d4 + 16 * nt + 4 * nt;
    return rows * 64 * sizeof(double);
}
static bool pooled_use_mfma(const mcmcx_engine *h)
{
    // (SCAM has its own)
    if (!h->pooled || (h->cfg.method != MCMCX_METHOD_DRAM && h->cfg.method != MCMCX_METHOD_RAM
        && h->cfg.method != MCMCX_METHOD_ER)) return false;
    // DR: with its dense tables, and above
    if (h->dodr) {
        // npar 20 (8.3e8 against 8.9e8 iterations/s
        if (h->cfg.method != MCMCX_METHOD_DRAM || !h->d_sharedR2T) return false;
        // for the lane kernel with its LDS vectors at 20;
        int dmin = 21;
        // 32: 5.6e8 / 2.5e8, 50: 3.2e8 / 0.8e8)
        if (h->sw.pooled_mfma_dr_min >= 0) dmin = h->sw.pooled_mfma_dr_min;
        if (h->d < dmin) return false;
    }
    if (h->sw.pooled_scalar > 0) return false;                                         // A/B switch for tests: the lane-per-chain kernel
    return pooled_mfma_lds(h->d) <= 160 * 1024;
}
// iteration cut at the evaluations
static bool phased(const mcmcx_engine *h) { return h->tkind == TGT_HOST || h->tkind == TGT_EXPCOLS || h->tkind == TGT_MODULE; }
// ... except where the DEVICE evaluates between the phases (the response-column target): the phases fused into one launch per segment
// (step_kernel_cols); MCMCX_COLS_PHASED=1 keeps the separate launches (A/B, tests)
static bool fused_cols(const mcmcx_engine *h) { return h->tkind == TGT_EXPCOLS && !(h->sw.cols_phased > 0); }
static bool phase_cut(const mcmcx_engine *h) { return phased(h) && !fused_cols(h); }
static size_t lds_bytes(const mcmcx_engine *h) { return (size_t)h->d * 64 * sizeof(double) * 2; }   // adapt / DR work vectors
static bool dr_fits_lds(const mcmcx_engine *h) { return lds_bytes(h) <= 160 * 1024; }          // npar <= 160
static size_t lds_step(const mcmcx_engine *h) { return (h->dodr && dr_fits_lds(h)) ? lds_bytes(h) : 0; }
// step_kernel_dr keeps the second stage's two vectors in LDS, step_kernel_dr_big in global scratch.  LDS pays while eight waves
// still fit a CU (npar <= 20: 6.9e8 against 6.0e8 iterations/s at 20); beyond, the waves it costs are worth more than the bytes
// it saves (npar 24: 3.1e8 against 4.1e8, 64: 3.2e7 / 4.6e7, 100: 0.8e7 / 1.7e7 -- tools/dr_sweep.py)
// (the pooled form, whose factors come through the scalar cache, is bound by that latency rather than by waves: LDS down to four
//  waves per CU -- npar 32: 2.5e8 against 2.2e8, 50: 7.4e7 / 8.1e7, 100: 0.8e7 / 1.3e7 -- tools/pooled_dr_probe.py)
static bool dr_vectors_in_lds(const mcmcx_engine *h, int min_waves = 8)
{
    if (!dr_fits_lds(h)) return false;
    if (h->sw.dr_big >= 0) return h->sw.dr_big == 0;                                  // A/B switch for tests: 1 = global scratch, 0 = LDS
    return lds_bytes(h) * (size_t)min_waves <= 160 * 1024;
}
static void launch_init(mcmcx_engine *h)
{ hipLaunchKernelGGL(init_kernel, dim3(h->ntiles), dim3(64), 0, h->stream, h->E); }
// ---- which sampling kernel runs: ONE table per launcher, walked in order -- the first entry whose predicate holds is launched and
// its name noted for mcmcx_last_kernel (bench.py labels its roofline with it).  The tables are exported through
// mcmcx_debug_kernel_table, so that tests/test_kernel_table.py can list every selectable instance and require a parity test that
// asserted each of them.  A configuration no entry accepts is an error of the run (launch_err), never a silent no-launch.
struct KernelEntry {
    const char *family;                                   // "step" (launch_step), "group" (launch_group), "scam" (launch_scam)
    const char *name;
    bool (*when)(const mcmcx_engine *);
    void (*launch)(mcmcx_engine *, int it0, int it1);
};
static void walk_table(mcmcx_engine *h, const KernelEntry *tab, size_t n, int it0, int it1)
{
    for (size_t i = 0; i < n; ++i)
        if (tab[i].when(h)) { h->last_kernel = tab[i].name; tab[i].launch(h, it0, it1); return; }
    h->launch_err = std::string("no ") + (n ? tab[0].family : "?") + " kernel covers this configuration";
}
// ---- the lane-group step kernel (mcx_group.hpp): four chains per wave, factors in registers
static const int GROUP_MAXSEG = 256;                  // iterations per launch (one accept byte per chain and iteration in d_accb)
static const int GROUP_MAX_NPAR = 64, GROUP_MAX_NPAR_DR = 32;      // (with delayed rejection three tables must fit a lane's registers)
// what the kernel covers: MCMC_run with per-chain Cholesky factors (method 'dram', with or without delayed rejection), one of
// the single-launch device targets, one response column
static bool group_covers(const mcmcx_engine *h)
{
    const mcmcx_config &c = h->cfg;
    return !h->pooled && (c.method == MCMCX_METHOD_DRAM || (c.method == MCMCX_METHOD_ER && !h->dodr)) && !h->usesvd && !phased(h)
        && h->ny == 1 &&
           (h->tkind == TGT_GAUSS || h->tkind == TGT_BANANA || h->tkind == TGT_EXPDATA) && h->d <= (h->dodr ? GROUP_MAX_NPAR_DR
               : GROUP_MAX_NPAR) &&
           !(h->tkind == TGT_BANANA && h->d < 2) && !(h->tkind == TGT_EXPDATA && h->d < 2);
}
// ... and where it is the faster one (tools/group_sweep.py, profiles/r04_a/group_sweep.txt: proposals/s of both kernel families over npar,
// target, delayed rejection and chain count).  Up to 16384 chains always: the chip is not full, a chain's iteration is latency, and
// sixteen lanes per chain with the factors on chip take 2-5 us where a lane takes 7-160 (4x-33x).  With the chip full, from npar 11
// on: 1.04-1.3x without delayed rejection up to npar 20 and 1.7-3x above, 1.1-3.7x with it; at npar <= 10 the lane kernels, which keep
// the factor in LDS there, stay ahead (group: 0.35-0.9x).
// ... and with which group width: four lanes per chain (sixteen chains per wave) for small npar with the chip full, where sixteen lanes
// would mostly idle (tools/quad_sweep.py, profiles/r04_b/quad_sweep.txt: without delayed rejection 1.3-3.6x the sixteen-lane form at npar
// <= 16 and 1.1-3.2x the lane kernels up to 131072 chains -- 1.5-1.7x at any count from npar 11 on; with it at npar <= 8)
static int group_width(const mcmcx_engine *h)
{
    if ((long long)h->cfg.nchains <= 16384 || h->d > 16 || h->cfg.updatesigma) return 16;
    if (!h->dodr) return 4;
    return h->d <= 8 ? 4 : 16;
}
static bool group_wins(const mcmcx_engine *h, int drm, int gw)
{
    const long long n = h->cfg.nchains;
    const int d = h->d;
    if (h->cfg.updatesigma) {
        // MCMC_updatesigma2's gamma sampler is a serial, data-dependent sequence of draws per chain: a group wave runs it for four chains,
        // a lane wave for 64 (tools/group_probe2.py: 2.3-12x up to 1024 chains, 0.9-5x at 16384, 0.45-0.8x beyond without delayed
        // rejection, 1.3-1.6x with it at npar 20)
        if (n <= 8192) return true;
        if (n <= 16384) return d >= 4;
        return h->dodr && d >= 11;
    }
    if (n <= 16384) return true;
    if (gw == 4) return h->dodr ? (drm == 2 || n <= 131072) : (d >= 11 || n <= 131072);
    return d >= 11 || (h->dodr && d >= 9 && n <= 131072);
}
// one instantiation per (group width, npar rounded up, delayed-rejection form, target kind); DRM 0 = none, 1 = the general form (R, R2, iC
// in registers), 2 = drscale a power of two (no R2; iC in LDS): the instantiation without R2 runs unless the device flag says that some
// factor leaves the range in which R'z / drscale is R2'z bit for bit; the general one is queued behind the same flag and returns at once
// otherwise
template <int GW, int D4, int DRM, int TK>
static void launch_group_inst(mcmcx_engine *h, int it0, int it1)
{
    const dim3 g(h->ntiles * (GW == 16 ? 16 : 4)), b(64);           // 64 / GW chains per wave
    const double *lam = h->E.tgt.lamT;
    if constexpr (DRM != 0 && D4 > GROUP_MAX_NPAR_DR) h->launch_err = "group kernel: no delayed-rejection instantiation above npar " +
        std::to_string(GROUP_MAX_NPAR_DR);
    else if constexpr (TK == TGT_EXPDATA && D4 != 4) h->launch_err = "group kernel: the expdata target has two parameters";
    else if constexpr (DRM == 0) hipLaunchKernelGGL((group_step_kernel<GW, D4, 0, TK>), g, b, 0, h->stream, h->E, it0, it1, lam, h->d_accb,
        (const int *)nullptr, 0);
    else if constexpr (DRM == 1) hipLaunchKernelGGL((group_step_kernel<GW, D4, 1, -1>), g, b, 0, h->stream, h->E, it0, it1, lam, h->d_accb,
        (const int *)nullptr, 0);
    else {
        hipLaunchKernelGGL((group_step_kernel<GW, D4, 2, TK>), g, b, 0, h->stream, h->E, it0, it1, lam, h->d_accb, (const int *)h->d_gflag,
            0);
        hipLaunchKernelGGL((group_step_kernel<GW, D4, 1, -1>), g, b, 0, h->stream, h->E, it0, it1, lam, h->d_accb, (const int *)h->d_gflag,
            1);
    }
}
template <int GW, int D4, int DRM>
static void launch_group_tk(mcmcx_engine *h, int it0, int it1)
{
    if (h->tkind == TGT_BANANA) launch_group_inst<GW, D4, DRM, TGT_BANANA>(h, it0, it1);
    else if (h->tkind == TGT_EXPDATA) launch_group_inst<GW, D4, DRM, TGT_EXPDATA>(h, it0, it1);
    else launch_group_inst<GW, D4, DRM, TGT_GAUSS>(h, it0, it1);
}
template <int GW, int DRM>
static void launch_group_d4(mcmcx_engine *h, int it0, int it1)
{
    if constexpr (GW == 4) {                              // quads: npar <= 16
        switch (h->group_d4) {
        case 4: launch_group_tk<4, 4, DRM>(h, it0, it1); break;
        case 8: launch_group_tk<4, 8, DRM>(h, it0, it1); break;
        case 12: launch_group_tk<4, 12, DRM>(h, it0, it1); break;
        case 16: launch_group_tk<4, 16, DRM>(h, it0, it1); break;
        default: h->launch_err = "group kernel (quads): npar > 16";
        }
    } else {
        switch (h->group_d4) {
        case 4: launch_group_tk<16, 4, DRM>(h, it0, it1); break;
        case 8: launch_group_tk<16, 8, DRM>(h, it0, it1); break;
        case 12: launch_group_tk<16, 12, DRM>(h, it0, it1); break;
        case 16: launch_group_tk<16, 16, DRM>(h, it0, it1); break;
        case 20: launch_group_tk<16, 20, DRM>(h, it0, it1); break;
        case 24: launch_group_tk<16, 24, DRM>(h, it0, it1); break;
        default:
            // DRM = 2 (iC as a square in LDS) is the engine's choice up to npar 24 only (mcmcx_init): larger sizes are not instantiated --
            // at 28 / 32 they could not hold the two waves per SIMD they would declare (Weak 12)
            if constexpr (DRM != 2) {
                switch (h->group_d4) {
                case 28: launch_group_tk<16, 28, DRM>(h, it0, it1); return;
                case 32: launch_group_tk<16, 32, DRM>(h, it0, it1); return;
                default: break;
                }
            }
            if constexpr (DRM == 0) {                                    // (above 32: sizes of eight, no delayed rejection)
                switch (h->group_d4) {
                case 40: launch_group_tk<16, 40, DRM>(h, it0, it1); return;
                case 48: launch_group_tk<16, 48, DRM>(h, it0, it1); return;
                case 56: launch_group_tk<16, 56, DRM>(h, it0, it1); return;
                case 64: launch_group_tk<16, 64, DRM>(h, it0, it1); return;
                default: break;
                }
            }
            h->launch_err = "group kernel: no instantiation for npar " + std::to_string(h->d) + " with delayed-rejection mode " +
                std::to_string(DRM);
        }
    }
}
static const KernelEntry GROUP_TABLE[] = {
    {"group", "group_step_kernel",            [](const mcmcx_engine *h) { return h->group_gw == 16 && h->group_drm == 0; },
        launch_group_d4<16, 0>},
    {"group", "group_step_kernel<DR>",        [](const mcmcx_engine *h) { return h->group_gw == 16 && h->group_drm == 1; },
        launch_group_d4<16, 1>},
    {"group", "group_step_kernel<DR2>",       [](const mcmcx_engine *h) { return h->group_gw == 16 && h->group_drm == 2; },
        launch_group_d4<16, 2>},
    {"group", "group_step_kernel<quad>",      [](const mcmcx_engine *h) { return h->group_gw == 4 && h->group_drm == 0; },
        launch_group_d4<4, 0>},
    {"group", "group_step_kernel<quad, DR>",  [](const mcmcx_engine *h) { return h->group_gw == 4 && h->group_drm == 1; },
        launch_group_d4<4, 1>},
    {"group", "group_step_kernel<quad, DR2>", [](const mcmcx_engine *h) { return h->group_gw == 4 && h->group_drm == 2; },
        launch_group_d4<4, 2>},
};
static void launch_group(mcmcx_engine *h, int it0, int it1)
{
    if (h->group_drm == 2 && h->group_check_due) {       // the factors have been rewritten since the last look
        (void)hipMemsetAsync(h->d_gflag, 0, sizeof(int), h->stream);
        hipLaunchKernelGGL(group_check_kernel, dim3(h->ntiles), dim3(64), 0, h->stream, h->E, h->d_gflag);
        h->group_check_due = false;
    }
    walk_table(h, GROUP_TABLE, sizeof(GROUP_TABLE) / sizeof(GROUP_TABLE[0]), it0, it1);
    if (h->d_accb && h->launch_err.empty()) {
        const long long n = (long long)(it1 - it0 + 1) * h->ntiles;
        hipLaunchKernelGGL(group_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->E, h->d_accb, it0, it1);
    }
}
// pooled_mfma_kernel<false, true>: two waves per SIMD (256 registers, some state spilled) pay with more tiles than SIMDs: from two per SIMD
// where the LDS vector lets eight waves on a CU (npar <= 39: +25 .. +55 %), from eight where it lets six (npar 50: 0.80 at 2048 tiles, 0.99
// at 4096, 1.14 at 16384); with one tile per SIMD the spills are all they buy (0.83 .. 0.90) -- tools/pooled_waves_probe.py
static bool pooled_two_waves(const mcmcx_engine *h)
{
    // (test switch: either instance on a small problem)
    if (h->sw.pooled_waves == 1 || h->sw.pooled_waves == 2) return h->sw.pooled_waves == 2;
    return h->ntiles >= (pooled_mfma_lds(h->d) * 8 <= 160 * 1024 ? 2048 : 8192);
}
#define STEP_ARGS h->stream, h->E, it0, it1
#define STEP_RS (h->d_ramscale + it0)
#define STEP_TGT h->E.tgt.mu, h->E.tgt.lamT
#define G1 dim3(h->ntiles), dim3(64)
// method = 'ram' with few chains: sixteen lanes per chain, the factor in registers, dchud / dchdd on it there (mcx_group_ram.hpp) where it
// is the faster one (tools/ram_group_sweep.py, profiles/r05_b/ram_group_sweep.txt: chain-iterations/s of both families over npar, chain
// count and regime): one wave per SIMD and four chains per wave, so the chip holds 4096 chains at once and the kernel saturates there
// (1.68e8 / 4.6e8 / 1.1e9 chain-iterations/s at npar 50 / 20 / 10) -- 9x / 6x / 6x the lane kernels up to 4096 chains, still 2.9x / 1.7x /
// 2.0x at 16384 and 1.6x / 1.1x / 1.2x at 32768; from 65536 chains on the streaming kernels are ahead (0.93 / 0.74 / 0.63)
static bool ram_group_wins(const mcmcx_engine *h)
{
    const long long n = h->cfg.nchains;
    return n <= 16384 || (n <= 32768 && h->d >= 17);
}
static bool ram_group_covers(const mcmcx_engine *h)
{
    return !h->pooled && h->cfg.method == MCMCX_METHOD_RAM && !h->usesvd && !phased(h) && h->ny == 1 && h->d <= 64 &&
           (h->tkind == TGT_GAUSS || h->tkind == TGT_BANANA || h->tkind == TGT_EXPDATA) && !(h->tkind == TGT_BANANA && h->d < 2);
}
template <int D4>
static void launch_group_ram_d4(mcmcx_engine *h, int it0, int it1)
{
    hipLaunchKernelGGL((group_ram_kernel<D4, -1>), dim3(h->ntiles * 16), dim3(64), 0, h->stream, h->E, it0, it1,
        (const double *)h->d_ramscale, h->E.tgt.lamT, h->d_accb);
}
static void launch_group_ram(mcmcx_engine *h, int it0, int it1)
{
    switch (h->ram_group_d4) {
    case 16: launch_group_ram_d4<16>(h, it0, it1); break;
    case 32: launch_group_ram_d4<32>(h, it0, it1); break;
    case 56: launch_group_ram_d4<56>(h, it0, it1); break;
    case 64: launch_group_ram_d4<64>(h, it0, it1); break;
    default: h->launch_err = "group_ram_kernel: no instantiation for npar " + std::to_string(h->d); return;
    }
    if (h->d_accb) {
        const long long n = (long long)(it1 - it0 + 1) * h->ntiles;
        hipLaunchKernelGGL(group_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->E, h->d_accb, it0, it1);
    }
}
#ifdef MCX_VARIANTS                // measured negatives, built only by tools/build_variant.sh -DMCX_VARIANTS -- never part of libmcmcx.so
#include "../../tools/variants/variants.inc"
#else
#define MCX_VARIANT_STEP_ENTRIES
#define MCX_VARIANT_SVD_SWEEP(h, lss) false
#define MCX_VARIANT_COV(h, g8, n10, noff, it, mode) false
#endif
static const KernelEntry STEP_TABLE[] = {
    // ---- a device target with response columns (nycol >= 1 sums of squares per point): the phases of an iteration in one launch
    // (one instantiation per method class: MCMC_run_ram's carries the rank-one update's panels, the others do not)
    {"step", "step_kernel_cols<ram>", [](const mcmcx_engine *h) { return fused_cols(h) && h->E.method == M_RAM; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_cols<1>, G1, lds_step(h), STEP_ARGS, (const double *)h->d_ramscale,
                                                                (const double *)nullptr, (const double *)nullptr, (const double *)nullptr); }},
    {"step", "step_kernel_cols", fused_cols,
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_cols<0>, G1, lds_step(h), STEP_ARGS, (const double *)h->d_ramscale,
                                                                (const double *)(h->pooled ? h->E.sharedR : nullptr),
                                                                (const double *)h->d_sharedR2, (const double *)h->d_sharediC); }},
    // ---- pooled mode (one shared factor)
    {"step", "pooled_mfma_kernel<true>", [](const mcmcx_engine *h) { return pooled_use_mfma(h) && h->dodr != 0; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(pooled_mfma_kernel<true>, G1, pooled_mfma_lds(h->d), STEP_ARGS, STEP_TGT,
         h->d_sharedRT, h->d_sharedR2T, h->d_sharediCd); }},
    MCX_VARIANT_STEP_ENTRIES
    {"step", "pooled_mfma_kernel<false, true>", [](const mcmcx_engine *h) { return pooled_use_mfma(h) && pooled_two_waves(h); },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL((pooled_mfma_kernel<false, true>), G1, pooled_mfma_lds(h->d), STEP_ARGS,
         STEP_TGT, h->d_sharedRT, (const double *)nullptr, (const double *)nullptr); }},
    {"step", "pooled_mfma_kernel<false>", [](const mcmcx_engine *h) { return pooled_use_mfma(h); },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(pooled_mfma_kernel<false>, G1, pooled_mfma_lds(h->d), STEP_ARGS, STEP_TGT,
         h->d_sharedRT, (const double *)nullptr, (const double *)nullptr); }},
    {"step", "step_kernel_pooled_dr_big", [](const mcmcx_engine *h) { return h->pooled && h->dodr && !dr_vectors_in_lds(h, 4); },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_pooled_dr_big, G1, 0, STEP_ARGS, STEP_RS, STEP_TGT,
         h->E.sharedR, h->d_sharedR2, h->d_sharediC); }},
    {"step", "step_kernel_pooled_dr", [](const mcmcx_engine *h) { return h->pooled && h->dodr; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_pooled_dr, G1, lds_step(h), STEP_ARGS, STEP_RS, STEP_TGT,
         h->E.sharedR, h->d_sharedR2, h->d_sharediC); }},
    {"step", "step_kernel<false, false, true>", [](const mcmcx_engine *h) { return h->pooled != 0; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL((step_kernel<false, false, true>), G1, 0, STEP_ARGS, STEP_RS, STEP_TGT,
         h->E.sharedR); }},
    // ---- method = 'ram', per-chain factors
    {"step", "group_ram_kernel", [](const mcmcx_engine *h) { return h->ram_group_d4 != 0; }, launch_group_ram},
    {"step", "step_kernel_ram_fullr", [](const mcmcx_engine *h) { return h->E.method == M_RAM && h->usesvd; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_ram_fullr, G1, 0, STEP_ARGS, STEP_RS, STEP_TGT, h->E.sharedR);
         }},
    {"step", "step_kernel_ram_ldsr", [](const mcmcx_engine *h) { return h->E.method == M_RAM && h->E.lds_scratch == 3; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_ram_ldsr, G1, (size_t)(2 * h->d + h->P) * 64 * sizeof(double),
         STEP_ARGS, STEP_RS, STEP_TGT, h->E.sharedR); }},
    {"step", "step_kernel_ram_wide", [](const mcmcx_engine *h) {
        return h->E.method == M_RAM && h->d > RAM_SMALL_MAX && h->sw.ram_wide != 0; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_ram_wide, G1, (size_t)NLC * 2 * 64 * sizeof(double), STEP_ARGS,
         STEP_RS, STEP_TGT, h->E.sharedR); }},
    {"step", "step_kernel<true, false, false>", [](const mcmcx_engine *h) { return h->E.method == M_RAM; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL((step_kernel<true, false, false>), G1,
         (size_t)NLC * 2 * 64 * sizeof(double), STEP_ARGS, STEP_RS, STEP_TGT, h->E.sharedR); }},
    // ---- delayed rejection, per-chain factors: the second stage's two vectors in global scratch / in LDS
    {"step", "step_kernel_dr_big", [](const mcmcx_engine *h) { return h->dodr && !dr_vectors_in_lds(h); },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_dr_big, G1, 0, STEP_ARGS, STEP_TGT); }},
    {"step", "step_kernel_dr", [](const mcmcx_engine *h) { return h->dodr != 0; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_dr, G1, lds_step(h), STEP_ARGS, STEP_TGT); }},
    // ---- AM / Metropolis / early rejection: state + factor in LDS, state in LDS, nothing in LDS
    {"step", "step_kernel_ldsr", [](const mcmcx_engine *h) { return h->E.lds_scratch == 2; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_ldsr, G1, (size_t)(2 * h->d + h->P) * 64 * sizeof(double),
         STEP_ARGS, STEP_RS, STEP_TGT, h->E.sharedR); }},
    {"step", "step_kernel_ldsv", [](const mcmcx_engine *h) { return h->E.lds_scratch != 0; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_ldsv, G1, (size_t)4 * h->d * 64 * sizeof(double), STEP_ARGS,
         STEP_RS, STEP_TGT, h->E.sharedR); }},
    {"step", "step_kernel<false, false, false>", [](const mcmcx_engine *) { return true; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL((step_kernel<false, false, false>), G1, 0, STEP_ARGS, STEP_RS, STEP_TGT,
         h->E.sharedR); }},
};
static void launch_step(mcmcx_engine *h, int it0, int it1)
{
    if (h->group_d4) { launch_group(h, it0, it1); return; }
    walk_table(h, STEP_TABLE, sizeof(STEP_TABLE) / sizeof(STEP_TABLE[0]), it0, it1);
}
// every chain's copy of a K-vector, filled on the device
static int dev_bcast(mcmcx_engine *h, double *dst, const std::vector<double> &v)
{
    double *tmp = nullptr;
    HIPCHK(hipMalloc(&tmp, v.size() * sizeof(double)));
    hipError_t e = hipMemcpy(tmp, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const unsigned gy = (unsigned)std::min<size_t>(v.size(), 64);
        hipLaunchKernelGGL(bcast_kernel, dim3(h->ntiles, gy), dim3(64), 0, h->stream, dst, tmp, v.size());
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(-100, hipGetErrorString(e));
    return 0;
}
static int upload_shared_rt(mcmcx_engine *h)
{
    const int d = h->d, d4 = (d + 3) & ~3;
    std::vector<double> m((size_t)d4 * d + PWS, 0.0);
    for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) m[(size_t)i * d + j] = h->pool_R[h_pidx(i, j, d)];
    HIPCHK(hipMemcpyAsync(h->d_sharedRT, m.data(), m.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
static int upload_shared_rf(mcmcx_engine *h)            // dense M[s*d + o] = Rf(o, s): the column-major factor as it stands, pad rows zero
{
    const int d = h->d, d4 = (d + 3) & ~3;
    std::vector<double> m((size_t)d4 * d + PWS, 0.0);
    memcpy(m.data(), h->pool_Rf.data(), (size_t)d * d * 8);
    HIPCHK(hipMemcpyAsync(h->d_sharedRT, m.data(), m.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
// d4 rows (pad rows zero) + slack
static size_t shared_u_stride(const mcmcx_engine *h) { return (size_t)((h->d + 3) & ~3) * h->d + PWS; }
// X [16 nt][64], Q [4 nt][64], zb, fl, mu [16 nt]
static size_t scam_pooled_lds(int d) {
    return ((size_t)((d + 15) / 16) * (16 + 4) * 64 + 128 + (size_t)((d + 15) / 16) * 16) * sizeof(double); }
static int upload_shared_u(mcmcx_engine *h)
{
    if (h->scam_replicated) {                           // every chain's own copy of the one rotation and its scales
        int rc = dev_bcast(h, h->E.Rf, h->pool_U); if (rc) return rc;
        return dev_bcast(h, h->E.qstd, h->pool_std);
    }
    const int d = h->d; const size_t st = shared_u_stride(h);
    std::vector<double> b(2 * st + d, 0.0);
    for (int j = 0; j < d; ++j) for (int i = 0; i < d; ++i) { b[(size_t)j * d + i] = h->pool_U[(size_t)j * d + i];
        b[st + (size_t)i * d + j] = h->pool_U[(size_t)j * d + i]; }
    for (int i = 0; i < d; ++i) b[2 * st + i] = h->pool_std[i];
    HIPCHK(hipMemcpyAsync(h->d_sharedU, b.data(), b.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
// 13..15 output blocks (npar 193..240): twelve waves of 170 registers (scam_pooled12_kernel) instead of sixteen of 128
static bool scam_use_12(const mcmcx_engine *h)
{
    const int nt = (h->d + 15) / 16;
    if (h->sw.scam_pooled_16 > 0) return false;                                       // A/B switch for tests: the sixteen-wave layout
    return nt >= 13 && nt <= 15;
}
// opt-in fast proposals with the Gaussian target: the workgroup-per-tile kernel of the pooled mode with each chain's own
// rotation column (g_U = nullptr) -- the target's d x d product on the matrix cores instead of lane by lane
static bool scam_fast_tile_kernel(const mcmcx_engine *h)
{
    return h->cfg.scam_fast && h->tkind == TGT_GAUSS && !h->has_lo && !h->has_hi && !h->has_pri && scam_pooled_lds(h->d) <= 160 * 1024 &&
           !(h->sw.scam_fast_lanes > 0);
}
// few tiles: several waves per tile (scam_mw_kernel), so that a sub-step is not bound by the latency of one wave's loads
// while most of the chip idles -- as many waves as keep the chip's ~2048 resident-wave slots busy, at most 8 (sixteen
// waves of 128 registers spill the products' panels)
static size_t scam_mw_lds(const mcmcx_engine *h) { return (size_t)(4 * ((h->d + 15) / 16) + 2) * 64 * sizeof(double); }
static int scam_tile_waves(const mcmcx_engine *h)
{
    int nw = 1;
    // (1024 tiles x npar 200: 8.4e6 at two waves per tile, 8.9e6 at four, 9.0e6 at eight -- profiles/r05_d/c5rep_waves.txt)
    while (nw < 8 && (long long)h->ntiles * nw * 2 <= 8192) nw *= 2;
    { const int v = h->sw.scam_waves; if (v == 1 || v == 2 || v == 4 || v == 8) nw = v; }   // A/B switch for tests
    // (npar > 1200: the one-wave kernel needs no LDS)
    if (scam_mw_lds(h) > 160 * 1024) nw = 1;
    return nw;
}
#define SCAM_POOLED_ARGS scam_pooled_lds(h->d), h->stream, h->E, it0, it1, h->E.tgt.mu, h->E.tgt.lamT
static const KernelEntry SCAM_TABLE[] = {
    {"scam", "step_kernel_cols<scam>", [](const mcmcx_engine *h) { return fused_cols(h) && !h->pooled; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(step_kernel_cols<2>, G1, 0, STEP_ARGS, (const double *)h->d_ramscale,
         (const double *)nullptr, (const double *)nullptr, (const double *)nullptr); }},
    // pooled: 16-row output blocks, min(12, 4*(nt/4)) block waves + 4 chain-group waves; twelve waves of 170 registers for 13..15 blocks
    {"scam", "scam_pooled12_kernel", [](const mcmcx_engine *h) { return h->pooled && !h->scam_replicated && scam_use_12(h); },
     [](mcmcx_engine *h, int it0, int it1) { const size_t st = shared_u_stride(h);
        hipLaunchKernelGGL(scam_pooled12_kernel, dim3(h->ntiles), dim3(768), SCAM_POOLED_ARGS, h->d_sharedU, h->d_sharedU + st,
            h->d_sharedU + 2 * st); }},
    {"scam", "scam_pooled_kernel", [](const mcmcx_engine *h) { return h->pooled && !h->scam_replicated; },
     [](mcmcx_engine *h, int it0, int it1) { const size_t st = shared_u_stride(h); const int nw = 4 + std::min(12, ((h->d + 15) / 16) & ~3);
        hipLaunchKernelGGL(scam_pooled_kernel, dim3(h->ntiles), dim3(64 * nw), SCAM_POOLED_ARGS, h->d_sharedU, h->d_sharedU + st,
            h->d_sharedU + 2 * st); }},
    // (the sixteen-wave layout whatever npar: every lane fetches the column of its own chain's factor per sub-step, and four
    //  waves per SIMD cover that better than three -- 4.62e8 against 4.45e8 proposals/s at npar 200)
    {"scam", "scam_pooled_kernel<per-chain>", scam_fast_tile_kernel,
     [](mcmcx_engine *h, int it0, int it1) { const int nw = 4 + std::min(12, ((h->d + 15) / 16) & ~3);
        hipLaunchKernelGGL(scam_pooled_kernel, dim3(h->ntiles), dim3(64 * nw), SCAM_POOLED_ARGS, (const double *)nullptr,
            (const double *)nullptr, (const double *)nullptr); }},
    {"scam", "scam_mw_kernel<8>", [](const mcmcx_engine *h) { return scam_tile_waves(h) == 8; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(scam_mw_kernel<8>, dim3(h->ntiles), dim3(512), scam_mw_lds(h), STEP_ARGS,
         STEP_TGT); }},
    {"scam", "scam_mw_kernel<4>", [](const mcmcx_engine *h) { return scam_tile_waves(h) == 4; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(scam_mw_kernel<4>, dim3(h->ntiles), dim3(256), scam_mw_lds(h), STEP_ARGS,
         STEP_TGT); }},
    {"scam", "scam_mw_kernel<2>", [](const mcmcx_engine *h) { return scam_tile_waves(h) == 2; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(scam_mw_kernel<2>, dim3(h->ntiles), dim3(128), scam_mw_lds(h), STEP_ARGS,
         STEP_TGT); }},
    {"scam", "scam_kernel", [](const mcmcx_engine *) { return true; },
     [](mcmcx_engine *h, int it0, int it1) { hipLaunchKernelGGL(scam_kernel, G1, 0, STEP_ARGS, STEP_TGT); }},
};
static void launch_scam(mcmcx_engine *h, int it0, int it1)
{
    walk_table(h, SCAM_TABLE, sizeof(SCAM_TABLE) / sizeof(SCAM_TABLE[0]), it0, it1);
}
// the adaptation's SVD one workgroup per chain (mcx_svd.hpp) where its rings and row groups are instantiated
static bool svd_blocked(const mcmcx_engine *h)
{
    if (!h->usesvd || h->pooled || h->cfg.method == MCMCX_METHOD_RAM) return false;
    // test switch: the lane SVD (the engine's form below npar 48 and above 256)
    if (h->sw.svd_lane > 0) return false;
    // (its rings and row groups are instantiated up to npar 256)
    return h->d >= 48 && h->d <= 256;
}
static void launch_adapt(mcmcx_engine *h, int it, int mode)
{
    // one d-vector / the Cholesky's diagonal block (18 kB: eight waves per CU up to npar 36)
    const size_t lds = std::max(lds_bytes(h) / 2, (size_t)36 * 64 * sizeof(double));
    hipLaunchKernelGGL(adapt_pre_kernel, dim3(h->ntiles), dim3(64), 0, h->stream, h->E, it, mode);
    // covmat's batch branch in blocks (adapt_covb_*): the AP window at every adaptation; with initcmatn = 0 the first AM adaptation and
    // the greedy restarts.  Which lanes take it is the lanes' own business (ADF_BATCH); a tick that cannot hold any skips the launches.
    const bool ap = (mode & AD_AM) && h->cfg.adapthist > 1;
    // (test switch: covmat_rows, the lane form)
    const int batch_done = (!(h->sw.cov_batch_rows > 0) &&
                            (ap || (h->cfg.initcmatn == 0 && ((mode & AD_FIRST) || ((mode & AD_BURN) && h->cfg.greedy != 0))))) ? 1 : 0;
    if (batch_done) {
        const int n10 = (h->d + TD - 1) / TD, noff = n10 * (n10 - 1) / 2;
        const unsigned g8 = (unsigned)(8 * ((h->ntiles + 7) / 8));
        hipLaunchKernelGGL(adapt_covb_diag_kernel, dim3(g8 * n10), dim3(64), 0, h->stream, h->E, it, n10);
        if (noff > 0) hipLaunchKernelGGL(adapt_covb_off_kernel, dim3(g8 * noff), dim3(64), 0, h->stream, h->E, it, noff);
    }
    if (!((mode & AD_AM) && h->cfg.adapthist > 1)) {                      // the AP window is a batch recompute: no blocked update
        // blocks of ten (triangular on the diagonal)
        const int n10 = (h->d + TD - 1) / TD, noff = n10 * (n10 - 1) / 2;
        const unsigned g8 = (unsigned)(8 * ((h->ntiles + 7) / 8));
        if (MCX_VARIANT_COV(h, g8, n10, noff, it, mode)) {}
        else {
            hipLaunchKernelGGL(adapt_cov_diag_kernel, dim3(g8 * n10), dim3(64), 0, h->stream, h->E, it, mode, n10);
            if (noff > 0) hipLaunchKernelGGL(adapt_cov_off_kernel, dim3(g8 * noff), dim3(64), 0, h->stream, h->E, it, mode, noff);
        }
    }
    if (!h->d_Gc) {
        if (lds > 160 * 1024) {                             // npar > 320: the work vector in global scratch (slower; no limit)
            if (h->usesvd) hipLaunchKernelGGL((adapt_post_kernel<true, true>), dim3(h->ntiles), dim3(64), 0, h->stream, h->E, it, mode, 0,
                (uint8_t *)nullptr, batch_done);
            else hipLaunchKernelGGL((adapt_post_kernel<false, true>), dim3(h->ntiles), dim3(64), 0, h->stream, h->E, it, mode, 0,
                (uint8_t *)nullptr, batch_done);
            return;
        }
        if (h->usesvd) hipLaunchKernelGGL(adapt_post_kernel<true>, dim3(h->ntiles), dim3(64), lds, h->stream, h->E, it, mode, 0,
            (uint8_t *)nullptr, batch_done);
        else if (h->tile_factor) {
            // dpotf2 (+ dtrti2 / dlauu2 with delayed rejection) with the packed matrices of 4 NW neighbouring chains in LDS, read and
            // written once (mcx_group.hpp: tile_factor_kernel); adapt_post_kernel keeps the covariance bookkeeping (phase 3)
            hipLaunchKernelGGL(adapt_post_kernel<false>, dim3(h->ntiles), dim3(64), lds, h->stream, h->E, it, mode, 3, (uint8_t *)nullptr,
                batch_done);
            const int nc = (h->d + 15) / 16, nw = nc <= 2 ? 4 : nc == 3 ? 2 : 1, ch = 4 * nw;
            const size_t tl = (size_t)ch * (h->P | 1) * sizeof(double) + (size_t)ch * sizeof(int);
            const dim3 tg((unsigned)(8 * ((h->ntiles + 7) / 8) * (64 / ch)));
            switch (nc) {
            case 1: hipLaunchKernelGGL((tile_factor_kernel<1, 4>), tg, dim3(256), tl, h->stream, h->E); break;
            case 2: hipLaunchKernelGGL((tile_factor_kernel<2, 4>), tg, dim3(256), tl, h->stream, h->E); break;
            case 3: hipLaunchKernelGGL((tile_factor_kernel<3, 2>), tg, dim3(128), tl, h->stream, h->E); break;
            default: hipLaunchKernelGGL((tile_factor_kernel<4, 1>), tg, dim3(64), tl, h->stream, h->E); break;
            }
        }
        else hipLaunchKernelGGL(adapt_post_kernel<false>, dim3(h->ntiles), dim3(64), lds, h->stream, h->E, it, mode, 0, (uint8_t *)nullptr,
            batch_done);
        return;
    }
    // large npar with an SVD factor: the factorisation runs one workgroup per chain on chain-major copies, one launch
    // pair per Jacobi sweep (the rotation log lives in Gw, which is free between tile2chain and the next tick)
    const size_t DD = (size_t)h->d * h->d;
    const dim3 tg((unsigned)((DD + 63) / 64), (unsigned)h->ntiles), tg1((unsigned)((h->d + 63) / 64), (unsigned)h->ntiles);
    hipLaunchKernelGGL(adapt_post_kernel<true>, dim3(h->ntiles), dim3(64), lds, h->stream, h->E, it, mode, 1, h->d_need, batch_done);
    hipLaunchKernelGGL(tile2chain_kernel, tg, dim3(256), 0, h->stream, h->E.Gw, h->d_Gc, DD, DD, h->d_need);
    hipLaunchKernelGGL(svd_init_kernel, dim3(h->nlanes), dim3(256), 0, h->stream, h->d_Vc, h->d_state, h->d_need, h->nlanes, h->d);
    for (int sweep = 0; sweep < 60; ++sweep) {
        (void)hipMemsetAsync(h->d_anyrot, 0, sizeof(int), h->stream);
        // the sweep: every later column streamed past a block's pair-lanes through an LDS ring (mcx_svd.hpp) -- all 32 lanes of a row group
        // on pairs up to npar 200 (svd_sweep_stream32_kernel), 24 pair-lanes and a wave of loaders above (svd_sweep_stream_kernel)
        if (h->d <= 200) {
            const int RLs = h->d <= 64 ? 8 : h->d <= 128 ? 16 : 25;
            const size_t lss = (size_t)33 * (8 * RLs + 2) * sizeof(double);
            if (MCX_VARIANT_SVD_SWEEP(h, lss)) {}
            else if (h->d <= 64) hipLaunchKernelGGL(svd_sweep_stream32_kernel<8>, dim3(h->nlanes), dim3(256), lss, h->stream, h->d_Gc,
                (mcx_d2 *)h->E.Gw, h->d_state, h->d_anyrot, h->nlanes, h->d);
            else if (h->d <= 128) hipLaunchKernelGGL(svd_sweep_stream32_kernel<16>, dim3(h->nlanes), dim3(256), lss, h->stream, h->d_Gc,
                (mcx_d2 *)h->E.Gw, h->d_state, h->d_anyrot, h->nlanes, h->d);
            else hipLaunchKernelGGL(svd_sweep_stream32_kernel<25>, dim3(h->nlanes), dim3(256), lss, h->stream, h->d_Gc, (mcx_d2 *)h->E.Gw,
                h->d_state, h->d_anyrot, h->nlanes, h->d);
        } else {
            // pair-lanes of the streamed sweep: whole waves of octets, one wave of loaders at least
            constexpr int svd_sb = 24;
            const int RLs = h->d <= 208 ? 26 : 32;
            const size_t lss = (size_t)(svd_sb + 2) * (8 * RLs + 2) * sizeof(double);
            if (h->d <= 208) hipLaunchKernelGGL(svd_sweep_stream_kernel<26>, dim3(h->nlanes), dim3(256), lss, h->stream, h->d_Gc,
                (mcx_d2 *)h->E.Gw, h->d_state, h->d_anyrot, h->nlanes, h->d, svd_sb);
            else hipLaunchKernelGGL(svd_sweep_stream_kernel<32>, dim3(h->nlanes), dim3(256), lss, h->stream, h->d_Gc, (mcx_d2 *)h->E.Gw,
                h->d_state, h->d_anyrot, h->nlanes, h->d, svd_sb);
        }
        {                                                // the log replayed on V: all 32 lanes of a row group on pairs, any npar
            // four one-wave workgroups per chain, a chain's on one XCD
            const unsigned gv = (unsigned)(32 * ((h->nlanes + 7) / 8));
            const int RP = h->d <= 64 ? 4 : h->d <= 128 ? 8 : h->d <= 208 ? 13 : 16;
            const size_t lsv2 = (size_t)33 * (4 * RP + 6) * sizeof(double);
            if (h->d <= 64) hipLaunchKernelGGL(svd_applyv_stream32_kernel<4>, dim3(gv), dim3(64), lsv2, h->stream, h->d_Vc,
                (const mcx_d2 *)h->E.Gw, h->d_state, h->nlanes, h->d);
            else if (h->d <= 128) hipLaunchKernelGGL(svd_applyv_stream32_kernel<8>, dim3(gv), dim3(64), lsv2, h->stream, h->d_Vc,
                (const mcx_d2 *)h->E.Gw, h->d_state, h->nlanes, h->d);
            else if (h->d <= 208) hipLaunchKernelGGL(svd_applyv_stream32_kernel<13>, dim3(gv), dim3(64), lsv2, h->stream, h->d_Vc,
                (const mcx_d2 *)h->E.Gw, h->d_state, h->nlanes, h->d);
            else hipLaunchKernelGGL(svd_applyv_stream32_kernel<16>, dim3(gv), dim3(64), lsv2, h->stream, h->d_Vc, (const mcx_d2 *)h->E.Gw,
                h->d_state, h->nlanes, h->d);
        }
        int any = 0;
        // (reported by the caller's hipGetLastError)
        if (hipMemcpyAsync(&any, h->d_anyrot, sizeof(int), hipMemcpyDeviceToHost,
            h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) break;
        if (!any) break;
    }
    hipLaunchKernelGGL(svd_finish_kernel, dim3(h->nlanes), dim3(256), 0, h->stream, h->d_Gc, h->d_Vc, h->d_svc, h->d_state, h->nlanes,
        h->d);
    hipLaunchKernelGGL(chain2tile_kernel, tg, dim3(256), 0, h->stream, h->d_Gc, h->E.Vw, DD, DD, h->d_need);
    hipLaunchKernelGGL(chain2tile_kernel, tg1, dim3(256), 0, h->stream, h->d_svc, h->E.cs, (size_t)h->d, (size_t)2 * h->d, h->d_need);
    hipLaunchKernelGGL(adapt_post_kernel<true>, dim3(h->ntiles), dim3(64), lds, h->stream, h->E, it, mode, 2, h->d_need, batch_done);
}

// Which branch of MCMC_adapt fires at iteration `it` (0 = none).  MCMC_adapt.F90:42-46, 60-61, 105.
static int adapt_mode(const mcmcx_config &c, int it)
{
    if (c.method == MCMCX_METHOD_RAM) return 0;
    if (c.doadapt == 0 && c.doburnin == 0) return 0;
    if (c.adaptend > 0 && it > c.adaptend) return 0;
    bool m1 = (c.adaptint != 0) && (it % c.adaptint == 0);
    bool m2 = (c.badaptint != 0) && (it % c.badaptint == 0);
    if (!m1 && !m2) return 0;
    if (it < c.burnintime && c.doburnin != 0 && m2) return AD_BURN;
    if (it >= c.burnintime + c.adaptint + c.adapthist && c.doadapt != 0)
        return AD_AM | ((it == c.burnintime + c.adaptint + c.adapthist) ? AD_FIRST : 0);
    return 0;
}

// pooled RAM: every adaptint iterations once the burn-in is over (MCMC_run_ram.F90:123-131), up to adaptend
static bool pooled_ram_due(const mcmcx_engine *h, int it)
{
    const mcmcx_config &c = h->cfg;
    if (!h->pooled || c.method != MCMCX_METHOD_RAM || c.doadapt == 0 || c.adaptint <= 0) return false;
    if (it < c.burnintime && c.doburnin != 0) return false;
    if (c.adaptend > 0 && it > c.adaptend) return false;
    return it % c.adaptint == 0;
}

// ------------------------------------------------------------------ getters
template <typename T>
static int fetch(mcmcx_engine *h, const T *dev, size_t n, std::vector<T> &out)
{
    out.resize(n);
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}
// gather element k of one chain from a tile-interleaved array
template <typename T>
static int fetch_chain_vec(mcmcx_engine *h, const T *dev, int K, int chain, std::vector<T> &out)
{
    int tile = chain / 64, lane = chain % 64;
    std::vector<T> tmp;
    int rc = fetch(h, dev + (size_t)tile * K * 64, (size_t)K * 64, tmp);
    if (rc) return rc;
    out.resize(K);
    for (int k = 0; k < K; ++k) out[k] = tmp[(size_t)k * 64 + lane];
    return 0;
}
static int check_chain(mcmcx_engine *h, int chain)
{
    if (!h) return fail(-1, "null handle");
    if (!h->inited) return fail(-40, "we have not inited");
    if (chain < 0 || chain >= h->cfg.nchains) return fail(-41, "chain index out of range");
    return 0;
}

static void unpack_upper(int d, const std::vector<double> &p, double *colmajor, bool symmetric)
{
    for (int j = 0; j < d; ++j)
        for (int i = 0; i < d; ++i) {
            double v = 0.0;
            if (i <= j) v = p[h_pidx(i, j, d)];
            else if (symmetric) v = p[h_pidx(j, i, d)];
            colmajor[(size_t)i + (size_t)j * d] = v;
        }
}

static volatile sig_atomic_t g_interrupt = 0;
static bool g_sig_installed = false;       // then mcmcx_run waits for each launch, so that a signal is seen at the next boundary
static void on_signal(int) { g_interrupt = 1; }

static int pooled_moments_launch(mcmcx_engine *h, double *dev_dst, int kind = 0, int it = 0);
static int pooled_vec_len(const mcmcx_engine *h, int kind);
#include "mcx_comm.hpp"

// Pooled moments of the chains of ALL ranks, left in h->d_pooled (asynchronous on the engine's stream): local tree ->
// slot `rank` of d_gather -> all-gather over the communicator -> the same pairwise tree over the ranks.  Every rank's
// slot carries one more element behind the vector, the rank's STOP FLAG (1 = a signal was caught here): its sum over the
// ranks comes back with the moments, so the decision to leave a run that has collectives ahead is taken by all ranks at
// the same tick (a rank that returned alone would leave its peers waiting in the next gather).
static int allreduce_moments_enqueue(mcmcx_engine *h, int stage /* 0 all, 1 local part, 2 gather, 3 tree */, int kind = 0, int it = 0,
    double flag = 0.0)
{
    const int len = pooled_vec_len(h, kind), st = len + 1;
    const int nr = h->comm ? h->comm->nranks : 1, rk = h->comm ? h->comm->rank : 0;
    HIPCHK(hipSetDevice(h->cfg.device));
    if (stage == 0 || stage == 1) {
        int rc = pooled_moments_launch(h, h->d_gather + (size_t)rk * st, kind, it); if (rc) return rc;
        h->h_flag = flag;
        hipLaunchKernelGGL(set_double_kernel, dim3(1), dim3(1), 0, h->stream, h->d_gather + (size_t)rk * st + len, flag);
        HIPCHK(hipGetLastError());
    }
    if ((stage == 0 || stage == 2) && h->comm) { int rc = comm_allgather(h->comm, h->d_gather, st, h->stream); if (rc) return rc; }
    if (stage == 0 || stage == 3) {
        hipLaunchKernelGGL(moments_tree_kernel, dim3((st + 255) / 256, 1), dim3(256), 0, h->stream, h->d_gather, nr, st, 1, h->d_pooled);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// several ranks meet in this engine's ticks: its run may only be left at a tick, by agreement (the stop flag above)
static bool collective_run(const mcmcx_engine *h) { return h->pooled && h->comm && h->comm->nranks > 1 && !h->xfn; }

// The pooled statistic vector of `kind` over the chains of ALL ranks, on the host.  With a communicator: local tree ->
// all-gather -> tree over ranks; with the caller's exchange hook (kind 0 only): the hook sums the device buffer.
// When some rank raised its stop flag, h->stop_seen is set: every rank reads the same sum at the same tick, APPLIES that tick
// like any other (all of them hold the same pooled vector) and leaves mcmcx_run behind it -- so a run resumed after
// mcmcx_clear_interrupt continues exactly like one that was never interrupted.
static int pooled_reduce(mcmcx_engine *h, int kind, int it, std::vector<double> &v)
{
    const int len = pooled_vec_len(h, kind);
    v.assign(len + 1, 0.0);
    if (h->xfn && kind != 0) return
        fail(-8, "pooled burn-in scaling and the pooled RAM variant exchange through a communicator (mcmcx_set_comm), not through the mcmcx_set_exchange hook");
    if (!h->xfn) {
        int rc = allreduce_moments_enqueue(h, 0, kind, it, (collective_run(h) && g_interrupt) ? 1.0 : 0.0); if (rc) return rc;
        if ((rc = comm_wait_stream(h->comm, h->stream))) return rc;
        HIPCHK(hipMemcpy(v.data(), h->d_pooled, (size_t)(len + 1) * 8, hipMemcpyDeviceToHost));
    } else {
        double *dst = h->xbuf ? h->xbuf : h->d_moments + (size_t)h->ntiles * len;  // tail of the moments workspace
        int rc = pooled_moments_launch(h, dst, kind, it); if (rc) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
        h->xfn(h->xuser);                                                            // the caller's own exchange
        HIPCHK(hipMemcpy(v.data(), dst, (size_t)len * 8, hipMemcpyDeviceToHost));
    }
    const double stop = v[len];
    v.resize(len);
    if (collective_run(h) && stop != 0.0) h->stop_seen = true;
    // the vector has been through an exchange: refuse to merge garbage (it would poison the pooled state for the rest of the run)
    if (!(v[0] >= 2.0) || !std::isfinite(v[0])) return fail(-46, "pooled adaptation needs at least 2 chains over all ranks (count = " +
        std::to_string(v[0]) + ")");
    for (int k = 1; k < len;
        ++k) if (!std::isfinite(v[k])) return fail(-46, "pooled adaptation: non-finite pooled statistic at iteration " +
        std::to_string(it));
    return 0;
}

// Delayed rejection in pooled mode (MCMC_adapt.F90:216-225 once for all chains): R2 = R / drscale, iC = dpotri('U', R) on
// the upper triangle of the factor as it stands.  fresh: a new factor (recompute both); otherwise the burn-in scaling
// has already been applied to the tables themselves, as MCMC_adapt.F90:66-78 does.
static int pooled_upload_dr(mcmcx_engine *h, bool fresh)
{
    if (!h->dodr) return 0;
    const int d = h->d, P = h->P;
    if (fresh) {
        std::vector<double> iC(P);
        if (h->usesvd) {
            h->pool_R2 = h->pool_Rf;
            for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) iC[h_pidx(i, j, d)] = h->pool_Rf[(size_t)j * d + i];
        } else { h->pool_R2 = h->pool_R; iC = h->pool_R; }
        for (auto &v : h->pool_R2) v = v / h->cfg.drscale;
        // the reference stops ("cannot invert cmat"); the old iC stays
        if (host_potri(d, iC) != 0) h->pool_status |= ST_POTRI_FAIL;
        else h->pool_iC = iC;
    }
    std::vector<double> r2 = h->pool_R2;
    if (h->usesvd) r2.resize((size_t)((d + 3) & ~3) * d + PWS, 0.0);
    HIPCHK(hipMemcpyAsync(h->d_sharedR2, r2.data(), r2.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_sharediC, h->pool_iC.data(), (size_t)P * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    // pooled_mfma_kernel<true>: R2 like d_sharedRT (M[s*d + o] = R2(s,o)), iC dense symmetric
    if (h->d_sharedR2T) {
        const int d4 = (d + 3) & ~3;
        std::vector<double> m((size_t)d4 * d + PWS, 0.0), q((size_t)d4 * d + PWS, 0.0);
        // the full factor as it stands: M[s*d + o] = R2f(o, s)
        if (h->usesvd) memcpy(m.data(), h->pool_R2.data(), (size_t)d * d * 8);
        else for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) m[(size_t)i * d + j] = h->pool_R2[h_pidx(i, j, d)];
        for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) { const double v = h->pool_iC[h_pidx(i, j, d)]; q[(size_t)i * d + j] = v;
            q[(size_t)j * d + i] = v; }
        HIPCHK(hipMemcpyAsync(h->d_sharedR2T, m.data(), m.size() * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_sharediCd, q.data(), q.size() * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

static int pooled_upload_R(mcmcx_engine *h)
{
    if (h->usesvd) return upload_shared_rf(h);
    HIPCHK(hipMemcpyAsync(h->d_sharedR, h->pool_R.data(), (size_t)h->P * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->d_sharedRT) return upload_shared_rt(h);
    return 0;
}

// chaincmat -> the shared proposal factor (MCMC_calculate_R): dpotf2 + 2.4/sqrt(d), or the pinned SVD for scam; on
// failure the old factor stays (MCMC_adapt.F90:168-171)
static int pooled_factor(mcmcx_engine *h)
{
    const mcmcx_config &c = h->cfg;
    const int d = h->d;
    std::vector<double> cm((size_t)d * d, 0.0), Rp, Cp;
    for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) cm[(size_t)i + (size_t)j * d] = h->pool_C[h_pidx(i, j, d)];
    if (c.method == MCMCX_METHOD_SCAM) {                // scam_svd of the pooled covariance, MCMC_adapt.F90:189-200
        std::vector<double> U, sd;
        if (host_initial_svd(d, cm, c.condmax, true, U, sd) == 0) { h->pool_U = U; h->pool_std = sd; return upload_shared_u(h); }
        return 0;
    }
    if (h->usesvd) {                                    // covtor_svd of the pooled covariance, MCMC_adapt.F90:203-209
        std::vector<double> Rf, sd, fc;
        if (host_initial_svd(d, cm, c.condmax, false, Rf, sd, &fc) == 0) {
            h->pool_Rf = Rf;
            if (!fc.empty()) for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) h->pool_C[h_pidx(i, j,
                d)] = fc[(size_t)i + (size_t)j * d];
            int rc = pooled_upload_R(h);
            return rc ? rc : pooled_upload_dr(h, true);
        }
        return 0;
    }
    if (host_initial_R(d, cm, Rp, Cp) == 0) { h->pool_R = Rp; int rc = pooled_upload_R(h); return rc ? rc : pooled_upload_dr(h, true); }
    return 0;
}

// merge the batch of n unit-weight rows (moments v about par0) into (chaincmat, chainmean, chainwsum): covmat's
// weighted update for a whole batch at once, or its batch branch when there is nothing to update (wsum = 0) or when
// `replace` (the AP window: covmat(..., update = .false.), MCMC_adapt.F90:131-133)
static void pooled_merge(mcmcx_engine *h, const std::vector<double> &v, bool replace)
{
    const int d = h->d, P = h->P;
    const double n = v[0];
    std::vector<double> m1(d), mb(d), Cb(P);
    for (int j = 0; j < d; ++j) { m1[j] = v[1 + j] / n; mb[j] = h->par0[j] + m1[j]; }
    for (int j = 0; j < d; ++j)
        for (int i = 0; i <= j; ++i) {
            double s2 = v[1 + d + j * (j + 1) / 2 + i];
            Cb[h_pidx(i, j, d)] = (s2 - n * m1[i] * m1[j]) / (n - 1.0);
        }
    if (replace || !(h->pool_W > 0.0)) {
        h->pool_C = Cb; h->pool_mean = mb; h->pool_W = n;
    } else {
        const double W = h->pool_W, Wn = W + n;
        std::vector<double> dl(d);
        for (int j = 0; j < d; ++j) dl[j] = mb[j] - h->pool_mean[j];
        const double f = W * n / Wn;
        for (int j = 0; j < d; ++j)
            for (int i = 0; i <= j; ++i) {
                const int e = h_pidx(i, j, d);
                h->pool_C[e] = ((W - 1.0) * h->pool_C[e] + (n - 1.0) * Cb[e] + f * dl[i] * dl[j]) / (Wn - 1.0);
            }
        const double g = n / Wn;
        for (int j = 0; j < d; ++j) h->pool_mean[j] = h->pool_mean[j] + g * dl[j];
        h->pool_W = Wn;
    }
}

static void pooled_restart(mcmcx_engine *h)             // chainwsum = initcmatn, chaincmat = cmat0, chainmean = par0
{
    const int d = h->d;
    h->pool_W = (double)h->cfg.initcmatn;
    for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) h->pool_C[h_pidx(i, j, d)] = h->cmat0[(size_t)i + (size_t)j * d];
    h->pool_mean = h->par0;
}

// Pooled tick: the multi-chain form of MCMC_adapt (MCMC_adapt.F90:60-170).  The N current states of all ranks are a
// batch of N unit-weight rows; `stayed` is summed over the chains.  Every operation below is restated in
// tests/test_gpu_pooled.py.
//   burn-in tick (:60-102): pooled rejection rate sum(stayed) / (N it) against scalelimit -> the shared factor is scaled
//       down / up; in between, greedy restarts from cmat0 and merges the batch, otherwise chaincmat stays, and the
//       factor is recomputed from chaincmat either way (which is what undoes earlier scalings in the reference too)
//   AM tick (:105-159): first time restart from cmat0; merge the batch; AP (adapthist > 1): the batch replaces the
//       covariance instead (the window of the single chain becomes the snapshot of the population)
static int pooled_tick(mcmcx_engine *h, int it, int mode)
{
    const mcmcx_config &c = h->cfg;
    std::vector<double> v;
    if (mode & AD_BURN) {
        int rc = pooled_reduce(h, 1, it, v); if (rc) return rc;
        const double staypc = v[pooled_vec_len(h, 1) - 1] / (v[0] * (double)it);
        const double sf = c.scalefactor;
        if (staypc > 1.0 - c.scalelimit || staypc < c.scalelimit) {
            const bool down = staypc > 1.0 - c.scalelimit;
            for (auto &r : (h->usesvd ? h->pool_Rf : h->pool_R)) r = down ? r / sf : r * sf;
            if (h->dodr) {                                  // R2 and iC are scaled themselves (MCMC_adapt.F90:66-78), not recomputed
                for (auto &r : h->pool_R2) r = down ? r / sf : r * sf;
                for (auto &r : h->pool_iC) r = down ? r * sf * sf : r / sf / sf;
            }
            int rc2 = pooled_upload_R(h);
            return rc2 ? rc2 : pooled_upload_dr(h, false);
        }
        if (c.greedy != 0) { pooled_restart(h); pooled_merge(h, v, false); }
        return pooled_factor(h);
    }
    int rc = pooled_reduce(h, 0, it, v); if (rc) return rc;
    if (it == c.burnintime + c.adaptint + c.adapthist) pooled_restart(h);           // first time: MCMC_adapt.F90:108-114
    pooled_merge(h, v, c.adapthist > 1);
    return pooled_factor(h);
}

// Pooled RAM tick (the multi-chain form of MCMC_adapt_ram, MCMC_run_ram.F90:104-179): every adaptint iterations the
// rank-one statistics of that iteration, one per chain, are averaged over all chains of all ranks and applied to the
// Gram matrix of the shared factor at once:   R'R  <-  R'R + (1/N) sum_c sign(a_c) x_c x_c',
// x_c = u_c / sum(u_c**2) * a_c,  a_c = (alpha_c - alphatarget) / it**nuparam  -- N Cholesky up/downdates of weight 1/N
// folded into one refactorisation.  A Gram matrix that stops being positive definite keeps the old factor (the
// single-chain code stops on a failed downdate; here one bad tick is skipped and flagged).
static int pooled_ram_tick(mcmcx_engine *h, int it)
{
    const int d = h->d, P = h->P;
    std::vector<double> v;
    int rc = pooled_reduce(h, 2, it, v); if (rc) return rc;
    const double n = v[0];
    if (h->usesvd) {
        // condmax > 0: the shared factor is the full matrix Rf of covtor_svd (matutils.F90:378-453), proposals are
        // matmulx(Rf, z) with covariance Rf Rf'.  The same fold on that Gram matrix, refactored the way this factor is made:
        // Rf <- U sqrt(s) of Rf Rf' + (1/N) sum_c sign(a_c) x_c x_c', singular values floored at s_1 / condmax (no 2.4/sqrt(d):
        // the Gram matrix carries the scale already, like the Cholesky form below)
        std::vector<double> S((size_t)d * d, 0.0), Rf, sd;
        for (int j = 0; j < d; ++j)
            for (int i = 0; i <= j; ++i) {
                double acc = 0.0;
                for (int k = 0; k < d; ++k) acc = std::fma(h->pool_Rf[(size_t)k * d + i], h->pool_Rf[(size_t)k * d + j], acc);
                S[(size_t)i + (size_t)j * d] = acc + v[2 + j * (j + 1) / 2 + i] / n;
            }
        if (host_initial_svd(d, S, h->cfg.condmax, false, Rf, sd, nullptr, false) != 0) { h->pool_status |= ST_CHOL_FAIL; return 0; }
        h->pool_Rf = Rf;
        h->pool_alpha = v[1] / n;
        return pooled_upload_R(h);
    }
    std::vector<double> S(P), A;
    for (int j = 0; j < d; ++j)
        for (int i = 0; i <= j; ++i) {
            double acc = 0.0;
            for (int k = 0; k <= i; ++k) acc = std::fma(h->pool_R[h_pidx(k, i, d)], h->pool_R[h_pidx(k, j, d)], acc);
            S[h_pidx(i, j, d)] = acc + v[2 + j * (j + 1) / 2 + i] / n;
        }
    A = S;
    for (int j = 0; j < d; ++j) {                       // dpotf2('U'), the order of host_initial_R
        double dot = 0.0;
        for (int i = 0; i < j; ++i) dot = std::fma(A[h_pidx(i, j, d)], A[h_pidx(i, j, d)], dot);
        double ajj = A[h_pidx(j, j, d)] - dot;
        if (!(ajj > 0.0)) { h->pool_status |= ST_CHOL_FAIL; return 0; }
        double rj = std::sqrt(ajj);
        A[h_pidx(j, j, d)] = rj;
        double rinv = 1.0 / rj;
        for (int k = j + 1; k < d; ++k) {
            double t = 0.0;
            for (int i = 0; i < j; ++i) t = std::fma(A[h_pidx(i, k, d)], A[h_pidx(i, j, d)], t);
            A[h_pidx(j, k, d)] = (A[h_pidx(j, k, d)] - t) * rinv;
        }
    }
    h->pool_R = A;
    h->pool_alpha = v[1] / n;
    return pooled_upload_R(h);
}

// Host-callback evaluation of one candidate vector per chain, in chain order, from the calling thread
// (the reference's callbacks keep SAVEd state and are not thread-safe: testcases/mcmcrun.F90:69-70).
// src: tile-interleaved device vector [T][stride][64]; only chains with want != 0 (hx slot) are evaluated.
// what: 0 = checkbounds, priorfun, ssfunction (MCMC_run.F90:47-56); 1 = checkbounds and priorfun only, 2 = ssfunction_er
// with each chain's threshold (the two halves of an early-rejection iteration, MCMC_run_er.F90:54-76)
static int host_eval(mcmcx_engine *h, const double *dev_src, int stride_k, bool use_stage2_flag, int what = 0)
{
    const int d = h->d, T = h->ntiles;
    if (h->tkind == TGT_EXPCOLS) {                      // device-resident response-column target: no host round trip
        hipLaunchKernelGGL(dev_eval_kernel, dim3(T), dim3(64), 0, h->stream, h->E, dev_src, stride_k, use_stage2_flag ? 1 : 0, what);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (h->tkind == TGT_MODULE) {                       // the user's own device code, loaded from a code object
        mcmcx_target_args a;
        a.src = dev_src; a.hev = h->E.hev; a.hx = h->E.hx; a.userdata = h->d_moddata;
        a.stride_k = stride_k; a.npar = d; a.ny = h->ny; a.nhe = NHE - 1 + h->ny; a.nhx = NHX; a.nchains = h->cfg.nchains;
        a.use_stage2 = use_stage2_flag ? 1 : 0; a.what = what;
        size_t asz = sizeof(a);
        void *cfgv[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
        HIPCHK(hipModuleLaunchKernel(h->mod_fn, (unsigned)T, 1, 1, 64, 1, 1, 0, h->stream, nullptr, cfgv));
        return 0;
    }
    const size_t L = (size_t)T * 64;
    const int ny = h->ny, nhe = NHE - 1 + ny;
    const bool src_mapped = h->host_mapped && (dev_src == h->E.cand || h->cs_mapped);
    const bool mapped = h->host_mapped;                  // flags and results in place
    if ((!src_mapped && h->h_cand.resize(L * stride_k)) || (!mapped && (h->h_ev.resize(L * nhe) || (use_stage2_flag
        && h->h_hx.resize(L * NHX)))))
        return fail(-100, "host callbacks: no page-locked memory for the candidates");
    std::vector<double> ssc(ny, 0.0);
    if (!src_mapped) HIPCHK(hipMemcpyAsync(h->h_cand.data(), dev_src, L * stride_k * 8, hipMemcpyDeviceToHost, h->stream));
    if (use_stage2_flag && !mapped) HIPCHK(hipMemcpyAsync(h->h_hx.data(), h->E.hx, h->h_hx.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));             // (also: the previous stage's results have left h_ev)
    const double *h_cand = src_mapped ? dev_src : h->h_cand.data();
    const double *hx = mapped ? h->E.hx : h->h_hx.data();
    double *h_ev = mapped ? h->E.hev : h->h_ev.data();
    memset(h_ev, 0, L * nhe * sizeof(double));
    std::vector<double> th(d);
    if (h->h_ss_batch && !(what == 2 && h->h_ss_er)) {
        // Batched form (opt-in): bounds and prior per chain on this thread, in chain order; then ONE call of the user's
        // ssfunction_batch per worker thread over the chains that need the sum of squares.
        h->h_bidx.clear(); h->h_bth.clear();
        for (int c = 0; c < h->cfg.nchains; ++c) {
            const int t = c / 64, l = c % 64;
            if (use_stage2_flag && hx[((size_t)t * NHX + HX_STAGE2) * 64 + l] == 0.0) continue;
            for (int k = 0; k < d; ++k) th[k] = h_cand[((size_t)t * stride_k + k) * 64 + l];
            int inb = 1; double pri = 0.0;
            if (what != 2) {
                inb = h->h_cb ? h->h_cb(th.data(), d, h->h_user) : 1;
                if (inb) pri = h->h_pri ? h->h_pri(th.data(), d, h->h_user) : 0.0;
            }
            h_ev[((size_t)t * nhe + HE_INB) * 64 + l] = inb ? 1.0 : 0.0;
            h_ev[((size_t)t * nhe + HE_PRI) * 64 + l] = pri;
            if ((what == 0 && inb) || what == 2) { h->h_bidx.push_back(c); h->h_bth.insert(h->h_bth.end(), th.begin(), th.end()); }
        }
        const int n = (int)h->h_bidx.size();
        h->h_bss.assign((size_t)n * ny, 0.0);
        // the first evaluation (MCMC_init's starting point) stays on the calling thread: user code commonly loads its
        // data on first call (testcases/mcmcrun.F90:69-70) -- after that concurrent calls only read it
        const int nt = h->inited ? std::max(1, std::min(h->h_threads, n)) : 1;
        if (nt <= 1) { if (n > 0) h->h_ss_batch(h->h_bth.data(), d, n, ny, h->h_bss.data(), h->h_user); }
        else {
            std::vector<std::thread> pool;
            for (int w = 0; w < nt; ++w) {
                const int lo = (int)((long long)n * w / nt), hi = (int)((long long)n * (w + 1) / nt);
                if (hi > lo) pool.emplace_back([=]() { h->h_ss_batch(h->h_bth.data() + (size_t)lo * d, d, hi - lo, ny,
                    h->h_bss.data() + (size_t)lo * ny, h->h_user); });
            }
            for (auto &t : pool) t.join();
        }
        for (int i = 0; i < n; ++i) {
            const int c = h->h_bidx[i], t = c / 64, l = c % 64;
            for (int j = 0; j < ny; ++j) h_ev[((size_t)t * nhe + HE_SS + j) * 64 + l] = h->h_bss[(size_t)i * ny + j];
        }
        if (!mapped) HIPCHK(hipMemcpyAsync(h->E.hev, h->h_ev.data(), h->h_ev.size() * 8, hipMemcpyHostToDevice, h->stream));
        return 0;
    }
    for (int c = 0; c < h->cfg.nchains; ++c) {
        const int t = c / 64, l = c % 64;
        if (use_stage2_flag && hx[((size_t)t * NHX + HX_STAGE2) * 64 + l] == 0.0) continue;
        for (int k = 0; k < d; ++k) th[k] = h_cand[((size_t)t * stride_k + k) * 64 + l];
        int inb = 1;
        double pri = 0.0;
        std::fill(ssc.begin(), ssc.end(), 0.0);
        if (what == 2) {                                                             // MCMC_ssfunction_er(newpar, sscrit)
            const double crit = hx[((size_t)t * NHX + HX_CRIT) * 64 + l];
            if (h->h_ss_er) h->h_ss_er(th.data(), d, ny, crit, ssc.data(), h->h_user);
            else h->h_ss(th.data(), d, ny, ssc.data(), h->h_user);                   // ssfunction_er0.f90: no er for ss
        } else {
            inb = h->h_cb ? h->h_cb(th.data(), d, h->h_user) : 1;                    // checkbounds0.f90: .true.
            if (inb) {                                                               // MCMC_run.F90:54-56: prior first
                pri = h->h_pri ? h->h_pri(th.data(), d, h->h_user) : 0.0;
                if (what == 0) h->h_ss(th.data(), d, ny, ssc.data(), h->h_user);
            }
        }
        h_ev[((size_t)t * nhe + HE_INB) * 64 + l] = inb ? 1.0 : 0.0;
        h_ev[((size_t)t * nhe + HE_PRI) * 64 + l] = pri;
        for (int j = 0; j < ny; ++j) h_ev[((size_t)t * nhe + HE_SS + j) * 64 + l] = ssc[j];
    }
    if (!mapped) HIPCHK(hipMemcpyAsync(h->E.hev, h->h_ev.data(), h->h_ev.size() * 8, hipMemcpyHostToDevice, h->stream));
    return 0;
}

// fuse_next: iteration it + 1 follows without a tick in between -- its proposal (phase 0; SCAM: component 0's phase 5) rides in this
// iteration's last launch, and h->p0_done tells the next call so (MCMCX_HOST_FUSE=0: one launch per phase, the A/B form the tests compare
// with)
static int host_iteration(mcmcx_engine *h, int it, bool fuse_next)
{
    const dim3 g(h->ntiles), b(64);
    const double *rs = h->d_ramscale + it, *rs0 = h->d_ramscale;
    const size_t lds = lds_step(h);
    const bool fuse = h->sw.host_fuse != 0;
    fuse_next = fuse_next && fuse;
    const bool p0_done = h->p0_done;
    h->p0_done = false;
    if (h->cfg.method == MCMCX_METHOD_SCAM) {           // MCMC_run_scam: npar componentwise proposals, each evaluated by the host
        for (int j = 0; j < h->d; ++j) {
            if (!(j == 0 ? p0_done : fuse)) { hipLaunchKernelGGL((host_phase_kernel<5>), g, b, 0, h->stream, h->E, it, rs, j);
                HIPCHK(hipGetLastError()); }
            int rc = host_eval(h, h->E.cand, h->d, false); if (rc) return rc;
            if (!fuse) hipLaunchKernelGGL((host_phase_kernel<6>), g, b, 0, h->stream, h->E, it, rs, j);
            else if (j + 1 < h->d) hipLaunchKernelGGL((host_phase_seq_kernel<6, 5, -1>), g, b, 0, h->stream, h->E, it, j, it, j + 1, 0, 0,
                rs0);
            else if (fuse_next) { hipLaunchKernelGGL((host_phase_seq_kernel<6, 7, 5>), g, b, 0, h->stream, h->E, it, j, it, 0, it + 1, 0,
                rs0); h->p0_done = true; }
            else hipLaunchKernelGGL((host_phase_seq_kernel<6, 7, -1>), g, b, 0, h->stream, h->E, it, j, it, 0, 0, 0, rs0);
            HIPCHK(hipGetLastError());
        }
        if (!fuse) { hipLaunchKernelGGL((host_phase_kernel<7>), g, b, 0, h->stream, h->E, it, rs, 0); HIPCHK(hipGetLastError()); }
        return 0;
    }
    if (!p0_done) { hipLaunchKernelGGL((host_phase_kernel<0>), g, b, 0, h->stream, h->E, it, rs, 0); HIPCHK(hipGetLastError()); }
    if (h->cfg.method == MCMCX_METHOD_ER) {             // MCMC_run_er: the threshold is drawn between priorfun and ssfunction_er
        int rc = host_eval(h, h->E.cand, h->d, false, 1); if (rc) return rc;
        hipLaunchKernelGGL((host_phase_kernel<3>), g, b, 0, h->stream, h->E, it, rs, 0);
        HIPCHK(hipGetLastError());
        rc = host_eval(h, h->E.cand, h->d, true, 2); if (rc) return rc;
        if (fuse_next) { hipLaunchKernelGGL((host_phase_seq_kernel<4, 0, -1>), g, b, 0, h->stream, h->E, it, 0, it + 1, 0, 0, 0, rs0);
            h->p0_done = true; }
        else hipLaunchKernelGGL((host_phase_kernel<4>), g, b, 0, h->stream, h->E, it, rs, 0);
        HIPCHK(hipGetLastError());
        return 0;
    }
    int rc = host_eval(h, h->E.cand, h->d, false); if (rc) return rc;
    if (fuse_next && !h->dodr) { hipLaunchKernelGGL((host_phase_seq_kernel<1, 0, -1>), g, b, 0, h->stream, h->E, it, 0, it + 1, 0, 0, 0,
        rs0); h->p0_done = true; }
    else hipLaunchKernelGGL((host_phase_kernel<1>), g, b, 0, h->stream, h->E, it, rs, 0);
    HIPCHK(hipGetLastError());
    if (h->dodr) {
        rc = host_eval(h, h->E.cs, 2 * h->d, true); if (rc) return rc;
        if (fuse_next) { hipLaunchKernelGGL((host_phase_seq_kernel<2, 0, -1>), g, b, lds, h->stream, h->E, it, 0, it + 1, 0, 0, 0, rs0);
            h->p0_done = true; }
        else hipLaunchKernelGGL((host_phase_kernel<2>), g, b, lds, h->stream, h->E, it, rs, 0);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// ---- MCMC_run1 / MCMC_run1_er: the arithmetic of one invocation (run1_kernel), all chains at once, vectors row-major per chain
static int run1_check(mcmcx_engine *h)
{
    if (!h) return fail(-1, "null handle");
    if (!h->inited) return fail(-40, "we have not inited");                           // MCMC_run1.F90:55
    if (!h->external) return
        fail(-42, "mcmcx_run1_*: needs mcmcx_set_target_external (the caller evaluates ssfunction / priorfun / checkbounds)");
    return 0;
}
static void run1_put(mcmcx_engine *h, int slot0, int K, const double *src /* [nchains][K] or nullptr */)
{
    const int n1 = 3 * h->d + 3 * h->ny + NR1;
    for (int c = 0; c < h->cfg.nchains; ++c) {
        const int t = c / 64, l = c % 64;
        for (int k = 0; k < K; ++k) h->h_r1[((size_t)t * n1 + slot0 + k) * 64 + l] = src ? src[(size_t)c * K + k] : 0.0;
    }
}
static void run1_get(mcmcx_engine *h, int slot0, int K, double *dst)
{
    const int n1 = 3 * h->d + 3 * h->ny + NR1;
    for (int c = 0; c < h->cfg.nchains; ++c) {
        const int t = c / 64, l = c % 64;
        for (int k = 0; k < K; ++k) dst[(size_t)c * K + k] = h->h_r1[((size_t)t * n1 + slot0 + k) * 64 + l];
    }
}
template <int MODE>
static int run1_launch(mcmcx_engine *h, int drstage)
{
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipMemcpyAsync(h->d_r1, h->h_r1.data(), h->h_r1.size() * 8, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL((run1_kernel<MODE>), dim3(h->ntiles), dim3(64), MODE == 0 ? lds_step(h) : 0, h->stream, h->E, h->d_r1, drstage);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h->h_r1.data(), h->d_r1, h->h_r1.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// ------------------------------------------------------------------ C ABI
extern "C" {

const char *mcmcx_last_error(void) { return g_err.c_str(); }
const char *mcmcx_version(void) { return "mcmcx 0.3 (gfx950)"; }
int32_t mcmcx_device_count(void) { int n = 0; return (hipGetDeviceCount(&n) == hipSuccess) ? n : 0; }
int mcmcx_device_info(int32_t device, char *buf, int32_t len)
{
    if (!buf || len < 2) return fail(-1, "mcmcx_device_info: bad argument");
    hipDeviceProp_t prop;
    char bus[64] = "?";
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { snprintf(buf, (size_t)len, "device %d: %s", device, hipGetErrorString(e)); return fail(-10, buf); }
    (void)hipDeviceGetPCIBusId(bus, (int)sizeof bus, device);
    snprintf(buf, (size_t)len, "%s %s, pci %s, %d CUs, %.0f GiB", prop.name, prop.gcnArchName, bus, prop.multiProcessorCount,
        (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
    return 0;
}

// what identifies the physical device and its clock limits (bench.py puts it in its JSON line: the pool's boxes differ by ~8 %)
int mcmcx_device_ident(int32_t device, char *uuid_hex, int32_t len, int32_t *out5)
{
    if (!uuid_hex || len < 33 || !out5) return fail(-1, "mcmcx_device_ident: bad argument (uuid buffer of >= 33 bytes, five integers)");
    hipUUID u;
    HIPCHK(hipDeviceGetUuid(&u, device));
    for (int i = 0; i < 16; ++i) snprintf(uuid_hex + 2 * i, 3, "%02x", (unsigned)(unsigned char)u.bytes[i]);
    int v = 0;
    HIPCHK(hipDeviceGetAttribute(&v, hipDeviceAttributeClockRate, device)); out5[0] = v;                 // kHz, the engine clock's limit
    HIPCHK(hipDeviceGetAttribute(&v, hipDeviceAttributeMemoryClockRate, device)); out5[1] = v;           // kHz
    HIPCHK(hipDeviceGetAttribute(&v, hipDeviceAttributeMemoryBusWidth, device)); out5[2] = v;            // bits
    HIPCHK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device)); out5[3] = v;
    HIPCHK(hipDeviceGetAttribute(&v, hipDeviceAttributeL2CacheSize, device)); out5[4] = v;
    return 0;
}

const char *mcmcx_last_kernel(mcmcx_handle h)
{
    static thread_local std::string name;
    name = h ? h->last_kernel : "";
    // template instances are launched as (k<...>)
    if (name.size() >= 2 && name.front() == '(' && name.back() == ')') name = name.substr(1, name.size() - 2);
    return name.c_str();
}

void mcmcx_config_defaults(mcmcx_config *c)                      // mcmcinit.F90:184-230
{
    memset(c, 0, sizeof *c);
    c->npar = 0; c->nchains = 1; c->method = MCMCX_METHOD_DRAM;
    c->nsimu = 0; c->doadapt = 1; c->doburnin = 0; c->burnintime = 0; c->badaptint = -1;
    c->greedy = 0; c->scalelimit = 0.05; c->scalefactor = 2.5; c->drscale = 0.0;
    c->adaptint = 100; c->adapthist = 0; c->adaptend = 0; c->initcmatn = 0;
    c->N0 = 1.0; c->S02 = 0.0; c->updatesigma = 1; c->condmax = 0.0;
    c->alphatarget = 0.234; c->nuparam = 0.7;
    c->seed = MCMCX_DEFAULT_SEED; c->chain_id0 = 0; c->record_accept = 0; c->record_chain = 0; c->device = 0; c->scam_fast = 0;
}

int mcmcx_create(const mcmcx_config *cfg_in, mcmcx_handle *out)
{
    if (!cfg_in || !out) return fail(-1, "mcmcx_create: null argument");
    mcmcx_config c = *cfg_in;
    // check_mcmcinit_parameters, mcmcinit.F90:235-368
    if (c.adapthist < 0) c.adapthist = 0;
    if (c.adaptint < 0) { c.adaptint = 0; c.doadapt = 0; }
    if (c.burnintime < 0) c.burnintime = 0;
    if (c.badaptint <= 0) c.badaptint = c.adaptint;
    if (c.badaptint == 0) c.doburnin = 0;
    if (c.initcmatn < 0) c.initcmatn = 0;
    if (c.scalelimit < 0.0 || c.scalelimit > 0.5)
        return fail(-2, "ERROR: Scalelimit control variable should be between [0,0.5]");
    if (c.scalefactor < 0.0) c.scalefactor = 1.0;
    if (c.method != MCMCX_METHOD_DRAM && c.method != MCMCX_METHOD_RAM && c.method != MCMCX_METHOD_SCAM &&
        c.method != MCMCX_METHOD_ER) return fail(-3, "unknown method");
    if (c.method == MCMCX_METHOD_ER) c.drscale = 0.0;                                // 'no dr with er', MCMC_run_er.F90:26-29
    if (c.method == MCMCX_METHOD_SCAM) {                                             // mcmcinit.F90:321-330
        if (c.condmax <= 0.0) c.condmax = 1.0e15;
        c.doburnin = 0; c.drscale = 0.0;
    }
    if (c.method == MCMCX_METHOD_RAM) c.drscale = 0.0;
    if (c.method != MCMCX_METHOD_SCAM) c.scam_fast = 0;
    if (c.nsimu < 1) return fail(-4, "nsimu <= 0 stopping");                         // mcmc_main.F90:22-25
    if (c.npar < 1 || c.npar > MCX_MAX_NPAR) return fail(-5, "npar must be in 1.." + std::to_string(MCX_MAX_NPAR));
    if (c.nchains < 1) return fail(-5, "nchains must be >= 1");
    if (c.doadapt && c.method != MCMCX_METHOD_RAM) {
        if (c.adaptint == 0) return fail(-7, "doadapt with adaptint = 0");
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) return fail(-10, "no HIP device: the mcmcx engine has no CPU fallback");
    if (c.device < 0 || c.device >= ndev) return fail(-10, "bad device ordinal");
    HIPCHK(hipSetDevice(c.device));
    mcmcx_engine *h = new mcmcx_engine();
    h->sw.read();
    h->cfg = c; h->d = c.npar; h->P = c.npar * (c.npar + 1) / 2;
    h->ntiles = (c.nchains + 63) / 64; h->nlanes = h->ntiles * 64;
    h->dodr = (c.drscale > 0.0) ? 1 : 0; h->usesvd = (c.condmax > 0.0) ? 1 : 0; h->pooled = c.pooled ? 1 : 0;
    e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(-100, hipGetErrorString(e)); }
    h->own_stream = true;
    *out = h;
    return 0;
}

int mcmcx_destroy(mcmcx_handle h)
{
    if (!h) return 0;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto &p : h->pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (void *p : h->allocs) (void)hipFree(p);
    for (void *p : h->hallocs) (void)hipHostFree(p);
    h->h_cand.release(); h->h_ev.release(); h->h_hx.release();
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    if (h->mod) (void)hipModuleUnload(h->mod);
    delete h;
    return 0;
}

int mcmcx_set_stream(mcmcx_handle h, void *s)
{
    if (!h) return fail(-1, "null handle");
    if (h->inited) return fail(-1, "mcmcx_set_stream after mcmcx_init");
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)s; h->own_stream = false;
    return 0;
}

int mcmcx_set_par0(mcmcx_handle h, const double *p, int32_t n)
{
    if (!h || !p) return fail(-1, "null argument");
    if (n != h->d) return fail(-20, "par0 and npar dont match");                     // MCMC_init.F90:173
    h->par0.assign(p, p + n);
    return 0;
}
int mcmcx_set_cmat0(mcmcx_handle h, const double *c, int32_t n)
{
    if (!h || !c) return fail(-1, "null argument");
    if (n != h->d) return fail(-20, "cmat0 and npar dont match");                    // MCMC_init.F90:207
    h->cmat0.assign(c, c + (size_t)n * n);
    return 0;
}
int mcmcx_set_sigma2nobs(mcmcx_handle h, const double *s2, const int32_t *nobs, int32_t nycol)
{
    if (!h || !s2 || !nobs) return fail(-1, "null argument");
    if (nycol < 1 || nycol > MCX_MAX_NYCOL) return fail(-21, "nycol must be in 1.." + std::to_string(MCX_MAX_NYCOL));
    if (h->inited) return fail(-20, "set sigma2 / nobs before mcmcx_init");
    h->ny = nycol; h->sigma2v.assign(s2, s2 + nycol); h->nobsv.assign(nobs, nobs + nycol);
    h->sigma2 = s2[0]; h->nobs = nobs[0]; h->sigma2ok = true;
    return 0;
}
int mcmcx_set_target_gauss(mcmcx_handle h, const double *mu, const double *lam)
{
    if (!h || !mu || !lam) return fail(-1, "null argument");
    h->tkind = TGT_GAUSS; h->tmu.assign(mu, mu + h->d); h->tlam.assign(lam, lam + (size_t)h->d * h->d);
    return 0;
}
int mcmcx_set_target_banana(mcmcx_handle h, double b)
{
    if (!h) return fail(-1, "null handle");
    if (h->d < 2) return fail(-22, "banana target needs npar >= 2");
    h->tkind = TGT_BANANA; h->tb = b;
    return 0;
}
int mcmcx_set_target_expdata(mcmcx_handle h, int32_t n, const double *x, const double *y)
{
    if (!h || !x || !y || n < 1) return fail(-1, "bad argument");
    if (h->d != 2) return fail(-22, "expdata target needs npar = 2");
    h->tkind = TGT_EXPDATA; h->tx.assign(x, x + n); h->ty.assign(y, y + n);
    return 0;
}
int mcmcx_set_target_expdata_cols(mcmcx_handle h, int32_t n, int32_t nycol, const double *x, const double *y)
{
    if (!h || !x || !y || n < 1) return fail(-1, "bad argument");
    if (nycol < 1 || nycol > MCX_MAX_NYCOL) return fail(-21, "nycol must be in 1.." + std::to_string(MCX_MAX_NYCOL));
    if (h->d != 1 + nycol) return fail(-22, "response-column target needs npar = 1 + nycol");
    h->tkind = TGT_EXPCOLS; h->tncols = nycol; h->tx.assign(x, x + n); h->ty.assign(y, y + (size_t)n * nycol);
    return 0;
}
int mcmcx_set_target_host(mcmcx_handle h, mcmcx_ssfun_t ss, mcmcx_priorfun_t pri, mcmcx_checkbounds_t cb, void *user)
{
    if (!h || !ss) return fail(-1, "mcmcx_set_target_host: ssfunction is required");     // ssfunction0.f90:10-14
    h->tkind = TGT_HOST; h->h_ss = ss; h->h_ss_batch = nullptr; h->h_pri = pri; h->h_cb = cb; h->h_user = user;
    return 0;
}

int mcmcx_set_target_host_batch(mcmcx_handle h, mcmcx_ssfun_batch_t ss_batch, mcmcx_priorfun_t pri, mcmcx_checkbounds_t cb, void *user,
    int32_t nthreads)
{
    if (!h || !ss_batch) return fail(-1, "mcmcx_set_target_host_batch: ssfunction_batch is required");
    if (h->inited) return fail(-20, "set the target before mcmcx_init");
    h->tkind = TGT_HOST; h->h_ss_batch = ss_batch; h->h_ss = nullptr; h->h_pri = pri; h->h_cb = cb; h->h_user = user;
    h->h_threads = nthreads < 1 ? 1 : nthreads;
    return 0;
}

int mcmcx_set_target_external(mcmcx_handle h)
{
    if (!h) return fail(-1, "null handle");
    if (h->inited) return fail(-2, "mcmcx_set_target_external after mcmcx_init");
    h->tkind = TGT_HOST; h->h_ss = nullptr; h->h_ss_batch = nullptr; h->h_pri = nullptr; h->h_cb = nullptr; h->h_user = nullptr;
    h->external = true;
    return 0;
}

int mcmcx_set_target_module(mcmcx_handle h, const char *code_object_path, const char *kernel_name, const void *userdata, int64_t nbytes)
{
    if (!h || !code_object_path || !kernel_name) return fail(-1, "mcmcx_set_target_module: null argument");
    if (h->inited) return fail(-20, "set the target before mcmcx_init");
    HIPCHK(hipSetDevice(h->cfg.device));
    hipError_t e = hipModuleLoad(&h->mod, code_object_path);
    if (e != hipSuccess) return fail(-37, std::string("cannot load the target module ") + code_object_path + ": " + hipGetErrorString(e));
    e = hipModuleGetFunction(&h->mod_fn, h->mod, kernel_name);
    if (e != hipSuccess) return fail(-37,
        std::string("target module: no kernel named ") + kernel_name + " (define it with MCMCX_DEFINE_TARGET, include/mcmcx_target.h)");
    for (const char *suffix : {"_abi", "_max_npar"}) {
        hipDeviceptr_t p = nullptr; size_t sz = 0; int v = 0;
        e = hipModuleGetGlobal(&p, &sz, h->mod, (std::string(kernel_name) + suffix).c_str());
        if (e != hipSuccess || sz != sizeof(int)) return fail(-37,
            std::string("target module: ") + kernel_name + suffix + " is missing (not built with MCMCX_DEFINE_TARGET?)");
        HIPCHK(hipMemcpy(&v, p, sizeof(int), hipMemcpyDeviceToHost));
        if (suffix[1] == 'a' && v != MCMCX_TARGET_ABI) return fail(-37, "target module: built against another mcmcx_target.h (abi " +
            std::to_string(v) + ")");
        if (suffix[1] == 'm' && h->d > v) return fail(-37, "target module: npar = " + std::to_string(h->d)
            + " but the module was compiled with MCMCX_TARGET_MAX_NPAR = " + std::to_string(v));
    }
    {   // response columns the module's kernel has room for (modules built before round 5 have no such symbol: eight)
        hipDeviceptr_t p = nullptr; size_t sz = 0; int v = 8;
        if (hipModuleGetGlobal(&p, &sz, h->mod, (std::string(kernel_name) + "_max_ny").c_str()) == hipSuccess && sz == sizeof(int))
            HIPCHK(hipMemcpy(&v, p, sizeof(int), hipMemcpyDeviceToHost));
        else (void)hipGetLastError();
        h->mod_max_ny = v;
    }
    if (userdata && nbytes > 0) {
        HIPCHK(hipMalloc(&h->d_moddata, (size_t)nbytes));
        h->allocs.push_back(h->d_moddata);
        HIPCHK(hipMemcpy(h->d_moddata, userdata, (size_t)nbytes, hipMemcpyHostToDevice));
    }
    h->tkind = TGT_MODULE;
    return 0;
}

int mcmcx_set_target_host_er(mcmcx_handle h, mcmcx_ssfun_er_t ss_er)
{
    if (!h) return fail(-1, "null handle");
    if (h->inited) return fail(-20, "set the target before mcmcx_init");
    h->h_ss_er = ss_er;
    return 0;
}

int mcmcx_set_bounds(mcmcx_handle h, const double *lo, const double *hi)
{
    if (!h) return fail(-1, "null handle");
    h->has_lo = lo != nullptr; h->has_hi = hi != nullptr;
    if (lo) h->tlo.assign(lo, lo + h->d);
    if (hi) h->thi.assign(hi, hi + h->d);
    return 0;
}
int mcmcx_set_priors(mcmcx_handle h, const double *mu, const double *sig)
{
    if (!h || !mu || !sig) return fail(-1, "null argument");
    h->has_pri = true; h->tpmu.assign(mu, mu + h->d); h->tpsig.assign(sig, sig + h->d);
    return 0;
}

int mcmcx_init(mcmcx_handle h)
{
    if (!h) return fail(-1, "null handle");
    if (h->inited) return fail(1, "Warning(mcmcinit): allready inited");            // MCMC_init.F90:24-27
    HIPCHK(hipSetDevice(h->cfg.device));
    h->sw.read();                                    // the A/B switches as the environment has them now: fixed for the engine's life
    const mcmcx_config &c = h->cfg;
    const int d = h->d, P = h->P, T = h->ntiles;
    if ((int)h->par0.size() != d) return fail(-30, "user initialization error: par0 not set");
    if (h->cmat0.empty()) {                                                           // MCMC_initcmat0: identity
        h->cmat0.assign((size_t)d * d, 0.0);
        for (int i = 0; i < d; ++i) h->cmat0[(size_t)i * d + i] = 1.0;
    }
    if (!h->sigma2ok) { h->sigma2 = 1.0; h->nobs = 1; h->ny = 1; }                    // MCMC_init.F90:52-59
    if (h->ny == 1) { h->sigma2v.assign(1, h->sigma2); h->nobsv.assign(1, h->nobs); }
    const int ny = h->ny;
    if (ny > 1 && !phased(h)) return
        fail(-36, "nycol > 1 needs the host-callback or the response-column target (the other built-in targets have one column)");
    if (h->tkind == TGT_MODULE && ny > h->mod_max_ny)
        return fail(-36, "target module: nycol = " + std::to_string(ny) + " but the module was compiled with MCMCX_TARGET_MAX_NY = " +
            std::to_string(h->mod_max_ny));
    if (h->tkind == TGT_EXPCOLS
        && h->tncols != ny) return fail(-36, "response-column target: mcmcx_set_sigma2nobs must give one sigma2 / nobs per column");
    if (ny > 1 && h->pooled && (!fused_cols(h) || c.method == MCMCX_METHOD_SCAM))
        return fail(-36, "nycol > 1 in pooled mode: the device-resident response-column target only, and not with method = 'scam'");
    if (h->tkind < 0) return fail(-31, "no target: the device engine needs mcmcx_set_target_*");
    std::vector<double> Rp, Cp, Rfull, qstd0;
    int info = host_initial_R(d, h->cmat0, Rp, Cp);
    if (h->usesvd) {                                                                  // Cp (packed cmat0) is still needed
        if (info != 0) { Rp.assign(P, 0.0); info = 0; }
        info = host_initial_svd(d, h->cmat0, c.condmax, c.method == MCMCX_METHOD_SCAM, Rfull, qstd0);
    }
    if (info != 0) return fail(-32, "could not factor the initial covariance");      // MCMC_init.F90:110
    h->S02eff = (c.S02 <= 0.0) ? h->sigma2 : c.S02;                                   // MCMC_init.F90:114-116
    double shape = c.N0 / 2.0 + (double)h->nobs / 2.0;

    EngineDev &E = h->E;
    E.d = d; E.P = P; E.ntiles = T;
    // pooled RAM adapts on the host side: the kernels see a plain Metropolis step
    E.method = (c.method == MCMCX_METHOD_RAM && !h->pooled) ? M_RAM : (c.method == MCMCX_METHOD_ER ? M_ER : M_DRAM);
    E.usesvd = h->usesvd; E.doscam = (c.method == MCMCX_METHOD_SCAM) ? 1 : 0; E.condmax = c.condmax;
    E.scam_fast = c.scam_fast ? 1 : 0;
    E.Rf = E.R2f = E.qstd = E.Gw = E.Vw = nullptr;
    E.greedy = c.greedy; E.adapthist = c.adapthist; E.initcmatn = (double)c.initcmatn;
    E.dodr = h->dodr; E.updatesigma = c.updatesigma; E.doadapt = c.doadapt; E.doburnin = c.doburnin; E.burnintime = c.burnintime;
    E.gam_shape = shape; E.N0S02 = c.N0 * h->S02eff;
    E.ny = ny; E.hs = d + ny; E.ssv = E.s2v = E.ss2v = nullptr; E.gshapev = nullptr;
    E.alphatarget = c.alphatarget; E.drscale = c.drscale; E.scalelimit = c.scalelimit; E.scalefactor = c.scalefactor;
    E.k0 = c.seed; E.chain_id0 = c.chain_id0;
    E.dr_lds = (h->dodr && dr_fits_lds(h)) ? 1 : 0;
    {   // plain AM / Metropolis / ER step kernel: state and scratch vectors in LDS (4 d x 512 bytes per wave) when that costs no
        // occupancy -- eight waves per CU still fit (npar <= 10), or all tiles are resident at once anyway (few chains)
        const char *ev = getenv("MCMCX_LDS_SCRATCH");                      // A/B switch: 0 = off
        const size_t per_wave = (size_t)4 * d * 64 * sizeof(double);
        const int per_cu = (int)((size_t)160 * 1024 / per_wave);
        hipDeviceProp_t prop;
        int cus = 256;
        if (hipGetDeviceProperties(&prop, c.device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        const bool fits = per_cu >= 8 || (per_cu >= 1 && (long long)T <= (long long)per_cu * cus);
        E.lds_scratch = (!h->pooled && !h->dodr && c.method != MCMCX_METHOD_RAM && c.method != MCMCX_METHOD_SCAM && fits && !(ev
            && atoi(ev) == 0)) ? 1 : 0;
        // ... and the packed factor with it (step_kernel_ldsr) where one column panel covers npar and state + factor of all tiles
        // are resident at once: (2 npar + npar (npar + 1) / 2) x 512 bytes per wave, 38 KiB at npar = 10 = four waves per CU
        const size_t per_wave_r = (size_t)(2 * d + P) * 64 * sizeof(double);
        const int per_cu_r = (int)((size_t)160 * 1024 / per_wave_r);
        const bool fits_r = per_cu_r >= 8 || (per_cu_r >= 1 && (long long)T <= (long long)per_cu_r * cus);
        // MCMCX_LDS_SCRATCH=1: the state only (A/B)
        if (E.lds_scratch && !h->usesvd && d <= TW && fits_r && !(ev && atoi(ev) == 1)) E.lds_scratch = 2;
        // method='ram' (per-chain factor, Cholesky form): the factor DCHUD / DCHDD rewrite every iteration, and their rotations, in LDS for
        // the launch where one column panel covers npar and all tiles are resident at once (step_kernel_ram_ldsr)
        if (!h->pooled && c.method == MCMCX_METHOD_RAM && !h->usesvd && d <= RW && fits_r && !(ev && atoi(ev) == 0)) E.lds_scratch = 3;
    }
    // target
    E.tgt.kind = phased(h) ? (int)TGT_HOST : h->tkind;   // the kernels know one phase-cut mode; who evaluates is the host's business
    E.tgt.b = h->tb; E.tgt.ndata = (int)h->tx.size(); E.tgt.ncols = h->tncols;
    E.tgt.mu = E.tgt.x = E.tgt.y = E.tgt.lo = E.tgt.hi = E.tgt.pmu = E.tgt.psig = nullptr;
    int rc;
    E.tgt.lamT = nullptr;
    if (h->tkind == TGT_GAUSS) {
        if ((rc = dev_upload(h, &E.tgt.mu, h->tmu))) return rc;
        std::vector<double> lt((size_t)(d + 4) * d + 64, 0.0);     // transpose; zero pad rows + slack for the panel / MFMA tile reads
        for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) lt[(size_t)j * d + i] = h->tlam[(size_t)i * d + j];
        if ((rc = dev_upload(h, &E.tgt.lamT, lt))) return rc;
    }
    if (h->tkind == TGT_EXPDATA || h->tkind == TGT_EXPCOLS) { if ((rc = dev_upload(h, &E.tgt.x, h->tx))) return rc; if ((rc = dev_upload(h,
        &E.tgt.y, h->ty))) return rc; }
    if (h->has_lo && (rc = dev_upload(h, &E.tgt.lo, h->tlo))) return rc;
    if (h->has_hi && (rc = dev_upload(h, &E.tgt.hi, h->thi))) return rc;
    if (h->has_pri) { if ((rc = dev_upload(h, &E.tgt.pmu, h->tpmu))) return rc; if ((rc = dev_upload(h, &E.tgt.psig, h->tpsig))) return rc;
        }
    if ((rc = dev_upload(h, &E.par0, h->par0))) return rc;
    if ((rc = dev_upload(h, &E.cmat0p, Cp))) return rc;

    // state
    const size_t L = (size_t)T * 64;
    if ((rc = dev_alloc(h, &E.theta, L * d))) return rc;
    // the user's functions on the host and at most sixteen tiles: candidate, results and flags in mapped host memory -- a phase kernel's
    // stores ARE the hand-over, the host's results are read by the next one in place (MCMCX_HOST_MAPPED=0: device buffers and copies)
    h->host_mapped = h->tkind == TGT_HOST && T <= 16 && h->sw.host_mapped != 0;
    // (RAM / SCAM use cs as sweep scratch)
    h->cs_mapped = h->host_mapped && (c.method == MCMCX_METHOD_DRAM || c.method == MCMCX_METHOD_ER);
    if ((rc = h->host_mapped ? host_alloc(h, &E.cand, L * d) : dev_alloc(h, &E.cand, L * d))) return rc;
    if ((rc = dev_alloc(h, &E.zs, L * 2 * d))) return rc;
    if ((rc = h->cs_mapped ? host_alloc(h, &E.cs, L * 2 * d) : dev_alloc(h, &E.cs, L * 2 * d))) return rc;
    E.xscr = nullptr;
    // step_kernel_pooled_dr_big's quadratic-form vectors; npar > 320: adapt_post_kernel's work vector
    if (((h->pooled && h->dodr) || d > 320) && (rc = dev_alloc(h, &E.xscr, L * 2 * d))) return rc;
    if ((rc = dev_alloc(h, &E.scal, L * NSCAL))) return rc;
    if ((rc = dev_alloc(h, &E.ictr, L * NICTR))) return rc;
    if ((rc = dev_alloc(h, &E.rngn, L))) return rc;
    E.R = nullptr;
    if (!h->pooled && (rc = dev_alloc(h, &E.R, L * P, false))) return rc;          // pooled: one shared factor instead
    if ((rc = dev_alloc(h, &E.basetheta, L * d))) return rc;
    if (h->usesvd && h->pooled && c.method == MCMCX_METHOD_SCAM) {     // pooled SCAM: one rotation for every chain
        // npar > 240: slower, not refused -- scam_kernel / scam_mw_kernel on per-chain copies
        h->scam_replicated = scam_pooled_lds(d) > 160 * 1024;
        if (h->scam_replicated) {
            if ((rc = dev_alloc(h, &E.Rf, L * (size_t)d * d, false))) return rc;
            if ((rc = dev_alloc(h, &E.qstd, L * d))) return rc;
        } else if ((rc = dev_alloc(h, &h->d_sharedU, 2 * shared_u_stride(h) + d, false))) return rc;
        h->pool_U = Rfull; h->pool_std = qstd0;
        if ((rc = upload_shared_u(h))) return rc;
    } else if (h->usesvd && h->pooled) {                // pooled AM with the SVD factor: one full matrix for every chain (below)
        h->pool_Rf = Rfull;
    } else if (h->usesvd) {
        const size_t DD = (size_t)d * d;
        if ((rc = dev_alloc(h, &E.Rf, L * DD, false))) return rc;
        if (c.method != MCMCX_METHOD_RAM) {                 // work space of the adaptation's SVD; RAM never refactors
            if ((rc = dev_alloc(h, &E.Gw, L * DD))) return rc;
            if ((rc = dev_alloc(h, &E.Vw, L * DD))) return rc;
            if (svd_blocked(h) && (c.doadapt != 0 || c.doburnin != 0)) {      // chain-major copies for svd_blocked_kernel
                if ((rc = dev_alloc(h, &h->d_Gc, L * DD, false))) return rc;
                if ((rc = dev_alloc(h, &h->d_Vc, L * DD, false))) return rc;
                if ((rc = dev_alloc(h, &h->d_svc, L * d, false))) return rc;
                if ((rc = dev_alloc(h, &h->d_need, L))) return rc;
                if ((rc = dev_alloc(h, &h->d_state, L))) return rc;
                if ((rc = dev_alloc(h, &h->d_anyrot, 1))) return rc;
            }
        }
        if ((rc = dev_alloc(h, &E.qstd, L * d))) return rc;
        if (h->dodr && (rc = dev_alloc(h, &E.R2f, L * DD, false))) return rc;
        if ((rc = dev_bcast(h, E.Rf, Rfull))) return rc;
        if ((rc = dev_bcast(h, E.qstd, qstd0))) return rc;
        if (h->dodr) {
            std::vector<double> r2 = Rfull;
            for (auto &v : r2) v = v / c.drscale;
            if ((rc = dev_bcast(h, E.R2f, r2))) return rc;
        }
    }
    E.R2 = E.iC = nullptr;
    if (h->dodr && !h->pooled) {                        // pooled: one R2 and one iC for every chain (below)
        if ((rc = dev_alloc(h, &E.R2, L * P, false))) return rc;
        if ((rc = dev_alloc(h, &E.iC, L * P, false))) return rc;
    }
    const bool am = (c.method != MCMCX_METHOD_RAM) && (c.doadapt != 0 || c.doburnin != 0) && !h->pooled;
    E.cmat = E.mean = E.Rtmp = nullptr; E.rowlist = nullptr;
    // history ring
    const bool need_hist = am || c.record_chain;
    h->wcap = 0; E.hist = E.s2hist = nullptr; E.wacc = nullptr; E.record_s2 = 0;
    if (need_hist) {
        // AP windows (adapthist > 1) may reach back to an arbitrarily old row: keep everything.  Otherwise the longest
        // window is the first AM one, rows 2 .. T1 with T1 the first multiple of adaptint / badaptint that is
        // >= burnintime + adaptint + adapthist (MCMC_adapt.F90:42-46,105): up to adaptint - 1 iterations past that
        // threshold when it is not a multiple itself (burn-in ticks only rescale and never restart the window)
        long long wc = (c.record_chain || (am && c.adapthist > 1)) ? (long long)c.nsimu + 1
                                      : (long long)c.burnintime + 2LL * c.adaptint + c.adapthist + 2;
        if (wc > (long long)c.nsimu + 1) wc = (long long)c.nsimu + 1;
        h->wcap = (int)wc;
        if ((rc = dev_alloc(h, &E.hist, L * (size_t)h->wcap * (d + ny), false))) return rc;
        if ((rc = dev_alloc(h, &E.wacc, (size_t)T * h->wcap))) return rc;
        if (c.record_chain && c.updatesigma) { E.record_s2 = 1; if ((rc = dev_alloc(h, &E.s2hist, L * (size_t)h->wcap * ny))) return rc; }
    }
    E.wcap = h->wcap > 0 ? h->wcap : 1;
    if (am) {
        if ((rc = dev_alloc(h, &E.cmat, L * P))) return rc;
        if ((rc = dev_alloc(h, &E.mean, L * d))) return rc;
        if ((rc = dev_alloc(h, &E.Rtmp, L * P))) return rc;
        if ((rc = dev_alloc(h, &E.rowlist, L * (size_t)(h->wcap + 1)))) return rc;
    }
    E.sharedR = nullptr;
    if (h->pooled) {
        if ((long long)c.nchains * (h->comm ? h->comm->nranks
            : 1) < 2) return fail(-8, "pooled mode needs at least 2 chains over all ranks");
        if (phase_cut(h) || (fused_cols(h) && c.method == MCMCX_METHOD_SCAM))
            return fail(-8, "pooled mode needs one of the single-launch device targets (gauss, banana, expdata, expcols; scam: not expcols)");
        if ((rc = dev_alloc(h, &h->d_sharedR, (size_t)P, false))) return rc;
        HIPCHK(hipMemcpy(h->d_sharedR, Rp.data(), (size_t)P * 8, hipMemcpyHostToDevice));
        E.sharedR = h->d_sharedR;
        h->pool_R = Rp; h->pool_C = Cp; h->pool_mean = h->par0; h->pool_W = (double)c.initcmatn;
        if (c.method == MCMCX_METHOD_DRAM || c.method == MCMCX_METHOD_RAM || c.method == MCMCX_METHOD_ER) {
            if ((rc = dev_alloc(h, &h->d_sharedRT, (size_t)((d + 3) & ~3) * d + PWS, false))) return rc;
            if ((rc = h->usesvd ? upload_shared_rf(h) : upload_shared_rt(h))) return rc;
            if (h->usesvd) E.sharedR = h->d_sharedRT;   // the lane-per-chain kernel reads the full matrix through the scalar cache
        }
    }
    E.hev = E.hx = nullptr;
    if (phased(h)) {
        if ((rc = h->host_mapped ? host_alloc(h, &E.hev, L * (NHE - 1 + ny)) : dev_alloc(h, &E.hev, L * (NHE - 1 + ny)))) return rc;
        if (ny > 1) {
            if ((rc = dev_alloc(h, &E.ssv, L * ny))) return rc;
            if ((rc = dev_alloc(h, &E.s2v, L * ny))) return rc;
            if ((rc = dev_alloc(h, &E.ss2v, L * ny))) return rc;
            if ((rc = dev_bcast(h, E.s2v, h->sigma2v))) return rc;
            std::vector<double> gs(ny);
            for (int j = 0; j < ny; ++j) gs[j] = c.N0 / 2.0 + (double)h->nobsv[j] / 2.0;
            if ((rc = dev_upload(h, &E.gshapev, gs))) return rc;
        }
        if ((rc = h->host_mapped ? host_alloc(h, &E.hx, L * NHX) : dev_alloc(h, &E.hx, L * NHX))) return rc;
    }
    E.accmask = nullptr;
    if (c.record_accept && (rc = dev_alloc(h, &E.accmask, (size_t)c.nsimu * T))) return rc;
    // the lane-group step kernel where it covers the configuration: MCMCX_GROUP = 1 / 0 forces it on / off (A/B, tests)
    h->group_d4 = 0;
    if (group_covers(h)) {
        const char *ev = getenv("MCMCX_GROUP");
        int ex = 0;
        const bool pow2 = h->dodr && c.drscale > 0.0 && std::frexp(c.drscale, &ex) == 0.5 && ex > -64 && ex < 64;
        // (the power-of-two form keeps iC in LDS: above npar 24 that leaves fewer waves per CU than the register form's four)
        const int drm = !h->dodr ? 0 : (pow2 && d <= 24 && !(getenv("MCMCX_GROUP_DR2") && atoi(getenv("MCMCX_GROUP_DR2")) == 0)) ? 2 : 1;
        const char *gwe = getenv("MCMCX_GROUP_GW");                                             // (4 / 16: A/B, tests)
        const int gw = (d <= 16 && gwe) ? (atoi(gwe) == 4 ? 4 : 16) : group_width(h);
        const bool on = ev ? atoi(ev) != 0 : group_wins(h, drm, gw);
        if (on) h->group_d4 = d <= 32 ? ((d + 3) & ~3) : ((d + 7) & ~7);
        if (h->group_d4 && (E.hist || E.accmask) && (rc = dev_alloc(h, &h->d_accb, L * (size_t)GROUP_MAXSEG))) return rc;
        if (h->group_d4) {
            h->group_gw = gw;
            h->group_drm = drm;
            h->group_check_due = true;
            if (h->group_drm == 2 && (rc = dev_alloc(h, &h->d_gflag, 1))) return rc;
        }
    }
    // method = 'ram' with few chains on the lane-group kernel (MCMCX_RAM_GROUP = 1 / 0: always where it covers / never; A/B, tests)
    h->ram_group_d4 = 0;
    if (ram_group_covers(h)) {
        const char *rg = getenv("MCMCX_RAM_GROUP");
        if (!rg) rg = getenv("MCMCX_GROUP");               // (the kernel-family switch of the tests covers it too)
        if (rg ? atoi(rg) != 0 : ram_group_wins(h)) {
            h->ram_group_d4 = d <= 16 ? 16 : d <= 32 ? 32 : d <= 56 ? 56 : 64;
            if ((E.hist || E.accmask) && (rc = dev_alloc(h, &h->d_accb, L * (size_t)GROUP_MAXSEG))) return rc;
        }
    }
    // the adaptation's factorisation (Cholesky branch) with the matrices in LDS (tile_factor_kernel): any chain count, npar <= 32 -- config
    // 3's size (npar 20, delayed rejection, 262144 chains): 2.11 -> 1.44 ms per tick, bound by VALU issue (~25 instructions per inner step
    // of four chains). Above npar 32 adapt_post_kernel's 8 x 8 register blocks stay: at npar 50 x 1 048 576 chains they stream the matrices
    // ~5.6 times (16.0 ms) and still beat the LDS form, whose three waves per CU issue ~15 x the instructions per chain (35.2 ms;
    // profiles/r05_a/tick_ab.txt). MCMCX_TILE_FACTOR = 0 / 1: never / up to npar 64 (test switch: both forms on one problem).
    {
        const char *tf = getenv("MCMCX_TILE_FACTOR");
        h->tile_factor = am && !h->usesvd && d <= ((tf && atoi(tf) == 1) ? 64 : 32) && !(tf && atoi(tf) == 0);
    }
    // 1/simuind**nuparam, computed like the reference: real(simuind) is default REAL (MCMC_run_ram.F90:166)
    {
        std::vector<double> rs((size_t)c.nsimu + 2, 0.0);
        for (int it = 1; it <= c.nsimu; ++it) rs[it] = 1.0 / std::pow((double)(float)it, c.nuparam);
        const double *p = nullptr;
        if ((rc = dev_upload(h, &p, rs))) return rc;
        h->d_ramscale = const_cast<double *>(p);
    }
    // longest pooled vector: 2 + d + P
    if ((rc = dev_alloc(h, &h->d_moments, (size_t)(T + 1) * (2 + d + P)))) return rc;
    // longest vector + the stop flag
    if ((rc = dev_alloc(h, &h->d_gather, (size_t)(h->comm ? h->comm->nranks : 1) * (3 + d + P)))) return rc;
    if ((rc = dev_alloc(h, &h->d_pooled, (size_t)(3 + d + P)))) return rc;

    // fill theta = par0, R = R(cmat0), chaincmat = cmat0, chainmean = par0, scalars
    {
        std::vector<double> sc(L * NSCAL, 0.0);
        std::vector<uint32_t> ic(L * NICTR, 0u);
        for (int t = 0; t < T; ++t)
            for (int l = 0; l < 64; ++l) {
                sc[((size_t)t * NSCAL + S_SIGMA2) * 64 + l] = h->sigma2;
                sc[((size_t)t * NSCAL + S_WSUM) * 64 + l] = (double)c.initcmatn;
                ic[((size_t)t * NICTR + I_CHAININD) * 64 + l] = 1;
                ic[((size_t)t * NICTR + I_CURCOUNT) * 64 + l] = 1;
                ic[((size_t)t * NICTR + I_BASECNT) * 64 + l] = 1;
                ic[((size_t)t * NICTR + I_WINSTART) * 64 + l] = 2;
            }
        HIPCHK(hipMemcpyAsync(E.scal, sc.data(), sc.size() * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(E.ictr, ic.data(), ic.size() * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if ((rc = dev_bcast(h, E.theta, h->par0))) return rc;
        if (!h->pooled && (rc = dev_bcast(h, E.R, Rp))) return rc;
        if (h->dodr) {                                   // iC = dpotri(R), R2 = R/drscale, MCMC_adapt.F90:216-225
            std::vector<double> iCp = Rp, R2p(P);
            if (h->usesvd) for (int j = 0; j < d; ++j) for (int i = 0; i <= j; ++i) iCp[h_pidx(i, j, d)] = Rfull[(size_t)j * d + i];
            if (host_potri(d, iCp) != 0) return fail(-34, "ERROR: cannot invert cmat");
            for (int e = 0; e < P; ++e) R2p[e] = Rp[e] / c.drscale;
            if (h->pooled) {
                h->pool_iC = iCp;
                if (h->usesvd) { h->pool_R2 = Rfull; for (auto &v : h->pool_R2) v = v / c.drscale; } else h->pool_R2 = R2p;
                if ((rc = dev_alloc(h, &h->d_sharedR2, h->usesvd ? (size_t)((d + 3) & ~3) * d + PWS : (size_t)P, false))) return rc;
                if ((rc = dev_alloc(h, &h->d_sharediC, (size_t)P, false))) return rc;
                if (c.method == MCMCX_METHOD_DRAM && h->d_sharedRT && pooled_mfma_lds(d) <= 160 * 1024 &&
                    !(h->sw.pooled_scalar > 0)) {          // the second stage on the matrix cores too
                    if ((rc = dev_alloc(h, &h->d_sharedR2T, (size_t)((d + 3) & ~3) * d + PWS, false))) return rc;
                    if ((rc = dev_alloc(h, &h->d_sharediCd, (size_t)((d + 3) & ~3) * d + PWS, false))) return rc;
                }
                if ((rc = pooled_upload_dr(h, false))) return rc;
            } else {
            if ((rc = dev_bcast(h, E.R2, R2p))) return rc;
            if ((rc = dev_bcast(h, E.iC, iCp))) return rc;
            }
        }
        if (am) {
            if ((rc = dev_bcast(h, E.cmat, Cp))) return rc;
            if ((rc = dev_bcast(h, E.mean, h->par0))) return rc;
        }
    }
    if (h->external) {                                  // MCMC_run1: every evaluation is the caller's, the first one included
        if (h->pooled) return fail(-8, "mcmcx_set_target_external: not in pooled mode");
        if ((rc = dev_alloc(h, &h->d_r1, L * (size_t)(3 * d + 3 * ny + NR1)))) return rc;
    } else if (phased(h)) {                             // first point: sspri1, ss1 from the callbacks (MCMC_run.F90:35-36)
        int rc2 = host_eval(h, E.theta, d, false);
        if (rc2) return rc2;
    }
    launch_init(h);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    h->simuind = 1;
    h->inited = true;
    return 0;
}

int mcmcx_install_signal_handlers(void)
{
    struct sigaction act;
    memset(&act, 0, sizeof(act));
    act.sa_handler = on_signal;
    sigemptyset(&act.sa_mask);
    const int sigs[] = {SIGHUP, SIGINT, SIGTERM, SIGTSTP, SIGUSR1, SIGUSR2};
    for (int sg : sigs) if (sigaction(sg, &act, nullptr) != 0) return fail(-1, "sigaction failed");
    g_sig_installed = true;
    return 0;
}
int mcmcx_interrupted(void) { return g_interrupt ? 1 : 0; }
void mcmcx_clear_interrupt(void) { g_interrupt = 0; }

static int run_impl(mcmcx_handle h, int32_t upto);
int mcmcx_run(mcmcx_handle h, int32_t upto)
{
    const int rc = run_impl(h, upto);
    // Several ranks meet in this engine's ticks: one that fails must not leave the others waiting in the next gather --
    // the communicator is marked failed, which the peers' waits poll (comm_wait_stream, shm_barrier)
    // -- once the run has entered its loop, that is: an argument or state error returned before that (-40 not inited, -41 external
    // target) strands nobody, and marking the communicator for it would abort peers that have nothing to do with it
    if (rc < 0 && h && h->inited && h->run_entered && collective_run(h)) comm_mark_failed(h->comm);
    if (h) h->run_entered = false;
    return rc;
}
static int run_impl(mcmcx_handle h, int32_t upto)
{
    if (!h) return fail(-1, "null handle");
    if (!h->inited) return fail(-40, "we have not inited");                           // MCMC_run.F90:22
    if (h->external) return fail(-41, "mcmcx_run: the target is external (mcmcx_set_target_external): drive the chain with mcmcx_run1_*");
    // (with the phases fused, iteration it + 1's proposal -- Philox draws included -- has already run when an error surfaces in iteration
    // it's evaluation; a second run on the handle would draw that proposal again and leave the reference's stream order silently: ADVICE
    // round 5)
    if (h->failed) return fail(-42, "mcmcx_run: an earlier run on this handle failed inside a host-callback iteration; its chains are in an undefined state -- destroy the handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    const mcmcx_config &c = h->cfg;
    if (upto > c.nsimu) upto = c.nsimu;
    const int maxseg = (h->group_d4 || h->ram_group_d4) ? GROUP_MAXSEG : (c.method == MCMCX_METHOD_RAM && !h->pooled) ? 4096 : 1 << 30;
    int it = h->simuind + 1;
    // Several ranks that meet in this engine's ticks (pooled mode with a communicator): a rank that left the loop alone --
    // on a signal it happened to see first, or on an error of its own -- would leave its peers waiting in the next gather.
    // So a signal only raises this rank's stop flag in the exchanged vector and the run is left at the tick where every
    // rank reads the summed flag (pooled_reduce); an error marks the communicator failed (mcmcx_run).
    const bool coll = collective_run(h);
    h->run_entered = true;
    // ... which needs a tick ahead.  Whether one lies in (it, upto] follows from the configuration alone, the same on every
    // rank: past the last one (doadapt = 0, iterations beyond adaptend, the tail of a run) no collective is left to strand a
    // peer in, and a signal is acted on at the next launch boundary like in a run of one rank.
    int last_tick = 0;
    if (coll) for (int i2 = upto; i2 >= it; --i2) if (adapt_mode(c, i2) != 0 || pooled_ram_due(h, i2)) { last_tick = i2; break; }
    h->stop_seen = false;
    while (it <= upto) {
        if (g_interrupt && (!coll || it > last_tick)) {     // a caught signal: stop at this launch boundary
            int rc = mcmcx_sync(h); if (rc) return rc;
            h->simuind = it - 1;
            return MCMCX_INTERRUPTED;
        }
        int end = it, mode = 0;
        bool ramtick = false;
        for (;; ++end) {                                    // extend the launch up to the next tick
            mode = adapt_mode(c, end);
            ramtick = pooled_ram_due(h, end);
            if (mode != 0 || ramtick || end == upto || end - it + 1 >= maxseg) break;
        }
        if (phase_cut(h)) {
            for (int i2 = it; i2 <= end; ++i2) { int rc = host_iteration(h, i2, i2 < end); if (rc) { h->failed = true; return rc; } }
        } else {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            hipError_t er = hipEventCreate(&e0);
            if (er == hipSuccess) er = hipEventCreate(&e1);
            if (er == hipSuccess) er = hipEventRecord(e0, h->stream);
            if (er == hipSuccess) {
                h->launch_err.clear();
                if (h->cfg.method == MCMCX_METHOD_SCAM) launch_scam(h, it, end); else launch_step(h, it, end);
                er = hipGetLastError();
            }
            // a cover predicate and the dispatch disagree: fail, never skip iterations silently
            if (er == hipSuccess && !h->launch_err.empty()) {
                if (e0) (void)hipEventDestroy(e0);
                if (e1) (void)hipEventDestroy(e1);
                return fail(-103, "step launch: " + h->launch_err);
            }
            if (er == hipSuccess) er = hipEventRecord(e1, h->stream);
            if (er != hipSuccess) {                       // nothing is left behind on the error path
                if (e0) (void)hipEventDestroy(e0);
                if (e1) (void)hipEventDestroy(e1);
                return fail(-100, std::string("step launch: ") + hipGetErrorString(er));
            }
            h->pending.emplace_back(e0, e1);
            h->launches += 1; h->steps += (end - it + 1);
        }
        int trc = 0;
        if (mode != 0) {
            if (h->pooled) trc = pooled_tick(h, end, mode);
            else { launch_adapt(h, end, mode); h->group_check_due = true; HIPCHK(hipGetLastError()); }
        }
        if (ramtick && trc == 0) trc = pooled_ram_tick(h, end);
        if (trc) return trc;
        if (h->stop_seen) {                                 // every rank read the same summed stop flag at this tick: leave together,
            h->stop_seen = false;                           // after iteration `end` WITH its adaptation applied (a resumed run
            int rc = mcmcx_sync(h); if (rc) return rc;      // continues at end + 1 like an uninterrupted one)
            h->simuind = end;
            return MCMCX_INTERRUPTED;
        }
        it = end + 1;
        if (g_sig_installed) { HIPCHK(hipStreamSynchronize(h->stream)); h->simuind = std::max(h->simuind, end); }
        if (h->pending.size() > 4096) { int rc = mcmcx_sync(h); if (rc) return rc; }
    }
    h->simuind = std::max(h->simuind, (int)upto);
    return 0;
}

int mcmcx_run1_decide(mcmcx_handle h, int32_t drstage, const double *oldpar2, const double *ssprev2, const double *sspri2,
                      const double *oldpar1, const double *ssprev1, const double *sspri1, const double *alpha12,
                      const double *newpar, const double *ss, const double *sspri, double *alpha_out, int32_t *reject_out)
{
    int rc = run1_check(h); if (rc) return rc;
    if (!oldpar1 || !ssprev1 || !sspri1 || !newpar || !ss || !sspri || !alpha_out
        || !reject_out) return fail(-1, "mcmcx_run1_decide: null argument");
    const bool dr2 = drstage > 1 && h->dodr;
    if (dr2 && (!oldpar2 || !ssprev2 || !sspri2
        || !alpha12)) return fail(-1, "mcmcx_run1_decide: the second stage needs oldpar2, ssprev2, sspri2, alpha12");
    const int d = h->d, ny = h->ny, n = h->cfg.nchains, s0 = 3 * d + 3 * ny;
    h->h_r1.assign((size_t)h->ntiles * 64 * (s0 + NR1), 0.0);
    run1_put(h, 0, d, dr2 ? oldpar2 : nullptr); run1_put(h, d, d, oldpar1); run1_put(h, 2 * d, d, newpar);
    run1_put(h, 3 * d, ny, dr2 ? ssprev2 : nullptr); run1_put(h, 3 * d + ny, ny, ssprev1); run1_put(h, 3 * d + 2 * ny, ny, ss);
    run1_put(h, s0 + R1_PRI2, 1, dr2 ? sspri2 : nullptr); run1_put(h, s0 + R1_PRI1, 1, sspri1); run1_put(h, s0 + R1_PRI, 1, sspri);
    run1_put(h, s0 + R1_A12, 1, dr2 ? alpha12 : nullptr);
    if ((rc = run1_launch<0>(h, drstage))) return rc;
    run1_get(h, s0 + R1_ALPHA, 1, alpha_out);
    std::vector<double> rj(n);
    run1_get(h, s0 + R1_REJECT, 1, rj.data());
    for (int c = 0; c < n; ++c) reject_out[c] = rj[c] != 0.0 ? 1 : 0;
    return 0;
}

int mcmcx_run1_propose(mcmcx_handle h, int32_t stage, const double *from, double *newpar_out)
{
    int rc = run1_check(h); if (rc) return rc;
    if (!from || !newpar_out) return fail(-1, "mcmcx_run1_propose: null argument");
    const int d = h->d, ny = h->ny;
    h->h_r1.assign((size_t)h->ntiles * 64 * (3 * d + 3 * ny + NR1), 0.0);
    run1_put(h, 0, d, from);
    if (stage > 1 && h->dodr) rc = run1_launch<2>(h, 2); else rc = run1_launch<1>(h, 1);
    if (rc) return rc;
    run1_get(h, 2 * d, d, newpar_out);
    return 0;
}

int mcmcx_run1_sscrit(mcmcx_handle h, const double *ssprev1, const double *sspri1, double *sscrit_out)
{
    int rc = run1_check(h); if (rc) return rc;
    if (!ssprev1 || !sspri1 || !sscrit_out) return fail(-1, "mcmcx_run1_sscrit: null argument");
    const int d = h->d, ny = h->ny, s0 = 3 * d + 3 * ny;
    h->h_r1.assign((size_t)h->ntiles * 64 * (s0 + NR1), 0.0);
    run1_put(h, 3 * d + ny, ny, ssprev1); run1_put(h, s0 + R1_PRI1, 1, sspri1);
    if ((rc = run1_launch<3>(h, 1))) return rc;
    run1_get(h, s0 + R1_CRIT, 1, sscrit_out);
    return 0;
}

int mcmcx_sync(mcmcx_handle h)
{
    if (!h) return fail(-1, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (auto &p : h->pending) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, p.first, p.second));
        h->ms_total += ms;
        (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second);
    }
    h->pending.clear();
    return 0;
}

int mcmcx_kernel_time(mcmcx_handle h, double *ms, int64_t *launches, int64_t *steps, int reset)
{
    int rc = mcmcx_sync(h);
    if (rc) return rc;
    if (ms) *ms = h->ms_total;
    if (launches) *launches = h->launches;
    if (steps) *steps = h->steps;
    if (reset) { h->ms_total = 0.0; h->launches = 0; h->steps = 0; }
    return 0;
}

int32_t mcmcx_simuind(mcmcx_handle h) { return h ? h->simuind : -1; }



int mcmcx_get_counters(mcmcx_handle h, int32_t chain, int32_t *out)
{
    int rc = check_chain(h, chain); if (rc) return rc;
    std::vector<uint32_t> v;
    if ((rc = fetch_chain_vec(h, h->E.ictr, NICTR, chain, v))) return rc;
    out[0] = (int32_t)v[I_STAYED]; out[1] = (int32_t)v[I_BNDSTAYED]; out[2] = (int32_t)v[I_DRACC]; out[3] = (int32_t)v[I_DRTRIES];
    out[4] = (int32_t)v[I_CHAININD]; out[5] = (int32_t)v[I_STATUS]; out[6] = (int32_t)v[I_ERSTAYED]; out[7] = (int32_t)v[I_CURCOUNT];
    return 0;
}

int mcmcx_get_totals_n(mcmcx_handle h, int64_t *out, int32_t n)
{
    if (!out || n < 1) return fail(-1, "mcmcx_get_totals_n: bad argument");
    int64_t t7[7];
    int rc = mcmcx_get_totals(h, t7); if (rc) return rc;
    for (int i = 0; i < n; ++i) out[i] = i < 7 ? t7[i] : 0;
    return 0;
}

int mcmcx_get_totals(mcmcx_handle h, int64_t *t7)
{
    if (!h || !h->inited) return fail(-40, "we have not inited");
    std::vector<uint32_t> v;
    int rc = fetch(h, h->E.ictr, (size_t)h->nlanes * NICTR, v); if (rc) return rc;
    for (int i = 0; i < 7; ++i) t7[i] = 0;
    for (int c = 0; c < h->cfg.nchains; ++c) {
        int t = c / 64, l = c % 64;
        auto at = [&](int k) { return (int64_t)v[((size_t)t * NICTR + k) * 64 + l]; };
        t7[0] += at(I_STAYED); t7[1] += at(I_BNDSTAYED); t7[2] += at(I_DRACC); t7[3] += at(I_DRTRIES);
        t7[5] += at(I_DOWNS); t7[6] |= at(I_STATUS);
    }
    t7[6] |= h->pool_status;                            // pooled RAM: a tick whose Gram matrix was not positive definite was skipped
    // proposals evaluated: one per iteration (d componentwise ones with method='scam') + the delayed-rejection tries
    t7[4] = (int64_t)h->cfg.nchains * (int64_t)(h->simuind - 1) * (h->cfg.method == MCMCX_METHOD_SCAM ? h->d : 1) + t7[3];
    return 0;
}

int mcmcx_get_theta(mcmcx_handle h, double *out)
{
    if (!h || !h->inited) return fail(-40, "we have not inited");
    std::vector<double> v;
    int rc = fetch(h, h->E.theta, (size_t)h->nlanes * h->d, v); if (rc) return rc;
    for (int c = 0; c < h->cfg.nchains; ++c)
        for (int k = 0; k < h->d; ++k) out[(size_t)c * h->d + k] = v[((size_t)(c / 64) * h->d + k) * 64 + (c % 64)];
    return 0;
}

int mcmcx_get_scalars(mcmcx_handle h, double *out)
{
    if (!h || !h->inited) return fail(-40, "we have not inited");
    std::vector<double> v;
    int rc = fetch(h, h->E.scal, (size_t)h->nlanes * NSCAL, v); if (rc) return rc;
    const int idx[4] = {S_SS1, S_PRI1, S_SIGMA2, S_ALPHA12};
    for (int c = 0; c < h->cfg.nchains; ++c)
        for (int k = 0; k < 4; ++k) out[(size_t)c * 4 + k] = v[((size_t)(c / 64) * NSCAL + idx[k]) * 64 + (c % 64)];
    return 0;
}

int mcmcx_get_rng(mcmcx_handle h, int32_t chain, uint64_t *n, int32_t *saved, double *saved_y)
{
    int rc = check_chain(h, chain); if (rc) return rc;
    std::vector<uint64_t> vn; std::vector<uint32_t> vi; std::vector<double> vs;
    if ((rc = fetch_chain_vec(h, h->E.rngn, 1, chain, vn))) return rc;
    if ((rc = fetch_chain_vec(h, h->E.ictr, NICTR, chain, vi))) return rc;
    if ((rc = fetch_chain_vec(h, h->E.scal, NSCAL, chain, vs))) return rc;
    if (n) *n = vn[0];
    if (saved) *saved = (int32_t)vi[I_SAVED];
    if (saved_y) *saved_y = vs[S_SAVEDY];
    return 0;
}

int mcmcx_get_R(mcmcx_handle h, int32_t chain, double *R)
{
    int rc = check_chain(h, chain); if (rc) return rc;
    std::vector<double> p;
    if (h->pooled && h->usesvd) { const auto &M = h->cfg.method == MCMCX_METHOD_SCAM ? h->pool_U : h->pool_Rf; memcpy(R, M.data(),
        sizeof(double) * M.size()); return 0; }
    if (h->pooled) { unpack_upper(h->d, h->pool_R, R, false); return 0; }
    if (h->usesvd) {                                     // full column-major factor
        if ((rc = fetch_chain_vec(h, h->E.Rf, h->d * h->d, chain, p))) return rc;
        memcpy(R, p.data(), sizeof(double) * p.size());
        return 0;
    }
    if ((rc = fetch_chain_vec(h, h->E.R, h->P, chain, p))) return rc;
    unpack_upper(h->d, p, R, false);
    return 0;
}

int mcmcx_get_qcovstd(mcmcx_handle h, int32_t chain, double *std)
{
    int rc = check_chain(h, chain); if (rc) return rc;
    if (h->pooled && h->usesvd && h->cfg.method == MCMCX_METHOD_SCAM) { memcpy(std, h->pool_std.data(),
        sizeof(double) * h->pool_std.size()); return 0; }
    if (!h->E.qstd) return fail(-45, "no SVD state (condmax = 0)");
    std::vector<double> p;
    if ((rc = fetch_chain_vec(h, h->E.qstd, h->d, chain, p))) return rc;
    memcpy(std, p.data(), sizeof(double) * p.size());
    return 0;
}

int mcmcx_get_dr(mcmcx_handle h, int32_t chain, double *R2, double *iC)
{
    int rc = check_chain(h, chain); if (rc) return rc;
    if (!h->dodr) return fail(-43, "drscale = 0: no delayed-rejection state");
    std::vector<double> p;
    if (h->pooled) {                                    // one pair of tables for every chain
        if (R2 && h->usesvd) memcpy(R2, h->pool_R2.data(), sizeof(double) * h->pool_R2.size());
        else if (R2) unpack_upper(h->d, h->pool_R2, R2, false);
        if (iC) unpack_upper(h->d, h->pool_iC, iC, false);
        return 0;
    }
    if (R2 && h->usesvd) { if ((rc = fetch_chain_vec(h, h->E.R2f, h->d * h->d, chain, p))) return rc; memcpy(R2, p.data(),
        sizeof(double) * p.size()); }
    else if (R2) { if ((rc = fetch_chain_vec(h, h->E.R2, h->P, chain, p))) return rc; unpack_upper(h->d, p, R2, false); }
    if (iC) { if ((rc = fetch_chain_vec(h, h->E.iC, h->P, chain, p))) return rc; unpack_upper(h->d, p, iC, false); }
    return 0;
}

int mcmcx_get_chaincov(mcmcx_handle h, int32_t chain, double *cmat, double *mean, double *wsum)
{
    int rc = check_chain(h, chain); if (rc) return rc;
    std::vector<double> s;
    if ((rc = fetch_chain_vec(h, h->E.scal, NSCAL, chain, s))) return rc;
    if (wsum) *wsum = s[S_WSUM];
    if (!h->E.cmat) {                                   // no AM state on the device: chaincmat = cmat0, chainmean = par0
        if (cmat) memcpy(cmat, h->cmat0.data(), sizeof(double) * (size_t)h->d * h->d);
        if (mean) memcpy(mean, h->par0.data(), sizeof(double) * (size_t)h->d);
        return 0;
    }
    std::vector<double> p, m;
    if ((rc = fetch_chain_vec(h, h->E.cmat, h->P, chain, p))) return rc;
    if ((rc = fetch_chain_vec(h, h->E.mean, h->d, chain, m))) return rc;
    if (cmat) unpack_upper(h->d, p, cmat, true);
    if (mean) memcpy(mean, m.data(), sizeof(double) * (size_t)h->d);
    return 0;
}

int mcmcx_get_accept_masks(mcmcx_handle h, uint64_t *masks, int32_t *ntiles)
{
    if (!h || !h->inited) return fail(-40, "we have not inited");
    if (ntiles) *ntiles = h->ntiles;
    if (!h->E.accmask) return fail(-42, "record_accept was not requested");
    if (!masks) return 0;
    std::vector<uint64_t> v;
    int rc = fetch(h, h->E.accmask, (size_t)h->simuind * h->ntiles, v); if (rc) return rc;
    memcpy(masks, v.data(), v.size() * 8);
    return 0;
}

int mcmcx_get_accepted(mcmcx_handle h, int32_t chain, uint8_t *acc)
{
    int rc = check_chain(h, chain); if (rc) return rc;
    const int tile = chain / 64, lane = chain % 64;
    if (h->E.accmask) {
        std::vector<uint64_t> v;
        if ((rc = fetch(h, h->E.accmask, (size_t)h->simuind * h->ntiles, v))) return rc;
        for (int it = 1; it <= h->simuind; ++it) acc[it - 1] = (uint8_t)((v[(size_t)(it - 1) * h->ntiles + tile] >> lane) & 1ull);
        return 0;
    }
    if (h->cfg.record_chain && h->E.wacc) {
        std::vector<uint64_t> v;
        if ((rc = fetch(h, h->E.wacc + (size_t)tile * h->wcap, (size_t)h->wcap, v))) return rc;
        for (int it = 1; it <= h->simuind; ++it) acc[it - 1] = (uint8_t)((v[it % h->wcap] >> lane) & 1ull);
        return 0;
    }
    return fail(-42, "record_accept / record_chain was not requested");
}

int mcmcx_get_chain(mcmcx_handle h, int32_t chain, double *chain_out, double *ss_out, double *s2_out, int32_t *nrows)
{
    int rc = check_chain(h, chain); if (rc) return rc;
    if (!h->cfg.record_chain || !h->E.hist) return fail(-42, "record_chain was not requested");
    const int tile = chain / 64, lane = chain % 64, d = h->d, W = h->wcap;
    std::vector<uint64_t> m;
    if ((rc = fetch(h, h->E.wacc + (size_t)tile * W, (size_t)W, m))) return rc;
    // this chain's lane of the history ring, gathered on the device: hv[slot*(d+1) + k]
    auto gather = [&](const double *src, size_t n, std::vector<double> &dst) -> int {
        double *tmp = nullptr;
        HIPCHK(hipMalloc(&tmp, n * sizeof(double)));       // (freed on every path below)
        hipLaunchKernelGGL(gather_lane_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, h->stream, src, tmp,
            n, lane);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        dst.resize(n);
        if (e == hipSuccess) e = hipMemcpy(dst.data(), tmp, n * sizeof(double), hipMemcpyDeviceToHost);
        (void)hipFree(tmp);
        if (e != hipSuccess) return fail(-100, hipGetErrorString(e));
        return 0;
    };
    std::vector<double> hv;
    const int ny = h->ny, hs = d + ny;                  // history row: theta, ss per column
    if ((rc = gather(h->E.hist + (size_t)tile * W * hs * 64, (size_t)W * hs, hv))) return rc;
    int row = -1;
    for (int it = 1; it <= h->simuind; ++it) {
        const size_t so = (size_t)(it % W) * hs;
        if ((m[it % W] >> lane) & 1ull) {
            ++row;
            if (chain_out) {
                for (int k = 0; k < d; ++k) chain_out[(size_t)row * (d + 1) + k] = hv[so + k];
                chain_out[(size_t)row * (d + 1) + d] = 1.0;
            }
            if (ss_out) { for (int j = 0; j < ny; ++j) ss_out[(size_t)row * (ny + 1) + j] = hv[so + d + j];
                ss_out[(size_t)row * (ny + 1) + ny] = 1.0; }
        } else {
            if (chain_out) chain_out[(size_t)row * (d + 1) + d] += 1.0;
            if (ss_out) ss_out[(size_t)row * (ny + 1) + ny] += 1.0;
        }
    }
    if (nrows) *nrows = row + 1;
    if (s2_out && h->E.s2hist) {
        std::vector<double> sv;
        if ((rc = gather(h->E.s2hist + (size_t)tile * W * ny * 64, (size_t)W * ny, sv))) return rc;
        for (int it = 1; it <= h->simuind; ++it) for (int j = 0; j < ny;
            ++j) s2_out[(size_t)(it - 1) * ny + j] = sv[(size_t)(it % W) * ny + j];
    }
    return 0;
}

int32_t mcmcx_pooled_moments_len(mcmcx_handle h) { return h ? 1 + h->d + h->P : -1; }

static int pooled_vec_len(const mcmcx_engine *h, int kind) { return kind == 2 ? 2 + h->P : 1 + h->d + h->P + (kind == 1 ? 1 : 0); }

static int pooled_moments_launch(mcmcx_engine *h, double *dev_dst, int kind, int it)
{   // (declared above)
    if (!h || !h->inited) return fail(-40, "we have not inited");
    HIPCHK(hipSetDevice(h->cfg.device));
    const int len = pooled_vec_len(h, kind), T = h->ntiles;
    const double rs = (kind == 2) ? 1.0 / std::pow((double)(float)it, h->cfg.nuparam) : 0.0;      // like d_ramscale (MCMC_run_ram.F90:166)
    const size_t mlds = ((size_t)64 * (h->d | 1) + 320) * sizeof(double);
    if (mlds <= 160 * 1024) hipLaunchKernelGGL(moments_kernel<false>, dim3(T), dim3(256), mlds, h->stream, h->E, h->d_moments,
        h->cfg.nchains, kind, it, rs);
    // npar >= 316: (64 (d | 1) + 320) 8 bytes exceed 160 KiB (317: 164 864)
    else hipLaunchKernelGGL(moments_kernel<true>, dim3(T), dim3(256), (size_t)320 * sizeof(double), h->stream, h->E, h->d_moments,
        h->cfg.nchains, kind, it, rs);
    for (long long stride = 1;; stride *= 64) {                            // six levels of the fixed pairwise tree per launch
        const long long groups = (T + 64 * stride - 1) / (64 * stride);
        hipLaunchKernelGGL(moments_tree_kernel, dim3((len + 255) / 256, (unsigned)groups), dim3(256), 0, h->stream, h->d_moments, T, len,
                           (int)stride, dev_dst);
        if (groups == 1) break;
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int mcmcx_pooled_moments(mcmcx_handle h, double *out)
{
    int rc = pooled_moments_launch(h, nullptr, 0, 0); if (rc) return rc;
    std::vector<double> v;
    if ((rc = fetch(h, h->d_moments, (size_t)(1 + h->d + h->P), v))) return rc;
    memcpy(out, v.data(), sizeof(double) * v.size());
    return 0;
}

int mcmcx_set_exchange(mcmcx_handle h, mcmcx_exchange_t fn, void *user, void *dev_buf)
{
    if (!h) return fail(-1, "null handle");
    if (fn && !dev_buf) return fail(-1, "mcmcx_set_exchange: dev_buf is required");
    h->xfn = fn; h->xuser = user; h->xbuf = (double *)dev_buf;
    return 0;
}

int mcmcx_get_pooled(mcmcx_handle h, double *cmat, double *mean, double *wsum, double *R)
{
    if (!h || !h->inited) return fail(-40, "we have not inited");
    if (!h->pooled) return fail(-44, "not in pooled mode");
    if (cmat) unpack_upper(h->d, h->pool_C, cmat, true);
    if (mean) memcpy(mean, h->pool_mean.data(), sizeof(double) * (size_t)h->d);
    if (wsum) *wsum = h->pool_W;
    // scam: the rotation U; condmax > 0: the full SVD factor; column-major
    if (R && h->usesvd) { const auto &M = h->cfg.method == MCMCX_METHOD_SCAM ? h->pool_U : h->pool_Rf; memcpy(R, M.data(),
        sizeof(double) * M.size()); }
    else if (R) unpack_upper(h->d, h->pool_R, R, false);
    return 0;
}

/* same, result left in device memory at dev_out (e.g. the buffer an RCCL all-reduce works on);
 * asynchronous on the engine's stream: call mcmcx_sync before another stream reads it */
int mcmcx_pooled_moments_dev(mcmcx_handle h, void *dev_out)
{
    if (!dev_out) return fail(-1, "null argument");
    return pooled_moments_launch(h, (double *)dev_out, 0, 0);
}

// ------------------------------------------------------------------ the node: several GPUs, one communicator
int mcmcx_set_comm(mcmcx_handle h, mcmcx_comm_t c)
{
    if (!h) return fail(-1, "null handle");
    if (h->inited) return fail(-1, "mcmcx_set_comm after mcmcx_init");
    if (c && c->device != h->cfg.device) return fail(-1, "mcmcx_set_comm: the communicator lives on device " + std::to_string(c->device)
        + ", the engine on " + std::to_string(h->cfg.device));
    h->comm = c;
    return 0;
}

int mcmcx_allreduce_moments(mcmcx_handle h, double *host_out)
{
    if (!h || !h->inited) return fail(-40, "we have not inited");
    if (h->comm && h->comm->single_process && h->comm->nranks > 1)
        return fail(-1, "mcmcx_allreduce_moments: this communicator drives several GPUs from one process; use mcmcx_allreduce_moments_all");
    int rc = allreduce_moments_enqueue(h, 0); if (rc) return rc;
    if (host_out) {
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(host_out, h->d_pooled, (size_t)(1 + h->d + h->P) * 8, hipMemcpyDeviceToHost));
    }
    return 0;
}

int mcmcx_allreduce_moments_all(mcmcx_handle *hs, int32_t n, double *host_out)
{
    if (!hs || n < 1) return fail(-1, "bad argument");
    for (int i = 0; i < n; ++i) if (!hs[i] || !hs[i]->inited) return fail(-40, "we have not inited");
    int rc;
    for (int i = 0; i < n; ++i) if ((rc = allreduce_moments_enqueue(hs[i], 1))) return rc;
    NCCLCHK(ncclGroupStart());                                  // one thread, several devices: the gathers go out as a group
    for (int i = 0; i < n; ++i) if ((rc = allreduce_moments_enqueue(hs[i], 2))) { (void)ncclGroupEnd(); return rc; }
    NCCLCHK(ncclGroupEnd());
    for (int i = 0; i < n; ++i) if ((rc = allreduce_moments_enqueue(hs[i], 3))) return rc;
    if (host_out) {
        HIPCHK(hipSetDevice(hs[0]->cfg.device));
        HIPCHK(hipStreamSynchronize(hs[0]->stream));
        HIPCHK(hipMemcpy(host_out, hs[0]->d_pooled, (size_t)(1 + hs[0]->d + hs[0]->P) * 8, hipMemcpyDeviceToHost));
    }
    return 0;
}

// mcmcx_run on n engines at once, one host thread per engine (each GPU has its own stream; in pooled mode the threads
// meet in the all-gather of every adaptation tick).  User host callbacks are not thread-safe: engines with
// a host target run one after the other on the calling thread instead.
int mcmcx_run_all(mcmcx_handle *hs, int32_t n, int32_t upto)
{
    if (!hs || n < 1) return fail(-1, "bad argument");
    bool serial = (n == 1);
    for (int i = 0; i < n; ++i) { if (!hs[i]) return fail(-1, "null handle"); if (phased(hs[i]) && hs[i]->tkind == TGT_HOST) serial = true;
        }
    if (serial) {
        int worst = 0;
        for (int i = 0; i < n; ++i) { int rc = mcmcx_run(hs[i], upto); if (rc < 0) return rc; worst = std::max(worst, rc); }
        return worst;
    }
    std::vector<int> rcs(n, 0); std::vector<std::string> errs(n);
    std::vector<std::thread> th;
    for (int i = 1; i < n; ++i) th.emplace_back([&, i]() { rcs[i] = mcmcx_run(hs[i], upto); if (rcs[i] < 0) errs[i] = g_err; });
    rcs[0] = mcmcx_run(hs[0], upto); if (rcs[0] < 0) errs[0] = g_err;
    for (auto &t : th) t.join();
    int worst = 0;
    for (int i = 0; i < n; ++i) { if (rcs[i] < 0) return fail(rcs[i], "engine " + std::to_string(i) + ": " + errs[i]);
        worst = std::max(worst, rcs[i]); }
    return worst;
}

} // extern "C"

extern "C" {
// ------------------------------------------------------------------ debug probes (tests only)
// The kernel-selection tables (launch_step / launch_group / launch_scam): entry `index` as "family:name" into buf; returns the number of
// entries (so index = -1 with buf = NULL just counts).  Needs no device.
int mcmcx_debug_kernel_table(int32_t index, char *buf, int32_t len)
{
    const KernelEntry *tabs[] = {STEP_TABLE, GROUP_TABLE, SCAM_TABLE};
    const size_t ns[] = {sizeof(STEP_TABLE) / sizeof(STEP_TABLE[0]), sizeof(GROUP_TABLE) / sizeof(GROUP_TABLE[0]),
        sizeof(SCAM_TABLE) / sizeof(SCAM_TABLE[0])};
    int total = 0, k = index;
    for (int t = 0; t < 3; ++t) {
        if (k >= 0 && k < (int)ns[t] && buf && len > 0) { snprintf(buf, (size_t)len, "%s:%s", tabs[t][k].family, tabs[t][k].name);
            k = -1 - total - (int)ns[t]; }
        else if (k >= 0) k -= (int)ns[t];
        total += (int)ns[t];
    }
    return total;
}

int mcmcx_debug_math(int32_t op, int32_t n, const double *a, const double *b, double *out)
{
    if (n < 1 || !a || !out) return fail(-1, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(-10, "no HIP device: the mcmcx engine has no CPU fallback");
    DevBufs g;
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    HIPCHK(g.alloc(&da, (size_t)n * 8)); HIPCHK(g.alloc(&dout, (size_t)n * 8));
    HIPCHK(hipMemcpy(da, a, (size_t)n * 8, hipMemcpyHostToDevice));
    if (b) { HIPCHK(g.alloc(&db, (size_t)n * 8)); HIPCHK(hipMemcpy(db, b, (size_t)n * 8, hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(debug_math_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, op, n, da, db, dout);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dout, (size_t)n * 8, hipMemcpyDeviceToHost));
    return 0;
}

// Replace the SVD proposal factor of EVERY chain (per-chain mode, condmax > 0): R column-major d x d (scam: the rotation U),
// qcovstd (scam; may be NULL).  Lets a test feed the factors an external LAPACK returned at an adaptation
// (tests/test_gpu_parity.py: the MKL-linked reference's logged dgesvd results at d = 200).
int mcmcx_debug_set_factor(mcmcx_handle h, const double *R_colmajor, const double *qcovstd)
{
    if (!h || !h->inited || !R_colmajor) return fail(-1, "mcmcx_debug_set_factor: bad argument");
    if (!h->usesvd || h->pooled
        || !h->E.Rf) return fail(-45, "mcmcx_debug_set_factor: per-chain SVD factor only (condmax > 0, not pooled)");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<double> r(R_colmajor, R_colmajor + (size_t)h->d * h->d);
    int rc = dev_bcast(h, h->E.Rf, r); if (rc) return rc;
    if (qcovstd && h->E.qstd) { std::vector<double> q(qcovstd, qcovstd + h->d); if ((rc = dev_bcast(h, h->E.qstd, q))) return rc; }
    return 0;
}

int mcmcx_debug_rng(uint32_t seed, uint32_t chain_id, int32_t kind, int32_t n, double a, double b, double *out, uint64_t *nused)
{
    if (n < 1 || !out) return fail(-1, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(-10, "no HIP device: the mcmcx engine has no CPU fallback");
    DevBufs g;
    double *dout = nullptr; uint64_t *dn = nullptr;
    HIPCHK(g.alloc(&dout, (size_t)n * 8)); HIPCHK(g.alloc(&dn, 8));
    hipLaunchKernelGGL(debug_rng_kernel, dim3(1), dim3(64), 0, 0, seed, chain_id, kind, n, a, b, dout, dn);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dout, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (nused) HIPCHK(hipMemcpy(nused, dn, 8, hipMemcpyDeviceToHost));
    return 0;
}

} // extern "C"
