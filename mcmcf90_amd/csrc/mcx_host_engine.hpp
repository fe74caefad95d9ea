// mcx_host_engine.hpp -- the engine object behind a mcmcx_handle: error reporting, the test switches, the host and device state of N
// chains, allocation helpers.
// Part of the ONE translation unit mcx_api.hip (included there, in this order: mcx_host_engine, mcx_host_linalg, mcx_host_launch,
// mcx_host_adapt, mcx_host_pooled, mcx_host_callbacks); not a stand-alone header.

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
        svd_lane = -1, cov_batch_rows = -1, ram_wide = -1, pooled_waves = -1, pooled_ks = -1, cols_phased = -1, host_mapped = -1,
            host_fuse = -1;
    static int get(const char *name) { const char *e = getenv(name); return e ? atoi(e) : -1; }
    void read()
    {
        pooled_mfma_dr_min = get("MCMCX_POOLED_MFMA_DR_MIN"); pooled_scalar = get("MCMCX_POOLED_SCALAR"); dr_big = get("MCMCX_DR_BIG");
        scam_pooled_16 = get("MCMCX_SCAM_POOLED_16"); scam_fast_lanes = get("MCMCX_SCAM_FAST_LANES");
        scam_waves = get("MCMCX_SCAM_WAVES"); svd_lane = get("MCMCX_SVD_LANE"); cov_batch_rows = get("MCMCX_COV_BATCH_ROWS");
        ram_wide = get("MCMCX_RAM_WIDE"); pooled_waves = get("MCMCX_POOLED_WAVES"); pooled_ks = get("MCMCX_POOLED_KS");
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
