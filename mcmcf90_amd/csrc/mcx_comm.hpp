// mcx_comm.hpp -- the one exchange step of the path across the GPUs of a node: the pooled
// empirical-moment vector [count, sum, upper second moments] of all chains.  Included by mcx_api.hip.
//
// Chains never talk to each other; every `adaptint` iterations (pooled mode) or whenever the host asks for the pooled
// posterior moments, each GPU reduces its own chains in the fixed pairwise tree of moments_kernel /
// moments_tree_kernel, and the per-GPU vectors (1 + d + d(d+1)/2 doubles: 10.6 kB at d = 50) are combined over RCCL.
// The combination is an ALL-GATHER of the per-rank vectors followed by the same pairwise tree over ranks on every GPU
// (moments_tree_kernel with "tiles" = ranks) instead of ncclAllReduce(ncclSum): a ring's summation order depends on
// the rank count, the tree's does not, so the pooled covariance -- and with it every proposal of a pooled-mode run --
// is bit-identical on 1, 2, 4 and 8 GPUs (power-of-two aligned shards).  The message is latency-sized either way.
// ncclAllReduce carries the host-side scalars (timing maxima, counters).
//
// Two transports behind one interface:
//   MCMCX_COMM_RCCL  one process per GPU (ncclCommInitRank; the ncclUniqueId travels through a POSIX shared-memory
//                    segment named by `key`: single node, no MPI / torch needed), or one process driving all GPUs
//                    (ncclCommInitAll, mcmcx_comm_create_all: what the Fortran shim's `ngpus` uses)
//   MCMCX_COMM_HOST  the gather staged through that shared-memory segment by the host: for ranks that share ONE
//                    GPU (RCCL refuses duplicate devices), i.e. checking the N > 1 control path on a one-GPU box
#pragma once
#include <rccl/rccl.h>
#include <atomic>
#include <memory>
#include <chrono>
#include <thread>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#define NCCLCHK(call)                                                                             \
    do {                                                                                          \
        ncclResult_t r_ = (call);                                                                 \
        if (r_ != ncclSuccess)                                                                    \
            return fail(-110, std::string(#call) + ": " + ncclGetErrorString(r_));                \
    } while (0)

// doubles per rank in the host segment (1 MiB): >= 3 + d + d(d+1)/2 up to d = 510; longer messages fail loudly (RCCL has no such limit)
static const int MCX_SHM_SLOT = 131072;
static const int MCX_COMM_MAXRANKS = 64;        // one level of moments_tree_kernel

struct mcx_shm_header {
    std::atomic<uint32_t> magic;                // set last by rank 0
    std::atomic<uint32_t> arrived, sense;       // sense-reversing barrier
    std::atomic<uint32_t> failed;               // a rank gave up: everybody else fails fast
    int32_t nranks;
    char uid[NCCL_UNIQUE_ID_BYTES];
};
static const uint32_t MCX_SHM_MAGIC = 0x6D637843u;

struct mcmcx_comm {
    int rank = 0, nranks = 1, device = 0, backend = MCMCX_COMM_RCCL;
    bool single_process = false;                // created by mcmcx_comm_create_all (ncclCommInitAll)
    ncclComm_t nccl = nullptr;
    hipStream_t stream = nullptr;               // for the host-scalar collectives
    double *d_scratch = nullptr;                // 512 doubles
    // shared-memory segment (bootstrap of the unique id; data path of the HOST transport)
    std::string shm_name; int shm_fd = -1; size_t shm_bytes = 0; mcx_shm_header *hdr = nullptr; double *slots = nullptr;
    uint32_t local_sense = 0;
    std::vector<double> hbuf;
    // the communicators of ONE mcmcx_comm_create_all call share this word (one process, a thread per engine): a rank that fails in a run
    // raises it, the waits of its siblings poll it.  Communicators of one process per rank use the segment's `failed` word.  Either way
    // the mark belongs to the communicator it concerns: other communicators of the process never see it.
    std::shared_ptr<std::atomic<int>> group_failed;
};

// An engine of this process failed inside a run whose ranks meet in collectives (mcmcx_run): the waits of its peers --
// other threads of a one-process node (mcmcx_run_all), other processes through the segment's `failed` word -- poll this
// and abort their side of the communicator instead of waiting for a gather that will never complete.
static void comm_mark_failed(mcmcx_comm *c)
{
    if (c && c->group_failed) c->group_failed->store(1);
    if (c && c->hdr) c->hdr->failed.store(1);
}
// wait for `stream` (which may hold a collective) without outliving a failed peer
static int comm_wait_stream(mcmcx_comm *c, hipStream_t stream)
{
    if (!c || c->nranks == 1) { HIPCHK(hipStreamSynchronize(stream)); return 0; }
    for (int spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(stream);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) return fail(-100, std::string("hipStreamQuery: ") + hipGetErrorString(e));
        if ((c->group_failed && c->group_failed->load(std::memory_order_relaxed)) || (c->hdr
            && c->hdr->failed.load(std::memory_order_relaxed))) {
            if (c->nccl) { (void)ncclCommAbort(c->nccl); c->nccl = nullptr; }     // frees this rank's pending collective
            return fail(-111, "mcmcx_comm: another rank failed; this rank's collective was aborted (rank " + std::to_string(c->rank) + ")");
        }
        if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        else std::this_thread::yield();
    }
}

static int shm_barrier(mcmcx_comm *c, double timeout_s = 300.0)
{
    mcx_shm_header *h = c->hdr;
    const uint32_t s = (c->local_sense ^= 1u);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->nranks) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->sense.store(s, std::memory_order_release);
        return 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (h->sense.load(std::memory_order_acquire) != s) {
        if (h->failed.load(std::memory_order_relaxed)) return fail(-111, "mcmcx_comm: another rank failed");
        if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        else std::this_thread::yield();
        if ((spins & 1023) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
            h->failed.store(1);
            return fail(-111, "mcmcx_comm: timed out waiting for the other ranks (rank " + std::to_string(c->rank) + ")");
        }
    }
    return 0;
}

static int shm_attach(mcmcx_comm *c, const char *key)
{
    c->shm_name = std::string("/mcmcx_") + key;
    for (auto &ch : c->shm_name) if (ch == '/' && &ch != &c->shm_name[0]) ch = '_';
    c->shm_bytes = sizeof(mcx_shm_header) + (size_t)c->nranks * MCX_SHM_SLOT * sizeof(double);
    const auto t0 = std::chrono::steady_clock::now();
    if (c->rank == 0) {
        shm_unlink(c->shm_name.c_str());                                    // a stale segment of a crashed run
        c->shm_fd = shm_open(c->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (c->shm_fd < 0) return fail(-112, "mcmcx_comm: shm_open(" + c->shm_name + ") failed");
        if (ftruncate(c->shm_fd, (off_t)c->shm_bytes) != 0) return fail(-112, "mcmcx_comm: ftruncate failed");
    } else {
        // A segment of the same name left by an earlier run (same key) may still be there until this run's rank 0 unlinks it.
        // Rank 0 clears `magic` once a communicator has formed (mcmcx_comm_create), so a left-over segment of a run that got
        // that far never looks ready; and a rank that sits on a segment rank 0 has meanwhile unlinked (st_nlink == 0) lets go
        // of it and opens the name again.  What remains is a run killed DURING its formation whose key recurs: callers make
        // keys unique per run (bench.py: a uuid, or the launcher's pid + start time).
        for (;;) {
            for (;;) {                                                      // wait for rank 0 to create and size it
                c->shm_fd = shm_open(c->shm_name.c_str(), O_RDWR, 0600);
                if (c->shm_fd >= 0) {
                    struct stat st;
                    if (fstat(c->shm_fd, &st) == 0 && (size_t)st.st_size >= c->shm_bytes && st.st_nlink > 0) break;
                    close(c->shm_fd); c->shm_fd = -1;
                }
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 300.0)
                    return fail(-112, "mcmcx_comm: rank 0 never created " + c->shm_name);
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
            void *q = mmap(nullptr, c->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->shm_fd, 0);
            if (q == MAP_FAILED) return fail(-112, "mcmcx_comm: mmap failed");
            mcx_shm_header *hq = (mcx_shm_header *)q;
            bool stale = false;
            while (hq->magic.load(std::memory_order_acquire) != MCX_SHM_MAGIC) {
                struct stat st;
                if (fstat(c->shm_fd, &st) != 0 || st.st_nlink == 0) { stale = true; break; }   // unlinked under us: not this run's segment
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 300.0)
                    return fail(-112, "mcmcx_comm: the segment was never initialised");
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            if (!stale) {
                c->hdr = hq;
                c->slots = (double *)((char *)q + sizeof(mcx_shm_header));
                if (c->hdr->nranks != c->nranks) return fail(-112, "mcmcx_comm: ranks disagree about the world size");
                return 0;
            }
            munmap(q, c->shm_bytes); close(c->shm_fd); c->shm_fd = -1;
        }
    }
    void *p = mmap(nullptr, c->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->shm_fd, 0);
    if (p == MAP_FAILED) return fail(-112, "mcmcx_comm: mmap failed");
    c->hdr = (mcx_shm_header *)p;
    c->slots = (double *)((char *)p + sizeof(mcx_shm_header));
    c->hdr->arrived.store(0); c->hdr->sense.store(0); c->hdr->failed.store(0); c->hdr->nranks = c->nranks;
    c->hdr->magic.store(MCX_SHM_MAGIC, std::memory_order_release);
    return 0;
}

static void comm_free(mcmcx_comm *c)
{
    if (!c) return;
    if (c->device >= 0) (void)hipSetDevice(c->device);
    if (c->nccl) (void)ncclCommDestroy(c->nccl);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->hdr) munmap((void *)c->hdr, c->shm_bytes);
    if (c->shm_fd >= 0) close(c->shm_fd);
    if (c->rank == 0 && !c->shm_name.empty()) shm_unlink(c->shm_name.c_str());
    delete c;
}

// gather `len` doubles per rank: dev_all[r*len ..] of every rank <- rank r's dev_all[rank*len ..]; on `stream`
static int comm_allgather(mcmcx_comm *c, double *dev_all, int len, hipStream_t stream)
{
    if (c->nranks == 1) return 0;
    if (c->backend == MCMCX_COMM_RCCL) {
        NCCLCHK(ncclAllGather(dev_all + (size_t)c->rank * len, dev_all, (size_t)len, ncclDouble, c->nccl, stream));
        return 0;
    }
    if (len > MCX_SHM_SLOT) return fail(-113, "mcmcx_comm: message too long for the host segment");
    HIPCHK(hipMemcpyAsync(c->slots + (size_t)c->rank * MCX_SHM_SLOT, dev_all + (size_t)c->rank * len, (size_t)len * 8,
        hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    int rc = shm_barrier(c); if (rc) return rc;
    c->hbuf.resize((size_t)c->nranks * len);
    for (int r = 0; r < c->nranks; ++r) memcpy(&c->hbuf[(size_t)r * len], c->slots + (size_t)r * MCX_SHM_SLOT, (size_t)len * 8);
    rc = shm_barrier(c); if (rc) return rc;                                 // every rank has read: the slots may be reused
    HIPCHK(hipMemcpyAsync(dev_all, c->hbuf.data(), c->hbuf.size() * 8, hipMemcpyHostToDevice, stream));
    HIPCHK(hipStreamSynchronize(stream));                                   // hbuf is reused by the next call
    return 0;
}

extern "C" {

int mcmcx_comm_create(const char *key, int32_t rank, int32_t nranks, int32_t device, int32_t backend, mcmcx_comm_t *out)
{
    if (!key || !out) return fail(-1, "mcmcx_comm_create: null argument");
    if (nranks < 1 || nranks > MCX_COMM_MAXRANKS || rank < 0 || rank >= nranks) return fail(-1, "mcmcx_comm_create: bad rank / nranks");
    if (backend != MCMCX_COMM_RCCL && backend != MCMCX_COMM_HOST) return fail(-1, "mcmcx_comm_create: unknown backend");
    // device < 0 with the HOST transport: no GPU is touched -- barrier and host scalars only (the bootstrap and the
    // rank bookkeeping can then be exercised on a box without a GPU); an engine cannot attach such a communicator
    const bool nogpu = (backend == MCMCX_COMM_HOST && device < 0);
    if (!nogpu) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(-10, "no HIP device: the mcmcx engine has no CPU fallback");
        if (device < 0 || device >= ndev) return fail(-10, "mcmcx_comm_create: bad device ordinal " + std::to_string(device) + " (" +
            std::to_string(ndev) + " visible)");
        HIPCHK(hipSetDevice(device));
    }
    mcmcx_comm *c = new mcmcx_comm();
    c->rank = rank; c->nranks = nranks; c->device = nogpu ? -1 : device; c->backend = backend;
    int rc = shm_attach(c, key);
    if (rc) { comm_free(c); return rc; }
    if (!nogpu) {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void **)&c->d_scratch, 512 * sizeof(double));
        if (e != hipSuccess) { c->hdr->failed.store(1); comm_free(c); return fail(-100, hipGetErrorString(e)); }
    }
    if (backend == MCMCX_COMM_RCCL) {
        ncclUniqueId id;
        if (rank == 0) {
            ncclResult_t r = ncclGetUniqueId(&id);
            if (r != ncclSuccess) { c->hdr->failed.store(1); comm_free(c); return fail(-110,
                std::string("ncclGetUniqueId: ") + ncclGetErrorString(r)); }
            memcpy(c->hdr->uid, id.internal, NCCL_UNIQUE_ID_BYTES);
        }
        if ((rc = shm_barrier(c))) { comm_free(c); return rc; }
        memcpy(id.internal, c->hdr->uid, NCCL_UNIQUE_ID_BYTES);
        ncclResult_t r = ncclCommInitRank(&c->nccl, nranks, id, rank);
        if (r != ncclSuccess) { c->hdr->failed.store(1); c->nccl = nullptr; comm_free(c); return fail(-110,
            std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
    }
    if ((rc = shm_barrier(c))) { comm_free(c); return rc; }
    // formed: nobody attaches to this segment any more (see shm_attach)
    if (rank == 0) c->hdr->magic.store(0, std::memory_order_release);
    *out = c;
    return 0;
}

int mcmcx_comm_create_all(int32_t ndev_want, const int32_t *devices, mcmcx_comm_t *out)
{
    if (!out || ndev_want < 1 || ndev_want > MCX_COMM_MAXRANKS) return fail(-1, "mcmcx_comm_create_all: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(-10, "no HIP device: the mcmcx engine has no CPU fallback");
    std::vector<int> dl(ndev_want);
    for (int i = 0; i < ndev_want; ++i) {
        dl[i] = devices ? devices[i] : i;
        if (dl[i] < 0 || dl[i] >= ndev) return fail(-10, "ngpus = " + std::to_string(ndev_want) + " but only " + std::to_string(ndev)
            + " HIP device(s) are visible");
        for (int k = 0; k < i; ++k) if (dl[k] == dl[i]) return fail(-10, "mcmcx_comm_create_all: duplicate device");
    }
    std::vector<ncclComm_t> comms(ndev_want);
    auto group_failed = std::make_shared<std::atomic<int>>(0);
    for (int i = 0; i < ndev_want; ++i) out[i] = nullptr;
    NCCLCHK(ncclCommInitAll(comms.data(), ndev_want, dl.data()));
    for (int i = 0; i < ndev_want; ++i) {
        mcmcx_comm *c = new mcmcx_comm();
        c->rank = i; c->nranks = ndev_want; c->device = dl[i]; c->backend = MCMCX_COMM_RCCL; c->single_process = true; c->nccl = comms[i];
            c->group_failed = group_failed;
        hipError_t e = hipSetDevice(dl[i]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void **)&c->d_scratch, 512 * sizeof(double));
        // nothing half-made is handed back: every rank so far, this one,
        if (e != hipSuccess) {
            const std::string msg = hipGetErrorString(e);                   // and the raw communicators of the ranks not reached yet
            comm_free(c);
            for (int k = 0; k < i; ++k) { comm_free(out[k]); out[k] = nullptr; }
            for (int k = i + 1; k < ndev_want; ++k) (void)ncclCommDestroy(comms[k]);
            return fail(-100, "mcmcx_comm_create_all: device " + std::to_string(dl[i]) + ": " + msg);
        }
        out[i] = c;
    }
    return 0;
}

int mcmcx_comm_destroy(mcmcx_comm_t c) { comm_free(c); return 0; }
int32_t mcmcx_comm_rank(mcmcx_comm_t c) { return c ? c->rank : -1; }
int32_t mcmcx_comm_size(mcmcx_comm_t c) { return c ? c->nranks : -1; }

// sum (op 0) or maximum (op 1) of n <= 512 host doubles over the ranks, in place (one process per rank)
int mcmcx_comm_allreduce_host(mcmcx_comm_t c, double *v, int32_t n, int32_t op)
{
    if (!c || !v || n < 1 || n > 512) return fail(-1, "mcmcx_comm_allreduce_host: bad argument");
    if (c->single_process) return
        fail(-1, "mcmcx_comm_allreduce_host: one process per rank only (reduce over the engines on the host instead)");
    if (c->nranks == 1) return 0;
    if (c->backend == MCMCX_COMM_RCCL) {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipMemcpyAsync(c->d_scratch, v, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
        NCCLCHK(ncclAllReduce(c->d_scratch, c->d_scratch, (size_t)n, ncclDouble, op == 1 ? ncclMax : ncclSum, c->nccl, c->stream));
        HIPCHK(hipMemcpyAsync(v, c->d_scratch, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return 0;
    }
    memcpy(c->slots + (size_t)c->rank * MCX_SHM_SLOT, v, (size_t)n * 8);
    int rc = shm_barrier(c); if (rc) return rc;
    std::vector<double> acc(n);
    for (int k = 0; k < n; ++k) {
        double a = c->slots[k];
        for (int r = 1; r < c->nranks; ++r) { const double b = c->slots[(size_t)r * MCX_SHM_SLOT + k];
            a = (op == 1) ? (b > a ? b : a) : a + b; }
        acc[k] = a;
    }
    rc = shm_barrier(c); if (rc) return rc;
    memcpy(v, acc.data(), (size_t)n * 8);
    return 0;
}

int mcmcx_comm_barrier(mcmcx_comm_t c)
{
    if (!c) return fail(-1, "null communicator");
    if (c->single_process || c->nranks == 1) return 0;
    // through the GPUs, like the data
    if (c->backend == MCMCX_COMM_RCCL) { double one = 1.0; return mcmcx_comm_allreduce_host(c, &one, 1, 0); }
    return shm_barrier(c);
}

} // extern "C"
