// Row-sharded Jacobi-PCG with the communication INSIDE the library: the iteration loop, the halo
// exchange of the search direction and the one all-reduce per iteration are issued from C++ on the
// context's stream - no Python between two iterations.
//
// Two bindings provide the two communication steps:
//   * RCCL (production: one process per GPU over xGMI).  librccl is resolved at run time from the copy the
//     process already holds (torch's), so that one RCCL runtime serves torch.distributed and this library.
//     ncclSend/ncclRecv of the boundary planes to the z-neighbours (rank-1 / rank+1) in one group,
//     ncclAllReduce in place on the device slot bank; everything on the context's stream.
//   * callbacks (tests: several ranks sharing one GPU over gloo, host-staged): the same C++ loop calls back
//     into the host for the two steps.
//
// The recurrence is the single-reduction (Chronopoulos-Gear) form of pgdrome_amd/dist.py, built from the
// same slot kernels (pgd_cg_*_slot, pgd_spmv_dot_slot), so a sharded solve walks the same Krylov iterates
// as the single-GPU one.
#include "pgd_internal.h"

#include <dlfcn.h>
#include <unistd.h>
#include <cstring>

#include <algorithm>
#include <chrono>
#include <thread>

namespace pgd {

struct NcclId { char internal[128]; };

struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(NcclId *) = nullptr;
    int (*CommInitRank)(void **, int, NcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommSplit)(void *, int, int, void **, void *) = nullptr;      // optional (RCCL >= 2.18)
};
constexpr int NCCL_F64 = 8, NCCL_SUM = 0;   // ncclFloat64, ncclSum (rccl.h)

static RcclApi *rccl_api(std::string &why) {
    static RcclApi api;
    static bool tried = false;
    static std::string err;
    if (!tried) {
        tried = true;
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (int pass = 0; pass < 2 && !api.lib; ++pass)          // first the copy already in the process
            for (const char *nm : names)
                if (!api.lib) api.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
        if (!api.lib) {
            err = "librccl.so not found";
        } else {
            auto sym = [&](const char *s) { void *p = dlsym(api.lib, s); if (!p && err.empty()) err = std::string("missing symbol ") + s; return p; };
            api.GetUniqueId = (int (*)(NcclId *))sym("ncclGetUniqueId");
            api.CommInitRank = (int (*)(void **, int, NcclId, int))sym("ncclCommInitRank");
            api.CommDestroy = (int (*)(void *))sym("ncclCommDestroy");
            api.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))sym("ncclAllReduce");
            api.Send = (int (*)(const void *, size_t, int, int, void *, hipStream_t))sym("ncclSend");
            api.Recv = (int (*)(void *, size_t, int, int, void *, hipStream_t))sym("ncclRecv");
            api.GroupStart = (int (*)())sym("ncclGroupStart");
            api.GroupEnd = (int (*)())sym("ncclGroupEnd");
            api.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
            api.CommSplit = (int (*)(void *, int, int, void **, void *))dlsym(api.lib, "ncclCommSplit");
        }
    }
    why = err;
    return err.empty() ? &api : nullptr;
}

#define PGD_NCCL(c, api, call)                                                                       \
    do {                                                                                             \
        const int rc__ = (call);                                                                     \
        if (rc__ != 0) return fail((c), PGD_ERR_HIP, "%s: %s", #call, (api)->GetErrorString(rc__)); \
    } while (0)

// the direct halo's mappings and flags (the work vectors stay: they belong to the binding)
static void push_drop(Ctx *c) {
    Comm &k = c->comm;
    for (void *&m : k.push_mapped) if (m) { (void)hipIpcCloseMemHandle(m); m = nullptr; }
    if (k.push_flags) { (void)hipFree(k.push_flags); k.push_flags = nullptr; }
    for (void *&m : k.ar_mapped) if (m) { (void)hipIpcCloseMemHandle(m); m = nullptr; }
    for (auto &q : k.ar_peer) q = nullptr;
    k.ar = k.ar_used = false;
    k.push = k.push_used = false;
    k.push_peer[0] = k.push_peer[1] = nullptr;
    k.push_peer_flags[0] = k.push_peer_flags[1] = nullptr;
    k.push_n = 0;
    (void)hipGetLastError();
}

void comm_release(Ctx *c) {
    Comm &k = c->comm;
    push_drop(c);
    if (k.kind == 2 && k.nccl) {
        std::string why;
        if (RcclApi *api = rccl_api(why)) {
            if (k.nccl_halo) (void)api->CommDestroy(k.nccl_halo);
            (void)api->CommDestroy(k.nccl);
        }
    }
    if (k.ev_ready) (void)hipEventDestroy(k.ev_ready);
    if (k.ev_halo) (void)hipEventDestroy(k.ev_halo);
    if (k.halo_stream) (void)hipStreamDestroy(k.halo_stream);
    for (hipEvent_t e : k.snap_ev) if (e) (void)hipEventDestroy(e);
    for (auto &set : k.mark) for (hipEvent_t e : set) if (e) (void)hipEventDestroy(e);
    if (k.snap_flags) (void)hipHostFree(k.snap_flags);
    for (pgd_handle &wh : k.work) if (wh) { (void)free_obj(c, wh, Obj::VEC); wh = 0; }
    // what pgd_tune / pgd_comm_timeout have set on this CONTEXT outlives a binding (PGD_TUNE=45=... given at context
    // creation was wiped here, before the first bind); the environment defaults of the next bind still apply on top
    const int64_t keep_rows = k.overlap_min_rows;
    const double keep_timeout = k.timeout_s;
    k = Comm();
    k.overlap_min_rows = keep_rows;
    k.timeout_s = keep_timeout;
}

static void comm_env_defaults(Comm &k) {
    if (const char *env = getenv("PGD_COMM_TIMEOUT_S")) {
        char *end = nullptr;
        const double v = strtod(env, &end);
        if (end != env) k.timeout_s = v;
    }
    // PGD_HALO_OVERLAP=1 asks for the overlapped exchange in EVERY sharded solve (it used to set up the second
    // communicator and then never use it); PGD_HALO_OVERLAP_MIN_ROWS names the rows per rank from which a solve takes it
    if (const char *env = getenv("PGD_HALO_OVERLAP")) {
        if (env[0] == '1') k.overlap_min_rows = 0;
    }
    if (const char *env = getenv("PGD_HALO_OVERLAP_MIN_ROWS")) {
        char *end = nullptr;
        const long long v = strtoll(env, &end, 10);
        if (end != env && v >= 0) k.overlap_min_rows = v;
    }
}

// ---- direct halo (opt-in; pgd_comm_push_export / pgd_comm_push_attach)
// k_halo_push: the two boundary planes of v go straight into the neighbours' ghost planes (dst_*: mapped addresses) as write-through
// stores at system scope - nothing of them stays behind in this device's L2, so no cache write-back stands between the data and the
// number posted behind it (a first version fenced every thread at system scope: 14 us for 1 MiB, as long as the RCCL kernel it
// replaced).  Every workgroup waits for its stores to be acknowledged and takes a ticket; the LAST one posts the sequence number into
// the neighbours' flag words (release, system scope) and then polls the flag words of THIS rank's ghost planes (acquire, system
// scope) until they carry the number too: when the kernel ends, the planes of both neighbours have arrived, and the kernel boundary
// orders the product's loads behind that.  Every rank posts before it waits, so nobody waits for somebody who waits.  A number that
// does not come within the deadline flags the solve done with PGD_ERR_TIMEOUT, like a breakdown, and the host reports it.
__device__ __forceinline__ void push_plane(const double *__restrict__ src, double *__restrict__ dst, int64_t n, int64_t t, int64_t step) {
    for (int64_t i = t; i < n; i += step)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst) + i, (unsigned long long)__double_as_longlong(src[i]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void k_halo_push(const double *__restrict__ src_lo, double *__restrict__ dst_lo, int64_t n_lo,
                                                   const double *__restrict__ src_hi, double *__restrict__ dst_hi, int64_t n_hi,
                                                   unsigned long long *post_lo, unsigned long long *post_hi,
                                                   const unsigned long long *wait_a, const unsigned long long *wait_b,
                                                   unsigned long long seq, unsigned long long *ticket, long long ticks, int *__restrict__ flags) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
    push_plane(src_lo, dst_lo, n_lo, t, step);
    push_plane(src_hi, dst_hi, n_hi, t, step);
    __builtin_amdgcn_s_waitcnt(0);                       // this wave's stores are acknowledged
    __syncthreads();                                     // ... and those of the whole workgroup
    if (threadIdx.x != 0) return;
    const unsigned long long done = __hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done + 1 != gridDim.x) return;
    __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (post_lo) __hip_atomic_store(post_lo, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (post_hi) __hip_atomic_store(post_hi, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long t0 = wall_clock64();
    for (int which = 0; which < 2; ++which) {
        const unsigned long long *f = which ? wait_b : wait_a;
        if (!f) continue;
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            if (wall_clock64() - t0 > ticks) {
                if (flags) { flags[2] = PGD_ERR_TIMEOUT; flags[0] = 1; }
                return;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
}

static long long push_ticks(const Comm &k) {
    const double s = k.timeout_s > 0.0 ? std::min(k.timeout_s, 3600.0) : 3600.0;
    return (long long)(s * 1e8);                                       // wall_clock64: 100 MHz
}

// the exchange itself: v must be the vector the export was made for (the sharded loop's p).  Always issued by every rank of a solve
// that voted for it - a rank that failed locally pushes whatever its p holds, so that nobody waits for a number that never comes.
// where this rank's boundary planes go.  To the LOWER neighbour: my bottom plane -> its ghost planes above (its last rows); to the UPPER
// one: my top plane -> its first rows.  (A rank that is its own neighbour: the ghost plane above is fed by the bottom plane, the one
// below by the top plane.)
struct PushTargets { int64_t n_lo, n_hi; double *dst_lo, *dst_hi; unsigned long long *post_lo, *post_hi, *wait_a, *wait_b; };
static PushTargets push_targets(const Comm &k) {
    const bool self = k.self_periodic && k.world == 1;
    PushTargets T;
    T.n_lo = self ? k.push_hi_g : k.push_lo_g;
    T.n_hi = self ? k.push_lo_g : k.push_hi_g;
    T.dst_lo = T.n_lo ? k.push_peer[0] + (k.push_peer_n[0] - T.n_lo) : nullptr;
    T.dst_hi = T.n_hi ? k.push_peer[1] : nullptr;
    T.post_lo = T.n_lo ? k.push_peer_flags[0] + 1 : nullptr;
    T.post_hi = T.n_hi ? k.push_peer_flags[1] + 0 : nullptr;
    T.wait_a = k.push_lo_g ? k.push_flags + 0 : nullptr;
    T.wait_b = k.push_hi_g ? k.push_flags + 1 : nullptr;
    return T;
}

static int comm_push_halo(Ctx *c, double *v, int *flags) {
    Comm &k = c->comm;
    if (!k.push) return fail(c, PGD_ERR_INVALID, "direct halo: not attached");
    const int64_t lo_g = k.push_lo_g, hi_g = k.push_hi_g, own0 = k.push_own0, own1 = k.push_own1;
    if (!lo_g && !hi_g) return PGD_OK;
    k.push_seq += 1;
    const PushTargets T = push_targets(k);
    const int64_t n_lo = T.n_lo, n_hi = T.n_hi;
    const double *src_lo = v + own0, *src_hi = v + own1 - n_hi;
    double *dst_lo = T.dst_lo, *dst_hi = T.dst_hi;
    const int64_t most = std::max(n_lo, n_hi);
    // (workgroups: enough write-through stores in flight for 2 x 512 KiB, few enough tickets on one address - measured on the
    // 256 x 256 plane: 16 / 32 / 64 / 128 / 256 workgroups = 50.4 / 45.0 / 40.9 / 40.8 / 41.9 us per iteration)
    const int grid = (int)std::min<int64_t>(std::max<int64_t>((most + 511) / 512, 1), 128);
    k_halo_push<<<grid, 256, 0, c->stream>>>(src_lo, dst_lo, n_lo, src_hi, dst_hi, n_hi, T.post_lo, T.post_hi, T.wait_a, T.wait_b, k.push_seq,
                                             k.push_flags + 2, push_ticks(k), flags);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

// ---- direct all-reduce of the single-sync loop's sums (opt-in with the direct halo)
// ONE workgroup: (1) the iteration's local sums exactly as k_pcg1_sums forms them (same lanes, same order) - or, `from_slots`, the
// five numbers another kernel has left in slots[base ..]; (2) lane t < world stores them into rank t's mailbox and posts the sequence
// number behind them; (3) lane t waits for rank t's number in this rank's block; (4) the contributions are added in RANK ORDER into
// slots[base .. base + 4]: every rank gets the same bits.  A launch that finds the solve done (every rank alike) only posts - a rank
// that failed locally and never sees the flag is not left waiting - and a rank that failed locally sends NaNs (`poison`), which is how
// the others learn of it, as with the binding's all-reduce.
struct ArArgs {
    unsigned long long *peer[PUSH_AR_MAXW];
    unsigned long long *own;
    int rank, world;
    unsigned long long seq;
    long long ticks;
};

__global__ __launch_bounds__(1024) void k_allreduce_direct(const double *__restrict__ prod, int nprod, const double *__restrict__ vecp, int nvec,
                                                           double *__restrict__ slots, int base, int *__restrict__ flags, int from_slots,
                                                           int poison, ArArgs A) {
    __shared__ double s_w[16];
    __shared__ double s_v[8];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (flags[0]) {
        if (tid < A.world) __hip_atomic_store(A.peer[tid] + PUSH_AR_FLAG0 + A.rank, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (tid == 0) s_bad = 0;
    if (!from_slots && !poison) {
        const int v = wv >> 2, t = tid & 255;
        const double *src = v < 2 ? prod : vecp;
        const int n = v < 2 ? nprod : nvec, off = v & 1;
        double a8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int i = t;
        for (; i + 7 * 256 < n; i += 8 * 256) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a8[u] += src[2 * (int64_t)(i + u * 256) + off];
        }
        for (int u = 0; i < n; i += 256, ++u) a8[u & 7] += src[2 * (int64_t)i + off];
        const double acc = wave_sum(((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7])));
        if (lane == 0) s_w[wv] = acc;
    }
    __syncthreads();
    if (tid < 5) {
        double v;
        if (poison) v = __longlong_as_double(0x7ff8000000000000ll);
        else if (from_slots) v = slots[base + tid];
        else v = tid < 4 ? (s_w[4 * tid] + s_w[4 * tid + 1]) + (s_w[4 * tid + 2] + s_w[4 * tid + 3]) : 0.0;
        s_v[tid] = v;
    }
    __syncthreads();
    const int par = (int)(A.seq & 1ull);
    if (tid < A.world) {
        double *box = reinterpret_cast<double *>(reinterpret_cast<char *>(A.peer[tid]) + PUSH_BOX_OFF) + ((int64_t)par * PUSH_AR_MAXW + A.rank) * 8;
#pragma unroll
        for (int j = 0; j < 5; ++j)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(box) + j, (unsigned long long)__double_as_longlong(s_v[j]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        __builtin_amdgcn_s_waitcnt(0);
        __hip_atomic_store(A.peer[tid] + PUSH_AR_FLAG0 + A.rank, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(A.own + PUSH_AR_FLAG0 + tid, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < A.seq) {
            if (wall_clock64() - t0 > A.ticks) { s_bad = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    if (s_bad) {
        if (tid == 0) { flags[2] = PGD_ERR_TIMEOUT; flags[0] = 1; }
        return;
    }
    if (tid < 5) {
        const double *box = reinterpret_cast<const double *>(reinterpret_cast<const char *>(A.own) + PUSH_BOX_OFF) + (int64_t)par * PUSH_AR_MAXW * 8;
        double sum = 0.0;
        for (int r = 0; r < A.world; ++r)
            sum += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(box + r * 8) + tid, __ATOMIC_RELAXED,
                                                                      __HIP_MEMORY_SCOPE_SYSTEM));
        slots[base + tid] = sum;
    }
}

// nprod < 0: the five local numbers are in slots[base ..] already
static int comm_allreduce_direct(Ctx *c, int nprod, int nvec, int base, bool poison) {
    Comm &k = c->comm;
    if (!k.ar) return fail(c, PGD_ERR_INVALID, "direct all-reduce: not attached");
    const double *prod = c->partials;
    int from_slots = nprod < 0 ? 1 : 0;
    if (!poison && nprod > 8192) {          // (k_pcg1_sums reduces that many pairs in two stages: let it, and exchange its result)
        PGD_TRY(pcg1_sums(c, nprod, nvec, base));
        from_slots = 1;
    }
    k.ar_seq += 1;
    ArArgs A;
    for (int r = 0; r < PUSH_AR_MAXW; ++r) A.peer[r] = r < k.world ? k.ar_peer[r] : nullptr;
    A.own = k.push_flags; A.rank = k.rank; A.world = k.world; A.seq = k.ar_seq; A.ticks = push_ticks(k);
    k_allreduce_direct<<<1, 1024, 0, c->stream>>>(prod, from_slots ? 0 : nprod, c->work[6], from_slots ? 0 : nvec, c->slots, base, c->flags, from_slots,
                                                  poison ? 1 : 0, A);
    PGD_LAUNCH_CHECK(c);
    return PGD_OK;
}

// neighbour planes -> ghost planes of v (local numbering: [0, lo_g) ghost below, [own0, own1) owned,
// [own1, own1 + hi_g) ghost above).  `async` (RCCL binding with a halo communicator only): the exchange runs on the
// halo stream behind an event of the compute stream, and comm_halo_wait makes the compute stream wait for it - rows
// that read no ghost entry can be multiplied in between.  Otherwise stream-ordered on the compute stream.
// WHICH communicator carries an exchange must be the same on both ends: callers pass an `async` that depends on nothing
// rank-local (inside a solve: every exchange of a product async, every other one not; k.overlap is agreed on at bind time).
static int comm_halo_begin(Ctx *c, pgd_handle vh, double *v, int64_t own0, int64_t own1, int64_t lo_g, int64_t hi_g, bool async) {
    Comm &k = c->comm;
    if (k.kind == 1) {
        const int rc = k.halo_cb(k.user, vh, own0, own1, lo_g, hi_g);
        return rc == 0 ? PGD_OK : fail(c, PGD_ERR_INVALID, "halo callback failed (%d)", rc);
    }
    if (k.kind != 2) return fail(c, PGD_ERR_INVALID, "no communication binding (pgd_comm_bind_*)");
    if (!lo_g && !hi_g) return PGD_OK;
    std::string why;
    RcclApi *api = rccl_api(why);
    if (!api) return fail(c, PGD_ERR_INVALID, "rccl: %s", why.c_str());
    const bool over = async && k.overlap;
    hipStream_t st = over ? k.halo_stream : c->stream;
    void *comm = over ? k.nccl_halo : k.nccl;
    if (over) {
        PGD_HIP(c, hipEventRecord(k.ev_ready, c->stream));               // v is complete on the compute stream
        PGD_HIP(c, hipStreamWaitEvent(k.halo_stream, k.ev_ready, 0));
    }
    PGD_NCCL(c, api, api->GroupStart());
    if (k.self_periodic && k.world == 1) {
        // tests: the rank is its own neighbour on both sides (periodic in z).  Transfers between the same pair of ranks are matched
        // in the order they are issued: the top plane first (-> the ghost plane below), then the bottom plane (-> the one above)
        if (lo_g) PGD_NCCL(c, api, api->Send(v + own1 - lo_g, (size_t)lo_g, NCCL_F64, 0, comm, st));
        if (lo_g) PGD_NCCL(c, api, api->Recv(v, (size_t)lo_g, NCCL_F64, 0, comm, st));
        if (hi_g) PGD_NCCL(c, api, api->Send(v + own0, (size_t)hi_g, NCCL_F64, 0, comm, st));
        if (hi_g) PGD_NCCL(c, api, api->Recv(v + own1, (size_t)hi_g, NCCL_F64, 0, comm, st));
    } else {
        if (lo_g) {
            PGD_NCCL(c, api, api->Send(v + own0, (size_t)lo_g, NCCL_F64, k.rank - 1, comm, st));
            PGD_NCCL(c, api, api->Recv(v, (size_t)lo_g, NCCL_F64, k.rank - 1, comm, st));
        }
        if (hi_g) {
            PGD_NCCL(c, api, api->Send(v + own1 - hi_g, (size_t)hi_g, NCCL_F64, k.rank + 1, comm, st));
            PGD_NCCL(c, api, api->Recv(v + own1, (size_t)hi_g, NCCL_F64, k.rank + 1, comm, st));
        }
    }
    PGD_NCCL(c, api, api->GroupEnd());
    if (over) PGD_HIP(c, hipEventRecord(k.ev_halo, k.halo_stream));
    return PGD_OK;
}

static int comm_halo_wait(Ctx *c, int64_t lo_g, int64_t hi_g, bool async) {
    Comm &k = c->comm;
    if (k.kind == 2 && async && k.overlap && (lo_g || hi_g)) PGD_HIP(c, hipStreamWaitEvent(c->stream, k.ev_halo, 0));
    return PGD_OK;
}

static int comm_halo(Ctx *c, pgd_handle vh, double *v, int64_t own0, int64_t own1, int64_t lo_g, int64_t hi_g) {
    return comm_halo_begin(c, vh, v, own0, own1, lo_g, hi_g, false);
}

static int comm_allreduce(Ctx *c, int first, int count) {
    Comm &k = c->comm;
    if (first < 0 || count < 1 || first + count > PGD_NSLOTS) return fail(c, PGD_ERR_INVALID, "allreduce: slot range");
    if (k.kind == 1) {
        const int rc = k.allreduce_cb(k.user, first, count);
        return rc == 0 ? PGD_OK : fail(c, PGD_ERR_INVALID, "allreduce callback failed (%d)", rc);
    }
    if (k.kind != 2) return fail(c, PGD_ERR_INVALID, "no communication binding (pgd_comm_bind_*)");
    std::string why;
    RcclApi *api = rccl_api(why);
    if (!api) return fail(c, PGD_ERR_INVALID, "rccl: %s", why.c_str());
    PGD_NCCL(c, api, api->AllReduce(c->slots + first, c->slots + first, (size_t)count, NCCL_F64, NCCL_SUM, k.nccl, c->stream));
    return PGD_OK;
}

constexpr int PUSH_TEST_ROUNDS = 8;
// entries of p[0, n) that differ from v are counted in *bad
__global__ void k_comm_compare(const double *p, int64_t n, double v, int *bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && p[i] != v) atomicAdd(bad, 1);
}

__global__ void k_comm_fill(double *p, int n, double v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// After binding: a ring shift (every rank sends its id to rank+1, receives from rank-1; one rank: to
// itself) and an all-reduce of the ids, both checked - a wrong transport is reported here, not as a wrong
// solution later.
static int comm_selftest(Ctx *c) {
    Comm &k = c->comm;
    std::string why;
    RcclApi *api = rccl_api(why);
    if (!api) return fail(c, PGD_ERR_INVALID, "rccl: %s", why.c_str());
    constexpr int N = 256;
    PGD_TRY(ensure_work(c, 5, 2 * N));
    double *snd = c->work[5], *rcv = c->work[5] + N;
    k_comm_fill<<<1, N, 0, c->stream>>>(snd, N, 1000.0 + k.rank);
    k_comm_fill<<<1, N, 0, c->stream>>>(rcv, N, -1.0);
    const int next = (k.rank + 1) % k.world, prev = (k.rank + k.world - 1) % k.world;
    PGD_NCCL(c, api, api->GroupStart());
    PGD_NCCL(c, api, api->Send(snd, N, NCCL_F64, next, k.nccl, c->stream));
    PGD_NCCL(c, api, api->Recv(rcv, N, NCCL_F64, prev, k.nccl, c->stream));
    PGD_NCCL(c, api, api->GroupEnd());
    k_comm_fill<<<1, 1, 0, c->stream>>>(c->slots + 48, 1, (double)k.rank + 1.0);
    PGD_NCCL(c, api, api->AllReduce(c->slots + 48, c->slots + 48, 1, NCCL_F64, NCCL_SUM, k.nccl, c->stream));
    double got[2] = {0.0, 0.0}, sum = 0.0;
    PGD_HIP(c, hipMemcpyAsync(got, rcv, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipMemcpyAsync(got + 1, rcv + N - 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipMemcpyAsync(&sum, c->slots + 48, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    const double want = 1000.0 + prev, wsum = 0.5 * k.world * (k.world + 1.0);
    if (got[0] != want || got[1] != want) return fail(c, PGD_ERR_HIP, "rccl self-test: ring shift delivered %g, expected %g", got[0], want);
    if (sum != wsum) return fail(c, PGD_ERR_HIP, "rccl self-test: all-reduce gave %g, expected %g", sum, wsum);
    return PGD_OK;
}

// The halo communicator (a split of the first) on its own stream, ordered against the compute stream by the two events
// exactly as the solve uses them: fill on the compute stream -> ring shift on the halo stream -> read on the compute
// stream.  Any failure simply leaves the overlap off (the exchange then stays on the compute stream).
static void comm_setup_overlap(Ctx *c) {
    Comm &k = c->comm;
    k.overlap = false;
    const char *env = getenv("PGD_HALO_OVERLAP");
    if (env && env[0] == '0') return;
    std::string why;
    RcclApi *api = rccl_api(why);
    if (!api || !api->CommSplit) return;
    if (api->CommSplit(k.nccl, 0, k.rank, &k.nccl_halo, nullptr) != 0 || !k.nccl_halo) { k.nccl_halo = nullptr; return; }
    if (hipStreamCreateWithFlags(&k.halo_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&k.ev_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&k.ev_halo, hipEventDisableTiming) != hipSuccess)
        return;
    constexpr int N = 256;
    if (ensure_work(c, 5, 2 * N) != PGD_OK) return;
    double *snd = c->work[5], *rcv = c->work[5] + N;
    k_comm_fill<<<1, N, 0, c->stream>>>(snd, N, 2000.0 + k.rank);
    k_comm_fill<<<1, N, 0, c->stream>>>(rcv, N, -1.0);
    const int next = (k.rank + 1) % k.world, prev = (k.rank + k.world - 1) % k.world;
    bool ok = hipEventRecord(k.ev_ready, c->stream) == hipSuccess && hipStreamWaitEvent(k.halo_stream, k.ev_ready, 0) == hipSuccess;
    ok = ok && api->GroupStart() == 0;
    ok = ok && api->Send(snd, N, NCCL_F64, next, k.nccl_halo, k.halo_stream) == 0;
    ok = ok && api->Recv(rcv, N, NCCL_F64, prev, k.nccl_halo, k.halo_stream) == 0;
    ok = ok && api->GroupEnd() == 0;
    ok = ok && hipEventRecord(k.ev_halo, k.halo_stream) == hipSuccess && hipStreamWaitEvent(c->stream, k.ev_halo, 0) == hipSuccess;
    double got = 0.0;
    ok = ok && hipMemcpyAsync(&got, rcv + N / 2, sizeof(double), hipMemcpyDeviceToHost, c->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(c->stream) == hipSuccess;
    k.overlap = ok && got == 2000.0 + prev;
    (void)hipGetLastError();
}

}  // namespace pgd

using namespace pgd;

extern "C" {

int pgd_comm_bind_callbacks(pgd_handle h, pgd_halo_fn halo, pgd_allreduce_fn allreduce, void *user, int rank, int world) {
    PGD_CTX(c, h);
    if (!halo || !allreduce || world < 1 || rank < 0 || rank >= world) return fail(c, PGD_ERR_INVALID, "comm_bind_callbacks: invalid arguments");
    comm_release(c);
    c->comm.kind = 1; c->comm.rank = rank; c->comm.world = world;
    comm_env_defaults(c->comm);
    c->comm.halo_cb = halo; c->comm.allreduce_cb = allreduce; c->comm.user = user;
    return PGD_OK;
}

int pgd_comm_unique_id(pgd_handle h, uint8_t *out128) {
    PGD_CTX(c, h);
    std::string why;
    RcclApi *api = rccl_api(why);
    if (!api || !out128) return fail(c, PGD_ERR_INVALID, "comm_unique_id: %s", api ? "null output" : why.c_str());
    NcclId id;
    PGD_NCCL(c, api, api->GetUniqueId(&id));
    std::copy(id.internal, id.internal + sizeof id.internal, (char *)out128);
    return PGD_OK;
}

int pgd_comm_bind_rccl(pgd_handle h, const uint8_t *id128, int rank, int world) {
    PGD_CTX(c, h);
    std::string why;
    RcclApi *api = rccl_api(why);
    if (!api) return fail(c, PGD_ERR_INVALID, "comm_bind_rccl: %s", why.c_str());
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(c, PGD_ERR_INVALID, "comm_bind_rccl: invalid arguments");
    comm_release(c);
    NcclId id;
    std::copy((const char *)id128, (const char *)id128 + sizeof id.internal, id.internal);
    void *comm = nullptr;
    PGD_NCCL(c, api, api->CommInitRank(&comm, world, id, rank));
    c->comm.kind = 2; c->comm.rank = rank; c->comm.world = world; c->comm.nccl = comm;
    comm_env_defaults(c->comm);
    const int rc = comm_selftest(c);
    if (rc != PGD_OK) { const std::string keep = c->err; comm_release(c); c->err = keep; return rc; }
    return rc;
}

int pgd_comm_overlap(pgd_handle h, int mode, int *state) {
    PGD_CTX(c, h);
    Comm &k = c->comm;
    if (mode == 1) {
        // COLLECTIVE over the bound RCCL communicator (ncclCommSplit + a ring shift on the halo stream): call it on every
        // rank, and only after every rank reported a successful pgd_comm_bind_rccl
        if (k.kind != 2) return fail(c, PGD_ERR_INVALID, "comm_overlap: needs the RCCL binding");
        if (!k.nccl_halo) comm_setup_overlap(c);
    } else if (mode == 0) {
        k.overlap = false;
    } else if (mode == -2) {
        if (state) *state = k.overlap_used ? 1 : 0;      // did the last pgd_pcg_solve_sharded take the second stream?
        return PGD_OK;
    } else if (mode != -1) {
        return fail(c, PGD_ERR_INVALID, "comm_overlap: mode must be 1 (enable), 0 (disable), -1 (query) or -2 (the last solve)");
    }
    if (state) *state = k.overlap ? 1 : 0;
    return PGD_OK;
}

int pgd_comm_unbind(pgd_handle h) {
    PGD_CTX(c, h);
    comm_release(c);
    return PGD_OK;
}

int pgd_comm_info(pgd_handle h, int *kind, int *rank, int *world) {
    PGD_CTX(c, h);
    if (kind) *kind = c->comm.kind;
    if (rank) *rank = c->comm.rank;
    if (world) *world = c->comm.world;
    return PGD_OK;
}

int pgd_comm_halo(pgd_handle h, pgd_handle vh, int64_t own0, int64_t own1, int64_t lo_g, int64_t hi_g) {
    PGD_CTX(c, h);
    Vec *v = get_vec(c, vh);
    if (!v || own0 != lo_g || own0 > own1 || own1 + hi_g != v->n || lo_g < 0 || hi_g < 0 || lo_g > own1 - own0 || hi_g > own1 - own0)
        return fail(c, PGD_ERR_INVALID, "comm_halo: partition does not fit the vector");
    if ((lo_g > 0) != (c->comm.rank > 0) || (hi_g > 0) != (c->comm.rank < c->comm.world - 1))
        return fail(c, PGD_ERR_INVALID, "comm_halo: ghost planes do not match the rank's position");
    return comm_halo(c, vh, v->d, own0, own1, lo_g, hi_g);
}

int pgd_comm_allreduce_slots(pgd_handle h, int first, int count) {
    PGD_CTX(c, h);
    return comm_allreduce(c, first, count);
}

// ---- direct halo: export -> (the caller carries the blobs to the neighbours) -> attach
struct PushBlob {
    hipIpcMemHandle_t vec, flags;
    int64_t n, own0, own1, lo_g, hi_g, pid;
    uint64_t vec_ptr, flags_ptr;         // addresses in the exporting process (used when the neighbour is that process)
};
static_assert(sizeof(PushBlob) <= PGD_PUSH_BLOB_BYTES, "the blob must fit its public size");

int pgd_comm_push_export(pgd_handle h, int64_t n, int64_t own0, int64_t own1, int64_t lo_g, int64_t hi_g, uint8_t *blob) {
    PGD_CTX(c, h);
    Comm &k = c->comm;
    if (!blob || n < 1 || n > 0x7fffffff || own0 != lo_g || own0 > own1 || own1 + hi_g != n || lo_g < 0 || hi_g < 0)
        return fail(c, PGD_ERR_INVALID, "comm_push_export: partition does not fit the vector");
    if (k.kind == 0) return fail(c, PGD_ERR_INVALID, "comm_push_export: no communication binding");
    push_drop(c);
    if (k.work_n != n) {
        for (pgd_handle &wh : k.work) { if (wh) (void)pgd_vec_free(h, wh); wh = 0; }
        k.work_n = 0;
        for (pgd_handle &wh : k.work) PGD_TRY(pgd_vec_alloc(h, n, &wh));
        k.work_n = n;
    }
    void *fl = nullptr;
    PGD_HIP(c, hipMalloc(&fl, PUSH_BLOCK_BYTES));      // flag words of the halo, of the all-reduce, and its mailbox (pgd_internal.h)
    k.push_flags = static_cast<unsigned long long *>(fl);
    PGD_HIP(c, hipMemsetAsync(fl, 0, PUSH_BLOCK_BYTES, c->stream));
    double *pv = get_vec(c, k.work[3])->d;
    // (the ghost planes start from a value no rank sends in the checked exchange of the attach step; set HERE, before any neighbour
    // can have the blob: a neighbour's planes may arrive while this rank is still attaching)
    k_comm_fill<<<(int)((n + 255) / 256), 256, 0, c->stream>>>(pv, (int)n, -1.0);
    PGD_LAUNCH_CHECK(c);
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    PushBlob b;
    memset(&b, 0, sizeof b);
    // (handles of memory another PROCESS will map; a neighbour inside this process takes the addresses - a failure here leaves the
    // handles zero, which only matters to a neighbour in another process: its attach fails and the exchange stays with the binding)
    if (hipIpcGetMemHandle(&b.vec, pv) != hipSuccess || hipIpcGetMemHandle(&b.flags, fl) != hipSuccess) {
        (void)hipGetLastError();
        memset(&b.vec, 0, sizeof b.vec);
        memset(&b.flags, 0, sizeof b.flags);
    }
    b.n = n; b.own0 = own0; b.own1 = own1; b.lo_g = lo_g; b.hi_g = hi_g; b.pid = (int64_t)getpid();
    b.vec_ptr = reinterpret_cast<uint64_t>(pv);
    b.flags_ptr = reinterpret_cast<uint64_t>(fl);
    memset(blob, 0, PGD_PUSH_BLOB_BYTES);
    memcpy(blob, &b, sizeof b);
    k.push_n = n; k.push_own0 = own0; k.push_own1 = own1; k.push_lo_g = lo_g; k.push_hi_g = hi_g;
    k.push_seq = 0;
    k.ar_seq = 0;
    return PGD_OK;
}

// lower / upper: the blobs of the ranks below and above (NULL where there is none; with ONE self-periodic rank both are its own).
// Collective over the neighbours: ends with a checked exchange - every rank pushes planes filled with 1000 + its rank and reads what
// arrived.  On any failure the direct halo stays off (state 0) and the call still returns PGD_OK unless an argument was wrong: the
// solves vote, so one rank without it switches it off for all.
int pgd_comm_push_attach(pgd_handle h, const uint8_t *lower, const uint8_t *upper, int *state) {
    PGD_CTX(c, h);
    Comm &k = c->comm;
    if (state) *state = 0;
    if (!k.push_flags || !k.push_n) return fail(c, PGD_ERR_INVALID, "comm_push_attach: pgd_comm_push_export first");
    const bool self = k.self_periodic && k.world == 1;
    if ((k.push_lo_g > 0) != (lower != nullptr) || (k.push_hi_g > 0) != (upper != nullptr))
        return fail(c, PGD_ERR_INVALID, "comm_push_attach: a blob for every side with ghost planes, and for no other");
    const uint8_t *blobs[2] = {lower, upper};
    bool ok = true;
    int nmap = 0;
    for (int side = 0; side < 2 && ok; ++side) {
        if (!blobs[side]) continue;
        PushBlob b;
        memcpy(&b, blobs[side], sizeof b);
        // the neighbour's ghost planes towards us must be as large as our boundary plane
        const int64_t want = self ? (side == 0 ? k.push_hi_g : k.push_lo_g) : (side == 0 ? k.push_lo_g : k.push_hi_g);
        const int64_t have = side == 0 ? b.hi_g : b.lo_g;
        if (have != want || b.n < want) return fail(c, PGD_ERR_INVALID, "comm_push_attach: the neighbour's ghost planes do not match this rank's boundary planes");
        if (b.pid == (int64_t)getpid()) {
            k.push_peer[side] = reinterpret_cast<double *>(b.vec_ptr);
            k.push_peer_flags[side] = reinterpret_cast<unsigned long long *>(b.flags_ptr);
        } else {
            void *pv = nullptr, *pf = nullptr;
            if (hipIpcOpenMemHandle(&pv, b.vec, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { ok = false; break; }
            k.push_mapped[nmap++] = pv;
            if (hipIpcOpenMemHandle(&pf, b.flags, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { ok = false; break; }
            k.push_mapped[nmap++] = pf;
            k.push_peer[side] = static_cast<double *>(pv);
            k.push_peer_flags[side] = static_cast<unsigned long long *>(pf);
        }
        k.push_peer_n[side] = b.n;
    }
    (void)hipGetLastError();
    if (!ok) {
        // (the neighbours still run the checked exchange below and would wait for this rank's number: it cannot be posted without a
        // mapping, so they find out through their deadline-free test wait - bounded at 2 s - and switch off as well)
        for (void *&m : k.push_mapped) if (m) { (void)hipIpcCloseMemHandle(m); m = nullptr; }
        k.push_peer[0] = k.push_peer[1] = nullptr;
        (void)hipGetLastError();
        return PGD_OK;
    }
    // checked exchanges, exactly as a solve does them - PUSH_TEST_ROUNDS of them with data that changes from round to round (a ghost
    // plane served from a stale cache line shows up as the previous round's value) and EVERY ghost entry compared on the device by a
    // kernel of its own, the way the product reads them.  Between two rounds the planes are pushed once more, unchanged, as an
    // acknowledgement: a neighbour fills its next pattern only after this rank has compared the current one.
    double *pv = get_vec(c, k.work[3])->d;
    const int64_t n = k.push_n;
    int *tflags = nullptr;
    PGD_TRY(ensure_work(c, 5, 16));              // (before anything is switched: an early return must leave the binding as it was)
    tflags = reinterpret_cast<int *>(c->work[5]);
    k.push = true;
    const double keep_timeout = k.timeout_s;
    k.timeout_s = 2.0;
    (void)hipMemsetAsync(tflags, 0, 4 * sizeof(int), c->stream);
    int rc = PGD_OK;
    int tf[4] = {0, 0, 0, 0};
    for (int round = 0; round < PUSH_TEST_ROUNDS && rc == PGD_OK; ++round) {
        const double base = 1000.0 * (round + 1);
        if (k.push_own1 > k.push_own0) {      // the OWNED rows only: the ghost planes belong to the neighbours from the export on
            const int64_t rows = k.push_own1 - k.push_own0;
            k_comm_fill<<<(int)((rows + 255) / 256), 256, 0, c->stream>>>(pv + k.push_own0, (int)rows, base + k.rank);
        }
        rc = comm_push_halo(c, pv, tflags);
        if (rc != PGD_OK) break;
        const double below = base + (self ? k.rank : k.rank - 1), above = base + (self ? k.rank : k.rank + 1);
        if (k.push_lo_g) k_comm_compare<<<(int)((k.push_lo_g + 255) / 256), 256, 0, c->stream>>>(pv, k.push_lo_g, below, tflags + 1);
        if (k.push_hi_g) k_comm_compare<<<(int)((k.push_hi_g + 255) / 256), 256, 0, c->stream>>>(pv + k.push_own1, k.push_hi_g, above, tflags + 1);
        rc = comm_push_halo(c, pv, tflags);                                   // the acknowledgement
        if (hipGetLastError() != hipSuccess) rc = PGD_ERR_HIP;
        // (a deadline that passed - a neighbour could not map this rank - ends the rounds: no point in waiting 2 s sixteen times)
        if (rc == PGD_OK) rc = hipMemcpyAsync(tf, tflags, sizeof tf, hipMemcpyDeviceToHost, c->stream) == hipSuccess ? PGD_OK : PGD_ERR_HIP;
        if (rc == PGD_OK) rc = hipStreamSynchronize(c->stream) == hipSuccess ? PGD_OK : PGD_ERR_HIP;
        if (tf[0] || tf[1]) break;
    }
    k.timeout_s = keep_timeout;
    (void)n;
    const bool good = rc == PGD_OK && tf[0] == 0 && tf[1] == 0;               // no deadline passed, no entry differed
    (void)hipGetLastError();
    if (!good) {
        for (void *&m : k.push_mapped) if (m) { (void)hipIpcCloseMemHandle(m); m = nullptr; }
        k.push = false;
        k.push_peer[0] = k.push_peer[1] = nullptr;
        (void)hipGetLastError();
        return PGD_OK;
    }
    if (state) *state = 1;
    return PGD_OK;
}

// mode -1: read; 0: off (the mappings stay); 1: on again if attached.  state: 1 = the next solves may use it; -2 reads what the LAST solve did
int pgd_comm_push(pgd_handle h, int mode, int *state) {
    PGD_CTX(c, h);
    Comm &k = c->comm;
    if (mode == 2) {          // probes: ONE exchange of the loop's search direction as it stands, queued on the stream (collective)
        if (!k.push) return fail(c, PGD_ERR_INVALID, "comm_push: not attached");
        PGD_TRY(comm_push_halo(c, get_vec(c, k.work[3])->d, nullptr));
        if (state) *state = 1;
        return PGD_OK;
    }
    if (mode == 0) k.push = false;
    else if (mode == 1) k.push = k.push_peer[0] != nullptr || k.push_peer[1] != nullptr || (k.push_n > 0 && !k.push_lo_g && !k.push_hi_g && k.push_flags);
    else if (mode != -1 && mode != -2) return fail(c, PGD_ERR_INVALID, "comm_push: mode must be 2 (one exchange), 1, 0, -1 (read) or -2 (what the last solve did)");
    if (state) *state = mode == -2 ? (k.push_used ? 1 : 0) : (k.push ? 1 : 0);
    return PGD_OK;
}

// The direct all-reduce: `blobs` = the export blobs of ALL ranks in rank order (world x PGD_PUSH_BLOB_BYTES; world <= 16).  Maps every
// rank's flag block (a neighbour's is mapped already) and ends with a checked exchange: the sum of rank + 1 over the ranks.
// Collective over all ranks; *state = 1 if usable here.  The solves vote on it like on the direct halo.
int pgd_comm_allreduce_attach(pgd_handle h, const uint8_t *blobs, int *state) {
    PGD_CTX(c, h);
    Comm &k = c->comm;
    if (state) *state = 0;
    if (!k.push_flags || !k.push_n) return fail(c, PGD_ERR_INVALID, "comm_allreduce_attach: pgd_comm_push_export first");
    if (!blobs || k.world > PUSH_AR_MAXW) return fail(c, PGD_ERR_INVALID, "comm_allreduce_attach: no blobs, or more than %d ranks", PUSH_AR_MAXW);
    const bool self = k.self_periodic && k.world == 1;
    bool ok = true;
    for (int r = 0; r < k.world && ok; ++r) {
        PushBlob b;
        memcpy(&b, blobs + (size_t)r * PGD_PUSH_BLOB_BYTES, sizeof b);
        if (r == k.rank || b.pid == (int64_t)getpid()) {
            k.ar_peer[r] = r == k.rank ? k.push_flags : reinterpret_cast<unsigned long long *>(b.flags_ptr);
        } else if (!self && r == k.rank - 1 && k.push_peer_flags[0]) {
            k.ar_peer[r] = k.push_peer_flags[0];              // (the halo's attach has mapped the neighbours' blocks)
        } else if (!self && r == k.rank + 1 && k.push_peer_flags[1]) {
            k.ar_peer[r] = k.push_peer_flags[1];
        } else {
            void *pf = nullptr;
            if (hipIpcOpenMemHandle(&pf, b.flags, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { ok = false; break; }
            k.ar_mapped[r] = pf;
            k.ar_peer[r] = static_cast<unsigned long long *>(pf);
        }
    }
    (void)hipGetLastError();
    if (!ok) {
        for (void *&m : k.ar_mapped) if (m) { (void)hipIpcCloseMemHandle(m); m = nullptr; }
        for (auto &q : k.ar_peer) q = nullptr;
        (void)hipGetLastError();
        return PGD_OK;
    }
    // checked exchange: slots[48] = rank + 1 -> the sum over the ranks (a 2 s deadline: a rank that could not map posts nothing)
    PGD_TRY(ensure_work(c, 5, 16));
    int *tflags = reinterpret_cast<int *>(c->work[5]);
    k.ar = true;
    const double keep_timeout = k.timeout_s;
    k.timeout_s = 2.0;
    (void)hipMemsetAsync(tflags, 0, 4 * sizeof(int), c->stream);
    // (rounds with changing numbers, both parities of the mailbox several times: slot 48 = (rank + 1) (round + 1) -> its sum over the ranks)
    int rc = PGD_OK;
    bool sums_ok = true;
    for (int round = 0; round < PUSH_TEST_ROUNDS && rc == PGD_OK; ++round) {
        const double mine[5] = {(k.rank + 1.0) * (round + 1.0), 0.0, 0.0, 0.0, 0.0};
        rc = pgd_slots_upload(h, mine, 48, 5);
        if (rc != PGD_OK) break;
        k.ar_seq += 1;
        ArArgs A;
        for (int r = 0; r < PUSH_AR_MAXW; ++r) A.peer[r] = r < k.world ? k.ar_peer[r] : nullptr;
        A.own = k.push_flags; A.rank = k.rank; A.world = k.world; A.seq = k.ar_seq; A.ticks = push_ticks(k);
        k_allreduce_direct<<<1, 1024, 0, c->stream>>>(nullptr, 0, nullptr, 0, c->slots, 48, tflags, 1, 0, A);
        rc = hipGetLastError() == hipSuccess ? PGD_OK : PGD_ERR_HIP;
        double got = 0.0;
        if (rc == PGD_OK) rc = pgd_slots_download(h, &got, 48, 1);
        sums_ok = sums_ok && got == 0.5 * k.world * (k.world + 1.0) * (round + 1.0);
    }
    k.timeout_s = keep_timeout;
    int tf[4] = {0, 0, 0, 0};
    if (rc == PGD_OK) rc = hipMemcpy(tf, tflags, sizeof tf, hipMemcpyDeviceToHost) == hipSuccess ? PGD_OK : PGD_ERR_HIP;
    (void)hipGetLastError();
    const bool good = rc == PGD_OK && tf[0] == 0 && sums_ok;
    if (!good) {
        for (void *&m : k.ar_mapped) if (m) { (void)hipIpcCloseMemHandle(m); m = nullptr; }
        for (auto &q : k.ar_peer) q = nullptr;
        k.ar = false;
        (void)hipGetLastError();
        return PGD_OK;
    }
    if (state) *state = 1;
    return PGD_OK;
}

// mode 1 / 0: on (if attached) / off; -1 reads the state; -2 what the last solve did
int pgd_comm_allreduce_direct(pgd_handle h, int mode, int *state) {
    PGD_CTX(c, h);
    Comm &k = c->comm;
    if (mode == 2) {          // probes: ONE direct all-reduce of slots 48 .. 52, queued on the stream (collective)
        if (!k.ar) return fail(c, PGD_ERR_INVALID, "comm_allreduce_direct: not attached");
        PGD_TRY(ensure_work(c, 5, 16));
        int *tflags = reinterpret_cast<int *>(c->work[5]);
        PGD_HIP(c, hipMemsetAsync(tflags, 0, 4 * sizeof(int), c->stream));
        k.ar_seq += 1;
        ArArgs A;
        for (int r = 0; r < PUSH_AR_MAXW; ++r) A.peer[r] = r < k.world ? k.ar_peer[r] : nullptr;
        A.own = k.push_flags; A.rank = k.rank; A.world = k.world; A.seq = k.ar_seq; A.ticks = push_ticks(k);
        k_allreduce_direct<<<1, 1024, 0, c->stream>>>(nullptr, 0, nullptr, 0, c->slots, 48, tflags, 1, 0, A);
        PGD_LAUNCH_CHECK(c);
        if (state) *state = 1;
        return PGD_OK;
    }
    if (mode == 0) k.ar = false;
    else if (mode == 1) k.ar = k.ar_peer[k.rank] != nullptr;
    else if (mode != -1 && mode != -2) return fail(c, PGD_ERR_INVALID, "comm_allreduce_direct: mode must be 2 (one all-reduce), 1, 0, -1 or -2");
    if (state) *state = mode == -2 ? (k.ar_used ? 1 : 0) : (k.ar ? 1 : 0);
    return PGD_OK;
}

}  // extern "C"

// ---- collective discipline of one sharded solve ------------------------------------------------------------------
// After the setup vote EVERY rank issues the same sequence of collectives whatever happens to it locally:
//   * a rank-local failure (an allocation, a launch, a copy) POISONS the rank: it stops launching local work, but keeps
//     issuing every collective of the protocol - all-reduces with NaN payloads, halo planes with whatever the buffers hold;
//   * every look at the flags (before the first chunk, after every chunk of 16 iterations) and the end of the solve is an
//     AGREEMENT: a one-slot all-reduce of the "vote" (0 from a healthy rank, NaN from a poisoned one) in front of the
//     snapshot the host reads.  A non-zero vote takes every rank out of the solve at that very point with an error
//     (PGD_ERR_PEER on the healthy ones, its own error on the poisoned one); decisions from the flags themselves are
//     functions of all-reduced numbers and identical everywhere;
//   * what cannot be handled cooperatively - a failing RCCL call, a failing snapshot, a stream that makes no progress - ends
//     the call at once (`fatal`), and the OTHER ranks come out through their deadline (pgd_comm_timeout).
namespace pgd {

constexpr int SH_VOTE = 23;      // slot in front of the recurrence's bank B = 24 .. 32

struct Shard {
    Ctx *c;
    int rc = PGD_OK;             // first rank-local failure
    std::string err;
    const char *last = "none";   // last collective issued, for the deadline's report
    int64_t ncoll = 0;
    int iter = -1;               // iteration being queued
    bool poisoned() const { return rc != PGD_OK; }
    void local(int r) {
        if (rc == PGD_OK && r != PGD_OK) { rc = r; err = c->err; }
    }
};

#define SH_LOCAL(S, expr)                           \
    do {                                            \
        if (!(S).poisoned()) (S).local((expr));     \
    } while (0)

static int sh_allreduce(Shard &S, int first, int count, const char *what) {
    Ctx *c = S.c;
    if (S.poisoned()) (void)hipMemsetAsync(c->slots + first, 0xFF, (size_t)count * sizeof(double), c->stream);     // NaNs
    S.last = what;
    S.ncoll += 1;
    return comm_allreduce(c, first, count);
}

static int sh_halo_begin(Shard &S, pgd_handle vh, double *v, int64_t own0, int64_t own1, int64_t lo_g, int64_t hi_g, bool async,
                         const char *what) {
    S.last = what;
    S.ncoll += 1;
    return comm_halo_begin(S.c, vh, v, own0, own1, lo_g, hi_g, async);
}

// kernel that keeps the stream busy for `ms` milliseconds (tests of the deadline; bounded)
__global__ void k_comm_stall(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

// wait for an event with the communication deadline: hipEventQuery polling (busy for the first 2 ms, then 50 us naps)
static int wait_deadline(Shard &S, hipEvent_t ev, const char *what) {
    Ctx *c = S.c;
    Comm &k = c->comm;
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) return fail(c, PGD_ERR_HIP, "pcg_solve_sharded: %s: %s", what, hipGetErrorString(e));
        const double el = std::chrono::duration<double>(clk::now() - t0).count();
        if (k.timeout_s > 0.0 && el > k.timeout_s) {
            const int rc = fail(c, PGD_ERR_TIMEOUT,
                                "pcg_solve_sharded: rank %d/%d: no progress for %.1f s waiting for %s; iteration %d queued, last "
                                "collective issued: %s (#%lld of this solve)%s",
                                k.rank, k.world, el, what, S.iter, S.last, (long long)S.ncoll,
                                S.poisoned() ? "; this rank had failed locally before" : "");
            fprintf(stderr, "[pgd_amd] %s\n", c->err.c_str());
            fflush(stderr);
            return rc;
        }
        if (el > 2e-3) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    k.prof_sum[7] += std::chrono::duration<double>(clk::now() - t0).count();
    return PGD_OK;
}

static int shard_resources(Ctx *c) {
    Comm &k = c->comm;
    if (!k.snap_flags) {
        void *p = nullptr;
        PGD_HIP(c, hipHostMalloc(&p, 8 * sizeof(int) + 2 * sizeof(double), hipHostMallocDefault));
        k.snap_flags = (int *)p;
        k.snap_vote = reinterpret_cast<double *>(k.snap_flags + 8);
    }
    for (hipEvent_t &e : k.snap_ev)
        if (!e) PGD_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (k.prof)
        for (auto &set : k.mark)
            for (hipEvent_t &e : set)
                if (!e) PGD_HIP(c, hipEventCreate(&e));
    return PGD_OK;
}

static void prof_collect(Ctx *c, int set) {
    Comm &k = c->comm;
    if (!k.mark_set[set]) return;
    k.mark_set[set] = false;
    float ms[6];
    for (int i = 0; i < 6; ++i)
        if (hipEventElapsedTime(&ms[i], k.mark[set][i], k.mark[set][i + 1]) != hipSuccess) { (void)hipGetLastError(); return; }
    k.prof_sum[0] += 1.0;
    k.prof_sum[2] += 1e-3 * ms[0];      // interior product
    k.prof_sum[1] += 1e-3 * ms[1];      // waiting for the ghost planes
    k.prof_sum[3] += 1e-3 * ms[2];      // boundary rows
    k.prof_sum[4] += 1e-3 * ms[3];      // local sums
    k.prof_sum[5] += 1e-3 * ms[4];      // all-reduce
    k.prof_sum[6] += 1e-3 * ms[5];      // update
}

}  // namespace pgd

extern "C" {

int pgd_comm_timeout(pgd_handle h, double seconds) {
    PGD_CTX(c, h);
    if (!(seconds == seconds)) return fail(c, PGD_ERR_INVALID, "comm_timeout: not a number");
    c->comm.timeout_s = seconds;
    return PGD_OK;
}

int pgd_comm_prof(pgd_handle h, int mode, double *out8) {
    PGD_CTX(c, h);
    Comm &k = c->comm;
    if (mode == 1) {
        k.prof = true;
        for (double &v : k.prof_sum) v = 0.0;
    } else if (mode == 0) {
        k.prof = false;
    } else if (mode != -1) {
        return fail(c, PGD_ERR_INVALID, "comm_prof: mode must be 1 (on + reset), 0 (off) or -1 (read)");
    }
    if (out8)
        for (int i = 0; i < 8; ++i) out8[i] = k.prof_sum[i];
    return PGD_OK;
}

int pgd_pcg_solve_sharded(pgd_handle h, pgd_handle oh, pgd_handle bh, pgd_handle xh, int64_t own0, int64_t own1,
                          int64_t lo_g, int64_t hi_g, double rtol, double atol, int maxit, int *iters, double *rel) {
    PGD_CTX(c, h);
    Csr *op = get_csr(c, oh);
    Vec *b = get_vec(c, bh), *x = get_vec(c, xh);
    if (!op || !b || !x || b->n != x->n) return fail(c, PGD_ERR_INVALID, "pcg_solve_sharded: invalid handles");
    const int64_t n = x->n;
    if (own0 != lo_g || own0 > own1 || own1 + hi_g != n || lo_g < 0 || hi_g < 0)
        return fail(c, PGD_ERR_INVALID, "pcg_solve_sharded: partition does not fit the vectors");
    Comm &k = c->comm;
    if (k.kind == 0) return fail(c, PGD_ERR_INVALID, "pcg_solve_sharded: no communication binding");
    if (!(k.self_periodic && k.world == 1) && ((lo_g > 0) != (k.rank > 0) || (hi_g > 0) != (k.rank < k.world - 1)))
        return fail(c, PGD_ERR_INVALID, "pcg_solve_sharded: ghost planes do not match the rank's position");
    if (k.self_periodic && k.world == 1 && lo_g != hi_g)
        return fail(c, PGD_ERR_INVALID, "pcg_solve_sharded: a rank that is its own neighbour needs ghost planes of equal size on both sides");
    constexpr int B = 24, CHECK = 16, V = SH_VOTE;       // slot base of the recurrence (pgdrome_amd/dist.py uses the same)
    const bool dbg_t = getenv("PGD_DEBUG_PCG") != nullptr;     // host timers of the call's phases on stderr (each behind a stream synchronisation)
    auto dbg_now = [&]() -> double {
        if (!dbg_t) return 0.0;
        (void)hipStreamSynchronize(c->stream);
        return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    const double dbg_t0 = dbg_now();
    static_assert(V == B - 1, "the vote rides in front of the recurrence's slots");
    Mesh *m = get_mesh(c, op->mesh);

    // ---- phase A: everything that can fail on ONE rank before the first collective (allocations, the symmetric copy), and
    // the choice of recurrence, are agreed on with one all-reduce: a rank whose operator did not qualify for the symmetric
    // storage must not take another branch (the scaled recurrence has one more halo exchange) than its neighbours.
    bool sym = false, ss_all = false, push_all = false, ar_all = false;
    double rows_all = 0.0;
    auto setup = [&]() -> int {
        if (!m) return fail(c, PGD_ERR_INVALID, "pcg_solve_sharded: operator without a mesh");
        if (k.work_n != n) {      // r, u, w, p, s, q, dinv as library vectors (the slot kernels take handles)
            push_drop(c);         // (the direct halo was exported for the old p: this rank votes it off below, for good)
            for (pgd_handle &wh : k.work) { if (wh) (void)pgd_vec_free(h, wh); wh = 0; }
            k.work_n = 0;
            for (pgd_handle &wh : k.work) PGD_TRY(pgd_vec_alloc(h, n, &wh));
            k.work_n = n;
        }
        PGD_TRY(shard_resources(c));
        PGD_TRY(ensure_sym(c, m, op, &sym));
        // the product's partial sums of ALL its row ranges lie side by side, in pairs in the single-sync form: room for the worst
        // case (row-order launches: one workgroup per 64 rows) now, so that no launch inside the loop has to move the buffer
        PGD_TRY(ensure_partials(c, std::max<int64_t>(2 * ((own1 - own0 + 63) / 64 + 8) + 64, 4 * (int64_t)MAX_VEC_BLOCKS)));
        PGD_TRY(ensure_work(c, 6, 2 * (int64_t)MAX_VEC_BLOCKS));
        PGD_TRY(pgd_flags_reset(h));
        return PGD_OK;
    };
    const int rc_setup = setup();
    const std::string err_setup = c->err;
    {
        const bool can_ss = rc_setup == PGD_OK && sym && c->pcg_scaled && c->pcg_single_sync && m && m->sym_nx > 0;
        // (the fourth number: the rows this rank owns - their sum decides, identically everywhere, whether the halo exchange of the
        // products takes the second stream)
        // (the fifth: the direct halo is attached here for exactly this vector - used only if it is on every rank)
        const bool can_push = k.push && k.push_n == n && k.push_own0 == own0 && k.push_own1 == own1 && k.push_lo_g == lo_g && k.push_hi_g == hi_g;
        const double vote[6] = {rc_setup != PGD_OK ? 1.0 : 0.0, (rc_setup == PGD_OK && sym && c->pcg_scaled) ? 0.0 : 1.0, can_ss ? 0.0 : 1.0,
                                (double)(own1 - own0), can_push ? 0.0 : 1.0, k.ar ? 0.0 : 1.0};      // (the sixth: the direct all-reduce)
        double got[6] = {1.0, 1.0, 1.0, 0.0, 1.0, 1.0};
        int rc = pgd_slots_upload(h, vote, B, 6);
        if (rc == PGD_OK) rc = comm_allreduce(c, B, 6);
        if (rc == PGD_OK) rc = pgd_slots_download(h, got, B, 6);
        if (rc_setup != PGD_OK) { c->err = err_setup; return rc_setup; }
        if (rc != PGD_OK) return rc;
        if (got[0] != 0.0) return fail(c, PGD_ERR_PEER, "pcg_solve_sharded: the setup failed on another rank");
        sym = sym && got[1] == 0.0;                     // scaled only if EVERY rank can
        ss_all = got[2] == 0.0;                         // ... and the single-sync recurrence only if every rank's slab is a grid
        rows_all = got[3];
        push_all = got[4] == 0.0;
        ar_all = got[5] == 0.0;
    }
    const double dbg_t1 = dbg_now();
    const bool scaled = sym && c->pcg_scaled;
    const bool ss = scaled && ss_all;                   // single-sync recurrence: 7 (here 8: the true norm every iteration) vector passes
    const pgd_handle r = k.work[0], u = k.work[1], w = k.work[2], p = k.work[3], s = k.work[4], q = k.work[5], dinv = k.work[6];

    // ---- from here on: the collective discipline described above
    Shard S{c};
    struct ProfIterGuard { Ctx *c; ~ProfIterGuard() { c->prof_iter = -1; } } prof_iter_guard{c};
    c->prof_pend.clear();
    auto fault_at = [&](int stage) {                    // tests (PGD_TUNE_FAULT_STAGE)
        if (c->fault_stage == stage && !S.poisoned()) {
            c->fault_stage = 0;
            S.local(fail(c, PGD_ERR_HIP, "pcg_solve_sharded: injected fault at stage %d (PGD_TUNE_FAULT_STAGE)", stage));
        }
    };
    fault_at(1);
    const double zeros[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    SH_LOCAL(S, pgd_slots_upload(h, zeros, V, 10));
    SH_LOCAL(S, pgd_op_diag_inv(h, oh, dinv));
    double *scp = get_vec(c, dinv)->d;
    double *xd = x->d, *rd = get_vec(c, r)->d, *wd = get_vec(c, w)->d, *pd = get_vec(c, p)->d, *sd = get_vec(c, s)->d,
           *qd = get_vec(c, q)->d;

    // On every exit after the operator and x were scaled: x back to sc x~, the slot arrays no longer taken for A.
    struct ScaleGuard {
        Ctx *c; Csr *op; double *x; const double *sc; int64_t n; bool x_scaled, op_scaled;
        ~ScaleGuard() {
            if (x_scaled) (void)vec_div_mul(c, x, sc, n, 1);
            if (op_scaled) { op->uvals_valid = false; op->uvals_scaled = false; }
        }
    } guard{c, op, xd, scp, n, false, false};

    // one agreement: vote -> all-reduce -> snapshot of (flags, vote) into pinned slot `sn` + its event
    auto agree_begin = [&](int sn, const char *what) -> int {
        PGD_TRY(sh_allreduce(S, V, 1, what));
        PGD_HIP(c, hipMemcpyAsync(k.snap_flags + 4 * sn, c->flags, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        PGD_HIP(c, hipMemcpyAsync(k.snap_vote + sn, c->slots + V, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        PGD_HIP(c, hipEventRecord(k.snap_ev[sn], c->stream));
        return PGD_OK;
    };
    // ... and its outcome: PGD_OK = everybody healthy (flags in f), else the error this rank leaves with
    auto agree_end = [&](int sn, const char *what, int f[4]) -> int {
        PGD_TRY(wait_deadline(S, k.snap_ev[sn], what));
        for (int i = 0; i < 4; ++i) f[i] = k.snap_flags[4 * sn + i];
        const double vote = k.snap_vote[sn];
        // (a poisoned rank leaves like everybody else: at the first agreement that CARRIES its vote.  With the loop pipelined the
        // agreement read here may have been issued before the failure - its vote is 0 and its flags are good on every rank)
        if (vote == 0.0) return PGD_OK;
        if (S.poisoned()) { c->err = S.err; return S.rc; }
        return fail(c, PGD_ERR_PEER, "pcg_solve_sharded: another rank failed locally (%s); every rank leaves the solve", what);
    };

    // w = A u on the owned rows, S[B + 2] (+ S[B + 3], S[B + 4]) <- local w.u: the rows that read no ghost entry first,
    // while the boundary planes travel (overlap binding) - the same three launches in the same order either way, so the
    // overlapped and the stream-ordered solve are bit-identical.  `folded`: partial sums of all three ranges side by
    // side in the scratch, one later reduction (k_reduce_two); else one reduction per range into its own slot.
    // Returns an error only for what ends the protocol (a failing collective); local failures poison S.
    int64_t glo = lo_g, ghi = hi_g;
    if (own1 - own0 < glo + ghi) { glo = own1 - own0; ghi = 0; }     // a rank that owns a single plane: nothing to overlap
    // the SAME choice on every rank: the capability was agreed on at bind time, the row count is all-reduced
    const bool async = k.overlap && rows_all >= (double)k.overlap_min_rows * (double)k.world;
    k.overlap_used = async;
    // the direct halo carries the search direction of the single-sync loop (one all-reduce between a product and the next update:
    // nobody overwrites ghost planes that are still being read), in stream order
    const bool pushing = push_all && ss && !async;
    k.push_used = pushing;
    // ... and the loop's five sums travel through the ranks' mailboxes (formed by the same kernel) instead of k_pcg1_sums + the binding's
    // all-reduce - still a full exchange: it completes on a rank only when every rank has contributed
    const bool direct_ar = ar_all && ss;
    k.ar_used = direct_ar;
    hipEvent_t *marks = nullptr;                                     // phase timing of the iteration being queued (or none)
    auto mark = [&](int i) { if (marks) (void)hipEventRecord(marks[i], c->stream); };
    // ... and leaves from the update kernel of the iteration before where that launch can carry it (fold, even rows and plane sizes):
    // `p_sent` says whether the planes of the CURRENT p are already on their way (else the product sends them itself - the first
    // iteration, a rank that skipped its update after a local failure)
    bool p_sent = false;
    auto product = [&](pgd_handle uh, double *ud, double *wdst, bool folded, int *np_total) -> int {
        mark(0);
        if (pushing && ud == pd) {
            S.last = "direct halo of the product";
            S.ncoll += 1;
            if (!p_sent) PGD_TRY(comm_push_halo(c, ud, c->flags));
            p_sent = false;
        } else PGD_TRY(sh_halo_begin(S, uh, ud, own0, own1, lo_g, hi_g, async, "halo exchange of the product"));
        const int64_t lo[3] = {own0 + glo, own0, own1 - ghi}, hi[3] = {own1 - ghi, own0 + glo, own1};
        const int mult = c->spmv_qq ? 2 : 1;      // pairs (w.y, y.y) per workgroup in the single-sync form
        int total = 0;
        auto range = [&](int part) -> int {
            if (!folded) return pgd_spmv_dot_slot(h, oh, uh, (wdst == wd) ? w : q, uh, lo[part], hi[part], B + 2 + part);
            int np = 0;
            c->partials_off = mult * (int64_t)total;
            const int rc = launch_spmv_op(c, m, op, ud, wdst, ud, lo[part], hi[part], true, true, c->flags, &np);
            c->partials_off = 0;
            PGD_TRY(rc);
            total += np;
            return PGD_OK;
        };
        // the exchange in stream order and the operator one stencil whose ghost planes are data planes (k_stencil_ghost): ALL owned
        // planes in one march - no boundary launch (7 us + a launch gap per iteration on the slab of an 8-GPU rank)
        if (folded && !async && c->shard_one_march && (glo || ghi) && !S.poisoned() && stencil_row_range(c, m, op, own0, own1)) {
            auto whole = [&]() -> int {
                int np = 0;
                c->partials_off = 0;
                PGD_TRY(launch_spmv_op(c, m, op, ud, wdst, ud, own0, own1, true, true, c->flags, &np));
                total = np;
                return PGD_OK;
            };
            SH_LOCAL(S, whole());
            mark(1); mark(2); mark(3);
            if (np_total) *np_total = total;
            return PGD_OK;
        }
        SH_LOCAL(S, range(0));
        mark(1);
        PGD_TRY(comm_halo_wait(c, lo_g, hi_g, async));
        mark(2);
        bool both = false;
        if (folded) {       // both boundary planes in one row-order launch where the operator is in diagonal form
            auto both_planes = [&]() -> int {
                int np = 0;
                c->partials_off = mult * (int64_t)total;
                const int rc = launch_spmv_dia_rows2(c, m, op, ud, wdst, ud, lo[1], hi[1], lo[2], hi[2], true, c->flags, &np, &both);
                c->partials_off = 0;
                PGD_TRY(rc);
                if (both) total += np;
                return PGD_OK;
            };
            SH_LOCAL(S, both_planes());
        }
        if (!both) { SH_LOCAL(S, range(1)); SH_LOCAL(S, range(2)); }
        mark(3);
        if (np_total) *np_total = total;
        return PGD_OK;
    };

    if (scaled) {
        SH_LOCAL(S, vec_sqrt(c, scp, n));
        PGD_TRY(sh_halo_begin(S, dinv, scp, own0, own1, lo_g, hi_g, false, "halo exchange of d^-1/2"));
        fault_at(2);
        if (!S.poisoned()) guard.op_scaled = true;
        SH_LOCAL(S, sym_scale(c, m, op, scp));
        {   // rank-local choice of kernel, same bits either way; the stencil form is verified on the OWNED planes (ghost rows are incomplete)
            const int64_t plane = m && m->sym_nx > 0 ? (int64_t)m->sym_nx * m->sym_ny : 0;
            const bool aligned = plane > 0 && own0 % plane == 0 && own1 % plane == 0;
            SH_LOCAL(S, dia_classify(c, m, op, aligned ? (int)(own0 / plane) : -1, aligned ? (int)(own1 / plane) : -1));
        }
        if (!S.poisoned()) guard.x_scaled = true;
        SH_LOCAL(S, vec_div_mul(c, xd, scp, n, 0));                      // x~ = x / sc on owned and ghost rows alike
    }
    PGD_TRY(sh_halo_begin(S, xh, xd, own0, own1, lo_g, hi_g, false, "halo exchange of the start vector"));
    if (scaled) SH_LOCAL(S, launch_spmv_op(c, m, op, xd, qd, nullptr, own0, own1, false, true, nullptr, nullptr));   // the scaled slots
    else SH_LOCAL(S, pgd_spmv(h, oh, xh, q, own0, own1));
    if (scaled) SH_LOCAL(S, cg_init_s(c, b->d, qd, scp, rd, pd, sd, own0, own1, B));
    else SH_LOCAL(S, pgd_cg_init_slot(h, bh, q, dinv, r, u, p, s, own0, own1, B));
    const pgd_handle mv = ss ? p : scaled ? r : u;                   // the vector the product is applied to
    double *mvd = get_vec(c, mv)->d;
    const int gvec = grid_for((own1 - own0 + 1) / 2);                // workgroups (= partial-sum pairs) of k_pcg1_update
    if (ss) {
        // textbook start p = r; the local (r~.r~, true r.r) of the initial residual become the first "previous update" sums
        if (own1 > own0 && !S.poisoned())
            S.local(hipMemcpyAsync(pd + own0, rd + own0, (size_t)(own1 - own0) * sizeof(double), hipMemcpyDeviceToDevice, c->stream) == hipSuccess
                        ? PGD_OK : fail(c, PGD_ERR_HIP, "pcg_solve_sharded: copy of the start direction failed"));
        SH_LOCAL(S, pcg1_seed(c, MAX_VEC_BLOCKS, B, B + 1));
        // (slot B + 7 rides along: the sum of s_i^16 over the owned rows, from which every rank forms the same lower bound of the
        // smallest diagonal entry - k_pcg1_tol - for the switch into the exact phase)
        if (c->pcg_exact_phase) SH_LOCAL(S, pcg1_aux(c, scp, nullptr, own0, own1, B + 7));
        PGD_TRY(sh_allreduce(S, B, 9, "all-reduce of the initial residual"));
        fault_at(3);
        SH_LOCAL(S, pcg1_tol(c, B, rtol, atol, c->pcg_exact_phase ? B + 7 : -1));
    } else {
        PGD_TRY(product(mv, mvd, wd, false, nullptr));
        PGD_TRY(sh_allreduce(S, B, 9, "all-reduce of the initial residual"));
        fault_at(3);
        SH_LOCAL(S, pgd_cg_scalars_slot(h, B, 1, rtol, atol));
        if (scaled && !S.poisoned()) {     // the folded form reads "previous alpha, previous r.r" from the slot set of its parity: seed set 0
            (void)hipMemcpyAsync(c->slots + B + 9, c->slots + B + 5, sizeof(double), hipMemcpyDeviceToDevice, c->stream);
            (void)hipMemcpyAsync(c->slots + B + 10, c->slots + B + 7, sizeof(double), hipMemcpyDeviceToDevice, c->stream);
        }
    }
    // an empty slab has no update launch to fold the scalar step into; every rank must take the same form (collective-free: the
    // decision depends on nothing rank-local but that, and a rank without rows runs k_pcg1_finish for its own books instead)
    const bool fold = ss && c->pcg_fold_finish != 0 && own1 > own0;
    // the direct halo inside the update launch (every rank decides from its own slab: a rank that cannot sends from k_halo_push, the
    // sequence numbers advance alike)
    PushArgs push_proto;
    bool push_fused = false;
    if (pushing && fold && c->push_in_update && (lo_g || hi_g)) {
        const PushTargets T = push_targets(k);
        const bool even = !(own0 & 1) && !((own1 - own0) & 1) && !(T.n_lo & 1) && !(T.n_hi & 1) && T.n_lo <= own1 - own0 && T.n_hi <= own1 - own0;
        if (even) {
            push_proto.dst_lo = T.dst_lo; push_proto.dst_hi = T.dst_hi;
            push_proto.lo_end = own0 + T.n_lo; push_proto.hi_begin = own1 - T.n_hi;
            push_proto.post_lo = T.post_lo; push_proto.post_hi = T.post_hi; push_proto.wait_a = T.wait_a; push_proto.wait_b = T.wait_b;
            push_proto.ticket = k.push_flags + 2;
            push_proto.ticks = push_ticks(k);
            push_proto.nblocks = pcg1_update_push_blocks(gvec, own0, own1, push_proto.lo_end, push_proto.hi_begin);
            push_fused = push_proto.nblocks > 0;
        }
    }
    // One iteration.  Returns an error only for what ends the protocol.
    auto iterate = [&](int kidx) -> int {
        S.iter = kidx;
        c->prof_iter = kidx;
        if (c->fault_iteration >= 0 && kidx == c->fault_iteration && !S.poisoned()) {      // tests: a rank-local failure in mid-solve
            c->fault_iteration = -1;
            S.local(fail(c, PGD_ERR_HIP, "pcg_solve_sharded: injected fault in iteration %d (PGD_TUNE_FAULT_ITERATION)", kidx));
        }
        if (ss) {
            // product (p.q and q.q partial sums) -> local sums -> ONE all-reduce -> stop test, alpha, beta -> x, r, p update
            int np = 0, nb = 0;
            c->spmv_qq = 1;
            const int rc = product(mv, pd, qd, true, &np);
            c->spmv_qq = 0;
            PGD_TRY(rc);
            if (direct_ar) {
                mark(4);
                S.last = "direct all-reduce of the iteration";
                S.ncoll += 1;
                PGD_TRY(comm_allreduce_direct(c, np, gvec, B, S.poisoned()));
            } else {
                SH_LOCAL(S, pcg1_sums(c, np, gvec, B));
                mark(4);
                PGD_TRY(sh_allreduce(S, B, 5, "all-reduce of the iteration"));
            }
            mark(5);
            // stop test, alpha, beta: by every workgroup of the update (fold), or by k_pcg1_finish in a launch of its own
            if (!fold) SH_LOCAL(S, pcg1_finish_slots(c, B));
            // (x is updated every other iteration, two terms at a time: k_pcg1_update; every rank reads the same beta)
            if (push_fused && !S.poisoned()) {
                PushArgs P = push_proto;
                P.seq = k.push_seq + 1;
                const int rc_u = pcg1_update(c, xd, rd, pd, qd, scp, own0, own1, B, &nb, c->pcg_lag_x ? 1 + (kidx & 1) : 0, kidx & 1, &P);
                if (rc_u == PGD_OK) { k.push_seq += 1; p_sent = true; }
                else S.local(rc_u);                                  // (nothing was launched: the next product sends the planes itself)
            } else
            SH_LOCAL(S, pcg1_update(c, xd, rd, pd, qd, scp, own0, own1, B, &nb, c->pcg_lag_x ? 1 + (kidx & 1) : 0, fold ? (kidx & 1) : -1));
            mark(6);
            return PGD_OK;
        }
        if (scaled && kidx > 0) {
            // 3 kernels per iteration: vector step (forms alpha / beta itself, counts, tests), product, one reduction
            int nb = 0, np = 0;
            SH_LOCAL(S, cg_update_s2(c, xd, rd, wd, pd, sd, scp, own0, own1, B, (kidx - 1) & 1, &nb));
            PGD_TRY(product(mv, rd, wd, true, &np));
            SH_LOCAL(S, reduce_two_slots(c, nb, np, B));
        } else {
            if (scaled) SH_LOCAL(S, cg_update_s(c, xd, rd, wd, pd, sd, scp, own0, own1, B));
            else SH_LOCAL(S, pgd_cg_update_slot(h, xh, r, u, w, p, s, dinv, own0, own1, B));
            PGD_TRY(product(mv, mvd, wd, false, nullptr));
        }
        mark(4);
        PGD_TRY(sh_allreduce(S, B, 5, "all-reduce of the iteration"));
        mark(5);
        if (!(scaled && kidx > 0) && !scaled) SH_LOCAL(S, pgd_cg_scalars_slot(h, B, 0, rtol, atol));
        mark(6);
        return PGD_OK;
    };

    // ---- the loop, PIPELINED: the next chunk is queued before the host waits for the agreement behind the previous one (the GPU
    // does not idle through the host's round trip).  Every rank queues exactly the same chunks: what is read at a boundary is
    // either all-reduced (the vote) or a function of all-reduced numbers (the flags), and a chunk queued behind the converged
    // iteration consists of no-op launches and matched collectives.
    int f[4] = {0, 0, 0, 0};
    int enq = 0, cur = 0, nchunks = 0;
    if (c->fault_stall_ms > 0) {      // tests: a stream that stops making progress
        const long long ticks = (long long)std::min(c->fault_stall_ms, 20000) * 100000LL;      // wall_clock64 ticks at 100 MHz
        c->fault_stall_ms = 0;
        k_comm_stall<<<1, 1, 0, c->stream>>>(ticks);
    }
    const double dbg_t2 = dbg_now();
    PGD_TRY(agree_begin(0, "agreement before the first chunk"));
    for (;;) {
        const int chunk = std::min(CHECK, maxit - enq);
        if (chunk > 0) {
            const int set = nchunks & 1;
            for (int j = 0; j < chunk; ++j) {
                const bool sample = k.prof && j == 5 + set && !S.poisoned();      // both parities of the iteration index: x is updated in every other one
                marks = sample ? k.mark[set] : nullptr;
                PGD_TRY(iterate(enq + j));
                if (sample) k.mark_set[set] = true;
                marks = nullptr;
            }
            enq += chunk;
            nchunks += 1;
            PGD_TRY(agree_begin(cur ^ 1, "agreement after a chunk of iterations"));
        }
        PGD_TRY(agree_end(cur, cur == 0 && nchunks <= 1 ? "the agreement before the first chunk" : "the agreement after a chunk", f));
        if (chunk > 0 && nchunks >= 2) prof_collect(c, nchunks & 1);      // the chunk before the one just queued has drained
        if (f[0] || chunk <= 0) break;                // converged (what is queued behind it does nothing), or nothing more to queue
        cur ^= 1;
    }
    c->prof_iter = -1;
    const double dbg_t3 = dbg_now();
    if (c->prof) prof_commit(c, f[1] + (ss ? 1 : 0), f[1]);      // samples of launches behind the converged iteration are dropped
    const int32_t done = f[0], it = f[1], status = f[2];
    if (scaled && !ss && !done && enq > 0) {
        // the last queued iteration's scalars are still unprocessed in the folded form: count and test them (same on every rank)
        SH_LOCAL(S, pgd_cg_scalars_slot(h, B, 0, rtol, atol));
    }
    fault_at(4);
    // ---- closing agreement: the true r.r of the final residual for the report (a solve that stops at maxit may never have
    // entered its exact phase) rides with the vote; whatever failed locally since the last boundary comes out here
    if (ss && c->pcg_exact_phase) SH_LOCAL(S, pcg1_aux(c, scp, rd, own0, own1, B));
    PGD_TRY(sh_allreduce(S, V, 2, "closing agreement"));
    if (ss && c->pcg_exact_phase && !S.poisoned())
        (void)hipMemcpyAsync(c->slots + 6, c->slots + B, sizeof(double), hipMemcpyDeviceToDevice, c->stream);
    PGD_HIP(c, hipMemcpyAsync(k.snap_flags, c->flags, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipMemcpyAsync(k.snap_vote, c->slots + V, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipEventRecord(k.snap_ev[0], c->stream));
    int f2[4];
    PGD_TRY(agree_end(0, "the closing agreement", f2));
    prof_collect(c, 0);
    prof_collect(c, 1);
    const int32_t it2 = f2[1], status2 = f2[2];
    (void)it;
    if (status == PGD_ERR_TIMEOUT || status2 == PGD_ERR_TIMEOUT)
        return fail(c, PGD_ERR_TIMEOUT, "pcg_solve_sharded: rank %d/%d: a neighbour's boundary planes did not arrive within the deadline (direct halo, "
                    "iteration %d)", k.rank, k.world, it2);
    if (status != 0 || status2 != 0) return fail(c, PGD_ERR_SINGULAR, "sharded PCG breakdown (NaN) after %d iterations", it2);
    // ---- everybody is healthy and has left the loop at the same iteration
    if (ss && c->pcg_lag_x && it2 > 0 && ((it2 - 1) & 1) == 0)          // the last update had an even index: its term of x may be outstanding
        PGD_TRY(pcg1_flush_x(c, xd, pd, rd, own0, own1, B, fold ? ((it2 - 1) & 1) : -1));
    if (scaled) {
        guard.x_scaled = false;
        PGD_TRY(vec_div_mul(c, xd, scp, n, 1));                      // back to x = sc x~ (ghosts too; refreshed below)
    }
    double sl[40];
    PGD_TRY(pgd_slots_download(h, sl, 0, 40));
    PGD_TRY(comm_halo(c, xh, x->d, own0, own1, lo_g, hi_g));     // the caller's x: ghosts current
    if (iters) *iters = it2;
    if (dbg_t) {
        const double dbg_t4 = dbg_now();
        fprintf(stderr, "[pcg_solve_sharded] n %lld rows %lld iterations %d queued %d | vote %.2f ms, setup %.2f ms, loop %.2f ms = %.1f us/it, end %.2f ms\n",
                (long long)n, (long long)(own1 - own0), it2, enq, 1e3 * (dbg_t1 - dbg_t0), 1e3 * (dbg_t2 - dbg_t1), 1e3 * (dbg_t3 - dbg_t2),
                1e6 * (dbg_t3 - dbg_t2) / (it2 > 0 ? it2 : 1), 1e3 * (dbg_t4 - dbg_t3));
    }
    const double bb = sl[B + 8], rr = sl[6];
    if (rel) *rel = bb > 0.0 ? sqrt(rr / bb) : 0.0;
    return PGD_OK;
}

}  // extern "C"
