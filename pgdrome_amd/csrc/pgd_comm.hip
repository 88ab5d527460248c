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

void comm_release(Ctx *c) {
    Comm &k = c->comm;
    if (k.kind == 2 && k.nccl) {
        std::string why;
        if (RcclApi *api = rccl_api(why)) (void)api->CommDestroy(k.nccl);
    }
    k = Comm();
}

// neighbour planes -> ghost planes of v (local numbering: [0, lo_g) ghost below, [own0, own1) owned,
// [own1, own1 + hi_g) ghost above); stream-ordered
static int comm_halo(Ctx *c, pgd_handle vh, double *v, int64_t own0, int64_t own1, int64_t lo_g, int64_t hi_g) {
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
    PGD_NCCL(c, api, api->GroupStart());
    if (lo_g) {
        PGD_NCCL(c, api, api->Send(v + own0, (size_t)lo_g, NCCL_F64, k.rank - 1, k.nccl, c->stream));
        PGD_NCCL(c, api, api->Recv(v, (size_t)lo_g, NCCL_F64, k.rank - 1, k.nccl, c->stream));
    }
    if (hi_g) {
        PGD_NCCL(c, api, api->Send(v + own1 - hi_g, (size_t)hi_g, NCCL_F64, k.rank + 1, k.nccl, c->stream));
        PGD_NCCL(c, api, api->Recv(v + own1, (size_t)hi_g, NCCL_F64, k.rank + 1, k.nccl, c->stream));
    }
    PGD_NCCL(c, api, api->GroupEnd());
    return PGD_OK;
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

}  // namespace pgd

using namespace pgd;

extern "C" {

int pgd_comm_bind_callbacks(pgd_handle h, pgd_halo_fn halo, pgd_allreduce_fn allreduce, void *user, int rank, int world) {
    PGD_CTX(c, h);
    if (!halo || !allreduce || world < 1 || rank < 0 || rank >= world) return fail(c, PGD_ERR_INVALID, "comm_bind_callbacks: invalid arguments");
    comm_release(c);
    c->comm.kind = 1; c->comm.rank = rank; c->comm.world = world;
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
    const int rc = comm_selftest(c);
    if (rc != PGD_OK) { const std::string keep = c->err; comm_release(c); c->err = keep; }
    return rc;
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
    if ((lo_g > 0) != (k.rank > 0) || (hi_g > 0) != (k.rank < k.world - 1))
        return fail(c, PGD_ERR_INVALID, "pcg_solve_sharded: ghost planes do not match the rank's position");
    if (k.work_n != n) {      // r, u, w, p, s, q, dinv as library vectors (the slot kernels take handles)
        for (pgd_handle &wh : k.work) { if (wh) (void)pgd_vec_free(h, wh); wh = 0; }
        for (pgd_handle &wh : k.work) PGD_TRY(pgd_vec_alloc(h, n, &wh));
        k.work_n = n;
    }
    const pgd_handle r = k.work[0], u = k.work[1], w = k.work[2], p = k.work[3], s = k.work[4], q = k.work[5], dinv = k.work[6];
    constexpr int B = 24, CHECK = 16;       // slot base of the recurrence (pgdrome_amd/dist.py uses the same)
    // The exchange is stream-ordered before the product, so the owned rows go in ONE launch; the slots of the
    // two boundary-row partials (kept for the overlapped host-driven variant) stay zero.
    auto spmv_dot3 = [&](pgd_handle uu, pgd_handle ww) -> int {
        PGD_TRY(comm_halo(c, uu, get_vec(c, uu)->d, own0, own1, lo_g, hi_g));
        return pgd_spmv_dot_slot(h, oh, uu, ww, uu, own0, own1, B + 2);
    };
    Mesh *m = get_mesh(c, op->mesh);
    bool sym = false;
    if (m) PGD_TRY(ensure_sym(c, m, op, &sym));
    PGD_TRY(pgd_flags_reset(h));
    const double zeros[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    PGD_TRY(pgd_slots_upload(h, zeros, B, 9));
    PGD_TRY(pgd_op_diag_inv(h, oh, dinv));
    // With the symmetric storage in place the recurrence runs on the diagonally scaled system (k_cg_update_s: u = r, no
    // dinv / u passes).  sc = d^-1/2 of the ghost rows comes from their owners (their local diagonals are partial sums).
    const bool scaled = sym && c->pcg_scaled;
    double *scp = get_vec(c, dinv)->d;
    double *xd = x->d, *rd = get_vec(c, r)->d, *wd = get_vec(c, w)->d, *pd = get_vec(c, p)->d, *sd = get_vec(c, s)->d,
           *qd = get_vec(c, q)->d;
    if (scaled) {
        PGD_TRY(vec_sqrt(c, scp, n));
        PGD_TRY(comm_halo(c, dinv, scp, own0, own1, lo_g, hi_g));
        PGD_TRY(sym_scale(c, m, op, scp));
        PGD_TRY(vec_div_mul(c, xd, scp, n, 0));                      // x~ = x / sc on owned and ghost rows alike
    }
    PGD_TRY(comm_halo(c, xh, xd, own0, own1, lo_g, hi_g));
    if (scaled) PGD_TRY(launch_spmv_op(c, m, op, xd, qd, nullptr, own0, own1, false, true, nullptr, nullptr));   // the scaled slots
    else PGD_TRY(pgd_spmv(h, oh, xh, q, own0, own1));
    if (scaled) PGD_TRY(cg_init_s(c, b->d, qd, scp, rd, pd, sd, own0, own1, B));
    else PGD_TRY(pgd_cg_init_slot(h, bh, q, dinv, r, u, p, s, own0, own1, B));
    const pgd_handle mv = scaled ? r : u;                            // the vector the product is applied to
    PGD_TRY(spmv_dot3(mv, w));
    PGD_TRY(comm_allreduce(c, B, 9));
    PGD_TRY(pgd_cg_scalars_slot(h, B, 1, rtol, atol));
    if (scaled) {     // the folded form reads "previous alpha, previous r.r" from the slot set of its parity: seed set 0
        PGD_HIP(c, hipMemcpyAsync(c->slots + B + 9, c->slots + B + 5, sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        PGD_HIP(c, hipMemcpyAsync(c->slots + B + 10, c->slots + B + 7, sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    int32_t done = 0, it = 0, status = 0;
    int kk = 0;
    for (;;) {
        PGD_TRY(pgd_flags_download(h, &done, &it, &status));     // the only host synchronisation of the loop
        if (done || kk >= maxit) break;
        const int chunk = std::min(CHECK, maxit - kk);
        for (int j = 0; j < chunk; ++j, ++kk) {
            if (scaled && kk > 0) {
                // 3 kernels per iteration: vector step (forms alpha / beta itself, counts, tests), product, one reduction
                int nb = 0, np = 0;
                PGD_TRY(cg_update_s2(c, xd, rd, wd, pd, sd, scp, own0, own1, B, (kk - 1) & 1, &nb));
                PGD_TRY(comm_halo(c, mv, rd, own0, own1, lo_g, hi_g));
                PGD_TRY(launch_spmv_op(c, m, op, rd, wd, rd, own0, own1, true, true, c->flags, &np));
                PGD_TRY(reduce_two_slots(c, nb, np, B));
                PGD_TRY(comm_allreduce(c, B, 5));
                continue;
            }
            if (scaled) PGD_TRY(cg_update_s(c, xd, rd, wd, pd, sd, scp, own0, own1, B));
            else PGD_TRY(pgd_cg_update_slot(h, xh, r, u, w, p, s, dinv, own0, own1, B));
            PGD_TRY(spmv_dot3(mv, w));
            PGD_TRY(comm_allreduce(c, B, 5));
            if (!scaled) PGD_TRY(pgd_cg_scalars_slot(h, B, 0, rtol, atol));
        }
    }
    if (scaled && !done && kk > 0) {
        // the last enqueued iteration's scalars are still unprocessed in the folded form: count and test them
        PGD_TRY(pgd_cg_scalars_slot(h, B, 0, rtol, atol));
        PGD_TRY(pgd_flags_download(h, &done, &it, &status));
    }
    if (scaled) {
        PGD_TRY(vec_div_mul(c, xd, scp, n, 1));                      // back to x = sc x~ (ghosts too; refreshed below)
        op->uvals_valid = false;                                     // the slot arrays hold the scaled operator
        op->uvals_scaled = false;
    }
    if (status != 0) return fail(c, PGD_ERR_SINGULAR, "sharded PCG breakdown (NaN) after %d iterations", it);
    double sl[40];
    PGD_TRY(pgd_slots_download(h, sl, 0, 40));
    PGD_TRY(comm_halo(c, xh, x->d, own0, own1, lo_g, hi_g));     // the caller's x: ghosts current
    if (iters) *iters = it;
    const double bb = sl[B + 8], rr = sl[6];
    if (rel) *rel = bb > 0.0 ? sqrt(rr / bb) : 0.0;
    return PGD_OK;
}

}  // extern "C"
