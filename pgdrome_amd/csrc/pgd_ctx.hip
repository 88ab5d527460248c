// Context, handle table, device vectors: the plumbing of libpgd_amd.so.
#include <algorithm>
#include <cstdarg>
#include <cstring>
#include <mutex>

#include "pgd_internal.h"

namespace pgd {

static std::mutex g_mu;
static std::vector<std::unique_ptr<Ctx>> g_ctx;   // handle = index + 1
static std::string g_global_err;

Ctx *get_ctx(pgd_handle h) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (h < 1 || h > (pgd_handle)g_ctx.size()) return nullptr;
    return g_ctx[h - 1].get();
}

int fail(Ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_global_err = buf;
    return code;
}

pgd_handle put_obj(Ctx *c, Obj *o) {
    o->ctx = c;
    o->serial = c->next_serial++;
    if (!c->free_list.empty()) {
        int64_t i = c->free_list.back();
        c->free_list.pop_back();
        c->objs[i].reset(o);
        return i + 1;
    }
    c->objs.emplace_back(o);
    return (pgd_handle)c->objs.size();
}

Obj *get_obj(Ctx *c, pgd_handle h, Obj::Kind k) {
    if (h < 1 || h > (pgd_handle)c->objs.size()) return nullptr;
    Obj *o = c->objs[h - 1].get();
    if (!o || o->kind != k) return nullptr;
    return o;
}

int free_obj(Ctx *c, pgd_handle h, Obj::Kind k) {
    if (!get_obj(c, h, k)) return fail(c, PGD_ERR_INVALID, "free: invalid handle %lld", (long long)h);
    if (k == Obj::MESH) (void)hipStreamSynchronize(c->stream);   // mesh arrays go straight back to HIP
    c->objs[h - 1].reset();
    c->free_list.push_back(h - 1);
    return PGD_OK;
}

void dev_release(Ctx *c, void *p, size_t bytes) {
    if (!p) return;
    if (c && bytes >= ((size_t)1 << 16) && c->pool_bytes + bytes <= Ctx::POOL_MAX) {
        c->pool.emplace(bytes, p);
        c->pool_bytes += bytes;
        return;
    }
    (void)hipFree(p);
}

int dev_alloc(Ctx *c, void **p, size_t bytes) {
    *p = nullptr;
    if (c) {
        auto it = c->pool.find(bytes);
        if (it != c->pool.end()) {
            *p = it->second;
            c->pool_bytes -= bytes;
            c->pool.erase(it);
            return PGD_OK;
        }
    }
    hipError_t e = hipMalloc(p, bytes + PAD_BYTES);
    if (e != hipSuccess && c && !c->pool.empty()) {      // out of memory: give the pooled buffers back and retry
        (void)hipStreamSynchronize(c->stream);
        for (auto &kv : c->pool) (void)hipFree(kv.second);
        c->pool.clear();
        c->pool_bytes = 0;
        e = hipMalloc(p, bytes + PAD_BYTES);
    }
    if (e != hipSuccess) {
        *p = nullptr;
        return fail(c, PGD_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    return PGD_OK;
}

template <class T>
static int ensure_buf(Ctx *c, T **p, int64_t *cap, int64_t n) {
    if (*cap >= n) return PGD_OK;
    (void)hipStreamSynchronize(c->stream);
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    void *q;
    PGD_TRY(dev_alloc(c, &q, (size_t)n * sizeof(T)));
    *p = (T *)q;
    *cap = n;
    return PGD_OK;
}

// The reduction scratch may be grown BETWEEN the launches of one reduction (several row ranges leaving their partial sums side by
// side, c->partials_off > 0): what the earlier launches wrote must survive the move.
int ensure_partials(Ctx *c, int64_t n) {
    if (c->partials_cap >= n) return PGD_OK;
    if (c->partials_off <= 0 || !c->partials) return ensure_buf(c, &c->partials, &c->partials_cap, n);
    void *q = nullptr;
    PGD_TRY(dev_alloc(c, &q, (size_t)n * sizeof(double)));
    const int64_t keep = std::min<int64_t>(c->partials_off, c->partials_cap);
    hipError_t e = hipMemcpyAsync(q, c->partials, (size_t)keep * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipFree(q); return fail(c, PGD_ERR_HIP, "ensure_partials: %s", hipGetErrorString(e)); }
    (void)hipFree(c->partials);
    c->partials = (double *)q;
    c->partials_cap = n;
    return PGD_OK;
}
int ensure_work(Ctx *c, int i, int64_t n) { return ensure_buf(c, &c->work[i], &c->work_cap[i], n); }
int pcg_flag_snapshots(Ctx *c) {
    if (!c->flags_host) {
        void *p = nullptr;
        PGD_HIP(c, hipHostMalloc(&p, 8 * sizeof(int), hipHostMallocDefault));
        c->flags_host = (int *)p;
    }
    for (hipEvent_t &e : c->flag_ev)
        if (!e) PGD_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return PGD_OK;
}

int ensure_mask(Ctx *c, int64_t n) { return ensure_buf(c, &c->mask, &c->mask_cap, n); }
int ensure_ibuf(Ctx *c, int64_t n) { return ensure_buf(c, &c->ibuf, &c->ibuf_cap, n); }

static void prof_keep(Ctx *c, const Ctx::ProfRec &r) {
    if (r.kind) { c->prof_upd_launches += 1; c->prof_upd_seconds += r.seconds; c->prof_upd_bytes += r.bytes; }
    else { c->prof_launches += 1; c->prof_seconds += r.seconds; c->prof_bytes += r.bytes; c->prof_own_bytes += r.own; }
}

void prof_flush(Ctx *c) {
    if (c->ev_used == 0) return;
    (void)hipStreamSynchronize(c->stream);
    for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]) != hipSuccess) continue;
        Ctx::ProfRec r = c->ev_rec[i / 2];
        r.seconds = 1e-3 * ms;
        if (r.iter < 0) prof_keep(c, r); else c->prof_pend.push_back(r);
    }
    c->ev_used = 0;
}

void prof_commit(Ctx *c, int valid_products, int valid_updates) {
    prof_flush(c);
    for (const Ctx::ProfRec &r : c->prof_pend) {
        if (r.iter < (r.kind ? valid_updates : valid_products)) prof_keep(c, r);
        else c->prof_dropped += 1;
    }
    c->prof_pend.clear();
}

}  // namespace pgd

using namespace pgd;

extern "C" {

int pgd_version(void) { return 100; }

int pgd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pgd_ctx_create(int device, void *stream, pgd_handle *out) {
    if (!out) return PGD_ERR_INVALID;
    *out = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, PGD_ERR_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(nullptr, PGD_ERR_INVALID, "device %d out of range", device);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, PGD_ERR_HIP, "hipSetDevice failed");
    std::unique_ptr<Ctx> c(new Ctx);
    c->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
            c->num_cu = prop.multiProcessorCount;
        }
    }
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
            return fail(nullptr, PGD_ERR_HIP, "hipStreamCreate failed");
        c->own_stream = true;
    }
    void *p;
    if (dev_alloc(c.get(), &p, PGD_NSLOTS * sizeof(double)) != PGD_OK) return PGD_ERR_NOMEM;
    c->slots = (double *)p;
    if (dev_alloc(c.get(), &p, 8 * sizeof(int)) != PGD_OK) return PGD_ERR_NOMEM;
    c->flags = (int *)p;
    (void)hipMemsetAsync(c->slots, 0, PGD_NSLOTS * sizeof(double), c->stream);
    (void)hipMemsetAsync(c->flags, 0, 8 * sizeof(int), c->stream);
    (void)hipStreamSynchronize(c->stream);
    std::lock_guard<std::mutex> lk(g_mu);
    g_ctx.emplace_back(std::move(c));
    *out = (pgd_handle)g_ctx.size();
    return PGD_OK;
}

int pgd_ctx_destroy(pgd_handle h) {
    Ctx *c = get_ctx(h);
    if (!c) return PGD_ERR_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    comm_release(c);
    mg_release(c);
    c->objs.clear();
    for (auto &kv : c->pool) (void)hipFree(kv.second);
    c->pool.clear();
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->timer_ev) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->flag_ev) if (e) (void)hipEventDestroy(e);
    if (c->flags_host) (void)hipHostFree(c->flags_host);
    if (c->cls_scratch) (void)hipFree(c->cls_scratch);
    if (c->gram_w) (void)hipFree(c->gram_w);
    for (void *p : {(void *)c->slots, (void *)c->flags, (void *)c->partials, (void *)c->mask,
                    (void *)c->ibuf, (void *)c->set_idx, (void *)c->set_vals})
        if (p) (void)hipFree(p);
    for (double *w : c->work)
        if (w) (void)hipFree(w);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    std::lock_guard<std::mutex> lk(g_mu);
    g_ctx[h - 1].reset();
    return PGD_OK;
}

int pgd_sync(pgd_handle h) {
    PGD_CTX(c, h);
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    return PGD_OK;
}

const char *pgd_last_error(pgd_handle h) {
    Ctx *c = get_ctx(h);
    return c ? c->err.c_str() : g_global_err.c_str();
}

// ------------------------------------------------------------------------- vectors
int pgd_vec_alloc(pgd_handle h, int64_t n, pgd_handle *out) {
    PGD_CTX(c, h);
    if (!out || n < 0) return fail(c, PGD_ERR_INVALID, "vec_alloc: bad arguments");
    std::unique_ptr<Vec> v(new Vec);
    v->kind = Obj::VEC;
    v->n = n;
    void *p;
    PGD_TRY(dev_alloc(c, &p, (size_t)(n > 0 ? n : 1) * sizeof(double)));
    v->d = (double *)p;
    PGD_HIP(c, hipMemsetAsync(v->d, 0, (size_t)(n > 0 ? n : 1) * sizeof(double) + PAD_BYTES, c->stream));
    *out = put_obj(c, v.release());
    return PGD_OK;
}

int pgd_vec_free(pgd_handle h, pgd_handle v) {
    PGD_CTX(c, h);
    return free_obj(c, v, Obj::VEC);
}

int pgd_vec_size(pgd_handle h, pgd_handle vh, int64_t *n) {
    PGD_CTX(c, h);
    Vec *v = get_vec(c, vh);
    if (!v || !n) return fail(c, PGD_ERR_INVALID, "vec_size: invalid handle");
    *n = v->n;
    return PGD_OK;
}

int pgd_vec_upload(pgd_handle h, pgd_handle vh, const double *host, int64_t off, int64_t cnt) {
    PGD_CTX(c, h);
    Vec *v = get_vec(c, vh);
    if (!v || !host || off < 0 || cnt < 0 || off + cnt > v->n)
        return fail(c, PGD_ERR_INVALID, "vec_upload: invalid handle or range");
    if (cnt == 0) return PGD_OK;
    PGD_HIP(c, hipMemcpyAsync(v->d + off, host, (size_t)cnt * sizeof(double), hipMemcpyHostToDevice, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));   // host buffer is caller-owned: copy before returning
    return PGD_OK;
}

int pgd_vec_download(pgd_handle h, pgd_handle vh, double *host, int64_t off, int64_t cnt) {
    PGD_CTX(c, h);
    Vec *v = get_vec(c, vh);
    if (!v || !host || off < 0 || cnt < 0 || off + cnt > v->n)
        return fail(c, PGD_ERR_INVALID, "vec_download: invalid handle or range");
    if (cnt == 0) return PGD_OK;
    PGD_HIP(c, hipMemcpyAsync(host, v->d + off, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    return PGD_OK;
}

int pgd_vec_ptr(pgd_handle h, pgd_handle vh, void **p) {
    PGD_CTX(c, h);
    Vec *v = get_vec(c, vh);
    if (!v || !p) return fail(c, PGD_ERR_INVALID, "vec_ptr: invalid handle");
    *p = v->d;
    return PGD_OK;
}

int pgd_vec_copy(pgd_handle h, pgd_handle dh, pgd_handle sh) {
    PGD_CTX(c, h);
    Vec *d = get_vec(c, dh), *s = get_vec(c, sh);
    if (!d || !s || d->n != s->n) return fail(c, PGD_ERR_INVALID, "vec_copy: invalid handles or size mismatch");
    if (d->n == 0 || d == s) return PGD_OK;
    PGD_HIP(c, hipMemcpyAsync(d->d, s->d, (size_t)d->n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return PGD_OK;
}

// ------------------------------------------------------------------ scalar bank
int pgd_slots_ptr(pgd_handle h, void **p) {
    PGD_CTX(c, h);
    if (!p) return PGD_ERR_INVALID;
    *p = c->slots;
    return PGD_OK;
}

int pgd_slots_download(pgd_handle h, double *out, int first, int count) {
    PGD_CTX(c, h);
    if (!out || first < 0 || count < 0 || first + count > PGD_NSLOTS)
        return fail(c, PGD_ERR_INVALID, "slots_download: bad range");
    PGD_HIP(c, hipMemcpyAsync(out, c->slots + first, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    return PGD_OK;
}

int pgd_slots_upload(pgd_handle h, const double *in, int first, int count) {
    PGD_CTX(c, h);
    if (!in || first < 0 || count < 0 || first + count > PGD_NSLOTS)
        return fail(c, PGD_ERR_INVALID, "slots_upload: bad range");
    PGD_HIP(c, hipMemcpyAsync(c->slots + first, in, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    return PGD_OK;
}

int pgd_flags_reset(pgd_handle h) {
    PGD_CTX(c, h);
    PGD_HIP(c, hipMemsetAsync(c->flags, 0, 8 * sizeof(int), c->stream));
    return PGD_OK;
}

int pgd_flags_download(pgd_handle h, int32_t *done, int32_t *iters, int32_t *status) {
    PGD_CTX(c, h);
    int f[4];
    PGD_HIP(c, hipMemcpyAsync(f, c->flags, sizeof f, hipMemcpyDeviceToHost, c->stream));
    PGD_HIP(c, hipStreamSynchronize(c->stream));
    if (done) *done = f[0];
    if (iters) *iters = f[1];
    if (status) *status = f[2];
    return PGD_OK;
}

// -------------------------------------------------------------------- profiling
// What a pair of HIP events adds to the kernel it brackets: t(n kernels between one pair) = overhead + n * kernel, so
// overhead = 2 t(1) - t(2) (medians of 15 samples, a kernel of a few microseconds on the slot bank's scratch).  Measured once per
// context; the bench subtracts it from its event timings so that they are the kernels' durations (what rocprofv3 reports).
__global__ void k_prof_calib(double *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

static void prof_calibrate(Ctx *c) {
    if (c->prof_overhead >= 0.0 || c->ev.size() < 4) return;
    c->prof_overhead = 0.0;
    if (ensure_work(c, 5, 1 << 16) != PGD_OK) return;
    double med[2] = {0.0, 0.0};
    for (int n = 1; n <= 2; ++n) {
        std::vector<float> t;
        for (int rep = 0; rep < 15; ++rep) {
            if (hipEventRecord(c->ev[0], c->stream) != hipSuccess) return;
            for (int k = 0; k < n; ++k) k_prof_calib<<<256, 256, 0, c->stream>>>(c->work[5], 1 << 16);
            if (hipEventRecord(c->ev[1], c->stream) != hipSuccess || hipEventSynchronize(c->ev[1]) != hipSuccess) return;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) t.push_back(ms);
        }
        if (t.size() < 5) return;
        std::sort(t.begin(), t.end());
        med[n - 1] = 1e-3 * t[t.size() / 2];
    }
    const double ov = 2.0 * med[0] - med[1];
    c->prof_overhead = ov > 0.0 && ov < 20e-6 ? ov : 0.0;
}

int pgd_prof_event_overhead(pgd_handle h, double *seconds) {
    PGD_CTX(c, h);
    if (seconds) *seconds = c->prof_overhead > 0.0 ? c->prof_overhead : 0.0;
    return PGD_OK;
}

int pgd_prof_enable(pgd_handle h, int on) {
    PGD_CTX(c, h);
    prof_flush(c);
    c->prof = on != 0;
    c->prof_pcg_only = on == 2;
    if (on) {
        c->prof_launches = 0;
        c->prof_seconds = 0.0;
        c->prof_bytes = 0.0;
        c->prof_own_bytes = 0.0;
        c->prof_upd_launches = 0;
        c->prof_upd_seconds = c->prof_upd_bytes = 0.0;
        c->prof_dropped = 0;
        c->prof_pend.clear();
        if (c->ev.empty()) {
            c->ev_rec.assign(1024, Ctx::ProfRec{0, -1, 0.0, 0.0, 0.0});
            c->ev.resize(2048);
            for (auto &e : c->ev) PGD_HIP(c, hipEventCreate(&e));
        }
        prof_calibrate(c);
    }
    return PGD_OK;
}

int pgd_prof_read(pgd_handle h, int64_t *launches, double *seconds, double *bytes) {
    PGD_CTX(c, h);
    prof_flush(c);
    if (launches) *launches = c->prof_launches;
    if (seconds) *seconds = c->prof_seconds;
    if (bytes) *bytes = c->prof_bytes;
    return PGD_OK;
}

int pgd_prof_read_update(pgd_handle h, int64_t *launches, double *seconds, double *bytes) {
    PGD_CTX(c, h);
    prof_flush(c);
    if (launches) *launches = c->prof_upd_launches;
    if (seconds) *seconds = c->prof_upd_seconds;
    if (bytes) *bytes = c->prof_upd_bytes;
    return PGD_OK;
}

int pgd_prof_read_dropped(pgd_handle h, int64_t *dropped) {
    PGD_CTX(c, h);
    prof_flush(c);
    if (dropped) *dropped = c->prof_dropped;
    return PGD_OK;
}

int pgd_prof_read_own(pgd_handle h, double *own_bytes) {
    PGD_CTX(c, h);
    prof_flush(c);
    if (own_bytes) *own_bytes = c->prof_own_bytes;
    return PGD_OK;
}

int pgd_kernel_counts(pgd_handle h, int64_t *out, int n) {
    PGD_CTX(c, h);
    if (!out || n < 0) return fail(c, PGD_ERR_INVALID, "kernel_counts: invalid arguments");
    for (int k = 0; k < n; ++k) out[k] = k < 8 ? c->kcount[k] : 0;
    return PGD_OK;
}

int pgd_timer_start(pgd_handle h) {
    PGD_CTX(c, h);
    for (auto &e : c->timer_ev)
        if (!e) PGD_HIP(c, hipEventCreate(&e));
    PGD_HIP(c, hipEventRecord(c->timer_ev[0], c->stream));
    return PGD_OK;
}

int pgd_timer_stop(pgd_handle h, double *seconds) {
    PGD_CTX(c, h);
    if (!c->timer_ev[0] || !c->timer_ev[1] || !seconds) return fail(c, PGD_ERR_INVALID, "timer_stop: no timer running");
    PGD_HIP(c, hipEventRecord(c->timer_ev[1], c->stream));
    PGD_HIP(c, hipEventSynchronize(c->timer_ev[1]));
    float ms = 0.f;
    PGD_HIP(c, hipEventElapsedTime(&ms, c->timer_ev[0], c->timer_ev[1]));
    *seconds = 1e-3 * (double)ms;
    return PGD_OK;
}

}  // extern "C"
