// multi.hip.h -- ONE batch of independent problems over several GPUs of this host: no exchange step, problem k lives on
// shard k mod G (interleaved, so that the spread of iteration counts averages out over the devices).
//
//   DAQPMultiBatch  = G device-resident shards (each an ordinary DAQPBatch on its own device and HIP stream) + one persistent host
//                     thread per shard.  create / setup / update / solve / free mirror the single-device calls; the factors, the
//                     working sets and the iterate stay on the devices between calls (config C5's warm sequence, the strong-scaled
//                     config C3 from one C process).
//   inputs/outputs  : either ONE host-resident batch in the caller's order (daqp_batch_*_multi: every shard gathers its problems
//                     k = g, g + G, ... into a pinned buffer, chunk by chunk, two buffers deep, and DMAs them into its own device
//                     slots; results come back the same way and are scattered to their k) -- or per-shard descriptors
//                     (daqp_batch_*_multi_shards: shard g's problems back to back, host- or device-resident ON THAT SHARD'S
//                     DEVICE, used in place).
//   daqp_quadprog_batch_multi = create + setup + solve + free on top of it.
// Included by daqp_amd.hip (host code only).
#pragma once
#include <condition_variable>
#include <functional>
#include <thread>

struct DAQPMultiBatch;

namespace {

constexpr size_t kPinChunk = 8u << 20;      // bytes per pinned staging buffer (two per shard)

struct MultiShard {
    int g = 0, G = 1, device = 0, Ng = 0;
    DAQPBatch *b = nullptr;
    hipStream_t stream = nullptr;
    char *pin[2] = {nullptr, nullptr};
    size_t pin_bytes = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
    int turn = 0;                               // which pinned buffer the next chunk takes
    // worker
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> task;
    bool has_task = false, done = false, stop = false;
    int rc = 0;
    std::string err;
    double secs = 0;
};

void shard_worker(MultiShard *s)
{
    (void)hipSetDevice(s->device);
    for (;;) {
        std::function<int()> t;
        {
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv.wait(lk, [&] { return s->has_task || s->stop; });
            if (s->stop) return;
            t = std::move(s->task);
            s->has_task = false;
        }
        g_err[0] = 0;
        const double t0 = now_s();
        const int rc = t();
        {
            std::lock_guard<std::mutex> lk(s->mu);
            s->rc = rc; s->err = rc ? std::string(g_err) : std::string(); s->secs = now_s() - t0; s->done = true;
        }
        s->cv.notify_all();
    }
}

// the pinned buffer for the next chunk, free again (its previous DMA has run)
int pin_acquire(MultiShard *s, size_t need, char **buf, int *which)
{
    if (s->pin_bytes < need) {
        for (int i = 0; i < 2; ++i) {
            if (s->ev[i]) HIPCHK(hipEventSynchronize(s->ev[i]));
            if (s->pin[i]) { (void)hipHostFree(s->pin[i]); s->pin[i] = nullptr; }
        }
        const size_t sz = need > kPinChunk ? need : kPinChunk;
        for (int i = 0; i < 2; ++i) HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&s->pin[i]), sz, hipHostMallocDefault));
        s->pin_bytes = sz;
    }
    for (int i = 0; i < 2; ++i) if (!s->ev[i]) HIPCHK(hipEventCreateWithFlags(&s->ev[i], hipEventDisableTiming));
    const int w = s->turn;
    s->turn ^= 1;
    HIPCHK(hipEventSynchronize(s->ev[w]));      // (an event never recorded is complete)
    *buf = s->pin[w]; *which = w;
    return 0;
}

// shard s takes rows k = g, g + G, ... of a host array of N rows of `per` elements into `dst` (device, Ng rows back to back).
// One shard: the rows are contiguous, one plain copy.  Several: by default a pitched copy straight out of the caller's memory (the
// runtime gathers the rows while it stages them; measured faster than gathering here: tools/pcie_rate.py); DAQP_AMD_MULTI_PINNED=1:
// gathered into this shard's pinned buffers chunk by chunk, two deep (what to use when the caller's arrays are pinned already or
// the runtime's pitched path is slow).
template <typename T>
int stage_rows_in(MultiShard *s, const T *src, size_t per, T *dst)
{
    if (!src || per == 0 || s->Ng == 0) return 0;
    const size_t row = per * sizeof(T);
    if (s->G == 1) { HIPCHK(hipMemcpyAsync(dst, src, row * (size_t)s->Ng, hipMemcpyHostToDevice, s->stream)); return 0; }
    static const bool pinned = [] { const char *e = getenv("DAQP_AMD_MULTI_PINNED"); return e && atoi(e) != 0; }();
    if (!pinned) {
        HIPCHK(hipMemcpy2DAsync(dst, row, src + (size_t)s->g * per, row * (size_t)s->G, row, (size_t)s->Ng, hipMemcpyHostToDevice, s->stream));
        return 0;
    }
    size_t rows_per = kPinChunk / row;
    if (rows_per == 0) rows_per = 1;
    for (size_t j0 = 0; j0 < (size_t)s->Ng; j0 += rows_per) {
        const size_t cnt = (j0 + rows_per <= (size_t)s->Ng) ? rows_per : (size_t)s->Ng - j0;
        char *buf = nullptr;
        int w = 0;
        if (pin_acquire(s, cnt * row, &buf, &w)) return DAQP_EXIT_UNSUPPORTED;
        for (size_t j = 0; j < cnt; ++j) memcpy(buf + j * row, src + ((size_t)s->g + (j0 + j) * s->G) * per, row);
        HIPCHK(hipMemcpyAsync(dst + j0 * per, buf, cnt * row, hipMemcpyHostToDevice, s->stream));
        HIPCHK(hipEventRecord(s->ev[w], s->stream));
    }
    return 0;
}
// ... and the way back: Ng rows on the device -> rows k = g, g + G, ... of a host array
template <typename T>
int stage_rows_out(MultiShard *s, const T *src_dev, size_t per, T *dst)
{
    if (!dst || per == 0 || s->Ng == 0) return 0;
    const size_t row = per * sizeof(T);
    if (s->G == 1) { HIPCHK(hipMemcpyAsync(dst, src_dev, row * (size_t)s->Ng, hipMemcpyDeviceToHost, s->stream)); return 0; }
    size_t rows_per = kPinChunk / row;
    if (rows_per == 0) rows_per = 1;
    // two chunks in flight: chunk c is scattered by the host while chunk c + 1 crosses the bus
    struct Pending { char *buf; int w; size_t j0, cnt; bool live; } pend = {nullptr, 0, 0, 0, false};
    auto scatter = [&](const Pending &p) -> int {
        HIPCHK(hipEventSynchronize(s->ev[p.w]));
        for (size_t j = 0; j < p.cnt; ++j) memcpy(dst + ((size_t)s->g + (p.j0 + j) * s->G) * per, p.buf + j * row, row);
        return 0;
    };
    for (size_t j0 = 0; j0 < (size_t)s->Ng; j0 += rows_per) {
        const size_t cnt = (j0 + rows_per <= (size_t)s->Ng) ? rows_per : (size_t)s->Ng - j0;
        char *buf = nullptr;
        int w = 0;
        if (pin_acquire(s, cnt * row, &buf, &w)) return DAQP_EXIT_UNSUPPORTED;
        HIPCHK(hipMemcpyAsync(buf, src_dev + j0 * per, cnt * row, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipEventRecord(s->ev[w], s->stream));
        if (pend.live && scatter(pend)) return DAQP_EXIT_UNSUPPORTED;
        pend = {buf, w, j0, cnt, true};
    }
    if (pend.live && scatter(pend)) return DAQP_EXIT_UNSUPPORTED;
    return 0;
}

} // namespace

struct DAQPMultiBatch {
    int N = 0, n = 0, m = 0, ms = 0, G = 0;
    std::vector<MultiShard *> sh;
    double last_s = 0;                          // the slowest shard's time of the last call
};

namespace {

// run fn(shard) on every shard's own thread, wait for all; the first failure is reported (with the shard it came from)
int multi_run(DAQPMultiBatch *mb, const std::function<int(MultiShard *)> &fn)
{
    for (MultiShard *s : mb->sh) {
        {
            std::lock_guard<std::mutex> lk(s->mu);
            s->task = [s, &fn] { return fn(s); };
            s->has_task = true; s->done = false;
        }
        s->cv.notify_all();
    }
    int rc = 0;
    mb->last_s = 0;
    for (MultiShard *s : mb->sh) {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv.wait(lk, [&] { return s->done; });
        if (s->rc && !rc) { rc = s->rc; set_err("shard %d (device %d): %s", s->g, s->device, s->err.c_str()); }
        if (s->secs > mb->last_s) mb->last_s = s->secs;
    }
    return rc;
}

int check_multi_problem(const DAQPMultiBatch *mb, const DAQPBatchProblem *p)
{
    if (!mb || !p) { set_err("null multi-device batch or problem"); return DAQP_EXIT_UNSUPPORTED; }
    if (p->N != mb->N || p->n != mb->n || p->m != mb->m || p->ms != mb->ms) {
        set_err("problem shape (N=%d n=%d m=%d ms=%d) does not match the multi-device batch (N=%d n=%d m=%d ms=%d)", p->N, p->n, p->m, p->ms, mb->N, mb->n, mb->m, mb->ms);
        return DAQP_EXIT_UNSUPPORTED;
    }
    if (p->memory != DAQP_MEM_HOST) {
        set_err("daqp_batch_*_multi take ONE host-resident batch; device-resident data goes in per shard (daqp_batch_*_multi_shards)");
        return DAQP_EXIT_UNSUPPORTED;
    }
    return 0;
}

// the shard's share of a host-resident batch -> its own device slots; `ps`: the device-resident descriptor of what was staged
// `what`: DAQP_UPDATE_* bits of the arrays to take over (Rinv: H, M: A, sense: sense; f and the bounds whenever they are given)
int shard_stage_problem(MultiShard *s, const DAQPBatchProblem *p, DAQPBatchProblem *ps, int what)
{
    DAQPBatch *b = s->b;
    const size_t Ng = s->Ng, n = p->n, m = p->m, mA = p->m - p->ms;
    *ps = *p;
    ps->N = s->Ng; ps->memory = DAQP_MEM_DEVICE;
    ps->H = nullptr; ps->A = nullptr; ps->f = nullptr; ps->bupper = nullptr; ps->blower = nullptr; ps->sense = nullptr;
    int rc = 0;
    if ((what & DAQP_UPDATE_Rinv) && p->H) { rc |= slot_reserve(b, &b->sH, &b->nH, Ng * n * n); if (!rc) rc |= stage_rows_in(s, p->H, n * n, b->sH); ps->H = b->sH; }
    if ((what & DAQP_UPDATE_M) && p->A && mA) { rc |= slot_reserve(b, &b->sA, &b->nA, Ng * mA * n); if (!rc) rc |= stage_rows_in(s, p->A, mA * n, b->sA); ps->A = b->sA; }
    if (p->f) { rc |= slot_reserve(b, &b->sf, &b->nf, Ng * n); if (!rc) rc |= stage_rows_in(s, p->f, n, b->sf); ps->f = b->sf; }
    if (p->bupper) { rc |= slot_reserve(b, &b->sbu, &b->nbu, Ng * m); if (!rc) rc |= stage_rows_in(s, p->bupper, m, b->sbu); ps->bupper = b->sbu; }
    if (p->blower) { rc |= slot_reserve(b, &b->sbl, &b->nbl, Ng * m); if (!rc) rc |= stage_rows_in(s, p->blower, m, b->sbl); ps->blower = b->sbl; }
    if ((what & DAQP_UPDATE_sense) && p->sense) { rc |= slot_reserve(b, &b->ssense, &b->nsense, Ng * m); if (!rc) rc |= stage_rows_in(s, p->sense, m, b->ssense); ps->sense = b->ssense; }
    return rc ? DAQP_EXIT_UNSUPPORTED : 0;
}

} // namespace

extern "C" {

void daqp_batch_free_multi(DAQPMultiBatch *mb)
{
    if (!mb) return;
    for (MultiShard *s : mb->sh) {
        if (s->th.joinable()) {
            { std::lock_guard<std::mutex> lk(s->mu); s->stop = true; }
            s->cv.notify_all();
            s->th.join();
        }
        (void)hipSetDevice(s->device);
        if (s->stream) (void)hipStreamSynchronize(s->stream);
        if (s->b) { s->b->stream = s->stream; daqp_batch_free(s->b); }
        for (int i = 0; i < 2; ++i) { if (s->ev[i]) (void)hipEventDestroy(s->ev[i]); if (s->pin[i]) (void)hipHostFree(s->pin[i]); }
        if (s->stream) (void)hipStreamDestroy(s->stream);
        delete s;
    }
    delete mb;
}

int daqp_batch_create_multi(DAQPMultiBatch **out, int N, int n, int m, int ms, int ns_max, const DAQPSettings *settings, const int *devices, int n_devices)
{
    if (!out) return DAQP_EXIT_UNSUPPORTED;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_err("no HIP device: libdaqp_amd has no CPU path"); return DAQP_EXIT_UNSUPPORTED; }
    if (N <= 0) { set_err("bad batch size N=%d", N); return DAQP_EXIT_UNSUPPORTED; }
    if (n_devices <= 0) { n_devices = ndev; devices = nullptr; }       // "every visible device": a list, if one was passed, is not read
    for (int g = 0; g < n_devices; ++g) {
        const int dev = devices ? devices[g] : g;
        if (dev < 0 || dev >= ndev) { set_err("device %d of the list does not exist (%d visible)", dev, ndev); return DAQP_EXIT_UNSUPPORTED; }
    }
    const int G = n_devices < N ? n_devices : N;
    DAQPMultiBatch *mb = new DAQPMultiBatch();
    mb->N = N; mb->n = n; mb->m = m; mb->ms = ms; mb->G = G;
    for (int g = 0; g < G; ++g) {
        MultiShard *s = new MultiShard();
        s->g = g; s->G = G; s->device = devices ? devices[g] : g; s->Ng = (N - g + G - 1) / G;
        mb->sh.push_back(s);
        if (hipSetDevice(s->device) != hipSuccess || hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
            set_err("shard %d: no stream on device %d", g, s->device);
            daqp_batch_free_multi(mb);
            return DAQP_EXIT_UNSUPPORTED;
        }
        const int rc = daqp_batch_create(&s->b, s->Ng, n, m, ms, ns_max, settings, s->device);
        if (rc) { const std::string e = g_err; daqp_batch_free_multi(mb); set_err("shard %d (device %d): %s", g, devices ? devices[g] : g, e.c_str()); return rc; }
        daqp_batch_set_stream(s->b, s->stream);
        s->th = std::thread(shard_worker, s);
    }
    *out = mb;
    return 0;
}

int daqp_batch_multi_shards(const DAQPMultiBatch *mb) { return mb ? mb->G : 0; }
DAQPBatch *daqp_batch_multi_shard(DAQPMultiBatch *mb, int g, int *shard_N, int *device)
{
    if (!mb || g < 0 || g >= mb->G) return nullptr;
    if (shard_N) *shard_N = mb->sh[g]->Ng;
    if (device) *device = mb->sh[g]->device;
    return mb->sh[g]->b;
}

int daqp_batch_setup_multi(DAQPMultiBatch *mb, const DAQPBatchProblem *p, int init_mask)
{
    int rc = check_multi_problem(mb, p);
    if (rc) return rc;
    if (!p->f || !p->bupper || !p->blower || (p->m > p->ms && !p->A)) { set_err("f, A, bupper, blower are required (H may be NULL: an LP)"); return DAQP_EXIT_UNSUPPORTED; }
    return multi_run(mb, [&](MultiShard *s) -> int {
        DAQPBatchProblem ps;
        int r = shard_stage_problem(s, p, &ps, DAQP_UPDATE_Rinv | DAQP_UPDATE_M | DAQP_UPDATE_sense);
        if (!r) r = daqp_batch_setup(s->b, &ps, init_mask);
        if (!r && hipStreamSynchronize(s->stream) != hipSuccess) { set_err("stream synchronisation failed"); r = DAQP_EXIT_UNSUPPORTED; }
        return r;
    });
}

int daqp_batch_update_multi(DAQPMultiBatch *mb, int mask, const DAQPBatchProblem *p)
{
    int rc = check_multi_problem(mb, p);
    if (rc) return rc;
    const int full = DAQP_UPDATE_Rinv | DAQP_UPDATE_M | DAQP_UPDATE_v | DAQP_UPDATE_d | DAQP_UPDATE_sense;
    const bool refactor = (mask & full) == full;
    if (refactor && (!p->f || !p->bupper || !p->blower || (p->m > p->ms && !p->A))) { set_err("a full re-setup of a multi-device batch needs every array"); return DAQP_EXIT_UNSUPPORTED; }
    return multi_run(mb, [&](MultiShard *s) -> int {
        DAQPBatchProblem ps;
        // the arrays the mask's steps read (utils.c:58-221): H with the Rinv bit, A with Rinv or M, sense with its own bit
        int r = shard_stage_problem(s, p, &ps, (mask & DAQP_UPDATE_Rinv) ? (mask | DAQP_UPDATE_M) : mask);
        if (!r) r = daqp_batch_update(s->b, mask, &ps);
        // the staging copies out of the caller's arrays have run when this returns (the header's contract: "every call returns when
        // all shards have finished it")
        if (!r && hipStreamSynchronize(s->stream) != hipSuccess) { set_err("stream synchronisation failed"); r = DAQP_EXIT_UNSUPPORTED; }
        return r;
    });
}

int daqp_batch_solve_multi(DAQPMultiBatch *mb, DAQPBatchResult *r)
{
    if (!mb || !r) { set_err("null multi-device batch or result"); return DAQP_EXIT_UNSUPPORTED; }
    if (r->memory != DAQP_MEM_HOST) { set_err("daqp_batch_solve_multi returns ONE host-resident result; device-resident results come per shard (daqp_batch_solve_multi_shards)"); return DAQP_EXIT_UNSUPPORTED; }
    const size_t n = mb->n, m = mb->m;
    const int rc = multi_run(mb, [&](MultiShard *s) -> int {
        DAQPBatchResult rs;
        memset(&rs, 0, sizeof(rs));
        rs.memory = DAQP_MEM_DEVICE;             // into the shard's own result buffers
        int q = daqp_batch_solve(s->b, &rs);
        if (q) return q;
        DAQPBatch *b = s->b;
        q |= stage_rows_out(s, b->ox, n, r->x);
        q |= stage_rows_out(s, b->olam, m, r->lam);
        q |= stage_rows_out(s, b->ofval, 1, r->fval);
        q |= stage_rows_out(s, b->osoft, 1, r->soft_slack);
        q |= stage_rows_out(s, b->oflag, 1, r->exitflag);
        q |= stage_rows_out(s, b->oiter, 1, r->iter);
        if (!q && hipStreamSynchronize(s->stream) != hipSuccess) { set_err("stream synchronisation failed"); q = DAQP_EXIT_UNSUPPORTED; }
        return q ? DAQP_EXIT_UNSUPPORTED : 0;
    });
    r->solve_time = mb->last_s;
    return rc;
}

// per-shard descriptors: ps[g] / rs[g] describe shard g's problems / results back to back (N = that shard's size), host-resident or
// resident on THAT shard's device; the G shards run side by side.  ps may be NULL for a solve, rs for a setup / update.
int daqp_batch_setup_multi_shards(DAQPMultiBatch *mb, const DAQPBatchProblem *ps, int init_mask)
{
    if (!mb || !ps) { set_err("null argument"); return DAQP_EXIT_UNSUPPORTED; }
    return multi_run(mb, [&](MultiShard *s) -> int { return daqp_batch_setup(s->b, &ps[s->g], init_mask); });
}
int daqp_batch_update_multi_shards(DAQPMultiBatch *mb, int mask, const DAQPBatchProblem *ps)
{
    if (!mb || !ps) { set_err("null argument"); return DAQP_EXIT_UNSUPPORTED; }
    return multi_run(mb, [&](MultiShard *s) -> int {
        int r = daqp_batch_update(s->b, mask, &ps[s->g]);
        if (!r && hipStreamSynchronize(s->stream) != hipSuccess) { set_err("stream synchronisation failed"); r = DAQP_EXIT_UNSUPPORTED; }
        return r;
    });
}
int daqp_batch_solve_multi_shards(DAQPMultiBatch *mb, DAQPBatchResult *rs)
{
    if (!mb || !rs) { set_err("null argument"); return DAQP_EXIT_UNSUPPORTED; }
    return multi_run(mb, [&](MultiShard *s) -> int {
        const int q = daqp_batch_solve(s->b, &rs[s->g]);
        if (!q && hipStreamSynchronize(s->stream) != hipSuccess) { set_err("stream synchronisation failed"); return DAQP_EXIT_UNSUPPORTED; }
        return q;
    });
}

int daqp_quadprog_batch_multi(DAQPBatchResult *r, const DAQPBatchProblem *p, const DAQPSettings *settings, const int *devices, int n_devices)
{
    if (!r || !p) { set_err("null argument"); return DAQP_EXIT_UNSUPPORTED; }
    if (p->memory != DAQP_MEM_HOST || r->memory != DAQP_MEM_HOST) { set_err("daqp_quadprog_batch_multi takes host-resident problems and results (each shard stages its own part)"); return DAQP_EXIT_UNSUPPORTED; }
    if (p->N == 0) { r->setup_time = r->solve_time = 0; return 0; }
    if (!p->f || !p->bupper || !p->blower || (p->m > p->ms && !p->A)) { set_err("f, A, bupper, blower are required"); return DAQP_EXIT_UNSUPPORTED; }
    int ns = 0;
    if (p->sense)
        for (int q = 0; q < p->N; ++q) {
            int c = 0;
            for (int i = 0; i < p->m; ++i) c += (p->sense[(size_t)q * p->m + i] & DAQP_SOFT) ? 1 : 0;
            if (c > ns) ns = c;
        }
    DAQPMultiBatch *mb = nullptr;
    int rc = daqp_batch_create_multi(&mb, p->N, p->n, p->m, p->ms, ns, settings, devices, n_devices);
    if (rc) return rc;
    rc = daqp_batch_setup_multi(mb, p, DAQP_UPDATE_unconstrained | DAQP_UPDATE_eliminate);
    r->setup_time = mb->last_s;
    if (!rc) rc = daqp_batch_solve_multi(mb, r);
    std::string e = rc ? std::string(g_err) : std::string();
    daqp_batch_free_multi(mb);
    if (rc) set_err("%s", e.c_str());
    return rc;
}

} // extern "C"
