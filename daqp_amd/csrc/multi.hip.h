// multi.hip.h -- daqp_quadprog_batch_multi: ONE host-resident batch solved on several GPUs.  Independent problems,
// no exchange step: problem k goes to shard k mod G (interleaved, so that the spread of iteration counts averages out), every shard
// has its own host thread, device, stream and device-resident workspaces, and writes its results back into the caller's arrays.
// Included by daqp_amd.hip (host code only).
#pragma once
#include <thread>

namespace {

struct ShardOut { int rc = 0; double setup_s = 0, solve_s = 0; std::string err; };

void run_shard(int g, int G, int device, const DAQPBatchProblem *p, DAQPBatchResult *r, const DAQPSettings *settings, int ns, ShardOut *out)
{
    const size_t n = p->n, m = p->m, mA = p->m - p->ms;
    const int Ng = (p->N - g + G - 1) / G;
    if (Ng <= 0) return;
    // this shard's problems, gathered back to back (k = g, g + G, ...)
    std::vector<double> H, f((size_t)Ng * n), A((size_t)Ng * mA * n), bu((size_t)Ng * m), bl((size_t)Ng * m);
    std::vector<int> sense;
    if (p->H) H.resize((size_t)Ng * n * n);
    if (p->sense) sense.resize((size_t)Ng * m);
    for (int j = 0; j < Ng; ++j) {
        const size_t k = (size_t)g + (size_t)j * G;
        if (p->H) memcpy(&H[(size_t)j * n * n], p->H + k * n * n, n * n * sizeof(double));
        memcpy(&f[(size_t)j * n], p->f + k * n, n * sizeof(double));
        if (mA) memcpy(&A[(size_t)j * mA * n], p->A + k * mA * n, mA * n * sizeof(double));
        memcpy(&bu[(size_t)j * m], p->bupper + k * m, m * sizeof(double));
        memcpy(&bl[(size_t)j * m], p->blower + k * m, m * sizeof(double));
        if (p->sense) memcpy(&sense[(size_t)j * m], p->sense + k * m, m * sizeof(int));
    }
    DAQPBatchProblem ps = *p;
    ps.N = Ng; ps.H = p->H ? H.data() : nullptr; ps.f = f.data(); ps.A = mA ? A.data() : nullptr;
    ps.bupper = bu.data(); ps.blower = bl.data(); ps.sense = p->sense ? sense.data() : nullptr; ps.memory = DAQP_MEM_HOST;
    std::vector<double> x(r->x ? (size_t)Ng * n : 0), lam(r->lam ? (size_t)Ng * m : 0), fval(r->fval ? Ng : 0), soft(r->soft_slack ? Ng : 0);
    std::vector<int> flag(r->exitflag ? Ng : 0), iter(r->iter ? Ng : 0);
    DAQPBatchResult rs;
    memset(&rs, 0, sizeof(rs));
    rs.x = r->x ? x.data() : nullptr; rs.lam = r->lam ? lam.data() : nullptr; rs.fval = r->fval ? fval.data() : nullptr;
    rs.soft_slack = r->soft_slack ? soft.data() : nullptr; rs.exitflag = r->exitflag ? flag.data() : nullptr; rs.iter = r->iter ? iter.data() : nullptr;
    rs.memory = DAQP_MEM_HOST;
    DAQPBatch *b = nullptr;
    hipStream_t stream = nullptr;
    int rc = daqp_batch_create(&b, Ng, p->n, p->m, p->ms, ns, settings, device);
    if (rc == 0 && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) { rc = DAQP_EXIT_UNSUPPORTED; set_err("hipStreamCreate failed on device %d", device); }
    if (rc == 0) {
        daqp_batch_set_stream(b, stream);
        const double t0 = now_s();
        rc = daqp_batch_setup(b, &ps, DAQP_UPDATE_unconstrained | DAQP_UPDATE_eliminate);
        if (rc == 0 && hipStreamSynchronize(stream) != hipSuccess) rc = DAQP_EXIT_UNSUPPORTED;
        out->setup_s = now_s() - t0;
        if (rc == 0) rc = daqp_batch_solve(b, &rs);
        out->solve_s = rs.solve_time;
    }
    if (rc) out->err = g_err;
    if (b) daqp_batch_free(b);
    if (stream) (void)hipStreamDestroy(stream);
    out->rc = rc;
    if (rc) return;
    for (int j = 0; j < Ng; ++j) {
        const size_t k = (size_t)g + (size_t)j * G;
        if (r->x) memcpy(r->x + k * n, &x[(size_t)j * n], n * sizeof(double));
        if (r->lam) memcpy(r->lam + k * m, &lam[(size_t)j * m], m * sizeof(double));
        if (r->fval) r->fval[k] = fval[j];
        if (r->soft_slack) r->soft_slack[k] = soft[j];
        if (r->exitflag) r->exitflag[k] = flag[j];
        if (r->iter) r->iter[k] = iter[j];
    }
}

} // namespace

extern "C" int daqp_quadprog_batch_multi(DAQPBatchResult *r, const DAQPBatchProblem *p, const DAQPSettings *settings, const int *devices, int n_devices)
{
    if (!r || !p) { set_err("null argument"); return DAQP_EXIT_UNSUPPORTED; }
    if (p->memory != DAQP_MEM_HOST || r->memory != DAQP_MEM_HOST) { set_err("daqp_quadprog_batch_multi takes host-resident problems and results (each shard stages its own part)"); return DAQP_EXIT_UNSUPPORTED; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_err("no HIP device: libdaqp_amd has no CPU path"); return DAQP_EXIT_UNSUPPORTED; }
    if (n_devices <= 0) n_devices = ndev;
    if (p->N == 0) { r->setup_time = r->solve_time = 0; return 0; }
    if (!p->f || !p->bupper || !p->blower || (p->m > p->ms && !p->A)) { set_err("f, A, bupper, blower are required"); return DAQP_EXIT_UNSUPPORTED; }
    for (int g = 0; g < n_devices; ++g) {
        const int dev = devices ? devices[g] : g;
        if (dev < 0 || dev >= ndev) { set_err("device %d of the list does not exist (%d visible)", dev, ndev); return DAQP_EXIT_UNSUPPORTED; }
    }
    int ns = 0;
    if (p->sense)
        for (int q = 0; q < p->N; ++q) {
            int c = 0;
            for (int i = 0; i < p->m; ++i) c += (p->sense[(size_t)q * p->m + i] & DAQP_SOFT) ? 1 : 0;
            if (c > ns) ns = c;
        }
    const int G = n_devices < p->N ? n_devices : p->N;
    std::vector<ShardOut> outs(G);
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g) th.emplace_back(run_shard, g, G, devices ? devices[g] : g, p, r, settings, ns, &outs[g]);
    for (auto &t : th) t.join();
    r->setup_time = r->solve_time = 0;
    for (int g = 0; g < G; ++g) {
        if (outs[g].rc) { set_err("shard %d (device %d): %s", g, devices ? devices[g] : g, outs[g].err.c_str()); return outs[g].rc; }
        if (outs[g].setup_s > r->setup_time) r->setup_time = outs[g].setup_s;
        if (outs[g].solve_s > r->solve_time) r->solve_time = outs[g].solve_s;
    }
    return 0;
}
