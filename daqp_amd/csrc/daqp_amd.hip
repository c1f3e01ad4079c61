// daqp_amd.hip -- host side of libdaqp_amd.so: the C ABI declared in include/daqp_amd.h.
// Thin by design: argument checks, device buffers, three kernel launches.  No numerical work
// happens on the host and there is no CPU fallback.
#include <hip/hip_runtime.h>

#include <sched.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <atomic>
#include <vector>
#include <string>
#include <thread>

#include "kernels.hip.h"
#include "reg_kernel.hip.h"
#include "setup_fast.hip.h"
#include "prox.hip.h"
#include "recheck.hip.h"
#include "setup_m.hip.h"
#include "setup_fact.hip.h"
#include "wg_layout.hip.h"
#include "tiny_setup.hip.h"
#include "setup_blk.hip.h"
// the workgroup-per-problem solve kernel lives in its own translation unit (wg_kernel.hip): a change to it does not rebuild
// everything else
namespace daqp_amd {
// the register-centric solve kernels: reg_kernel.hip
#define DAQP_REG_SHAPE(NB, NP) \
    extern template __global__ void k_ldp_reg<NB, NP, false>(const BatchDev *__restrict__, int); \
    extern template __global__ void k_ldp_reg<NB, NP, true>(const BatchDev *__restrict__, int);
extern template __global__ void k_ldp_reg<3, 25, true, 2>(const BatchDev *__restrict__, int);
extern template __global__ void k_ldp_reg<2, 32, true, 1>(const BatchDev *__restrict__, int);
extern template __global__ void k_ldp_reg<3, 25, true, 1>(const BatchDev *__restrict__, int);
extern template __global__ void k_ldp_reg<4, 32, true, 1>(const BatchDev *__restrict__, int);
extern template __global__ void k_ldp_reg<8, 16, true, 1>(const BatchDev *__restrict__, int);
extern template __global__ void k_ldp_reg<6, 25, true, 1>(const BatchDev *__restrict__, int);
extern template __global__ void k_ldp_reg<5, 32, true, 1>(const BatchDev *__restrict__, int);
DAQP_REG_SHAPE(1, 6)
DAQP_REG_SHAPE(1, 8)
DAQP_REG_SHAPE(3, 25)
#ifndef DAQP_AMD_FEW_VARIANTS
DAQP_REG_SHAPE(1, 13)
DAQP_REG_SHAPE(1, 16)
DAQP_REG_SHAPE(2, 16)
DAQP_REG_SHAPE(3, 8)
DAQP_REG_SHAPE(4, 8)
DAQP_REG_SHAPE(1, 25)
DAQP_REG_SHAPE(2, 32)
#endif
#undef DAQP_REG_SHAPE
#define DAQP_SETUP_SIZE(NMAX) \
    extern template __global__ void k_setup_fast<NMAX, false, false>(BatchDev, int); \
    extern template __global__ void k_setup_fast<NMAX, false, true>(BatchDev, int); \
    extern template __global__ void k_setup_fast<NMAX, true, false>(BatchDev, int);
DAQP_SETUP_SIZE(16)
DAQP_SETUP_SIZE(32)
DAQP_SETUP_SIZE(56)
DAQP_SETUP_SIZE(64)
#undef DAQP_SETUP_SIZE
extern template __global__ void k_setup_tiny<4>(BatchDev, int);
#define DAQP_BLK_SHAPE(NT, NW, TAIL) extern template __global__ void k_setup_blk<NT, NW, TAIL>(const BatchDev *__restrict__, int);
DAQP_BLK_SHAPES
#undef DAQP_BLK_SHAPE   // (the 16-per-wave SETUP kernel is the default for these shapes: setup_kernel.hip)
template <int C, bool EX, bool TIER = false> __global__ void k_ldp_wg(BatchDev b, int mode);
extern template __global__ void k_ldp_wg<4, false, true>(BatchDev, int);
extern template __global__ void k_ldp_wg<2, false, true>(BatchDev, int);
extern template __global__ void k_ldp_wg<2, false>(BatchDev, int);
extern template __global__ void k_ldp_wg<2, true>(BatchDev, int);
extern template __global__ void k_ldp_wg<4, false>(BatchDev, int);
extern template __global__ void k_ldp_wg<4, true>(BatchDev, int);
}

using namespace daqp_amd;

namespace {

thread_local char g_err[512] = "";
void set_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return DAQP_EXIT_UNSUPPORTED;                                                     \
        }                                                                                     \
    } while (0)

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void default_settings(DAQPSettings *s) // reference constants.h:15-29 / api.c:505-527
{
    s->primal_tol = 1e-6; s->dual_tol = 1e-12; s->zero_tol = 1e-11; s->pivot_tol = 1e-6;
    s->progress_tol = 1e-14; s->cycle_tol = 10; s->iter_limit = 10000; s->fval_bound = DAQP_INF;
    s->eps_prox = -1e-6; s->eta_prox = -1.0; s->rho_soft = 1e-6; s->rel_subopt = 0; s->abs_subopt = 0;
    s->sing_tol = 3.7e-11; s->refactor_tol = 1e-9; s->time_limit = 0;
}

} // namespace

struct DAQPBatch {
    int device = 0;
    hipStream_t stream = nullptr;
    BatchDev d{};
    std::vector<void *> owned;
    unsigned long long bytes = 0;
    // staging copies of host inputs (allocated on first use)
    double *sH = nullptr, *sf = nullptr, *sA = nullptr, *sbu = nullptr, *sbl = nullptr;
    int *ssense = nullptr;
    size_t nH = 0, nf = 0, nA = 0, nbu = 0, nbl = 0, nsense = 0;   // elements each staging slot can hold
    bool was_shared = false;   // last set up by daqp_batch_setup_shared: d.H / d.A are ONE matrix, not N
    // library-owned result buffers
    double *ox = nullptr, *olam = nullptr, *ofval = nullptr, *osoft = nullptr;
    int *oflag = nullptr, *oiter = nullptr;
    DAQPSettings *st_dev = nullptr;
    BatchDev *d_dev = nullptr;
    int C = 1;
    bool spill = false;
    int NB = 0, NP = 0;   // register-resident M variant (0: stream M from HBM)
    bool tiny_setup = false;   // n <= 12, m <= 48: the 16-problems-per-wave setup kernel (tiny_setup.hip.h)
    bool fast_setup = false, setup_spill = false;
    bool fact_gs = false;           // 64 < n, factors that WOULD fit the LDS of k_setup: the default arithmetic's full setup still takes k_fact_wg + the scratch variant
    double *fact_buf = nullptr;     // [N][4] records of k_fact_wg (setup_fact.hip.h), allocated for the shapes it serves
    // workgroup-per-problem solve kernel (wg_kernel.hip.h): shapes without a register variant and more than 64 working-set rows
    bool in_prox_loop = false;      // launches of the proximal outer loop (solve_with_prox)
    bool use_wg = false;
    int wg_W = 0, wg_C = 0, wg_grid = 0;
    int wg_tier_grid = 0; size_t lds_wg_tier = 0;   // tiered launch of the workgroup kernel in front of a cold solve (0: none)
    bool reg_handover = false;   // k_ldp_reg<2,32,*> may flag problems (more working-set rows than lanes: n = 64) for k_ldp right behind it
    size_t lds_fb = 0;           // ... and that launch's LDS
    bool img32 = false;          // default arithmetic: the solve launch is k_ldp_reg<NB, NP, true, 1> -- an fp32 image of M in the registers, two waves per
                                 // SIMD, at most d.img_rows working-set rows -- with k_ldp_reg<NB, NP, true, 0> right behind it for the problems it flags
    size_t lds_img = 0;          // ... and the image kernel's LDS
    int img_kind = 0;            // its IMG template argument (2: the last row block split over two lanes per row; 1: full blocks)
    // the carve-up of a WARM launch (daqp_update_ldp(UPDATE_v|UPDATE_d) fused into the solve: an MPC step starts from ~20 rows and moves a few): fewer
    // rows held at all, so that at the same eight workgroups per CU more of them sit in LDS (the launch argument carries rows | cache, see k_ldp_reg)
    int img_rows_warm = 0, img_cache_warm = 0;
    size_t lds_img_warm = 0;
    // hand-overs of the last solve launch, counted on the device and copied to pinned memory without waiting: when most of a batch goes to the
    // full-register kernel anyway (working sets beyond what the image kernel holds), the next launches go there directly; every 16th tries again
    int *img_ho_pin = nullptr;
    unsigned img_skipped = 0;
    bool img_only = false;       // no register shape holds M, but its fp32 image fits (k_ldp_reg<NB,NP,true,1> of kImgOnlyShapes, one wave per SIMD): default-mode solves run it in front of k_ldp
    int io_nb = 0, io_np = 0;
    int img_min_warm = 16384;
    size_t lds_wg = 0;
    double *wide_u = nullptr, *wide_l = nullptr;   // daqp_batch_setup_shared: +-1e30 bounds of the one factorisation
    int *structural = nullptr, *shared_flag = nullptr;
    bool exact_sticky = false;   // the sense of some working-set rows was replaced without the working set being rebuilt (a sense update without a
                                 // sense array, utils.c:85-86; an update that took the new sense and then failed its bound check, utils.c:88-96): the
                                 // reference iterates on from that state, and only its own arithmetic does the same (the default mode's carried
                                 // intermediate results assume the flags they were formed with) -- solves use the exact kernels until the next update
    int part_mask = 0;      // the last (re-)setup was a partial daqp_batch_update with this mask (k_setup<..., PART>): its regularising re-runs use the same kernel
    int pending_mask = 0;   // daqp_batch_update(UPDATE_v|UPDATE_d) not yet applied: the next solve launch does it (k_ldp_reg mode 2)
    size_t lds_setup = 0, lds_ldp = 0, lds_update = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool timed_setup = false, timed_solve = false;
    bool is_setup = false;
    // proximal outer loop (prox.hip.h): buffers appear with the first singular Hessian
    ProxDev px{};
    bool prox_ready = false;
    int counter_host[4] = {0, 0, 0, 0};   // read-back of px.counter
    // "is any Hessian singular?" after a setup: counted on the device, copied to pinned memory without waiting.  The next solve
    // goes out FIRST (flagged problems sit it out), then the host looks at the count: no gap between the setup and the solve
    // launch on the device.  Anything else that needs the final setup flags resolves the question first (resolve_setup).
    int *pin_count = nullptr;
    hipEvent_t ev_count = nullptr;
    bool reg_pending = false;
    int reg_mask = 0;
    int n_prox_qps = 0;            // problems of the current setup that go through the outer loop
    int prox_outer = 0;            // outer iterations of the last solve (the longest loop of the batch)
    double *ident = nullptr;       // LP batches (H == NULL): the one n x n identity the setup pass reads as H
    // single-problem workspaces: results of the last daqp_ldp, waiting for daqp_extract_result
    // single-problem batches (N == 1, the daqp_quadprog / setup_daqp path): one pinned host slab and one device slab each for
    // the inputs and the results (one copy each way instead of six), and the pool key of a parked batch
    char *pin_in = nullptr, *dev_in = nullptr, *pin_out = nullptr;
    size_t in_cap = 0, out_bytes = 0;
    double *dev_mir = nullptr, *pin_mir = nullptr;   // host mirrors of a single-problem workspace: gathered slab (k_mirror) and its pinned landing zone
    // One problem, latency: the result slab and the mirror slab are MAPPED host memory that the kernels write themselves, and the host
    // learns that they are there from a mapped word written by a one-thread kernel at the end of the stream's work -- no hipMemcpy, no
    // stream synchronisation on the way of daqp_quadprog / daqp_solve / daqp_update_ldp (0.34 / 0.14 ms per call were
    // launches, small copies and synchronisations around ~0.1 ms / ~0.02 ms of kernels)
    BatchDev d_pushed{};        // what d_dev holds (the register kernels read the descriptor through it): pushed again only when it differs
    bool pushed_valid = false;
    bool mapped_out = false, mapped_mir = false;     // ox .. oiter / dev_mir point into pin_out / pin_mir through their device addresses
    int *pin_sig = nullptr, *pin_sig_dev = nullptr;  // the completion word and its device address
    int sig_seq = 0;
    bool lean = false;          // daqp_quadprog: no count of flagged Hessians, no activation launch (the solve launch activates; a flagged problem comes back with the internal code and takes the full path)
    bool defer_wait = false;    // daqp_batch_solve only enqueues: the caller puts more work behind it, waits once and collects (daqp_ldp)
    bool recheck_due = false;   // one problem: the second pass of an INFEASIBLE verdict is decided after the wait, on the host
    char *pin_upd = nullptr, *pin_upd_dev = nullptr;  // f / bupper / blower of a deferred daqp_update_ldp(v|d): mapped, read by the solve launch in place
    bool mirror_ldp_due = false;   // a deferred update's v / d have not reached the host mirrors yet: the next solve's gather brings them
    hipEvent_t ev_in = nullptr;     // the last packed host->device copy (the pinned slab is free again once it has run)
    std::string env_key;
    int ns_max = 0;
    unsigned long long *tstart = nullptr;   // [N] start stamps of the current daqp_batch_solve (allocated when a time limit is first armed)
    std::vector<double> one_lam;
    double one_fval = 0, one_soft = 0;
    int one_flag = 0, one_iter = 0;
    bool one_valid = false;   // one_* belong to the workspace's current LDP (cleared by setup / update / a failed solve)
    // default arithmetic: INFEASIBLE verdicts of a solve straight after a setup are re-derived in the reference's arithmetic (recheck.hip.h)
    bool fresh = false;        // the state is that of the last daqp_batch_setup (own H / A per problem, not an LP): nothing solved or updated since
    int fresh_mask = 0;        // its init_mask
    bool recheck = true;       // DAQP_AMD_NO_RECHECK=1 switches the second pass off
    int recheck_device = -1;   // the same for inputs that were ADOPTED (DAQP_MEM_DEVICE: the second pass reads them again at solve time): -1 = not said
                               // (on, unless DAQP_AMD_RECHECK_DEVICE=0), 0 / 1 = daqp_batch_set_recheck
    bool inputs_adopted = false;   // the last setup's inputs were device arrays of the caller's, used in place
    bool quiet_setup = false;  // the setup that runs now is the second pass of a one-problem batch: the caller's setup events stay as they are
    hipEvent_t ev_r[2] = {nullptr, nullptr};   // around the second pass (daqp_batch_recheck_ms)
    bool timed_recheck = false;
    DAQPBatch *redo = nullptr; // companion batch in the exact mode (created with the first infeasible verdict, grown on demand)
    int *redo_list = nullptr, *redo_count = nullptr, *pin_redo = nullptr, *pin_redo_dev = nullptr;   // pin_redo: a mapped host word the marking kernel writes
    int rechecked = 0;         // problems the last solve sent through the second pass
};

namespace {

template <typename T>
int dev_alloc(DAQPBatch *b, T **p, size_t count)
{
    if (count == 0) count = 1;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)));
    b->owned.push_back(*p);
    b->bytes += count * sizeof(T);
    return 0;
}

__global__ void k_signal(int *host_word, int seq) { __hip_atomic_store(host_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
// everything queued on the batch's stream so far has run when this returns: a one-thread kernel behind it writes the next sequence
// number into a mapped host word (kernels of a stream run in order; a kernel's writes to host memory are visible when it has ended),
// the host polls that word.  Falls back to a stream synchronisation when the word cannot be had.
int wait_stream(DAQPBatch *b)
{
    if (!b->pin_sig) {
        if (hipHostMalloc(reinterpret_cast<void **>(&b->pin_sig), 64, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer(reinterpret_cast<void **>(&b->pin_sig_dev), b->pin_sig, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (b->pin_sig) { (void)hipHostFree(b->pin_sig); b->pin_sig = nullptr; }
            HIPCHK(hipStreamSynchronize(b->stream));
            return 0;
        }
        *b->pin_sig = 0;
    }
    const int seq = ++b->sig_seq;
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, b->stream, b->pin_sig_dev, seq);
    HIPCHK(hipGetLastError());
    unsigned spins = 0;
    while (__atomic_load_n(b->pin_sig, __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 0xffff) == 0) {      // (a failed stream never writes the word)
            const hipError_t qe = hipStreamQuery(b->stream);
            if (qe != hipSuccess && qe != hipErrorNotReady) { set_err("a launch on the batch's stream failed: %s", hipGetErrorString(qe)); return DAQP_EXIT_UNSUPPORTED; }
            if (qe == hipSuccess && __atomic_load_n(b->pin_sig, __ATOMIC_ACQUIRE) != seq) { HIPCHK(hipStreamSynchronize(b->stream)); break; }
        }
    }
    return 0;
}

// A small host structure into device memory, stream-ordered, as a KERNEL ARGUMENT: no copy command out of pageable host memory.  (Such a
// copy is staged by the runtime and its staging is released by the runtime's completion thread some time after the stream has gone
// idle -- work for that thread right where a process that exits at once races with it (INTEGRATION.md "Process exit").  And a launch
// costs half of what the copy did.)
template <class T>
__global__ void k_put(T v, T *dst)
{
    static_assert(sizeof(T) % 4 == 0 && sizeof(T) <= 3072, "travels as a kernel argument, word by word");
    const int *s = reinterpret_cast<const int *>(&v);
    int *o = reinterpret_cast<int *>(dst);
    for (int i = (int)threadIdx.x; i < (int)(sizeof(T) / 4); i += (int)blockDim.x) o[i] = s[i];
}
template <class T>
int put_async(T *dst, const T &v, hipStream_t stream)
{
    hipLaunchKernelGGL(k_put<T>, dim3(1), dim3(64), 0, stream, v, dst);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// the descriptor as the kernels that take it through a pointer see it: written (stream-ordered) when it has changed since the last time
int push_descriptor(DAQPBatch *b)
{
    if (b->pushed_valid && memcmp(&b->d_pushed, &b->d, sizeof(BatchDev)) == 0) return 0;
    if (put_async(b->d_dev, b->d, b->stream)) { set_err("descriptor launch failed"); return DAQP_EXIT_UNSUPPORTED; }
    memcpy(&b->d_pushed, &b->d, sizeof(BatchDev));
    b->pushed_valid = true;
    return 0;
}

typedef void (*ldp_kernel_t)(BatchDev, int);
typedef void (*ldp_reg_kernel_t)(const BatchDev *, int);
// register-resident variants: (row blocks, k-pairs) held per lane; needs cap <= 64 and L/rows in LDS
struct RegShape { int nb, np; };
#ifdef DAQP_AMD_FEW_VARIANTS   // development builds: fewer instantiations, faster compile
const RegShape kRegShapes[] = {{1, 6}, {1, 8}, {3, 25}};
#else
const RegShape kRegShapes[] = {{1, 6}, {1, 8}, {1, 13}, {1, 16}, {2, 16}, {3, 8}, {4, 8}, {1, 25}, {3, 25}, {2, 32}};   // ((3,8), (4,8): few variables, many rows -- n <= 16, m <= 192; (1,25): n <= 50 with m <= 64 -- both at two waves per SIMD)
#endif
// the shapes served by an fp32 image ALONE (no full-register kernel exists: M itself would not fit 512 registers), first fit:
// (4,32) n <= 63, m <= 256 | (8,16) n <= 32, m <= 512 | (6,25) n <= 50, m <= 384 | (5,32) n <= 63, m <= 320 -- 256 ... 320 image registers, one wave per SIMD
const RegShape kImgOnlyShapes[] = {{4, 32}, {8, 16}, {6, 25}, {5, 32}};
ldp_reg_kernel_t pick_img_only(const DAQPBatch *b)
{
    if (b->io_nb == 4 && b->io_np == 32) return k_ldp_reg<4, 32, true, 1>;
    if (b->io_nb == 8 && b->io_np == 16) return k_ldp_reg<8, 16, true, 1>;
    if (b->io_nb == 6 && b->io_np == 25) return k_ldp_reg<6, 25, true, 1>;
    if (b->io_nb == 5 && b->io_np == 32) return k_ldp_reg<5, 32, true, 1>;
    return nullptr;
}
// exact: the reference's arithmetic (two roundings per multiply-add); otherwise fused multiply-adds (default mode)
ldp_reg_kernel_t pick_ldp_reg_img(const DAQPBatch *b)
{
    if (b->NB == 3 && b->NP == 25) return b->img_kind == 2 ? k_ldp_reg<3, 25, true, 2>      // (IMG = 2: the third row block holds at most 32 rows)
                                                            : k_ldp_reg<3, 25, true, 1>;     // (161 <= m <= 192: three full blocks, 150 image registers, 23 of the rest in scratch)
    if (b->NB == 2 && b->NP == 32) return k_ldp_reg<2, 32, true, 1>;    // (51 <= n <= 63, m <= 128: two full row blocks, 128 image registers)
    return nullptr;
}
ldp_reg_kernel_t pick_ldp_reg(const DAQPBatch *b, bool exact)
{
#define DAQP_REG_PICK(nb, np) if (b->NB == nb && b->NP == np) return exact ? k_ldp_reg<nb, np, false> : k_ldp_reg<nb, np, true>;
    DAQP_REG_PICK(1, 6)
    DAQP_REG_PICK(1, 8)
    DAQP_REG_PICK(3, 25)
#ifndef DAQP_AMD_FEW_VARIANTS
    DAQP_REG_PICK(1, 13)
    DAQP_REG_PICK(1, 16)
    DAQP_REG_PICK(2, 16)
    DAQP_REG_PICK(3, 8)
    DAQP_REG_PICK(4, 8)
    DAQP_REG_PICK(1, 25)
    DAQP_REG_PICK(2, 32)
#endif
#undef DAQP_REG_PICK
    return nullptr;
}
ldp_kernel_t pick_ldp(const DAQPBatch *b)
{
    const int C = b->C;
    const bool spill = b->spill;
#ifdef DAQP_AMD_FEW_VARIANTS
    if (!spill) return k_ldp<4, false, 0, 0>;
    return C == 8 ? k_ldp<8, true, 0, 0> : k_ldp<4, true, 0, 0>;
#else
    if (!spill) {
        if (C == 1) return k_ldp<1, false, 0, 0>;
        if (C == 2) return k_ldp<2, false, 0, 0>;
        return k_ldp<4, false, 0, 0>;
    }
    if (C == 1) return k_ldp<1, true, 0, 0>;
    if (C == 2) return k_ldp<2, true, 0, 0>;
    if (C == 8) return k_ldp<8, true, 0, 0>;
    return k_ldp<4, true, 0, 0>;
#endif
}

int launch_ldp(DAQPBatch *b, int mode, bool descriptor_changed = true)
{
    if (b->use_wg) {
        // persistent workgroups pull problems from a counter; whatever outgrows the LDS-resident L is flagged and solved by
        // the one-wave kernel right behind (mode | 4: flagged problems only -- an empty pass costs a few microseconds)
        typedef void (*wg_kernel_t)(BatchDev, int);
        BatchDev dd = b->d;
        if (b->in_prox_loop || b->exact_sticky) dd.exact_setup = 1;   // problems of the proximal outer loop keep the reference's arithmetic in both modes
        wg_kernel_t kw = dd.exact_setup ? (b->wg_C == 2 ? k_ldp_wg<2, true> : k_ldp_wg<4, true>) : (b->wg_C == 2 ? k_ldp_wg<2, false> : k_ldp_wg<4, false>);
        int wg_mode = mode;
        if (b->wg_tier_grid > 0 && b->fresh && mode == 0 && !dd.exact_setup) {
            // a first solve in the default arithmetic: two four-wave workgroups per CU, the inverse factor tiered (LDS + the problem's slot of the
            // stored factor); what it flags -- warm working sets, soft rows, a factor that left the inverse representation -- is solved by the
            // launch behind it, which holds the whole factor in LDS
            HIPCHK(hipMemsetAsync(b->d.wg_counter, 0, sizeof(int), b->stream));
            wg_kernel_t kt = b->wg_C == 2 ? k_ldp_wg<2, false, true> : k_ldp_wg<4, false, true>;
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kt), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_wg_tier));
            hipLaunchKernelGGL(kt, dim3(b->wg_tier_grid), dim3(256), b->lds_wg_tier, b->stream, dd, mode);
            HIPCHK(hipGetLastError());
            wg_mode = mode | 4;
        }
        HIPCHK(hipMemsetAsync(b->d.wg_counter, 0, sizeof(int), b->stream));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kw), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_wg));
        hipLaunchKernelGGL(kw, dim3(b->wg_grid), dim3(64 * b->wg_W), b->lds_wg, b->stream, dd, wg_mode);
        HIPCHK(hipGetLastError());
        ldp_kernel_t kf = pick_ldp(b);
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_ldp));
        hipLaunchKernelGGL(kf, dim3(b->d.N), dim3(64), b->lds_ldp, b->stream, b->d, mode | 4);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (b->NB > 0) {
        // problems that run the proximal outer loop keep the reference's arithmetic in both modes (their setup passes do as well)
        const bool exact_kernels = b->d.exact_setup != 0 || b->in_prox_loop || b->exact_sticky;
        ldp_reg_kernel_t kr = pick_ldp_reg(b, exact_kernels);
        // the descriptor travels through device memory: stream-ordered copy, then the launch
        (void)descriptor_changed;
        if (push_descriptor(b)) return DAQP_EXIT_UNSUPPORTED;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_ldp));
        bool use_img = b->img32 && !exact_kernels;
        if (use_img && b->img_ho_pin && (long long)*b->img_ho_pin * 2 > b->d.N && (++b->img_skipped & 15) != 0) use_img = false;
        if (use_img && (mode & 3) == 2 && b->d.N < b->img_min_warm) use_img = false;      // (a warm step is short: the crossover sits higher)
        if (use_img) {
            // the image kernel first (two waves per SIMD); the problems whose working set outgrows its LDS are flagged and solved, from the
            // state they were stored in, by the full-register kernel right behind (mode | 4: flagged problems only)
            ldp_reg_kernel_t ki = pick_ldp_reg_img(b);
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(ki), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(b->lds_img > b->lds_img_warm ? b->lds_img : b->lds_img_warm)));
            const bool warm = (mode & 3) == 2 && b->img_rows_warm > 0;
            const size_t lds = warm ? b->lds_img_warm : b->lds_img;
            if (b->d.img_ho) HIPCHK(hipMemsetAsync(b->d.img_ho, 0, sizeof(int), b->stream));
            HIPCHK(hipMemsetAsync(b->d.fallback, 0, (size_t)b->d.N * sizeof(int), b->stream));      // (the kernels only ever SET a flag)
            hipLaunchKernelGGL(ki, dim3(b->d.N), dim3(64), lds, b->stream, (const BatchDev *)b->d_dev, warm ? (mode | b->img_rows_warm << 16 | b->img_cache_warm << 22) : mode);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(kr, dim3(b->d.N), dim3(64), b->lds_ldp, b->stream, (const BatchDev *)b->d_dev, mode | 4);
            HIPCHK(hipGetLastError());
            if (b->d.img_ho && (mode & 3) != 1) HIPCHK(hipMemcpyAsync(b->img_ho_pin, b->d.img_ho, sizeof(int), hipMemcpyDeviceToHost, b->stream));
            return 0;
        }
        if (b->reg_handover && b->img_only && mode == 0 && !exact_kernels) {
            // n = 64 in the default arithmetic: the image-only kernel (64 rows held), k_ldp behind it for the problems that need a 65th row
            ldp_reg_kernel_t ki = pick_img_only(b);
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(ki), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_img));
            HIPCHK(hipMemsetAsync(b->d.fallback, 0, (size_t)b->d.N * sizeof(int), b->stream));
            hipLaunchKernelGGL(ki, dim3(b->d.N), dim3(64), b->lds_img, b->stream, (const BatchDev *)b->d_dev, mode);
            HIPCHK(hipGetLastError());
            ldp_kernel_t kf = pick_ldp(b);
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_fb));
            hipLaunchKernelGGL(kf, dim3(b->d.N), dim3(64), b->lds_fb, b->stream, b->d, mode | 4);
            HIPCHK(hipGetLastError());
            return 0;
        }
        if (b->reg_handover) HIPCHK(hipMemsetAsync(b->d.fallback, 0, (size_t)b->d.N * sizeof(int), b->stream));
        hipLaunchKernelGGL(kr, dim3(b->d.N), dim3(64), b->lds_ldp, b->stream, (const BatchDev *)b->d_dev, mode);
        HIPCHK(hipGetLastError());
        if (b->reg_handover) {   // the problems it flagged (mode | 4: nobody else is touched; an empty pass costs a few microseconds)
            if ((mode & 3) == 2) { set_err("fused update + solve launch on a hand-over shape"); return DAQP_EXIT_UNSUPPORTED; }
            BatchDev dd = b->d;
            if (b->in_prox_loop || b->exact_sticky) dd.exact_setup = 1;
            ldp_kernel_t kf = pick_ldp(b);
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_fb));
            hipLaunchKernelGGL(kf, dim3(b->d.N), dim3(64), b->lds_fb, b->stream, dd, mode | 4);
            HIPCHK(hipGetLastError());
        }
        return 0;
    }
    ldp_kernel_t k = pick_ldp(b);
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_ldp));
    if (b->img_only && mode == 0 && !b->d.exact_setup && !b->in_prox_loop && !b->exact_sticky) {
        // The shapes whose M fits no register file but whose fp32 IMAGE does (kImgOnlyShapes): the image kernel alone at one wave per SIMD --
        // the scan runs out of registers instead of streaming 8 (m - ms) n bytes per iteration from L2 / HBM --, and this kernel behind it for
        // whatever it flags (mode | 4; the image kernel holds every working-set row the shape can have, so that is an empty pass)
        if (push_descriptor(b)) return DAQP_EXIT_UNSUPPORTED;
        ldp_reg_kernel_t ki = pick_img_only(b);
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(ki), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_img));
        if (b->d.img_ho) HIPCHK(hipMemsetAsync(b->d.img_ho, 0, sizeof(int), b->stream));
        HIPCHK(hipMemsetAsync(b->d.fallback, 0, (size_t)b->d.N * sizeof(int), b->stream));
        hipLaunchKernelGGL(ki, dim3(b->d.N), dim3(64), b->lds_img, b->stream, (const BatchDev *)b->d_dev, mode);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k, dim3(b->d.N), dim3(64), b->lds_ldp, b->stream, b->d, mode | 4);
        HIPCHK(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(k, dim3(b->d.N), dim3(64), b->lds_ldp, b->stream, b->d, mode);
    HIPCHK(hipGetLastError());
    return 0;
}

// a deferred daqp_batch_update is applied now by the stand-alone kernel (someone wants to see the state before a solve)
int flush_update(DAQPBatch *b)
{
    if (!b->pending_mask) return 0;
    const int mask = b->pending_mask;
    b->pending_mask = 0;
    hipLaunchKernelGGL(k_update, dim3(b->d.N), dim3(64), b->lds_update, b->stream, b->d, mask);
    HIPCHK(hipGetLastError());
    return launch_ldp(b, 1);
}

// copy (host) or adopt (device) one input array
// a staging slot that can hold `count` elements (grown when a later call needs more: setup_shared stages ONE H / A,
// a later per-problem setup N of them)
template <typename T>
int slot_reserve(DAQPBatch *b, T **slot, size_t *cap, size_t count)
{
    if (*slot != nullptr && *cap >= count) return 0;
    if (*slot != nullptr) {
        HIPCHK(hipStreamSynchronize(b->stream));   // nothing in flight may still read the old buffer
        for (auto &o : b->owned) if (o == *slot) { o = b->owned.back(); b->owned.pop_back(); break; }
        (void)hipFree(*slot);
        b->bytes -= *cap * sizeof(T);
        *slot = nullptr; *cap = 0;
    }
    if (dev_alloc(b, slot, count)) return DAQP_EXIT_UNSUPPORTED;
    *cap = count ? count : 1;
    return 0;
}
template <typename T>
int stage(DAQPBatch *b, const T *src, int memory, size_t count, T **slot, size_t *cap, const T **out)
{
    if (src == nullptr) { *out = nullptr; return 0; }
    if (memory == DAQP_MEM_DEVICE) { *out = src; return 0; }
    if (slot_reserve(b, slot, cap, count)) return DAQP_EXIT_UNSUPPORTED;
    HIPCHK(hipMemcpyAsync(*slot, src, count * sizeof(T), hipMemcpyHostToDevice, b->stream));
    *out = *slot;
    return 0;
}

// ---- singular Hessians: regularising setup passes and the proximal outer loop (prox.hip.h) -------------------------
// the one-wave setup kernel of n <= 64 by size class; fused: the default arithmetic (fused multiply-adds in the factorisation sweep)
typedef void (*setup_kernel_t)(BatchDev, int);
setup_kernel_t pick_setup_fast(int n, bool fused)
{
    if (fused) return (n <= 16) ? k_setup_fast<16, false, true> : (n <= 32 ? k_setup_fast<32, false, true> : (n <= 56 ? k_setup_fast<56, false, true> : k_setup_fast<64, false, true>));
    return (n <= 16) ? k_setup_fast<16> : (n <= 32 ? k_setup_fast<32> : (n <= 56 ? k_setup_fast<56> : k_setup_fast<64>));
}

// k_setup<..., PART>: daqp_update_ldp masks that are neither within v|d nor everything (kernels.hip.h).  The generic one-wave kernel
// serves every shape here (these updates are not on the path the configurations are timed on), always in the reference's arithmetic.
setup_kernel_t pick_setup_part(const DAQPBatch *b)
{
    return b->setup_spill ? (b->d.n > 256 ? k_setup<true, 8, false, true> : k_setup<true, 4, false, true>) : k_setup<false, 4, false, true>;
}
int read_counters(DAQPBatch *b)
{
    HIPCHK(hipMemcpyAsync(b->counter_host, b->px.counter, 4 * sizeof(int), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}
int prox_buffers(DAQPBatch *b)
{
    if (b->prox_ready) return 0;
    const size_t N = b->d.N, n = b->d.n;
    ProxDev &p = b->px;
    int rc = 0;
    rc |= dev_alloc(b, &p.eps, N); rc |= dev_alloc(b, &p.hshift, N); rc |= dev_alloc(b, &p.tries, N);
    rc |= dev_alloc(b, &p.center, N * n); rc |= dev_alloc(b, &p.xold, N * n); rc |= dev_alloc(b, &p.feff, N * n);
    rc |= dev_alloc(b, &p.state, N * 4); rc |= dev_alloc(b, &p.saved_flag, N);
    rc |= dev_alloc(b, &p.t_flag, N); rc |= dev_alloc(b, &p.t_iter, N); rc |= dev_alloc(b, &p.t_fval, N); rc |= dev_alloc(b, &p.t_soft, N);
    rc |= dev_alloc(b, &b->d.prox_mask, N * n);
    if (rc) return DAQP_EXIT_UNSUPPORTED;
    HIPCHK(hipMemsetAsync(p.center, 0, N * n * sizeof(double), b->stream));   // api.c:318: the first centre is the origin
    HIPCHK(hipMemsetAsync(p.state, 0, N * 4 * sizeof(int), b->stream));
    b->d.hshift = p.hshift;
    b->prox_ready = true;
    return 0;
}
// After the first setup pass: problems whose Hessian Cholesky found singular (or all of them when eps_prox > 0) are set up
// again from H + eps*I, eps doubling while the shifted factor is still ill-conditioned (utils.c:354-377).
// the count of problems that the first setup pass flagged as singular: launched here, looked at later (resolve_setup)
int count_flagged_async(DAQPBatch *b, int mask)
{
    BatchDev &d = b->d;
    const int tpb = 128, nb = (d.N + tpb - 1) / tpb;
    b->n_prox_qps = 0;
    b->px.lp = 0;
    HIPCHK(hipMemsetAsync(b->px.counter, 0, 4 * sizeof(int), b->stream));
    hipLaunchKernelGGL(k_prox_shift, dim3(nb), dim3(tpb), 0, b->stream, d, b->px, 2);
    HIPCHK(hipGetLastError());
    if (!b->pin_count) HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&b->pin_count), 4 * sizeof(int), hipHostMallocDefault));
    if (!b->ev_count) HIPCHK(hipEventCreateWithFlags(&b->ev_count, hipEventDisableTiming));
    HIPCHK(hipMemcpyAsync(b->pin_count, b->px.counter, 4 * sizeof(int), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipEventRecord(b->ev_count, b->stream));
    b->reg_pending = true; b->reg_mask = mask;
    return 0;
}
// `d`: the batch's own descriptor, or the one-problem descriptor of a shared factorisation (daqp_batch_setup_shared)
int regularise(DAQPBatch *b, BatchDev &d, int mask, bool lp, bool counted = false)
{
    const int tpb = 128, nb = (d.N + tpb - 1) / tpb;
    b->n_prox_qps = 0;
    b->px.lp = lp ? 1 : 0;
    if (!lp && !counted) {
        HIPCHK(hipMemsetAsync(b->px.counter, 0, 4 * sizeof(int), b->stream));
        hipLaunchKernelGGL(k_prox_shift, dim3(nb), dim3(tpb), 0, b->stream, d, b->px, 2);
        HIPCHK(hipGetLastError());
        if (read_counters(b)) return DAQP_EXIT_UNSUPPORTED;
        if (b->counter_host[0] == 0) return 0;
    }
    if (prox_buffers(b)) return DAQP_EXIT_UNSUPPORTED;
    d.hshift = b->d.hshift; d.prox_mask = b->d.prox_mask;
    hipLaunchKernelGGL(k_prox_shift, dim3(nb), dim3(tpb), 0, b->stream, d, b->px, lp ? 3 : 0);
    HIPCHK(hipGetLastError());
    // the shifted passes keep M in the reference's operation order whatever the arithmetic mode of the batch (their
    // problems are then bit-identical to the reference in both modes); an LP's one pass is the generic kernel's diagonal path
    const bool gs = b->setup_spill;
    setup_kernel_t ks = gs ? (d.n > 256 ? k_setup<true, 8> : k_setup<true>) : k_setup<false>;
    size_t lds = (size_t)setup_lds(d.n, d.m, gs).total_bytes;
    if (b->part_mask && !lp) ks = pick_setup_part(b);       // (a partial update's flagged problems: the kernel that knows which sense / bounds it took)
    else if (b->fast_setup && !lp) {
        ks = (d.n <= 16) ? k_setup_fast<16, true> : (d.n <= 32 ? k_setup_fast<32, true> : (d.n <= 56 ? k_setup_fast<56, true> : k_setup_fast<64, true>));
        lds = (size_t)fast_lds(d.n, d.m, 1, d.mA).total_bytes;
    }
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int exact_saved = d.exact_setup;
    for (int pass = 0; pass < 18; ++pass) {
        d.prox_pass = lp ? 2 : 1;
        d.exact_setup = 1;
        hipLaunchKernelGGL(ks, dim3(d.N), dim3(64), lds, b->stream, d, mask & ~DAQP_UPDATE_unconstrained);   // utils.c:622
        d.prox_pass = 0;
        d.exact_setup = exact_saved;
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemsetAsync(b->px.counter, 0, 4 * sizeof(int), b->stream));
        hipLaunchKernelGGL(k_prox_shift, dim3(nb), dim3(tpb), 0, b->stream, d, b->px, 1);
        HIPCHK(hipGetLastError());
        if (read_counters(b)) return DAQP_EXIT_UNSUPPORTED;
        if (b->counter_host[0] == 0) break;
    }
    HIPCHK(hipMemsetAsync(b->px.counter, 0, 4 * sizeof(int), b->stream));
    hipLaunchKernelGGL(k_prox_final, dim3(nb), dim3(tpb), 0, b->stream, d, b->px);
    HIPCHK(hipGetLastError());
    if (read_counters(b)) return DAQP_EXIT_UNSUPPORTED;
    b->n_prox_qps = b->counter_host[1];
    return 0;
}
// the generic setup's general rows as their own launch right behind k_setup (setup_m.hip.h): default arithmetic, n <= 208, not an LP /
// regularising pass (those keep the reference's arithmetic inside k_setup)
bool defers_m(const DAQPBatch *b, const BatchDev &d)
{
    static const bool off = [] { const char *e = getenv("DAQP_AMD_NO_SETUP_M"); return e && atoi(e) != 0; }();
    return !off && !b->fast_setup && !d.exact_setup && d.setup_sq != nullptr && d.m_tick != nullptr && d.n <= 208 && d.mA > 0 && d.prox_pass == 0;
}
int launch_setup_m(DAQPBatch *b, const BatchDev &d)
{
    const int nrb = (d.mA + kSetupMRows - 1) / kSetupMRows;
    const size_t lds = (size_t)setup_m_lds().total_bytes;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_setup_m), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_setup_m, dim3((unsigned)((size_t)((d.N + 7) / 8) * nrb * 8)), dim3(256), lds, b->stream, d, nrb);   // (whole groups of eight problems: the XCD-aware mapping in the kernel)
    HIPCHK(hipGetLastError());
    return 0;
}
// daqp_update_ldp(mask within UPDATE_v|UPDATE_d) followed by daqp_solve, whichever kernels the shape uses
int launch_update_solve(DAQPBatch *b, int mask, bool descriptor_changed)
{
    if (b->NB > 0 && !b->reg_handover) return launch_ldp(b, 2 | (mask << 4), descriptor_changed);
    hipLaunchKernelGGL(k_update, dim3(b->d.N), dim3(64), b->lds_update, b->stream, b->d, mask);
    HIPCHK(hipGetLastError());
    if (launch_ldp(b, 1)) return DAQP_EXIT_UNSUPPORTED;
    return launch_ldp(b, 0);
}
// daqp_solve for a batch that holds proximal problems: the ordinary ones are solved by one launch as usual, then the
// outer iterations of daqp_prox.c:60-198 run for the others until each has stopped.
// ordinary_done: the ordinary problems have been solved already (the solve launch that went out before the host knew of any
// singular Hessian: resolve_setup)
int solve_with_prox(DAQPBatch *b, int mode, bool ordinary_done = false)
{
    BatchDev &d = b->d;
    const int tpb = 128, nb = (d.N + tpb - 1) / tpb;
    hipLaunchKernelGGL(k_prox_mark, dim3(nb), dim3(tpb), 0, b->stream, d, b->px, 0);
    HIPCHK(hipGetLastError());
    if (!ordinary_done && b->n_prox_qps < d.N) { if (launch_ldp(b, mode)) return DAQP_EXIT_UNSUPPORTED; }
    hipLaunchKernelGGL(k_prox_mark, dim3(nb), dim3(tpb), 0, b->stream, d, b->px, 1);
    HIPCHK(hipGetLastError());
    // the inner launches report into scratch; a problem's outputs are written when its loop ends
    const double *f_user = d.f;
    double *o_fval = d.fval, *o_soft = d.soft;
    int *o_flag = d.exitflag, *o_iter = d.iter;
    d.f = b->px.feff; d.fval = b->px.t_fval; d.soft = b->px.t_soft; d.exitflag = b->px.t_flag; d.iter = b->px.t_iter;
    const ProxOut po = {f_user, d.lam, o_fval, o_soft, o_flag, o_iter};
    const size_t lds_grad = (size_t)ldp_lds(d.n, d.m, d.cap, b->spill).total_bytes;
    typedef void (*grad_kernel_t)(BatchDev, ProxDev, double *, ProxOut);
    const grad_kernel_t kgrad = b->C == 8 ? k_lp_gradient<8, true> : (b->spill ? k_lp_gradient<4, true> : k_lp_gradient<4, false>);
    if (b->px.lp) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kgrad), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_grad));
    int rc = 0, outer = 0;
    b->in_prox_loop = true;
    hipLaunchKernelGGL(k_prox_pre, dim3(d.N), dim3(64), 0, b->stream, d, b->px, f_user);
    if (hipGetLastError() != hipSuccess) rc = 1;
    for (; !rc; ++outer) {
        if (hipMemsetAsync(b->px.counter, 0, 4 * sizeof(int), b->stream) != hipSuccess) { rc = 1; break; }
        if (launch_update_solve(b, DAQP_UPDATE_v | DAQP_UPDATE_d, outer == 0)) { rc = 1; break; }
        // the fixed-point test; problems that go on get their next input here (or from the gradient step below)
        hipLaunchKernelGGL(k_prox_post, dim3(d.N), dim3(64), 0, b->stream, d, b->px, (const double *)d.x, po);
        if (hipGetLastError() != hipSuccess) { rc = 1; break; }
        if (read_counters(b)) { rc = 1; break; }
        if (b->counter_host[2] == 0) break;
        if (b->counter_host[3] > 0) {   // LP iterates off a vertex walk to the next constraint
            hipLaunchKernelGGL(kgrad, dim3(d.N), dim3(64), lds_grad, b->stream, d, b->px, d.x, po);
            if (hipGetLastError() != hipSuccess) { rc = 1; break; }
        }
    }
    b->prox_outer = outer + 1;
    b->in_prox_loop = false;
    d.f = f_user; d.fval = o_fval; d.soft = o_soft; d.exitflag = o_flag; d.iter = o_iter;
    hipLaunchKernelGGL(k_prox_mark, dim3(nb), dim3(tpb), 0, b->stream, d, b->px, 2);
    HIPCHK(hipGetLastError());
    if (rc) { set_err("proximal outer loop: a launch failed"); return DAQP_EXIT_UNSUPPORTED; }
    return 0;
}

// The count started by the last daqp_batch_setup is looked at now; if the first pass flagged singular Hessians, their regularising
// setup passes and their activation pass run here (utils.c:354-377).  Every entry point that needs final setup flags comes through.
// had_flagged: whether the first pass had flagged any problem (the caller's early solve launch then reported the internal code for them)
int resolve_setup(DAQPBatch *b, bool *had_flagged = nullptr)
{
    if (had_flagged) *had_flagged = false;
    if (!b->reg_pending) return 0;
    HIPCHK(hipEventSynchronize(b->ev_count));
    if (b->pin_count[0] == 0) { b->reg_pending = false; return 0; }
    if (had_flagged) *had_flagged = true;
    int rc = regularise(b, b->d, b->reg_mask, false, true);
    if (!rc) rc = launch_ldp(b, 1);
    // a failed pass leaves problems with the internal code and nothing that would ever resolve it: the batch is not set up
    if (rc) { b->is_setup = false; b->reg_pending = false; return rc; }
    b->reg_pending = false;
    return 0;
}

// recheck.hip.h: after the solve launches of a default-mode solve that directly follows a setup.  One 4-byte read-back; nothing
// else happens unless some problem was declared infeasible.
int batch_setup(DAQPBatch *b, const DAQPBatchProblem *p, int init_mask, bool fresh);
void destroy_batch(DAQPBatch *b);
// one problem: the same workspace again, in the exact mode; its inputs are still where the setup read them
int redo_one_exact(DAQPBatch *b)
{
    BatchDev &d = b->d;
    DAQPBatchProblem pp;
    pp.N = 1; pp.n = d.n; pp.m = d.m; pp.ms = d.ms;
    pp.H = const_cast<double *>(d.H); pp.f = const_cast<double *>(d.f); pp.A = const_cast<double *>(d.A);
    pp.bupper = const_cast<double *>(d.bu); pp.blower = const_cast<double *>(d.bl); pp.sense = const_cast<int *>(d.sense_in);
    pp.memory = DAQP_MEM_DEVICE;
    d.exact_setup = 1;
    b->quiet_setup = true;             // (daqp_batch_kernel_ms keeps reporting the caller's setup)
    const bool lean = b->lean;
    b->lean = false;
    int rc = batch_setup(b, &pp, b->fresh_mask, false);
    b->quiet_setup = false;
    b->lean = lean;
    if (!rc) rc = launch_ldp(b, 0);
    if (!rc && b->reg_pending) rc = resolve_setup(b);      // (the count of singular Hessians of this pass: none, the first pass had none)
    d.exact_setup = 0;
    b->rechecked = 1;
    // batch_setup armed the second pass again (fresh = true): this WAS the second pass, and the solve it belongs to is over -- a further
    // daqp_solve on this workspace continues from the stored working set, as the reference does, instead of re-running setup + solve
    b->fresh = false;
    return rc;
}
int recheck_infeasible(DAQPBatch *b)
{
    BatchDev &d = b->d;
    b->rechecked = 0;
    if (d.N == 1 && b->mapped_out && d.x == b->ox) { b->recheck_due = true; return 0; }     // (the host sees the verdict itself when the results are there)
    if (!b->redo_list) {
        if (dev_alloc(b, &b->redo_list, (size_t)d.N) || dev_alloc(b, &b->redo_count, 2)) return DAQP_EXIT_UNSUPPORTED;
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&b->pin_redo), sizeof(int), hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&b->pin_redo_dev), b->pin_redo, 0));
    }
    HIPCHK(hipMemsetAsync(b->redo_count, 0, 2 * sizeof(int), b->stream));
    __atomic_store_n(b->pin_redo, -1, __ATOMIC_RELEASE);
    hipLaunchKernelGGL(k_mark_infeasible, dim3((d.N + 127) / 128), dim3(128), 0, b->stream, d, b->redo_list, b->redo_count, b->pin_redo_dev);
    HIPCHK(hipGetLastError());
    {   // poll the mapped word (written by the kernel's last block); a stream that has failed is noticed through hipStreamQuery
        unsigned spins = 0;
        while (__atomic_load_n(b->pin_redo, __ATOMIC_ACQUIRE) < 0) {
            if ((spins & 0xff) == 0xff) sched_yield();          // (one core per shard thread otherwise spins flat out while the solve kernel runs)
            if ((++spins & 0x3fff) == 0) {
                const hipError_t qe = hipStreamQuery(b->stream);
                if (qe != hipSuccess && qe != hipErrorNotReady) { set_err("the solve launch failed: %s", hipGetErrorString(qe)); return DAQP_EXIT_UNSUPPORTED; }
                if (qe == hipSuccess && __atomic_load_n(b->pin_redo, __ATOMIC_ACQUIRE) < 0) {   // (everything has run and the word never arrived)
                    HIPCHK(hipMemcpy(b->pin_redo, b->redo_count, sizeof(int), hipMemcpyDeviceToHost));
                    break;
                }
            }
        }
    }
    const int count = *b->pin_redo;
    if (count == 0) return 0;
    if (!b->ev_r[0]) { HIPCHK(hipEventCreate(&b->ev_r[0])); HIPCHK(hipEventCreate(&b->ev_r[1])); }
    HIPCHK(hipEventRecord(b->ev_r[0], b->stream));
    struct Stamp { DAQPBatch *b; ~Stamp() { if (hipEventRecord(b->ev_r[1], b->stream) == hipSuccess) b->timed_recheck = true; } } stamp{b};
    if (d.N == 1) return redo_one_exact(b);
    int cap = 2;
    while (cap < count) cap *= 2;
    if (cap > d.N) cap = d.N;
    if (b->redo && b->redo->d.N < cap) { b->redo->stream = nullptr; destroy_batch(b->redo); b->redo = nullptr; }
    if (!b->redo) {
        DAQPBatch *rb = nullptr;
        // no memory for the companion: the first pass's verdicts stand (a valid solve does not fail over its second opinion)
        if (daqp_batch_create(&rb, cap < 2 ? 2 : cap, d.n, d.m, d.ms, b->ns_max, &d.st, b->device)) { (void)hipGetLastError(); b->rechecked = -1; return 0; }
        rb->recheck = false;
        b->redo = rb;
    }
    DAQPBatch *rb = b->redo;
    BatchDev &rd = rb->d;
    rb->stream = b->stream;
    rd.exact_setup = 1;
    rd.st = d.st;
    if (put_async(rb->st_dev, rd.st, rb->stream)) return DAQP_EXIT_UNSUPPORTED;
    if (d.trace && (!rd.trace || rd.trace_cap != d.trace_cap)) { if (daqp_batch_enable_trace(rb, d.trace_cap)) return DAQP_EXIT_UNSUPPORTED; }
    if (!d.trace) { rd.trace = nullptr; rd.trace_cap = 0; }
    const size_t R = rd.N, n = d.n, m = d.m;
    int rc = 0;
    rc |= slot_reserve(rb, &rb->sH, &rb->nH, R * n * n); rc |= slot_reserve(rb, &rb->sf, &rb->nf, R * n);
    rc |= slot_reserve(rb, &rb->sA, &rb->nA, R * d.mA * n); rc |= slot_reserve(rb, &rb->sbu, &rb->nbu, R * m);
    rc |= slot_reserve(rb, &rb->sbl, &rb->nbl, R * m);
    if (d.sense_in) rc |= slot_reserve(rb, &rb->ssense, &rb->nsense, R * m);
    if (rc) { (void)hipGetLastError(); b->rechecked = -1; return 0; }
    hipLaunchKernelGGL(k_gather_problems, dim3(rd.N), dim3(256), 0, b->stream, d, (const int *)b->redo_list, (const int *)b->redo_count,
                       rb->sH, rb->sf, rb->sA, rb->sbu, rb->sbl, d.sense_in ? rb->ssense : (int *)nullptr);
    HIPCHK(hipGetLastError());
    DAQPBatchProblem pp;
    pp.N = rd.N; pp.n = d.n; pp.m = d.m; pp.ms = d.ms;
    pp.H = rb->sH; pp.f = rb->sf; pp.A = rb->sA; pp.bupper = rb->sbu; pp.blower = rb->sbl; pp.sense = d.sense_in ? rb->ssense : nullptr;
    pp.memory = DAQP_MEM_DEVICE;
    rc = daqp_batch_setup(rb, &pp, b->fresh_mask);
    if (rc) return rc;
    DAQPBatchResult rr;
    memset(&rr, 0, sizeof(rr));
    rr.memory = DAQP_MEM_DEVICE;
    rc = daqp_batch_solve(rb, &rr);
    if (rc) return rc;
    hipLaunchKernelGGL(k_scatter_problems, dim3(count), dim3(256), 0, b->stream, d, rd, (const int *)b->redo_list, (const int *)b->redo_count);
    HIPCHK(hipGetLastError());
    b->rechecked = count;
    return 0;
}

// the second pass reads the setup's inputs again.  Host inputs were staged into the batch's own buffers: always there.  Device inputs
// were adopted: the caller says whether they are still what the setup read (daqp_batch_set_recheck), or the process does
// (DAQP_AMD_RECHECK_DEVICE=0: never for adopted inputs); unsaid, they are taken to be (include/daqp_amd.h spells the contract out).
bool recheck_allowed(const DAQPBatch *b)
{
    if (!b->inputs_adopted) return true;
    if (b->recheck_device >= 0) return b->recheck_device != 0;
    static const bool off = [] { const char *e = getenv("DAQP_AMD_RECHECK_DEVICE"); return e && atoi(e) == 0; }();
    return !off;
}
bool blk_setup_enabled()
{
    static const bool off = [] { const char *e = getenv("DAQP_AMD_NO_BLK_SETUP"); return e && atoi(e) != 0; }();
    return !off;
}
int check_problem(const DAQPBatch *b, const DAQPBatchProblem *p)
{
    if (!b || !p) { set_err("null batch or problem"); return DAQP_EXIT_UNSUPPORTED; }
    if (p->N != b->d.N || p->n != b->d.n || p->m != b->d.m || p->ms != b->d.ms) {
        set_err("problem shape (N=%d n=%d m=%d ms=%d) does not match the batch (N=%d n=%d m=%d ms=%d)", p->N, p->n, p->m,
                p->ms, b->d.N, b->d.n, b->d.m, b->d.ms);
        return DAQP_EXIT_UNSUPPORTED;
    }
    return 0;
}

} // namespace

namespace {
// Parked single-problem batches.  daqp_quadprog creates and frees a workspace per call (api.c:61-104); on the GPU that is
// ~40 hipMalloc / hipFree and four events per call, an order of magnitude more than the solve itself.  A freed batch of ONE
// problem is parked here instead and handed to the next create with the same shape, device and environment switches.
std::mutex g_pool_mu;
std::atomic<unsigned long long> g_devices_used{0};
std::vector<DAQPBatch *> g_pool;
constexpr size_t kPoolMax = 8;
bool pool_enabled() { const char *e = getenv("DAQP_AMD_NO_POOL"); return !(e && atoi(e) != 0); }
std::string env_signature()
{
    static const char *names[] = {"DAQP_AMD_LDS_LIMIT", "DAQP_AMD_FORCE_SPILL", "DAQP_AMD_STREAM_M", "DAQP_AMD_NO_WG", "DAQP_AMD_WG_WAVES",
                                  "DAQP_AMD_WG_CAPL", "DAQP_AMD_WG_GRID", "DAQP_AMD_NO_WG_TIER", "DAQP_AMD_WG_R0", "DAQP_AMD_WG_TIER_GRID", "DAQP_AMD_WG_TIER_MIN_BATCH", "DAQP_AMD_SLOW_SETUP", "DAQP_AMD_NO_SCAN32", "DAQP_AMD_WG_INVERSE", "DAQP_AMD_NO_TINY_SETUP", "DAQP_AMD_NO_RECHECK", "DAQP_AMD_NO_SETUP_M", "DAQP_AMD_NO_BLK_SETUP",
                                  "DAQP_AMD_REG_ROWS", "DAQP_AMD_NO_REG_HANDOVER", "DAQP_AMD_NO_FACT_WG", "DAQP_AMD_NO_FACT_SMALL", "DAQP_AMD_NO_IMG32", "DAQP_AMD_IMG_ROWS", "DAQP_AMD_IMG_MIN_BATCH", "DAQP_AMD_IMG_WAVES", "DAQP_AMD_IMG_CACHE", "DAQP_AMD_IMG_WARM_ROWS", "DAQP_AMD_IMG_WARM_WAVES", "DAQP_AMD_NO_IMG_ONLY", "DAQP_AMD_IMG_ONLY_MIN_BATCH", "DAQP_AMD_NO_BLK_BOUNDS"};
    std::string k;
    for (const char *nme : names) { const char *v = getenv(nme); k += v ? v : "-"; k += '|'; }
    return k;
}
void destroy_batch(DAQPBatch *b)
{
    (void)hipSetDevice(b->device);
    (void)hipStreamSynchronize(b->stream);
    if (b->redo) { b->redo->stream = nullptr; destroy_batch(b->redo); b->redo = nullptr; }
    if (b->pin_redo) (void)hipHostFree(b->pin_redo);
    if (b->img_ho_pin) (void)hipHostFree(b->img_ho_pin);
    for (void *p : b->owned) (void)hipFree(p);
    if (b->pin_in) (void)hipHostFree(b->pin_in);
    if (b->pin_out) (void)hipHostFree(b->pin_out);
    if (b->pin_mir) (void)hipHostFree(b->pin_mir);
    if (b->pin_sig) (void)hipHostFree(b->pin_sig);
    if (b->pin_upd) (void)hipHostFree(b->pin_upd);
    if (b->ev_in) (void)hipEventDestroy(b->ev_in);
    if (b->ev_count) (void)hipEventDestroy(b->ev_count);
    if (b->pin_count) (void)hipHostFree(b->pin_count);
    for (auto &e : b->ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : b->ev_r) if (e) (void)hipEventDestroy(e);
    delete b;
}
} // namespace

extern "C" {

const char *daqp_amd_last_error(void) { return g_err; }
#ifdef DAQP_AMD_FEW_VARIANTS
const char *daqp_amd_version(void) { return "daqp_amd 0.4 (gfx950, fp64: one wavefront per QP, one workgroup per QP beyond 64 working-set rows, up to 512 rows) [dev build: few template instantiations]"; }
#else
const char *daqp_amd_version(void) { return "daqp_amd 0.4 (gfx950, fp64: one wavefront per QP, one workgroup per QP beyond 64 working-set rows, up to 512 rows)"; }
#endif
int daqp_amd_has_tiny(void) { return 0; }   // (the sixteen-problems-per-wave SOLVE kernel of rounds 3-4 was measured slower than the register kernel and is gone; the symbol stays for callers that asked)
int daqp_amd_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}
void daqp_default_settings(DAQPSettings *settings) { default_settings(settings); }

int daqp_batch_create(DAQPBatch **out, int N, int n, int m, int ms, int ns_max, const DAQPSettings *settings, int device)
{
    if (!out) return DAQP_EXIT_UNSUPPORTED;
    *out = nullptr;
    if (N <= 0 || n <= 0 || m < 0 || ms < 0 || ms > m || ms > n || ns_max < 0) {
        set_err("bad dimensions N=%d n=%d m=%d ms=%d ns=%d", N, n, m, ms, ns_max);
        return DAQP_EXIT_UNSUPPORTED;
    }
    const int cap = n + ns_max + 1;
    if (cap > 512 || n > 511) { set_err("n + ns + 1 = %d exceeds the 512-row working-set limit of this build", cap); return DAQP_EXIT_UNSUPPORTED; }
    int ndev = 0;
    const hipError_t dev_rc = hipGetDeviceCount(&ndev);
    if (dev_rc == hipSuccess && ndev > 0 && device >= 0 && device < 64) {
        // (registered AFTER the HIP runtime initialised itself -- the call above --, so it runs BEFORE that runtime's own exit handlers)
        static std::once_flag once;
        std::call_once(once, [] { if (!getenv("DAQP_AMD_NO_EXIT_SYNC")) atexit(daqp_amd_shutdown); });
        g_devices_used.fetch_or(1ull << device);
    }
    if (dev_rc != hipSuccess || ndev == 0) {
        set_err("no HIP device: libdaqp_amd has no CPU path");
        return DAQP_EXIT_UNSUPPORTED;
    }
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    const std::string env_key = env_signature();
    if (N == 1 && pool_enabled()) {
        DAQPBatch *hit = nullptr;
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            for (size_t i = 0; i < g_pool.size(); ++i) {
                DAQPBatch *c = g_pool[i];
                if (c->device == device && c->d.n == n && c->d.m == m && c->d.ms == ms && c->ns_max == ns_max && c->env_key == env_key) {
                    hit = c; g_pool.erase(g_pool.begin() + i); break;
                }
            }
        }
        if (hit) {
            if (hipSetDevice(device) != hipSuccess) { destroy_batch(hit); set_err("hipSetDevice(%d) failed", device); return DAQP_EXIT_UNSUPPORTED; }
            BatchDev &hd = hit->d;
            if (settings) hd.st = *settings; else default_settings(&hd.st);
            const char *ex = getenv("DAQP_AMD_EXACT");
            hd.exact_setup = (ex && atoi(ex) != 0) ? 1 : 0;
            // the iterate of the previous owner: gone (a fresh batch starts from zeros too)
            if (put_async(hit->st_dev, hd.st, hit->stream) ||
                hipMemsetAsync(hd.vecs, 0, (size_t)5 * hd.cap * sizeof(double), hit->stream) != hipSuccess ||
                hipMemsetAsync(hd.qs, 0, sizeof(QState), hit->stream) != hipSuccess) { destroy_batch(hit); set_err("reset of a pooled workspace failed"); return DAQP_EXIT_UNSUPPORTED; }
            *out = hit;
            return 0;
        }
    }
    DAQPBatch *b = new DAQPBatch();
    b->device = device;
    b->env_key = env_key; b->ns_max = ns_max;
    { const char *nr = getenv("DAQP_AMD_NO_RECHECK"); b->recheck = !(nr && atoi(nr) != 0); }
    if (hipSetDevice(device) != hipSuccess) { set_err("hipSetDevice(%d) failed", device); delete b; return DAQP_EXIT_UNSUPPORTED; }
    BatchDev &d = b->d;
    d.N = N; d.n = n; d.m = m; d.ms = ms; d.cap = cap; d.mA = m - ms;
    d.npair = (n + 1) / 2; d.nblk = (m + 63) / 64; d.ldr = n | 1; d.nquad = (d.npair + 1) / 2;
    d.ltri = round_up(cap * (cap + 1) / 2, 2); d.rtri = n * (n + 1) / 2;   // even stride: 16-byte aligned rows for the direct HBM->LDS copy
    if (settings) d.st = *settings; else default_settings(&d.st);
    b->C = cap <= 64 ? 1 : (cap <= 128 ? 2 : (cap <= 256 ? 4 : 8));
#ifdef DAQP_AMD_FEW_VARIANTS
    if (b->C < 4) b->C = 4;
#endif
    const char *env = getenv("DAQP_AMD_LDS_LIMIT");
    const int lds_limit = env ? atoi(env) : 80 * 1024;
    b->spill = ldp_lds(n, m, cap, false).total_bytes > lds_limit;
    if (getenv("DAQP_AMD_FORCE_SPILL") || cap > 256) b->spill = true;
    if (!b->spill && cap <= 65 && !getenv("DAQP_AMD_STREAM_M"))
        for (const RegShape &rs : kRegShapes)
            if (d.nblk <= rs.nb && d.npair <= rs.np) { b->NB = rs.nb; b->NP = rs.np; break; }
    // one lane per working-set row: 64 rows.  n = 64 without soft rows may hold 65 (only while a 65th constraint is being exchanged at a full
    // vertex): the (2,32) shape takes it and hands the few problems that get there to k_ldp (launch_ldp); every other shape needs cap <= 64
    d.reg_rows = 64;
    if (const char *re = getenv("DAQP_AMD_REG_ROWS")) { const int v = atoi(re); if (v >= 1 && v < 64) d.reg_rows = v; }   // tests: force the hand-over
    if (b->NB > 0 && d.reg_rows < cap) {
        if (b->NB == 2 && b->NP == 32 && !getenv("DAQP_AMD_NO_REG_HANDOVER")) b->reg_handover = true;
        else if (cap > 64) { b->NB = 0; b->NP = 0; }
    }
    if (b->reg_handover) b->lds_fb = (size_t)ldp_lds(n, m, cap, b->spill).total_bytes;
    // The shapes whose M fills the register file (one wave per SIMD) run, in the default arithmetic and on batches that fill the device twice
    // over, as an fp32 IMAGE of M at two waves per SIMD (reg_kernel.hip.h, IMG = 1).  Its LDS holds img_rows working-set rows (C2: the peak is
    // 27 rows on average, above 40 on 1.3 % of the problems -- those are handed to the full-register kernel behind it)
    b->img_kind = (b->NB == 3 && b->NP == 25) ? (m <= 160 ? 2 : 1) : ((b->NB == 2 && b->NP == 32) ? 1 : 0);
    if (b->img_kind && cap <= 64 && !b->reg_handover && !getenv("DAQP_AMD_NO_IMG32")) {
        int min_batch = 10240, rows = 42;      // (tools/img_threshold.py: below ~10 000 problems -- warm launches: ~16 000, see launch_ldp -- the device is not full twice over and a problem's latency decides: one wave per SIMD is faster per problem)
        if (const char *e = getenv("DAQP_AMD_IMG_MIN_BATCH")) min_batch = atoi(e);
        if (const char *e = getenv("DAQP_AMD_IMG_ROWS")) { const int v = atoi(e); if (v >= 2 && v <= 64) rows = v; }
        if (getenv("DAQP_AMD_IMG_MIN_BATCH")) b->img_min_warm = min_batch;      // (the tests' override applies to every launch)
        if (N >= min_batch) { b->img32 = true; d.img_rows = rows < cap ? rows : cap; }
    }
    d.ldrc = 0;
    if (b->NB > 0) {   // stride == 2 (mod 4): rows 16-byte aligned and 16 consecutive rows hit 16 distinct 4-bank groups
        int l = n > 2 * b->NP ? n : 2 * b->NP;
        while ((l & 3) != 2) ++l;
        d.ldrc = l;
    }
    b->lds_ldp = b->NB > 0 ? (size_t)reg_lds_bytes(b->NB, n, m, cap, d.ldrc) : (size_t)ldp_lds(n, m, cap, b->spill, d.ldrc).total_bytes;
    // No register shape for (n, m), but the fp32 image of M fits one (kImgOnlyShapes: n <= 63 with m <= 256 ... 512, one wave per SIMD): the default
    // arithmetic's solves run on it (launch_ldp); everything else about the batch stays the generic one-wave path's.  Every working-set row the
    // shape can have is held (img_rows = cap), as many of them in LDS as leave four workgroups per CU, the rest in the scratch tier.
    // (n <= 16: M streamed is as fast -- n = 8, m = 256: 0.86 against 1.02 ms per 20 000 -- a scan is 16 column pairs of L2 hits)
    int io_nb = 0, io_np = 0;
    for (const RegShape &rs : kImgOnlyShapes)
        if (d.nblk <= rs.nb && d.npair <= rs.np) { io_nb = rs.nb; io_np = rs.np; break; }
    // (cap = 65 is n = 64 without soft rows: a 65th row exists only while a constraint is exchanged at a full vertex -- the image kernel holds 64
    //  and flags such a problem for k_ldp behind it, as the (2,32) registers do for m <= 128)
    // (n = 64 with m <= 128 keeps its (2,32) registers for the exact mode, fused updates and one-problem calls; the default arithmetic's plain
    //  solves of a batch take the image-only kernel there as well: 6.5 -> 4.0 ms per 20 000 at m = 128 -- the fp32 scan is half the issue slots)
    const bool n64 = b->NB == 2 && b->NP == 32 && b->reg_handover && N > 1;
    if ((b->NB == 0 || n64) && !b->spill && n > 16 && cap <= 65 && n <= 64 && io_nb > 0 && !getenv("DAQP_AMD_STREAM_M") && !getenv("DAQP_AMD_NO_IMG32") && !getenv("DAQP_AMD_NO_IMG_ONLY")) {
        int min_batch = 1;
        if (const char *e = getenv("DAQP_AMD_IMG_ONLY_MIN_BATCH")) min_batch = atoi(e);
        if (N >= min_batch) {
            b->img_only = true; b->io_nb = io_nb; b->io_np = io_np;
            int l = n > 2 * io_np ? n : 2 * io_np;
            while ((l & 3) != 2) ++l;
            d.ldrc = l;                      // (the row-cache stride of the register kernels; k_ldp has its own)
            d.img_rows = cap < 64 ? cap : 64;
            const int budget = (160 * 1024 / 4) / 512 * 512;
            int cache = d.img_rows;
            while (cache > 2 && reg_img_lds_bytes(io_nb, 1, n, m, d.img_rows, cache, d.ldrc) > budget) --cache;
            d.img_cache = cache;
            b->lds_img = (size_t)reg_img_lds_bytes(io_nb, 1, n, m, d.img_rows, cache, d.ldrc);
        }
    }
    if (b->img32) {
        // rows of the active-row cache in LDS: as many as leave `waves` workgroups per CU (the rest of a working set lives in rowc_g, L2-resident);
        // a workgroup's share of the CU's 160 KB is granted in 512-byte steps
        int waves = 8;
        if (const char *e = getenv("DAQP_AMD_IMG_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= 8) waves = v; }
        const int budget = (160 * 1024 / waves) / 512 * 512;
        int cache = d.img_rows;
        while (cache > 2 && reg_img_lds_bytes(b->NB, b->img_kind, n, m, d.img_rows, cache, d.ldrc) > budget) --cache;
        if (const char *e = getenv("DAQP_AMD_IMG_CACHE")) { const int v = atoi(e); if (v >= 1) cache = v < d.img_rows ? v : d.img_rows; }
        d.img_cache = cache;
        b->lds_img = (size_t)reg_img_lds_bytes(b->NB, b->img_kind, n, m, d.img_rows, cache, d.ldrc);
        if (!getenv("DAQP_AMD_IMG_ROWS") && !getenv("DAQP_AMD_IMG_CACHE")) {      // (the tests' overrides apply to every launch)
            int wrows = 40;
            if (const char *e = getenv("DAQP_AMD_IMG_WARM_ROWS")) wrows = atoi(e);
            if (wrows >= 2 && wrows < d.img_rows) {
                int wwaves = waves;
                if (const char *e = getenv("DAQP_AMD_IMG_WARM_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= 8) wwaves = v; }
                const int wbudget = (160 * 1024 / wwaves) / 512 * 512;
                int wc = wrows;
                while (wc > 2 && reg_img_lds_bytes(b->NB, b->img_kind, n, m, wrows, wc, d.ldrc) > wbudget) --wc;
                b->img_rows_warm = wrows; b->img_cache_warm = wc;
                b->lds_img_warm = (size_t)reg_img_lds_bytes(b->NB, b->img_kind, n, m, wrows, wc, d.ldrc);
            }
        }
    }
    if (b->NB == 0 && !b->img_only && cap > 64 && cap <= 256 && !getenv("DAQP_AMD_NO_WG")) {     // (beyond 256 rows: the one-wave kernel with eight chunks, everything large in HBM scratch)
        int W = d.nblk < 4 ? 4 : (d.nblk > kWgMaxWaves ? kWgMaxWaves : d.nblk);
        const int lds_max = 160 * 1024 - 512;          // (the kernel's static LDS -- 320 bytes by the code generator's report -- comes on top of the
                                                       //  dynamic allocation: with 256 bytes of reserve a shape whose packed factor ended within 64 bytes
                                                       //  of the limit, (n, m) = (187, 371), could not be launched at all)
        const int Cw = cap <= 128 ? 2 : 4;
        // Two workgroups per CU beat one with more waves wherever the whole factor fits half the LDS: one problem's serial master phases then run
        // under the other's bandwidth phases (n = 100, m = 300: solve launch 29.8 -> 17.9 ms per 4 096; n = 110, m = 330: 17.1 -> 10.8 ms,
        // profiles/r06t_wg_two_per_cu.txt).  The register file holds eight waves per CU, so two workgroups means four waves each.
        if (W > 4 && wg_lds_bytes(Cw, m, cap) <= lds_max / 2 - 256) W = 4;
        if (const char *we = getenv("DAQP_AMD_WG_WAVES")) { const int v = atoi(we); if (v >= 4 && v <= kWgMaxWaves) W = v; }
        int capL = cap;
        while (capL > 16 && wg_lds_bytes(Cw, m, capL) > lds_max) --capL;
        if (const char *ce = getenv("DAQP_AMD_WG_CAPL")) { const int v = atoi(ce); if (v >= 2 && v < capL) capL = v; }   // (tests: force the hand-over)
        if (wg_lds_bytes(Cw, m, capL) <= lds_max && capL >= (cap < 48 ? cap : 48)) {
            b->use_wg = true; b->wg_W = W; b->wg_C = Cw;
            { const char *iv = getenv("DAQP_AMD_WG_INVERSE"); d.wg_inverse = (iv && atoi(iv) == 0) ? 0 : 1; }   // default mode: L^-1 instead of L (wg_ldp.hip.h)
            d.wg_capL = capL; d.wg_capT = round_up(cap, 8);
            b->lds_wg = (size_t)wg_lds_bytes(Cw, m, capL);
        }
    }
    b->fast_setup = (n <= 64) && !getenv("DAQP_AMD_SLOW_SETUP");
    b->tiny_setup = b->fast_setup && n <= TNC && m <= TMR && !getenv("DAQP_AMD_NO_TINY_SETUP");
    {   // DAQP_AMD_EXACT=1: keep the reference's summation order in M = A R^-1 (bit-exact LDP); default: MFMA
        const char *ex = getenv("DAQP_AMD_EXACT");
        d.exact_setup = (ex && atoi(ex) != 0) ? 1 : 0;
    }
    b->setup_spill = !b->fast_setup && (setup_lds(n, m).total_bytes > 150 * 1024 || getenv("DAQP_AMD_FORCE_SPILL") || n > 256);
    b->lds_setup = b->fast_setup ? (size_t)fast_lds(n, m, 1, m - ms).total_bytes : (size_t)setup_lds(n, m, b->setup_spill).total_bytes;
    // 64 < n <= ~134: both factors fit the LDS of k_setup, where ONE wave factors and inverts in the reference's order -- 13.5 ms per 4 096 problems at
    // n = 100 against 5.2 ms at n = 150, where k_fact_wg (a workgroup per problem, matrix cores) does it.  The default arithmetic's full setup takes
    // k_fact_wg + the scratch variant of k_setup for these shapes too (profiles/r06v_shape_map.txt -> r06w); exact mode, partial updates and the
    // regularising passes keep the LDS variant.
    b->fact_gs = !b->fast_setup && !b->setup_spill && n > 64 && n <= kFactMaxN && !getenv("DAQP_AMD_NO_FACT_WG") && !getenv("DAQP_AMD_NO_FACT_SMALL");
    b->lds_update = (size_t)round_up(n, 2) * 16;
    if (b->lds_ldp > 160 * 1024 || b->lds_setup > 160 * 1024) {
        set_err("problem too large for the LDS-staged setup (needs %zu / %zu bytes)", b->lds_setup, b->lds_ldp);
        delete b;
        return DAQP_EXIT_UNSUPPORTED;
    }
    const size_t Nn = (size_t)N;
    int rc = 0;
    rc |= dev_alloc(b, &d.Mblk, Nn * d.nblk * d.npair * 128);
    rc |= dev_alloc(b, &d.Rinv, Nn * d.rtri);
    rc |= dev_alloc(b, &d.v, Nn * n);
    rc |= dev_alloc(b, &d.scaling, Nn * m);
    rc |= dev_alloc(b, &d.dupper, Nn * m);
    rc |= dev_alloc(b, &d.dlower, Nn * m);
    rc |= dev_alloc(b, &d.sense, Nn * m);
    rc |= dev_alloc(b, &d.xunc, Nn * n);
    rc |= dev_alloc(b, &d.L, Nn * d.ltri);
    rc |= dev_alloc(b, &d.vecs, Nn * 5 * cap);
    rc |= dev_alloc(b, &d.WS, Nn * cap);
    rc |= dev_alloc(b, &d.qs, Nn);
    if (b->spill) rc |= dev_alloc(b, &d.rowc_g, Nn * cap * d.ldr);
    if (b->img_only && d.img_rows > d.img_cache) rc |= dev_alloc(b, &d.rowc_g, Nn * (size_t)((d.img_rows - d.img_cache) * d.ldrc));
    if (b->img32) {
        int t2 = d.img_rows - d.img_cache;
        if (b->img_rows_warm - b->img_cache_warm > t2) t2 = b->img_rows_warm - b->img_cache_warm;
        if (t2 > 0) rc |= dev_alloc(b, &d.rowc_g, Nn * (size_t)(t2 * d.ldrc));
    }
    if (b->setup_spill || b->fact_gs) rc |= dev_alloc(b, &d.setup_g, Nn * 2 * (size_t)round_up(d.rtri, 2));
    // fp32 image of M for the workgroup kernel's screening scan (generic setup kernel only: it is the one that writes it)
    if (b->use_wg && !b->fast_setup && !getenv("DAQP_AMD_NO_SCAN32")) {
        const size_t cnt = Nn * (size_t)d.nblk * d.nquad * 256;
        rc |= dev_alloc(b, &d.M32, cnt);
        if (!rc) HIPCHK(hipMemset(d.M32, 0, cnt * sizeof(float)));
    }
    if (!b->fast_setup) {   // zeroed once: the kernel only ever writes the upper triangle
        const size_t sq = Nn * (size_t)round_up(n, 32) * (size_t)round_up(n, 16);
        rc |= dev_alloc(b, &d.setup_sq, sq);
        if (!rc) HIPCHK(hipMemset(d.setup_sq, 0, sq * sizeof(double)));
        rc |= dev_alloc(b, &d.m_tick, Nn);
        if ((b->setup_spill || b->fact_gs) && n <= kFactMaxN && !getenv("DAQP_AMD_NO_FACT_WG")) rc |= dev_alloc(b, &b->fact_buf, Nn * 4);
        if (!rc) HIPCHK(hipMemset(d.m_tick, 0, Nn * sizeof(int)));
    }
    if (N == 1) {   // one slab: x[n] lam[m] fval soft | flag iter -- in mapped host memory, written by the solve kernel itself (see mapped_out)
        double *slab = nullptr;
        b->out_bytes = ((size_t)n + m + 3) * sizeof(double);
        void *dp = nullptr;
        static const bool no_map = [] { const char *e = getenv("DAQP_AMD_NO_MAPPED_RESULTS"); return e && atoi(e) != 0; }();
        if (!no_map && hipHostMalloc(reinterpret_cast<void **>(&b->pin_out), b->out_bytes, hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer(&dp, b->pin_out, 0) == hipSuccess) {
            slab = static_cast<double *>(dp);
            memset(b->pin_out, 0, b->out_bytes);
            b->mapped_out = true;
        } else {        // (no mapped memory to be had: a device slab and one copy per solve, as before)
            (void)hipGetLastError();
            if (b->pin_out) { (void)hipHostFree(b->pin_out); b->pin_out = nullptr; }
            rc |= dev_alloc(b, &slab, (size_t)n + m + 3);
            if (hipHostMalloc(reinterpret_cast<void **>(&b->pin_out), b->out_bytes, hipHostMallocDefault) != hipSuccess) { b->pin_out = nullptr; b->out_bytes = 0; }
        }
        if (!rc) {
            b->ox = slab; b->olam = slab + n; b->ofval = slab + n + m; b->osoft = slab + n + m + 1;
            b->oflag = reinterpret_cast<int *>(slab + n + m + 2); b->oiter = b->oflag + 1;
        }
    } else {
        rc |= dev_alloc(b, &b->ox, Nn * n);
        rc |= dev_alloc(b, &b->olam, Nn * m);
        rc |= dev_alloc(b, &b->ofval, Nn);
        rc |= dev_alloc(b, &b->osoft, Nn);
        rc |= dev_alloc(b, &b->oflag, Nn);
        rc |= dev_alloc(b, &b->oiter, Nn);
    }
    rc |= dev_alloc(b, &b->st_dev, 1);
    rc |= dev_alloc(b, &b->d_dev, 1);
    rc |= dev_alloc(b, &b->px.counter, 4);
    {   // period of the constant device clock behind s_memrealtime (settings->time_limit): asked of the runtime, not assumed
        int khz = 0;
        d.tick_s = (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) ? 1.0 / (1e3 * (double)khz) : 1e-8;
        d.tstart = nullptr;
    }
    if (b->use_wg && !rc) {
        typedef void (*wg_kernel_t)(BatchDev, int);
        wg_kernel_t kw = b->wg_C == 2 ? k_ldp_wg<2, false> : k_ldp_wg<4, false>;
        wg_kernel_t kwx = b->wg_C == 2 ? k_ldp_wg<2, true> : k_ldp_wg<4, true>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kwx), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_wg);
        (void)hipGetLastError();
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kw), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_wg) != hipSuccess) {
            (void)hipGetLastError();      // (not sticky: the next launch check must not report this)
            b->use_wg = false;            // the kernel cannot be launched with this much LDS: the one-wave kernel takes the shape
        }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kw), 64 * b->wg_W, b->lds_wg) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            per_cu = 1;
        }
        long long g = (long long)cus * per_cu;
        if (const char *ge = getenv("DAQP_AMD_WG_GRID")) { const long long v = atoll(ge); if (v >= 1) g = v; }   // tuning: problems in flight
        b->wg_grid = (int)(g < N ? g : N);
        // Tiered launch (wg_kernel.hip.h, TIER): where the whole factor allows only one workgroup per CU, cold solves run two four-wave workgroups
        // per CU with the rows of the inverse factor beyond what half the LDS holds in HBM.  Worth it only while enough problems are in flight.
        d.wg_r0 = 0;
        long long tier_min = 2LL * cus;
        if (const char *te = getenv("DAQP_AMD_WG_TIER_MIN_BATCH")) { const long long v = atoll(te); if (v >= 1) tier_min = v; }    // (tests: small batches through the tiered launch)
        if (b->use_wg && per_cu == 1 && d.wg_inverse && !getenv("DAQP_AMD_NO_WG_TIER") && N >= tier_min) {
            const int half = (160 * 1024 - 512) / 2 - 256;
            wg_kernel_t kt = b->wg_C == 2 ? k_ldp_wg<2, false, true> : k_ldp_wg<4, false, true>;
            int r0 = cap;
            while (r0 > 16 && wg_lds_bytes(b->wg_C, m, r0) > half) --r0;
            if (const char *re = getenv("DAQP_AMD_WG_R0")) { const int v = atoi(re); if (v >= 8 && v < r0) r0 = v; }    // (tests: more rows in the HBM tier)
            int per2 = 0;
            const size_t lds2 = (size_t)wg_lds_bytes(b->wg_C, m, r0);
            if (r0 >= 32 && r0 < cap && hipFuncSetAttribute(reinterpret_cast<const void *>(kt), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) == hipSuccess
                && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per2, reinterpret_cast<const void *>(kt), 256, lds2) == hipSuccess && per2 >= 2) {
                d.wg_r0 = r0; b->lds_wg_tier = lds2;
                long long g2 = (long long)cus * per2;
                if (const char *ge = getenv("DAQP_AMD_WG_TIER_GRID")) { const long long v = atoll(ge); if (v >= 1) g2 = v; }
                b->wg_tier_grid = (int)(g2 < N ? g2 : N);
            } else (void)hipGetLastError();
        }
        const int wg_slots = b->wg_grid > b->wg_tier_grid ? b->wg_grid : b->wg_tier_grid;   // scratch per workgroup in flight, whichever launch has more
        rc |= dev_alloc(b, &d.wg_counter, 1);
        {   // row-major active rows per workgroup in flight, whole 32-column chunks per row; the pad columns are zero and stay zero
            const size_t cnt = (size_t)wg_slots * cap * wg_row_stride(n);
            rc |= dev_alloc(b, &d.wg_rowc, cnt);
            if (!rc && hipMemset(d.wg_rowc, 0, cnt * sizeof(double)) != hipSuccess) rc = 1;
        }
        rc |= dev_alloc(b, &d.wg_rowcT, (size_t)wg_slots * n * d.wg_capT);
        rc |= dev_alloc(b, &d.fallback, Nn);
        if (!rc && hipMemset(d.fallback, 0, Nn * sizeof(int)) != hipSuccess) rc = 1;
    }
    if (b->img32 && !rc) {
        rc |= dev_alloc(b, &d.img_ho, 1);
        if (!rc && hipHostMalloc(reinterpret_cast<void **>(&b->img_ho_pin), sizeof(int), hipHostMallocDefault) != hipSuccess) rc = 1;
        if (!rc) *b->img_ho_pin = 0;
    }
    if ((b->reg_handover || b->img32 || b->img_only) && !rc) {
        rc |= dev_alloc(b, &d.fallback, Nn);
        if (!rc && hipMemset(d.fallback, 0, Nn * sizeof(int)) != hipSuccess) rc = 1;
    }
    if (!rc && hipMemcpy(b->st_dev, &d.st, sizeof(DAQPSettings), hipMemcpyHostToDevice) != hipSuccess) rc = 1;
    d.st_dev = b->st_dev;
    if (rc) { daqp_batch_free(b); return DAQP_EXIT_UNSUPPORTED; }
    // padding rows/columns of the blocked M image are never written by the kernels: keep them defined
    if (hipMemset(d.Mblk, 0, Nn * d.nblk * d.npair * 128 * sizeof(double)) != hipSuccess ||
        hipMemset(d.vecs, 0, Nn * 5 * cap * sizeof(double)) != hipSuccess ||
        hipMemset(d.qs, 0, Nn * sizeof(QState)) != hipSuccess) {
        set_err("hipMemset failed");
        daqp_batch_free(b);
        return DAQP_EXIT_UNSUPPORTED;
    }
    for (auto &e : b->ev) if (hipEventCreate(&e) != hipSuccess) { set_err("hipEventCreate failed"); daqp_batch_free(b); return DAQP_EXIT_UNSUPPORTED; }
    *out = b;
    return 0;
}

void daqp_batch_free(DAQPBatch *b)
{
    if (!b) return;
    if (b->d.N == 1 && b->ev[3] != nullptr && pool_enabled()) {   // (fully constructed single-problem batch: park it)
        (void)hipSetDevice(b->device);
        (void)hipStreamSynchronize(b->stream);
        b->stream = nullptr;
        if (b->d.trace || b->d.prof) { destroy_batch(b); return; }     // (debug buffers were attached: not worth keeping, and not kept)
        b->one_lam.clear(); b->one_valid = false;
        b->d.shared = 0; b->d.prox_pass = 0;
        b->was_shared = false; b->in_prox_loop = false; b->pending_mask = 0; b->part_mask = 0; b->exact_sticky = false; b->is_setup = false; b->reg_pending = false; b->fresh = false; b->rechecked = 0; b->recheck_device = -1; b->inputs_adopted = false; b->timed_recheck = false; b->lean = false; b->defer_wait = false; b->recheck_due = false; b->mirror_ldp_due = false;
        b->timed_setup = b->timed_solve = false; b->n_prox_qps = 0; b->prox_outer = 0;
        b->one_fval = b->one_soft = 0; b->one_flag = b->one_iter = 0;
        DAQPBatch *evict = nullptr;
        // (Parked workspaces are NOT released when the process exits: the operating system takes the memory back.  Freeing them from an
        //  atexit handler -- some forty hipFree / hipHostFree / hipEventDestroy per workspace right in front of the HIP runtime's own
        //  teardown -- is more work for that runtime's completion thread right where a process that exits at once races with it (about 2 of 1 000
        //  runs of tests/c/mask_caller.c ended with SIGSEGV inside libamdhip64 after their last line of output, tools/stress_caller.py).  daqp_amd_release_pool() is there for a host that wants
        //  the memory back while it runs.)
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            g_pool.push_back(b);
            if (g_pool.size() > kPoolMax) { evict = g_pool.front(); g_pool.erase(g_pool.begin()); }
        }
        if (evict) destroy_batch(evict);
        return;
    }
    destroy_batch(b);
}
// What a process owes the HIP runtime before it exits: nothing of this library's still in flight.  The one-problem path polls a mapped word
// instead of synchronising its stream, so the runtime retires a call sequence's commands late -- a process that returned from main() within
// microseconds of its last call met the runtime's own static teardown half way (about 2 deaths in 1 000 runs of tests/c/mask_caller.c inside
// libamdhip64's completion thread; none with a synchronising call path: INTEGRATION.md "Process exit").  daqp_amd_shutdown() waits for every
// device this library touched; it is registered with atexit() at the first batch creation (DAQP_AMD_NO_EXIT_SYNC=1: not), frees nothing
// (parked workspaces are left to the operating system) and may be called by the host at any time, any number of times.
void daqp_amd_shutdown(void)
{
    const unsigned long long used = g_devices_used.load();
    int cur = -1;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (int dv = 0; dv < 64; ++dv)
        if ((used >> dv) & 1ull) { if (hipSetDevice(dv) == hipSuccess) (void)hipDeviceSynchronize(); }
    if (have) (void)hipSetDevice(cur);
}
// really release every parked single-problem workspace (tests; a host that wants its device memory back)
void daqp_amd_release_pool(void)
{
    std::vector<DAQPBatch *> all;
    { std::lock_guard<std::mutex> lk(g_pool_mu); all.swap(g_pool); }
    for (DAQPBatch *c : all) destroy_batch(c);
}

void daqp_batch_set_exact(DAQPBatch *b, int exact) { if (b) b->d.exact_setup = exact ? 1 : 0; }
void daqp_batch_set_stream(DAQPBatch *b, void *hip_stream)
{
    if (!b) return;
    hipStream_t ns = reinterpret_cast<hipStream_t>(hip_stream);
    if (ns != b->stream) {
        // work already queued for this batch on the old stream (the reset of a pooled workspace, a setup) stays ahead of whatever
        // the new stream is given: a non-blocking stream does not order itself against the null stream
        hipEvent_t e = nullptr;
        (void)hipSetDevice(b->device);
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess) {
            if (hipEventRecord(e, b->stream) == hipSuccess) (void)hipStreamWaitEvent(ns, e, 0);
            (void)hipEventDestroy(e);
        }
    }
    b->stream = ns;
}
void daqp_batch_set_settings(DAQPBatch *b, const DAQPSettings *settings)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    (void)resolve_setup(b);    // regularising passes of the last setup still owed: they belong to the settings that setup ran with (eps_prox)
    if (settings) b->d.st = *settings; else default_settings(&b->d.st);
    (void)put_async(b->st_dev, b->d.st, b->stream);
}
unsigned long long daqp_batch_device_bytes(const DAQPBatch *b) { return b ? b->bytes + (b->redo ? b->redo->bytes : 0) : 0; }
int daqp_batch_rechecked(const DAQPBatch *b) { return b ? b->rechecked : 0; }
void daqp_batch_set_recheck(DAQPBatch *b, int on) { if (b) { b->recheck = on != 0; b->recheck_device = on != 0 ? 1 : 0; } }
int daqp_batch_recheck_ms(DAQPBatch *b, float *ms)
{
    if (!b || !ms) return DAQP_EXIT_UNSUPPORTED;
    *ms = 0;
    if (!b->timed_recheck) return 0;
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipEventSynchronize(b->ev_r[1]));
    HIPCHK(hipEventElapsedTime(ms, b->ev_r[0], b->ev_r[1]));
    return 0;
}

// debugging aid used by the parity tests: per-problem add/remove event trace (cap ints each;
// the last slot receives the event count).  Pass cap 0 to switch it off.
int daqp_batch_enable_trace(DAQPBatch *b, int cap)
{
    if (!b) return DAQP_EXIT_UNSUPPORTED;
    (void)hipSetDevice(b->device);
    if (cap <= 0) { b->d.trace = nullptr; b->d.trace_cap = 0; return 0; }
    int *t = nullptr;
    if (dev_alloc(b, &t, (size_t)b->d.N * cap)) return DAQP_EXIT_UNSUPPORTED;
    HIPCHK(hipMemset(t, 0, (size_t)b->d.N * cap * sizeof(int)));
    b->d.trace = t; b->d.trace_cap = cap;
    return 0;
}
// debugging aid: per-problem cycle counters of the solve kernel's phases (8 x int64 per problem:
// csp, blocking test, primal, scan, add, remove, -, -).  on=0 switches it off.
int daqp_batch_enable_profile(DAQPBatch *b, int on)
{
    if (!b) return DAQP_EXIT_UNSUPPORTED;
    (void)hipSetDevice(b->device);
    if (!on) { b->d.prof = nullptr; return 0; }
    if (!kProfile) { set_err("cycle-counter probes are not compiled in (build with -DDAQP_AMD_PROFILE)"); return DAQP_EXIT_UNSUPPORTED; }
    long long *p = nullptr;
    if (dev_alloc(b, &p, (size_t)b->d.N * 32)) return DAQP_EXIT_UNSUPPORTED;
    HIPCHK(hipMemset(p, 0, (size_t)b->d.N * 32 * sizeof(long long)));
    b->d.prof = p;
    return 0;
}
int daqp_batch_read_profile(DAQPBatch *b, long long *host)
{
    if (!b || !b->d.prof) return DAQP_EXIT_UNSUPPORTED;
    (void)hipSetDevice(b->device);
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipMemcpy(host, b->d.prof, (size_t)b->d.N * 32 * sizeof(long long), hipMemcpyDeviceToHost));
    return 0;
}
int daqp_batch_read_trace(DAQPBatch *b, int *host)
{
    if (!b || !b->d.trace) return DAQP_EXIT_UNSUPPORTED;
    (void)hipSetDevice(b->device);
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipMemcpy(host, b->d.trace, (size_t)b->d.N * b->d.trace_cap * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}
// debugging aid: copy the LDP of problem q to the host (M as the reference stores it: (m-ms) x n
// row-major; R packed upper; v; dupper; dlower; scaling).  Any pointer may be NULL.
int daqp_batch_read_ldp(DAQPBatch *b, int q, double *M, double *R, double *v, double *dupper, double *dlower, double *scaling)
{
    if (!b || q < 0 || q >= b->d.N) return DAQP_EXIT_UNSUPPORTED;
    (void)hipSetDevice(b->device);
    if (resolve_setup(b)) return DAQP_EXIT_UNSUPPORTED;
    if (flush_update(b)) return DAQP_EXIT_UNSUPPORTED;
    const BatchDev &d = b->d;
    HIPCHK(hipStreamSynchronize(b->stream));
    if (M) {
        const size_t per = (size_t)d.nblk * d.npair * 128;
        std::vector<double> blk(per);
        HIPCHK(hipMemcpy(blk.data(), d.Mblk + per * q, per * sizeof(double), hipMemcpyDeviceToHost));
        for (int r = d.ms; r < d.m; ++r)
            for (int k = 0; k < d.n; ++k)
                M[(size_t)(r - d.ms) * d.n + k] = blk[(((size_t)(r >> 6) * d.npair + (k >> 1)) * 64 + (r & 63)) * 2 + (k & 1)];
    }
    if (R) HIPCHK(hipMemcpy(R, d.Rinv + (size_t)q * d.rtri, d.rtri * sizeof(double), hipMemcpyDeviceToHost));
    if (v) HIPCHK(hipMemcpy(v, d.v + (size_t)q * d.n, d.n * sizeof(double), hipMemcpyDeviceToHost));
    if (dupper) HIPCHK(hipMemcpy(dupper, d.dupper + (size_t)q * d.m, d.m * sizeof(double), hipMemcpyDeviceToHost));
    if (dlower) HIPCHK(hipMemcpy(dlower, d.dlower + (size_t)q * d.m, d.m * sizeof(double), hipMemcpyDeviceToHost));
    if (scaling) HIPCHK(hipMemcpy(scaling, d.scaling + (size_t)q * d.m, d.m * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

// setup_daqp_main on fresh workspaces (api.c:93-151): the iterate of the proximal loop starts at the origin again (api.c:318)
int daqp_batch_setup(DAQPBatch *b, const DAQPBatchProblem *p, int init_mask) { return batch_setup(b, p, init_mask, true); }
} // extern "C"
namespace {
int batch_setup(DAQPBatch *b, const DAQPBatchProblem *p, int init_mask, bool fresh)
{
    int rc = check_problem(b, p);
    if (rc) return rc;
    if (!p->f || !p->bupper || !p->blower || (b->d.mA > 0 && !p->A)) {
        set_err("f, A, bupper, blower are required (H may be NULL: an LP)");
        return DAQP_EXIT_UNSUPPORTED;
    }
    const bool lp = p->H == nullptr || (b->ident && p->H == b->ident);   // api.c:183-185
    b->pending_mask = 0;   // a full setup supersedes any deferred update
    b->part_mask = 0;
    b->exact_sticky = false;
    HIPCHK(hipSetDevice(b->device));
    if (fresh && b->prox_ready) HIPCHK(hipMemsetAsync(b->px.center, 0, (size_t)b->d.N * b->d.n * sizeof(double), b->stream));
    BatchDev &d = b->d;
    d.shared = 0;
    const size_t N = d.N;
    if (lp) {
        if (!b->ident) {
            if (dev_alloc(b, &b->ident, (size_t)d.n * d.n)) return DAQP_EXIT_UNSUPPORTED;
            std::vector<double> eye((size_t)d.n * d.n, 0.0);
            for (int i = 0; i < d.n; ++i) eye[(size_t)i * d.n + i] = 1.0;
            HIPCHK(hipMemcpy(b->ident, eye.data(), eye.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        d.H = b->ident;
    } else if (!(N == 1 && p->memory == DAQP_MEM_HOST)) rc |= stage(b, p->H, p->memory, N * d.n * d.n, &b->sH, &b->nH, &d.H);
    if (N == 1 && p->memory == DAQP_MEM_HOST) {
        // one problem from host memory: everything through ONE pinned slab and ONE copy (six pageable copies cost more than
        // the kernels of a small problem)
        const size_t nH = (lp ? 0 : (size_t)d.n * d.n), nf = d.n, nA = (size_t)d.mA * d.n, nb = d.m;
        const size_t dbl = nH + nf + nA + 2 * nb, bytes = dbl * sizeof(double) + (p->sense ? nb * sizeof(int) : 0);
        if (b->in_cap < bytes || b->pin_in == nullptr) {
            if (b->ev_in) HIPCHK(hipEventSynchronize(b->ev_in));
            if (b->pin_in) { (void)hipHostFree(b->pin_in); b->pin_in = nullptr; }
            HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&b->pin_in), bytes ? bytes : 8, hipHostMallocDefault));
            if (slot_reserve(b, &b->dev_in, &b->in_cap, bytes ? bytes : 8)) return DAQP_EXIT_UNSUPPORTED;
        }
        if (b->ev_in == nullptr) HIPCHK(hipEventCreateWithFlags(&b->ev_in, hipEventDisableTiming));
        else HIPCHK(hipEventSynchronize(b->ev_in));      // the previous copy out of the slab has run
        double *hp = reinterpret_cast<double *>(b->pin_in);
        const double *dp = reinterpret_cast<const double *>(b->dev_in);
        size_t o = 0;
        if (!lp) { memcpy(hp + o, p->H, nH * sizeof(double)); d.H = dp + o; o += nH; }
        if (p->f) { memcpy(hp + o, p->f, nf * sizeof(double)); d.f = dp + o; } else d.f = nullptr;
        o += nf;
        if (nA) memcpy(hp + o, p->A, nA * sizeof(double));
        d.A = dp + o; o += nA;
        if (nb) { memcpy(hp + o, p->bupper, nb * sizeof(double)); memcpy(hp + o + nb, p->blower, nb * sizeof(double)); }
        d.bu = dp + o; d.bl = dp + o + nb; o += 2 * nb;
        if (p->sense) { memcpy(hp + o, p->sense, nb * sizeof(int)); d.sense_in = reinterpret_cast<const int *>(dp + o); }
        else d.sense_in = nullptr;
        HIPCHK(hipMemcpyAsync(b->dev_in, b->pin_in, bytes, hipMemcpyHostToDevice, b->stream));
        HIPCHK(hipEventRecord(b->ev_in, b->stream));
    } else {
        rc |= stage(b, p->f, p->memory, N * d.n, &b->sf, &b->nf, &d.f);
        rc |= stage(b, p->A, p->memory, N * d.mA * d.n, &b->sA, &b->nA, &d.A);
        rc |= stage(b, p->bupper, p->memory, N * d.m, &b->sbu, &b->nbu, &d.bu);
        rc |= stage(b, p->blower, p->memory, N * d.m, &b->sbl, &b->nbl, &d.bl);
        rc |= stage(b, p->sense, p->memory, N * d.m, &b->ssense, &b->nsense, &d.sense_in);
    }
    if (rc) return DAQP_EXIT_UNSUPPORTED;
    b->was_shared = false;
    // DAQP_UPDATE_eliminate (daqp_quadprog, eq_elim.c): the reference projects many equalities out of the LDP first.  That
    // reduction is not built; such problems are solved on the full LDP instead (what setup_daqp + daqp_solve do): same exit
    // flag and active set, x and lam equal to ~1e-13, only the iteration count may differ (tests/test_gpu_reference_cases.py).
    const int mask = (init_mask & ~DAQP_UPDATE_eliminate) | DAQP_UPDATE_Rinv | DAQP_UPDATE_M | DAQP_UPDATE_v | DAQP_UPDATE_d | DAQP_UPDATE_sense;
    setup_kernel_t ks = b->setup_spill ? (d.n > 256 ? k_setup<true, 8> : k_setup<true>) : k_setup<false>;
    size_t lds_setup = b->lds_setup;
    if (b->fast_setup) {
        ks = pick_setup_fast(d.n, d.exact_setup == 0);
        lds_setup = (size_t)fast_lds(d.n, d.m, d.exact_setup, d.mA).total_bytes;   // the MFMA path overlays the A tile on R^-1
    }
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_setup));
    if (!b->quiet_setup) HIPCHK(hipEventRecord(b->ev[0], b->stream));
    b->inputs_adopted = p->memory == DAQP_MEM_DEVICE;
    if (!lp && b->tiny_setup) {     // sixteen problems per wavefront; singular Hessians leave flagged for the regularising re-run below
        hipLaunchKernelGGL(k_setup_tiny<4>, dim3((d.N + 15) / 16), dim3(64), 0, b->stream, d, mask);
        HIPCHK(hipGetLastError());
    } else if (!lp && b->fast_setup && (d.ms == 0 || !getenv("DAQP_AMD_NO_BLK_BOUNDS")) && d.n > 16 && d.exact_setup == 0 && blk_setup_enabled()) {
        // the factorisation on the matrix cores (setup_blk.hip.h); what it does not call clearly regular it marks, and the ordered
        // kernel right behind it takes exactly those problems (every other wave of that launch leaves at its first scalar load)
        const int NT = (d.n + 15) / 16;
        ldp_reg_kernel_t kb = nullptr;
#define DAQP_BLK_SHAPE(nt, nw, tail) \
        if (!kb && NT == nt && d.n <= nw && (tail) == (d.n - 16 * (nt - 1) <= 4)) kb = k_setup_blk<nt, nw, tail>;
        DAQP_BLK_SHAPES
#undef DAQP_BLK_SHAPE
        const size_t lds_blk = (size_t)(NT == 2 ? blk_lds<2>(d.n, d.m) : (NT == 3 ? blk_lds<3>(d.n, d.m) : blk_lds<4>(d.n, d.m))).total_bytes;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_blk));
        if (push_descriptor(b)) return DAQP_EXIT_UNSUPPORTED;      // (it reads the descriptor through a pointer)
        hipLaunchKernelGGL(kb, dim3(d.N), dim3(64), lds_blk, b->stream, (const BatchDev *)b->d_dev, mask);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(ks, dim3(d.N), dim3(64), lds_setup, b->stream, d, mask | kSetupOnlyMarked);
        HIPCHK(hipGetLastError());
    } else if (!lp) {
        // what belongs to THESE launches only (who forms the general rows, whose factorisation records) travels in a copy of the
        // descriptor: the batch's own never carries it, whichever way this function is left (a later shared setup copies b->d)
        BatchDev l = d;
        l.defer_m = defers_m(b, d) ? 1 : 0;
        const bool gs_now = b->setup_spill || (b->fact_gs && l.defer_m && b->fact_buf && d.setup_g && !(d.st.eps_prox > 0.0));
        if (l.defer_m) {    // the instantiation without the general rows (k_setup_m follows): more waves per SIMD
            ks = gs_now ? k_setup<true, 4, true> : k_setup<false, 4, true>;
            if (gs_now && !b->setup_spill) lds_setup = (size_t)setup_lds(d.n, d.m, true).total_bytes;
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_setup));
        }
        // default arithmetic, factors out of LDS (the n = 200 class): Cholesky and inverse by k_fact_wg, a workgroup per problem with
        // the triangle in LDS and the matrix cores behind each panel (setup_fact.hip.h); k_setup takes R^-1 from the scratch
        l.fact = nullptr;
        if (l.defer_m && gs_now && b->fact_buf && d.n <= kFactMaxN && !(d.st.eps_prox > 0.0)) {
            const size_t lds = fact_lds_bytes(d.n);
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_fact_wg), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) {
                l.fact = b->fact_buf;
                hipLaunchKernelGGL(k_fact_wg, dim3(d.N), dim3(512), lds, b->stream, l);
                HIPCHK(hipGetLastError());
            } else (void)hipGetLastError();
        }
        hipLaunchKernelGGL(ks, dim3(d.N), dim3(64), lds_setup, b->stream, l, mask);
        HIPCHK(hipGetLastError());
        if (l.defer_m) { if (launch_setup_m(b, l)) return DAQP_EXIT_UNSUPPORTED; }
    } else HIPCHK(hipMemsetAsync(d.qs, 0, (size_t)d.N * sizeof(QState), b->stream));   // fresh records: the LP pass below fills them
    // numerically singular Hessians: shifted re-runs for those problems (one tiny counting kernel when there are none);
    // an LP batch: its one setup pass
    b->reg_pending = false;
    if (b->lean && !lp) {
        // daqp_quadprog's one problem: the solve launch that follows activates by itself (need_activate in the record) and reports a
        // flagged Hessian with the internal code -- the caller then takes the full path; no counting kernel, no activation launch
        b->n_prox_qps = 0;
    } else {
    rc = lp ? regularise(b, d, mask, true) : count_flagged_async(b, mask);
    if (rc) return rc;
    // initial working set from sense (utils.c:199-211); a no-op per problem unless flagged
    rc = launch_ldp(b, 1);
    if (rc) return rc;
    }
    if (!b->quiet_setup) { HIPCHK(hipEventRecord(b->ev[1], b->stream)); b->timed_setup = true; }
    b->is_setup = true;
    // (a daqp_batch_update with every bit set comes through here as well and arms the second pass like a setup does: what follows it
    //  IS a first solve on a fresh LDP, and the arrays it read are the ones the descriptor points at now)
    b->fresh = !lp; b->fresh_mask = init_mask;
    return 0;
}
} // namespace
extern "C" {

// N problems that share H and A (condensed MPC: one plant, many states): p->H is ONE n x n matrix, p->A ONE (m-ms) x n
// matrix; f, bupper, blower (and sense, if given) are per problem as usual.  Semantics: the reference's MPC usage -- one
// setup_daqp (open bounds), then daqp_update_ldp(UPDATE_v|UPDATE_d) per problem -- for N problems at once.  The
// factorisation (Cholesky, R^-1, M = A R^-1, scaling) runs ONCE; v and d of every problem are formed by the update path
// (fused into the first solve launch for the register kernel), and every later solve reads the one shared image of M.
int daqp_batch_setup_shared(DAQPBatch *b, const DAQPBatchProblem *p, int init_mask)
{
    int rc = check_problem(b, p);
    if (rc) return rc;
    if (!p->H || !p->f || !p->bupper || !p->blower || (b->d.mA > 0 && !p->A)) {
        set_err("H, f, A, bupper, blower are required (LPs / missing linear term are outside this path)");
        return DAQP_EXIT_UNSUPPORTED;
    }
    (void)init_mask;   // the unconstrained shortcut / elimination are per-problem decisions of daqp_quadprog: not taken here
    b->pending_mask = 0;
    b->part_mask = 0;
    b->n_prox_qps = 0;
    b->reg_pending = false;
    HIPCHK(hipSetDevice(b->device));
    if (b->prox_ready) HIPCHK(hipMemsetAsync(b->px.center, 0, (size_t)b->d.N * b->d.n * sizeof(double), b->stream));   // api.c:318
    BatchDev &d = b->d;
    const size_t N = d.N;
    rc |= stage(b, p->H, p->memory, (size_t)d.n * d.n, &b->sH, &b->nH, &d.H);
    rc |= stage(b, p->A, p->memory, (size_t)d.mA * d.n, &b->sA, &b->nA, &d.A);
    rc |= stage(b, p->f, p->memory, N * d.n, &b->sf, &b->nf, &d.f);
    rc |= stage(b, p->bupper, p->memory, N * d.m, &b->sbu, &b->nbu, &d.bu);
    rc |= stage(b, p->blower, p->memory, N * d.m, &b->sbl, &b->nbl, &d.bl);
    rc |= stage(b, p->sense, p->memory, N * d.m, &b->ssense, &b->nsense, &d.sense_in);
    if (rc) return DAQP_EXIT_UNSUPPORTED;
    b->was_shared = true;
    b->fresh = false;
    if (!b->wide_u) {
        if (dev_alloc(b, &b->wide_u, d.m) || dev_alloc(b, &b->wide_l, d.m) || dev_alloc(b, &b->structural, d.m) || dev_alloc(b, &b->shared_flag, 4))
            return DAQP_EXIT_UNSUPPORTED;
        std::vector<double> hu(d.m, 1e30), hl(d.m, -1e30);
        HIPCHK(hipMemcpy(b->wide_u, hu.data(), d.m * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(b->wide_l, hl.data(), d.m * sizeof(double), hipMemcpyHostToDevice));
    }
    // ---- the one factorisation: problem 0's slots, wide-open bounds, no sense => only structural bits in sense[0..m)
    BatchDev t = d;
    t.N = 1; t.shared = 0; t.bu = b->wide_u; t.bl = b->wide_l; t.sense_in = nullptr;
    const int mask = DAQP_UPDATE_Rinv | DAQP_UPDATE_M | DAQP_UPDATE_v | DAQP_UPDATE_d | DAQP_UPDATE_sense;
    setup_kernel_t ks = b->setup_spill ? (d.n > 256 ? k_setup<true, 8> : k_setup<true>) : k_setup<false>;
    size_t lds_setup = b->lds_setup;
    if (b->fast_setup) {
        ks = pick_setup_fast(d.n, d.exact_setup == 0);
        lds_setup = (size_t)fast_lds(d.n, d.m, d.exact_setup, d.mA).total_bytes;
    }
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_setup));
    HIPCHK(hipEventRecord(b->ev[0], b->stream));
    t.defer_m = defers_m(b, t) ? 1 : 0;
    if (t.defer_m) {
        ks = b->setup_spill ? k_setup<true, 4, true> : k_setup<false, 4, true>;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_setup));
    }
    hipLaunchKernelGGL(ks, dim3(1), dim3(64), lds_setup, b->stream, t, mask);
    HIPCHK(hipGetLastError());
    if (t.defer_m) { if (launch_setup_m(b, t)) return DAQP_EXIT_UNSUPPORTED; t.defer_m = 0; }
    // a numerically singular (or forcibly shifted) shared Hessian: the regularising passes of utils.c:354-377 on the one
    // factorisation (this is a setup that happens once per plant: the host looks at the count right away); every problem of the
    // batch then runs the proximal outer loop on the one shifted factor, each with its own centre (daqp_prox.c:21-221)
    rc = regularise(b, t, mask, false);
    if (rc) return rc;
    const bool all_prox = b->n_prox_qps > 0;
    b->n_prox_qps = all_prox ? d.N : 0;
    HIPCHK(hipMemcpyAsync(b->structural, d.sense, d.m * sizeof(int), hipMemcpyDeviceToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->shared_flag, &d.qs[0].setup_flag, sizeof(int), hipMemcpyDeviceToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->shared_flag + 1, &d.qs[0].diag_h, sizeof(int), hipMemcpyDeviceToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->shared_flag + 2, &d.qs[0].n_prox, sizeof(int), hipMemcpyDeviceToDevice, b->stream));
    // ---- per-problem state, then v and d of every problem through the update path
    d.shared = 1;
    hipLaunchKernelGGL(k_init_shared, dim3(d.N), dim3(64), 0, b->stream, d, (const int *)b->structural, (const int *)b->shared_flag);
    HIPCHK(hipGetLastError());
    if (all_prox) {
        hipLaunchKernelGGL(k_prox_share, dim3((d.N + 127) / 128), dim3(128), 0, b->stream, d, b->px);
        HIPCHK(hipGetLastError());
    }
    b->is_setup = true;
    const int upd = DAQP_UPDATE_v | DAQP_UPDATE_d;
    const bool lazy = b->NB > 0 && !b->reg_handover && !getenv("DAQP_AMD_EAGER_UPDATE") && p->sense == nullptr;
    if (lazy) b->pending_mask = upd;
    else {   // a given working set is activated now, and activation needs d
        hipLaunchKernelGGL(k_update, dim3(d.N), dim3(64), b->lds_update, b->stream, d, upd);
        HIPCHK(hipGetLastError());
        rc = launch_ldp(b, 1);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(b->ev[1], b->stream));
    b->timed_setup = true;
    return 0;
}

// daqp_update_ldp with the Rinv and / or M bit but not every bit (utils.c:58-221): a new Hessian, new general rows, or both, on a
// workspace that keeps what the mask leaves out -- its sense (stale ACTIVE bits included), its normalised R^-1, its v.  One launch
// of k_setup<..., PART> in the reference's operation order, then (Rinv bit) the count of Hessians that need the shift, then the
// activation pass for the problems whose update asks for it.  Arrays of `p` that the mask's steps read and that are NULL stay
// as the batch has them (device-resident ones were adopted: they must still be valid).
static int partial_update(DAQPBatch *b, int mask, const DAQPBatchProblem *p)
{
    BatchDev &d = b->d;
    const bool withR = (mask & DAQP_UPDATE_Rinv) != 0, lp = b->ident && d.H == b->ident;
    if (b->was_shared) {
        set_err("update mask %d after daqp_batch_setup_shared: the batch holds ONE H and A (call daqp_batch_setup_shared again)", mask);
        return DAQP_EXIT_UNSUPPORTED;
    }
    if (lp && withR) { set_err("update mask %d: an LP batch has no Hessian to update (set the batch up again)", mask); return DAQP_EXIT_UNSUPPORTED; }
    if (flush_update(b)) return DAQP_EXIT_UNSUPPORTED;      // a deferred v|d update comes first, as it was called first
    const size_t N = d.N;
    int rc = 0;
    const double *tmp = nullptr;
    const int *itmp = nullptr;
    // what the mask's steps read (utils.c:95,103,123,136,161): H with Rinv; A with Rinv or M; f with Rinv or v; the bounds always
    if (withR && p->H) { rc |= stage(b, p->H, p->memory, N * d.n * d.n, &b->sH, &b->nH, &tmp); d.H = tmp; }
    if (p->A && d.mA > 0) { rc |= stage(b, p->A, p->memory, N * d.mA * d.n, &b->sA, &b->nA, &tmp); d.A = tmp; }
    if ((withR || (mask & DAQP_UPDATE_v)) && p->f) { rc |= stage(b, p->f, p->memory, N * d.n, &b->sf, &b->nf, &tmp); d.f = tmp; }
    if (p->bupper) { rc |= stage(b, p->bupper, p->memory, N * d.m, &b->sbu, &b->nbu, &tmp); d.bu = tmp; }
    if (p->blower) { rc |= stage(b, p->blower, p->memory, N * d.m, &b->sbl, &b->nbl, &tmp); d.bl = tmp; }
    if (mask & DAQP_UPDATE_sense) { rc |= stage(b, p->sense, p->memory, N * d.m, &b->ssense, &b->nsense, &itmp); d.sense_in = itmp; }
    if (rc) return DAQP_EXIT_UNSUPPORTED;
    const setup_kernel_t ks = pick_setup_part(b);
    const size_t lds = (size_t)setup_lds(d.n, d.m, b->setup_spill).total_bytes;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    BatchDev l = d;                    // (per-launch state stays off the batch's own descriptor)
    l.exact_setup = 1; l.defer_m = 0; l.fact = nullptr; l.prox_pass = 0;
    const int kmask = mask & ~(DAQP_UPDATE_eliminate | DAQP_UPDATE_hierarchy);
    HIPCHK(hipEventRecord(b->ev[0], b->stream));
    hipLaunchKernelGGL(ks, dim3(d.N), dim3(64), lds, b->stream, l, kmask);
    HIPCHK(hipGetLastError());
    b->part_mask = kmask;
    if (withR) {                       // numerically singular Hessians: their shifted re-runs, as after a setup (resolve_setup)
        rc = count_flagged_async(b, kmask);
        if (rc) return rc;
    }
    rc = launch_ldp(b, 1);             // utils.c:199-211 for the problems that asked for it
    if (rc) return rc;
    HIPCHK(hipEventRecord(b->ev[1], b->stream));
    b->timed_setup = true;
    return 0;
}

int daqp_batch_update(DAQPBatch *b, int mask, const DAQPBatchProblem *p)
{
    int rc = check_problem(b, p);
    if (rc) return rc;
    if (!b->is_setup) { set_err("daqp_batch_update before daqp_batch_setup"); return DAQP_EXIT_UNSUPPORTED; }
    HIPCHK(hipSetDevice(b->device));
    if (resolve_setup(b)) return DAQP_EXIT_UNSUPPORTED;
    b->fresh = false;
    // (see the member's comment.  A call that updates nothing on this path -- mask 0, or only the eliminate / unconstrained / hierarchy bits --
    //  leaves the flag as it is: the workspace is still in the state the earlier sense update put it in)
    if (mask & (DAQP_UPDATE_Rinv | DAQP_UPDATE_M | DAQP_UPDATE_v | DAQP_UPDATE_d | DAQP_UPDATE_sense))
        b->exact_sticky = (mask & DAQP_UPDATE_sense) != 0 && p->sense == nullptr && (mask & (DAQP_UPDATE_Rinv | DAQP_UPDATE_M)) == 0;
    const int full = DAQP_UPDATE_Rinv | DAQP_UPDATE_M | DAQP_UPDATE_v | DAQP_UPDATE_d | DAQP_UPDATE_sense;
    if ((mask & full) == full) {
        DAQPBatchProblem pp = *p;   // unchanged arrays may be omitted: reuse what the batch already has
        BatchDev &d = b->d;
        if (!pp.H) { pp.H = d.H; } if (!pp.f) pp.f = d.f; if (!pp.A) pp.A = d.A;
        if (!pp.bupper) pp.bupper = d.bu; if (!pp.blower) pp.blower = d.bl;
        const bool lp = b->ident && d.H == b->ident;   // an LP batch has no H to resend
        const bool need_A = d.mA > 0;                   // only simple bounds: A is legitimately NULL
        if ((!p->H && !lp) || !p->f || (need_A && !p->A) || !p->bupper || !p->blower) {
            if (p->memory != DAQP_MEM_DEVICE) { set_err("full re-setup from host memory needs every array"); return DAQP_EXIT_UNSUPPORTED; }
        }
        if (b->was_shared && ((!p->H && !lp) || (need_A && !p->A))) {   // the batch holds ONE H / A: nothing per problem to reuse
            set_err("full re-setup after daqp_batch_setup_shared needs per-problem H and A (or call daqp_batch_setup_shared again)");
            return DAQP_EXIT_UNSUPPORTED;
        }
        return batch_setup(b, &pp, mask & (DAQP_UPDATE_unconstrained | DAQP_UPDATE_eliminate), false);   // daqp_update_ldp keeps work->x
    }
    // every other mask of utils.c:58-221, step by step (the reference's bindings build them field by field: daqp.pyx:513-571)
    if (mask & (DAQP_UPDATE_Rinv | DAQP_UPDATE_M)) return partial_update(b, mask, p);
    if (!(mask & (DAQP_UPDATE_v | DAQP_UPDATE_d | DAQP_UPDATE_sense))) return 0;    // nothing of this path to update (utils.c: only sing_ind is reset; the next update or solve does that here)
    BatchDev &d = b->d;
    const size_t N = d.N;
    const double *tmp = nullptr;
    const bool with_sense = (mask & DAQP_UPDATE_sense) != 0;
    // The register solve kernel applies the update itself at its next launch (it has the rows of M in registers anyway:
    // one pass over M per MPC step instead of two).  Because that reads f / the bounds later than this call, device
    // arrays are copied into the batch's own buffers (stream-ordered, a few tens of microseconds) instead of adopted.
    // A new sense is taken over right away, and what comes with it in the same call as well (utils.c:84-98: the bound check reads
    // the new sense, the activation at the end the new d).
    const bool lazy = b->NB > 0 && !b->reg_handover && !getenv("DAQP_AMD_EAGER_UPDATE") && !with_sense;
    const int mem = p->memory;
    auto take = [&](const double *src, size_t count, double **slot, size_t *cap, const double **out) -> int {
        if (!lazy || mem != DAQP_MEM_DEVICE) return stage(b, src, mem, count, slot, cap, out);
        if (src != *slot && slot_reserve(b, slot, cap, count)) return DAQP_EXIT_UNSUPPORTED;
        if (src != *slot) HIPCHK(hipMemcpyAsync(*slot, src, count * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
        *out = *slot;
        return 0;
    };
    if (with_sense) {
        if (flush_update(b)) return DAQP_EXIT_UNSUPPORTED;    // a deferred v|d update comes first, as it was called first
        const int *itmp = nullptr;
        rc |= stage(b, p->sense, mem, N * d.m, &b->ssense, &b->nsense, &itmp);
        if (rc) return DAQP_EXIT_UNSUPPORTED;
        d.sense_in = itmp;
    }
    if (mask & DAQP_UPDATE_v) {
        if (!p->f) { set_err("DAQP_UPDATE_v needs f"); return DAQP_EXIT_UNSUPPORTED; }
        rc |= take(p->f, N * d.n, &b->sf, &b->nf, &tmp); d.f = tmp;
    }
    if ((mask & (DAQP_UPDATE_v | DAQP_UPDATE_d)) && p->bupper) { rc |= take(p->bupper, N * d.m, &b->sbu, &b->nbu, &tmp); d.bu = tmp; }
    if ((mask & (DAQP_UPDATE_v | DAQP_UPDATE_d)) && p->blower) { rc |= take(p->blower, N * d.m, &b->sbl, &b->nbl, &tmp); d.bl = tmp; }
    if (rc) return DAQP_EXIT_UNSUPPORTED;
    if (lazy) { b->pending_mask |= mask & (DAQP_UPDATE_v | DAQP_UPDATE_d); b->timed_setup = false; return 0; }
    HIPCHK(hipEventRecord(b->ev[0], b->stream));
    if (with_sense) {
        hipLaunchKernelGGL(k_update_sense, dim3(d.N), dim3(256), 0, b->stream, d);
        HIPCHK(hipGetLastError());
    }
    if (mask & (DAQP_UPDATE_v | DAQP_UPDATE_d)) {
        hipLaunchKernelGGL(k_update, dim3(d.N), dim3(64), b->lds_update, b->stream, d, mask & (DAQP_UPDATE_v | DAQP_UPDATE_d));
        HIPCHK(hipGetLastError());
    }
    rc = launch_ldp(b, 1);
    if (rc) return rc;
    HIPCHK(hipEventRecord(b->ev[1], b->stream));
    b->timed_setup = true;
    return 0;
}

} // extern "C"
namespace {
// one problem: wait for the stream, take the results out of the slab; an INFEASIBLE verdict of a first solve in the default arithmetic is
// re-derived in the reference's (recheck.hip.h) -- decided here, on the host, which has the verdict in its hands
int collect_one(DAQPBatch *b, DAQPBatchResult *r)
{
    BatchDev &d = b->d;
    for (int pass = 0; pass < 2; ++pass) {
        if (b->mapped_out) { if (wait_stream(b)) return DAQP_EXIT_UNSUPPORTED; }
        else {
            HIPCHK(hipMemcpyAsync(b->pin_out, b->ox, b->out_bytes, hipMemcpyDeviceToHost, b->stream));
            HIPCHK(hipStreamSynchronize(b->stream));
        }
        const int *oi0 = reinterpret_cast<const int *>(reinterpret_cast<const double *>(b->pin_out) + d.n + d.m + 2);
        if (!(pass == 0 && b->recheck_due && oi0[0] == DAQP_EXIT_INFEASIBLE && oi0[1] > 0 && b->n_prox_qps == 0)) break;
        b->recheck_due = false;
        if (redo_one_exact(b)) return DAQP_EXIT_UNSUPPORTED;
    }
    b->recheck_due = false;
    const double *o = reinterpret_cast<const double *>(b->pin_out);
    const int *oi = reinterpret_cast<const int *>(o + d.n + d.m + 2);
    // a solve counts from 1: flag < 0 with iter 0 is a setup / update flag, and -6 only ever comes out of an activation (auxiliary.c:399-479:
    // part of the setup / update in the reference) -- x and lam are not results then (api.c:70-78)
    const bool setup_failed = oi[0] < 0 && (oi[1] == 0 || oi[0] == DAQP_EXIT_OVERDETERMINED_INITIAL);
    if (!setup_failed) {
        if (r->x) memcpy(r->x, o, d.n * sizeof(double));
        if (r->lam && d.m) memcpy(r->lam, o + d.n, d.m * sizeof(double));
    }
    if (r->fval) *r->fval = o[d.n + d.m];
    if (r->soft_slack) *r->soft_slack = o[d.n + d.m + 1];
    if (r->exitflag) *r->exitflag = oi[0];
    if (r->iter) *r->iter = oi[1];
    return 0;
}
} // namespace
extern "C" {
int daqp_batch_solve(DAQPBatch *b, DAQPBatchResult *r)
{
    if (!b || !r) { set_err("null batch or result"); return DAQP_EXIT_UNSUPPORTED; }
    if (!b->is_setup) { set_err("daqp_batch_solve before daqp_batch_setup"); return DAQP_EXIT_UNSUPPORTED; }
    HIPCHK(hipSetDevice(b->device));
    BatchDev &d = b->d;
    const bool dev = r->memory == DAQP_MEM_DEVICE;
    d.x = (dev && r->x) ? r->x : b->ox;
    d.lam = (dev && r->lam) ? r->lam : b->olam;
    d.fval = (dev && r->fval) ? r->fval : b->ofval;
    d.soft = (dev && r->soft_slack) ? r->soft_slack : b->osoft;
    d.exitflag = (dev && r->exitflag) ? r->exitflag : b->oflag;
    d.iter = (dev && r->iter) ? r->iter : b->oiter;
    const double t0 = now_s();
    // settings->time_limit (daqp.c:95-103: ONE timer per daqp_solve): every problem's start stamp of this call is cleared here;
    // the first launch that reaches a problem sets it, later launches of the same solve (one-wave fallback behind the workgroup
    // kernel, the launches of the proximal outer loop) inherit it
    if (d.st.time_limit > 0) {
        if (!b->tstart && dev_alloc(b, &b->tstart, (size_t)d.N)) return DAQP_EXIT_UNSUPPORTED;
        HIPCHK(hipMemsetAsync(b->tstart, 0, (size_t)d.N * sizeof(unsigned long long), b->stream));
        d.tstart = b->tstart;
    } else d.tstart = nullptr;
    HIPCHK(hipEventRecord(b->ev[2], b->stream));
    const int mode = b->pending_mask ? (2 | (b->pending_mask << 4)) : 0;
    b->pending_mask = 0;
    int rc;
    if (b->reg_pending) {
        // straight after a setup: the solve launch goes out before the host knows whether any Hessian was singular (such
        // problems sit it out with their internal flag); only then is the count read.  No gap on the device in the common case.
        bool had_flagged = false;
        rc = launch_ldp(b, mode);
        if (!rc) rc = resolve_setup(b, &had_flagged);
        if (!rc && b->n_prox_qps > 0) rc = solve_with_prox(b, mode, true);
        // the early launch wrote the internal "needs the shift" code for the flagged problems; those whose regularising passes
        // failed (-5 after the doublings, -1 for a zero row) report that now; the others were rewritten by their outer loop
        if (!rc && had_flagged) {
            hipLaunchKernelGGL(k_report_failed_setups, dim3((d.N + 127) / 128), dim3(128), 0, b->stream, d);
            HIPCHK(hipGetLastError());
        }
    } else rc = b->n_prox_qps > 0 ? solve_with_prox(b, mode) : launch_ldp(b, mode);
    if (rc) return rc;
    // default arithmetic, first solve after a setup: INFEASIBLE verdicts are re-derived in the reference's arithmetic (recheck.hip.h)
    b->timed_recheck = false;
    if (b->fresh && b->recheck && d.exact_setup == 0 && !b->was_shared && recheck_allowed(b)) { rc = recheck_infeasible(b); if (rc) return rc; }
    else b->rechecked = 0;
    b->fresh = false;
    HIPCHK(hipEventRecord(b->ev[3], b->stream));
    b->timed_solve = true;
    if (!dev && d.N == 1 && b->pin_out != nullptr) {   // one problem: the result slab, mapped (the kernels wrote it in place) or in one copy
        if (b->defer_wait) return 0;                    // (daqp_ldp: more work goes behind this, one wait for all of it, then collect_one)
        rc = collect_one(b, r);
        if (rc) return rc;
        r->solve_time = now_s() - t0;
    } else if (!dev) {
        const size_t N = d.N;
        if (r->x) HIPCHK(hipMemcpyAsync(r->x, b->ox, N * d.n * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        if (r->lam) HIPCHK(hipMemcpyAsync(r->lam, b->olam, N * d.m * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        if (r->fval) HIPCHK(hipMemcpyAsync(r->fval, b->ofval, N * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        if (r->soft_slack) HIPCHK(hipMemcpyAsync(r->soft_slack, b->osoft, N * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        if (r->exitflag) HIPCHK(hipMemcpyAsync(r->exitflag, b->oflag, N * sizeof(int), hipMemcpyDeviceToHost, b->stream));
        if (r->iter) HIPCHK(hipMemcpyAsync(r->iter, b->oiter, N * sizeof(int), hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        r->solve_time = now_s() - t0;
    }
    return 0;
}

// api.c:636-641 for every problem: the point the proximal iterations start from (x: N*n).  Only problems whose Hessian
// needed the shift ever read it; a later daqp_batch_solve continues from the previous solution, as the reference does.
int daqp_batch_set_primal_start(DAQPBatch *b, const c_float *x, int memory)
{
    if (!b || !x) return DAQP_EXIT_UNSUPPORTED;
    HIPCHK(hipSetDevice(b->device));
    if (resolve_setup(b)) return DAQP_EXIT_UNSUPPORTED;
    if (!b->prox_ready) return 0;   // no singular Hessian seen: nothing would read it
    HIPCHK(hipMemcpyAsync(b->px.center, x, (size_t)b->d.N * b->d.n * sizeof(double),
                          memory == DAQP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, b->stream));
    if (memory != DAQP_MEM_DEVICE) HIPCHK(hipStreamSynchronize(b->stream));
    return 0;
}

// Which problems of the last setup go through the proximal outer loop: n_prox (types.h:229; host int[N], 0 = ordinary),
// outer iterations of the last solve (host int[N]) and the shift eps (host double[N]).  Any pointer may be NULL.
// Returns the number of proximal problems (>= 0) or a negative flag.
int daqp_batch_prox_info(DAQPBatch *b, int *n_prox_host, int *outer_host, c_float *eps_host)
{
    if (!b) return DAQP_EXIT_UNSUPPORTED;
    HIPCHK(hipSetDevice(b->device));
    if (resolve_setup(b)) return DAQP_EXIT_UNSUPPORTED;
    HIPCHK(hipStreamSynchronize(b->stream));
    const int N = b->d.N;
    if (n_prox_host) {
        std::vector<QState> qs(N);
        HIPCHK(hipMemcpy(qs.data(), b->d.qs, sizeof(QState) * N, hipMemcpyDeviceToHost));
        for (int i = 0; i < N; ++i) n_prox_host[i] = qs[i].setup_flag > 0 ? qs[i].n_prox : 0;
    }
    if (outer_host) {
        for (int i = 0; i < N; ++i) outer_host[i] = 0;
        if (b->prox_ready) {
            std::vector<int> st(4 * (size_t)N);
            HIPCHK(hipMemcpy(st.data(), b->px.state, sizeof(int) * 4 * N, hipMemcpyDeviceToHost));
            for (int i = 0; i < N; ++i) outer_host[i] = st[4 * (size_t)i + 3];
        }
    }
    if (eps_host) {
        for (int i = 0; i < N; ++i) eps_host[i] = 0;
        if (b->prox_ready && b->n_prox_qps > 0) HIPCHK(hipMemcpy(eps_host, b->px.eps, sizeof(double) * N, hipMemcpyDeviceToHost));
    }
    return b->n_prox_qps;
}

int daqp_batch_setup_flags(DAQPBatch *b, int *flags_host)
{
    if (!b || !flags_host) return DAQP_EXIT_UNSUPPORTED;
    HIPCHK(hipSetDevice(b->device));
    if (resolve_setup(b)) return DAQP_EXIT_UNSUPPORTED;
    if (flush_update(b)) return DAQP_EXIT_UNSUPPORTED;
    HIPCHK(hipStreamSynchronize(b->stream));
    std::vector<QState> qs(b->d.N);
    HIPCHK(hipMemcpy(qs.data(), b->d.qs, sizeof(QState) * b->d.N, hipMemcpyDeviceToHost));
    // (a daqp_update_ldp whose bound check failed reports its flag here until the next update, as the call itself does in the reference)
    for (int i = 0; i < b->d.N; ++i) flags_host[i] = (qs[i].setup_flag > 0 && qs[i].upd_flag < 0) ? qs[i].upd_flag : qs[i].setup_flag;
    return 0;
}

int daqp_batch_working_sets(DAQPBatch *b, int *n_active_host, int *ws_host)
{
    if (!b) return DAQP_EXIT_UNSUPPORTED;
    HIPCHK(hipSetDevice(b->device));
    if (resolve_setup(b)) return DAQP_EXIT_UNSUPPORTED;
    if (flush_update(b)) return DAQP_EXIT_UNSUPPORTED;
    HIPCHK(hipStreamSynchronize(b->stream));
    if (n_active_host) {
        std::vector<QState> qs(b->d.N);
        HIPCHK(hipMemcpy(qs.data(), b->d.qs, sizeof(QState) * b->d.N, hipMemcpyDeviceToHost));
        for (int i = 0; i < b->d.N; ++i) n_active_host[i] = qs[i].n_active;
    }
    if (ws_host) HIPCHK(hipMemcpy(ws_host, b->d.WS, sizeof(int) * (size_t)b->d.N * b->d.cap, hipMemcpyDeviceToHost));
    return 0;
}

int daqp_batch_kernel_ms(DAQPBatch *b, float *setup_ms, float *solve_ms)
{
    if (!b) return DAQP_EXIT_UNSUPPORTED;
    HIPCHK(hipSetDevice(b->device));
    if (setup_ms) {
        *setup_ms = 0;
        if (b->timed_setup) { HIPCHK(hipEventSynchronize(b->ev[1])); HIPCHK(hipEventElapsedTime(setup_ms, b->ev[0], b->ev[1])); }
    }
    if (solve_ms) {
        *solve_ms = 0;
        if (b->timed_solve) { HIPCHK(hipEventSynchronize(b->ev[3])); HIPCHK(hipEventElapsedTime(solve_ms, b->ev[2], b->ev[3])); }
    }
    return 0;
}

int daqp_quadprog_batch(DAQPBatchResult *r, const DAQPBatchProblem *p, const DAQPSettings *settings)
{
    if (!r || !p) { set_err("null argument"); return DAQP_EXIT_UNSUPPORTED; }
    if (p->N == 0) { r->setup_time = r->solve_time = 0; return 0; }   // empty batch: nothing to do, not an error
    int ns = 0;
    if (p->sense && p->memory == DAQP_MEM_HOST) {
        for (int q = 0; q < p->N; ++q) {
            int c = 0;
            for (int i = 0; i < p->m; ++i) c += (p->sense[(size_t)q * p->m + i] & DAQP_SOFT) ? 1 : 0;
            if (c > ns) ns = c;
        }
    } else if (p->sense) ns = p->m < 512 - p->n - 1 ? p->m : 512 - p->n - 1; // device-resident sense: size for the worst case
    DAQPBatch *b = nullptr;
    int rc = daqp_batch_create(&b, p->N, p->n, p->m, p->ms, ns, settings, -1);
    if (rc) return rc;
    const double t0 = now_s();
    rc = daqp_batch_setup(b, p, DAQP_UPDATE_unconstrained | DAQP_UPDATE_eliminate);
    if (rc == 0) {
        if (hipStreamSynchronize(b->stream) != hipSuccess) rc = DAQP_EXIT_UNSUPPORTED;
        r->setup_time = now_s() - t0;
    }
    if (rc == 0) rc = daqp_batch_solve(b, r);
    if (rc == 0 && r->memory == DAQP_MEM_DEVICE) { if (hipStreamSynchronize(b->stream) != hipSuccess) rc = DAQP_EXIT_UNSUPPORTED; }
    daqp_batch_free(b);
    return rc;
}

} // extern "C"
#include "multi.hip.h"
extern "C" {

// ------------------------------------------------------------------------------------
// single-problem drop-in entry points: a batch of one behind the reference's workspace struct
// ------------------------------------------------------------------------------------
static DAQPBatch *ws_batch(DAQPWorkspace *w) { return reinterpret_cast<DAQPBatch *>(w->timer); }
static void free_daqp_workspace_keep_settings(DAQPWorkspace *work);

static DAQPBatchProblem one_problem(const DAQPProblem *qp)
{
    DAQPBatchProblem p;
    p.N = 1; p.n = qp->n; p.m = qp->m; p.ms = qp->ms;
    p.H = qp->H; p.f = qp->f; p.A = qp->A; p.bupper = qp->bupper; p.blower = qp->blower; p.sense = qp->sense;
    p.memory = DAQP_MEM_HOST;
    return p;
}

// Host mirrors of the device state behind a single-problem workspace.  `ldp`: also the LDP itself (M, R^-1, v, d, scaling:
// after a setup or an update; read-only copies for bindings that inspect them -- interfaces/daqp-eigen/daqp.cpp:250-271
// reads Rinv / RinvD / v / sense -- writing to them does not reach the device).
// enqueue the gather (k_mirror) into the mirror slab -- mapped host memory that the kernel writes in place, or a device slab plus copies
static int mirror_enqueue(DAQPBatch *b, bool ldp, bool full)
{
    if (!ldp) full = false;
    (void)hipSetDevice(b->device);
    const BatchDev &d = b->d;
    const size_t dbl = mirror_doubles(d.n, d.m, d.ms, d.cap, d.rtri);
    if (b->dev_mir == nullptr) {
        void *dp = nullptr;
        static const bool no_map = [] { const char *e = getenv("DAQP_AMD_NO_MAPPED_RESULTS"); return e && atoi(e) != 0; }();
        if (!no_map && hipHostMalloc(reinterpret_cast<void **>(&b->pin_mir), dbl * sizeof(double), hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer(&dp, b->pin_mir, 0) == hipSuccess) {
            b->dev_mir = static_cast<double *>(dp);
            b->mapped_mir = true;
        } else {
            (void)hipGetLastError();
            if (b->pin_mir) { (void)hipHostFree(b->pin_mir); b->pin_mir = nullptr; }
            if (dev_alloc(b, &b->dev_mir, dbl) || hipHostMalloc(reinterpret_cast<void **>(&b->pin_mir), dbl * sizeof(double), hipHostMallocDefault) != hipSuccess) return 1;
        }
    }
    // one gather launch: the LDP part only when it may have changed (after a setup or an update), and of it R^-1 and M only when the
    // caller says those changed too
    const int blocks = full ? (int)(((size_t)(d.m - d.ms) * d.n + 255) / 256 > 64 ? 64 : ((size_t)(d.m - d.ms) * d.n + 255) / 256 + 1) : 1;
    hipLaunchKernelGGL(k_mirror, dim3(blocks), dim3(256), 0, b->stream, d, b->dev_mir, ldp ? (full ? 2 : 1) : 0);
    if (hipGetLastError() != hipSuccess) return 1;
    if (b->mapped_mir) return 0;
    const size_t io = mirror_int_off(d.n, d.m, d.ms, d.cap, d.rtri);
    if (ldp && full) {
        if (hipMemcpyAsync(b->pin_mir, b->dev_mir, dbl * sizeof(double), hipMemcpyDeviceToHost, b->stream) != hipSuccess) return 1;
    } else {   // head (scalars, lam_star and, with ldp, v / d / scaling) and tail (WS, sense)
        const size_t head = mirror_ldp_off(d.cap) + (ldp ? (size_t)d.n + 3 * (size_t)d.m : 0);
        if (hipMemcpyAsync(b->pin_mir, b->dev_mir, head * sizeof(double), hipMemcpyDeviceToHost, b->stream) != hipSuccess) return 1;
        if (hipMemcpyAsync(b->pin_mir + io, b->dev_mir + io, (dbl - io) * sizeof(double), hipMemcpyDeviceToHost, b->stream) != hipSuccess) return 1;
    }
    return 0;
}
// the slab has arrived: into the workspace's fields
static void mirror_parse(DAQPWorkspace *w, DAQPBatch *b, bool ldp, bool full)
{
    if (!ldp) full = false;
    const BatchDev &d = b->d;
    const size_t io = mirror_int_off(d.n, d.m, d.ms, d.cap, d.rtri);
    QState qs;
    memcpy(&qs, b->pin_mir, sizeof(QState));
    w->n_active = qs.n_active; w->reuse_ind = qs.reuse_ind; w->sing_ind = qs.sing_ind;
    w->iterations = qs.iterations; w->fval = qs.fval; w->soft_slack = qs.soft_slack;
    const int *ints = reinterpret_cast<const int *>(b->pin_mir + io);
    if (w->WS) memcpy(w->WS, ints, sizeof(int) * d.cap);
    if (w->sense) memcpy(w->sense, ints + d.cap, sizeof(int) * d.m);
    if (w->lam_star) memcpy(w->lam_star, b->pin_mir + 16, sizeof(double) * d.cap);
    if (!ldp || qs.setup_flag < 0) return;
    const bool lp = b->ident && d.H == b->ident;
    const double *o = b->pin_mir + mirror_ldp_off(d.cap);
    if (w->v) memcpy(w->v, o, sizeof(double) * d.n);
    o += d.n;
    if (w->dupper) memcpy(w->dupper, o, sizeof(double) * d.m);
    if (w->dlower) memcpy(w->dlower, o + d.m, sizeof(double) * d.m);
    if (w->scaling) memcpy(w->scaling, o + 2 * (size_t)d.m, sizeof(double) * d.m);
    o += 3 * (size_t)d.m;
    if (!full) return;
    if (!lp) {
        if (qs.diag_h) {   // the reference's RinvD branch (utils.c:245-312): Rinv == NULL, RinvD = 1/sqrt(H_ii)
            if (w->RinvD) for (int i = 0; i < d.n; ++i) w->RinvD[i] = o[((2 * d.n - i - 1) * i) / 2 + i];
        } else if (w->Rinv) memcpy(w->Rinv, o, sizeof(double) * d.rtri);
    }
    o += d.rtri;
    if (w->M && d.mA > 0) memcpy(w->M, o, sizeof(double) * (size_t)d.mA * d.n);   // (m - ms) x n row-major, rows normalised
}
static void refresh_mirrors(DAQPWorkspace *w, bool ldp = false, bool full = true)
{
    DAQPBatch *b = ws_batch(w);
    if (!b) return;
    if (mirror_enqueue(b, ldp, full)) return;
    if (b->mapped_mir ? wait_stream(b) != 0 : hipStreamSynchronize(b->stream) != hipSuccess) return;
    mirror_parse(w, b, ldp, full);
}

void allocate_daqp_settings(DAQPWorkspace *work)
{
    if (work->settings == nullptr) {
        work->settings = static_cast<DAQPSettings *>(malloc(sizeof(DAQPSettings)));
        default_settings(work->settings);
    }
}

int setup_daqp_main(DAQPProblem *qp, DAQPWorkspace *work, c_float *setup_time, int init_mask)
{
    const double t0 = now_s();
    if (setup_time) *setup_time = 0;
    int own_settings = 1;
    if (qp->problem_type != 0 || qp->nh > 1 || qp->break_points != nullptr || qp->f == nullptr) {
        set_err("AVI / hierarchical problems and problems without a linear term are outside this path");
        return DAQP_EXIT_UNSUPPORTED;
    }
    int ns = 0;
    if (qp->sense)
        for (int i = 0; i < qp->m; ++i) {
            if (qp->sense[i] & DAQP_SOFT) ns++;
            if (qp->sense[i] & DAQP_BINARY) { set_err("binary constraints are outside this path"); return DAQP_EXIT_UNSUPPORTED; }
        }
    if (work->settings == nullptr) allocate_daqp_settings(work); else own_settings = 0;
    DAQPBatch *b = nullptr;
    int rc = daqp_batch_create(&b, 1, qp->n, qp->m, qp->ms, ns, work->settings, -1);
    if (rc == 0) {
        DAQPBatchProblem p = one_problem(qp);
        rc = daqp_batch_setup(b, &p, init_mask);
        int flag = 1;
        if (rc == 0) rc = daqp_batch_setup_flags(b, &flag);
        if (rc == 0 && flag < 0) rc = flag;
    }
    if (rc < 0) {
        if (b) daqp_batch_free(b);
        if (own_settings) { free(work->settings); }
        work->settings = own_settings ? nullptr : work->settings;
        return rc;
    }
    // host-visible part of the workspace (everything numerical stays on the device)
    work->qp = qp; work->n = qp->n; work->m = qp->m; work->ms = qp->ms;
    work->xold = work->lam = work->u = nullptr;
    work->L = work->D = work->xldl = work->zldl = work->Mu = nullptr;
    {   // read-only host mirrors of the LDP (api.c:343-371 allocates the originals)
        const size_t n = qp->n, m = qp->m, mA = qp->m - qp->ms;
        work->M = mA ? static_cast<c_float *>(calloc(mA * n, sizeof(c_float))) : nullptr;
        work->dupper = static_cast<c_float *>(calloc(m ? m : 1, sizeof(c_float)));
        work->dlower = static_cast<c_float *>(calloc(m ? m : 1, sizeof(c_float)));
        work->scaling = static_cast<c_float *>(calloc(m ? m : 1, sizeof(c_float)));
        work->v = static_cast<c_float *>(calloc(n, sizeof(c_float)));
        QState q0;
        const bool diag = hipMemcpy(&q0, b->d.qs, sizeof(QState), hipMemcpyDeviceToHost) == hipSuccess && q0.diag_h;
        const bool lp = qp->H == nullptr;
        work->Rinv = (!lp && !diag) ? static_cast<c_float *>(calloc(n * (n + 1) / 2, sizeof(c_float))) : nullptr;
        work->RinvD = (!lp && diag) ? static_cast<c_float *>(calloc(n, sizeof(c_float))) : nullptr;
    }
    work->prox_mask = nullptr; work->n_prox = 0; work->bnb = nullptr; work->avi = nullptr; work->eq = nullptr;
    work->nh = 1; work->break_points = nullptr;
    work->sense = static_cast<int *>(calloc(qp->m > 0 ? qp->m : 1, sizeof(int)));
    work->x = static_cast<c_float *>(calloc(qp->n, sizeof(c_float)));
    work->lam_star = static_cast<c_float *>(calloc(b->d.cap, sizeof(c_float)));
    work->WS = static_cast<int *>(calloc(b->d.cap, sizeof(int)));
    work->timer = b;
    refresh_mirrors(work, true);
    (void)daqp_batch_prox_info(b, &work->n_prox, nullptr, nullptr);   // types.h:229: > 0 sends daqp_solve through daqp_prox
    if (setup_time) *setup_time = now_s() - t0;
    return 1;
}

int setup_daqp(DAQPProblem *qp, DAQPWorkspace *work, c_float *setup_time) { return setup_daqp_main(qp, work, setup_time, 0); }

// daqp_update_ldp with a mask within v | d on a workspace whose solve kernel applies such an update itself (the register kernel: it has the
// rows of M in registers anyway).  Nothing goes to the device here: the new f / bounds are put into mapped host memory that the next solve
// launch reads in place, and the one thing the call has to RETURN -- the bound check's verdict, utils.c:94-98 -- is formed on the host
// from the same numbers the device will look at (the workspace's sense as the last solve / update left it: the host mirror).  A check
// that fails here takes the ordinary path (the device's own check decides, marks what the reference marks and records the flag).
// The host mirrors of v / dupper / dlower follow with the next daqp_solve's gather (they are copies for inspection, not inputs).
static bool update_one_deferred(DAQPWorkspace *work, DAQPBatch *b, int m, const DAQPProblem *qp)
{
    const BatchDev &dd = b->d;
    if ((m & ~(DAQP_UPDATE_v | DAQP_UPDATE_d)) || !(m & (DAQP_UPDATE_v | DAQP_UPDATE_d))) return false;
    if (dd.N != 1 || b->NB == 0 || b->reg_handover || !b->is_setup || b->reg_pending || b->was_shared || !work->sense || getenv("DAQP_AMD_EAGER_UPDATE")) return false;
    if (!qp->bupper || !qp->blower || ((m & DAQP_UPDATE_v) && !qp->f)) return false;
    const int n = dd.n, mm = dd.m;
    for (int i = 0; i < mm; ++i) {      // utils.c:546-567 on the mirror of the workspace's sense
        if (work->sense[i] & DAQP_IMMUTABLE) continue;
        const double diff = qp->bupper[i] - qp->blower[i];
        if (diff < -dd.st.primal_tol) return false;                                        // crossed: the device's check decides and records
        if (diff < dd.st.zero_tol && !(work->sense[i] & DAQP_SOFT)) return false;          // an unmarked equality: marked and ACTIVATED by the update (its flag may be -6)
    }
    if (!b->pin_upd) {
        void *dp = nullptr;
        if (hipSetDevice(b->device) != hipSuccess) return false;
        if (hipHostMalloc(reinterpret_cast<void **>(&b->pin_upd), ((size_t)n + 2 * (size_t)mm) * sizeof(double), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer(&dp, b->pin_upd, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (b->pin_upd) { (void)hipHostFree(b->pin_upd); b->pin_upd = nullptr; }
            return false;
        }
        b->pin_upd_dev = static_cast<char *>(dp);
    }
    // (a solve launch that still reads an earlier update's numbers has been waited for by its daqp_solve: the buffer is free)
    double *h = reinterpret_cast<double *>(b->pin_upd);
    const double *dv = reinterpret_cast<const double *>(b->pin_upd_dev);
    BatchDev &d = b->d;
    if (m & DAQP_UPDATE_v) { memcpy(h, qp->f, n * sizeof(double)); d.f = dv; }
    memcpy(h + n, qp->bupper, mm * sizeof(double));
    memcpy(h + n + mm, qp->blower, mm * sizeof(double));
    d.bu = dv + n; d.bl = dv + n + mm;
    b->fresh = false;
    b->exact_sticky = false;
    b->pending_mask |= m;
    b->timed_setup = false;
    b->mirror_ldp_due = true;
    return true;
}

int daqp_update_ldp(const int mask, DAQPWorkspace *work, DAQPProblem *qp)
{
    DAQPBatch *b = ws_batch(work);
    if (!b) { set_err("workspace has not been set up"); return DAQP_EXIT_UNSUPPORTED; }
    if (qp->n != work->n || qp->m != work->m || qp->ms != work->ms) { set_err("daqp_update_ldp: the problem's dimensions differ from the workspace's"); return DAQP_EXIT_UNSUPPORTED; }
    if ((mask & DAQP_UPDATE_hierarchy) && (qp->nh > 1 || qp->break_points != nullptr)) { set_err("hierarchies are outside this path"); return DAQP_EXIT_UNSUPPORTED; }
    work->qp = qp;
    DAQPBatchProblem p = one_problem(qp);
    const int m = mask & ~(DAQP_UPDATE_hierarchy);
    b->one_valid = false;
    if (update_one_deferred(work, b, m, qp)) return 0;
    int rc = daqp_batch_update(b, m, &p);
    if (rc < 0) return rc;
    int flag = 1;
    rc = daqp_batch_setup_flags(b, &flag);
    if (rc < 0) return rc;
    if (flag < 0) {
        // An update that ends at its bound check (or at a vanishing row whose bounds exclude 0) has told its caller so; the
        // reference's workspace is then what it was (utils.c:95-96 comes before anything is recomputed) and a daqp_solve on it
        // solves the previous LDP.  The batch API reports the flag from the next solve instead (a batch caller has no per-problem
        // return value to look at); here the caller has it, so the record is cleared and the next daqp_solve runs as the reference's.
        QState q0;
        if (hipMemcpy(&q0, b->d.qs, sizeof(QState), hipMemcpyDeviceToHost) == hipSuccess && q0.setup_flag > 0 && q0.upd_flag < 0) {
            (void)hipMemsetAsync(&b->d.qs[0].upd_flag, 0, sizeof(int), b->stream);
            if (m & DAQP_UPDATE_sense) b->exact_sticky = true;      // the new sense is in, the working set was not rebuilt
        }
    }
    if ((m & DAQP_UPDATE_Rinv) && flag > 0) {
        // a new Hessian may be of the other kind (dense <-> diagonal: Rinv <-> RinvD, utils.c:245-312) and may or may not need the shift
        QState q0;
        if (hipMemcpy(&q0, b->d.qs, sizeof(QState), hipMemcpyDeviceToHost) == hipSuccess) {
            const size_t n = (size_t)work->n;
            if (q0.diag_h && work->RinvD == nullptr) { free(work->Rinv); work->Rinv = nullptr; work->RinvD = static_cast<c_float *>(calloc(n, sizeof(c_float))); }
            if (!q0.diag_h && work->Rinv == nullptr && qp->H != nullptr) { free(work->RinvD); work->RinvD = nullptr; work->Rinv = static_cast<c_float *>(calloc(n * (n + 1) / 2, sizeof(c_float))); }
        }
        (void)daqp_batch_prox_info(b, &work->n_prox, nullptr, nullptr);
    }
    refresh_mirrors(work, true, (m & (DAQP_UPDATE_Rinv | DAQP_UPDATE_M)) != 0);
    return flag < 0 ? flag : 0;
}

// daqp.h:12.  Here the device launch that iterates also back-transforms (ldp2qp_solution, daqp.c:111-139) and assembles the
// per-constraint multipliers; what daqp_extract_result needs is parked behind the workspace.  Returns the exit flag.
int daqp_ldp(DAQPWorkspace *work)
{
    DAQPBatch *b = ws_batch(work);
    if (!b) { set_err("workspace has not been set up"); return DAQP_EXIT_UNSUPPORTED; }
    if (work->settings) daqp_batch_set_settings(b, work->settings);
    DAQPBatchResult r;
    memset(&r, 0, sizeof(r));
    b->one_lam.resize(work->m > 0 ? work->m : 1);
    r.x = work->x; r.lam = b->one_lam.data(); r.fval = &b->one_fval; r.soft_slack = &b->one_soft; r.exitflag = &b->one_flag; r.iter = &b->one_iter;
    r.memory = DAQP_MEM_HOST;
    b->one_valid = false;
    // the solve launch(es), then the gather of the workspace's host mirrors behind them -- with the v / d part when a deferred
    // daqp_update_ldp has just been applied by that launch --, ONE wait for all of it
    const bool with_ldp = b->mirror_ldp_due;
    const bool fuse = b->mapped_out && b->pin_out != nullptr && !(b->fresh && b->recheck && b->d.exact_setup == 0);   // (a first solve may be re-derived: collect first)
    b->defer_wait = fuse;
    int rc = daqp_batch_solve(b, &r);
    b->defer_wait = false;
    if (rc < 0) { b->one_flag = rc; return rc; }
    if (fuse) {
        if (mirror_enqueue(b, with_ldp, false) || !b->mapped_mir) {     // (no mapped mirror slab: its copies are behind the launch; wait for everything)
            if (hipStreamSynchronize(b->stream) != hipSuccess) { b->one_flag = DAQP_EXIT_UNSUPPORTED; return DAQP_EXIT_UNSUPPORTED; }
        }
        rc = collect_one(b, &r);
        if (rc < 0) { b->one_flag = rc; return rc; }
        mirror_parse(work, b, with_ldp, false);
    } else refresh_mirrors(work, with_ldp, false);
    b->mirror_ldp_due = false;
    work->iterations = b->one_iter;
    b->one_valid = true;
    return b->one_flag;
}
void ldp2qp_solution(DAQPWorkspace *work) { (void)work; }   // daqp.h:13: done on the device by daqp_ldp (work->x already holds x)

// api.c:455-495: package the last daqp_ldp / daqp_solve of this workspace
// Everything comes from the workspace's own fields, as in the reference: x, the multipliers scattered from WS / lam_star /
// n_active, fval = 1/2 (work->fval - |v|^2), iterations, soft_slack -- the host mirrors that daqp_ldp refreshed.  (A proximal
// workspace reports the objective its outer loop formed on the device; an LP f'x.)
void daqp_extract_result(DAQPResult *res, DAQPWorkspace *work)
{
    DAQPBatch *b = ws_batch(work);
    if (!b || !res) return;
    if (res->x && work->x) for (int i = 0; i < work->n; ++i) res->x[i] = work->x[i];
    if (res->lam) {
        for (int i = 0; i < work->m; ++i) res->lam[i] = 0;
        if (work->WS && work->lam_star)
            for (int i = 0; i < work->n_active; ++i)
                if (work->WS[i] >= 0 && work->WS[i] < work->m) res->lam[work->WS[i]] = work->lam_star[i];
    }
    if (work->n_prox > 0 && b->one_valid) res->fval = b->one_fval;
    else if (work->v != nullptr && (work->Rinv != nullptr || work->RinvD != nullptr)) {
        c_float fv = work->fval;
        for (int i = 0; i < work->n; ++i) fv -= work->v[i] * work->v[i];
        res->fval = fv * 0.5;
    } else if (work->qp != nullptr && work->qp->f != nullptr && work->x != nullptr) {
        c_float fv = 0;
        for (int i = 0; i < work->n; ++i) fv += work->qp->f[i] * work->x[i];
        res->fval = fv;
    }
    res->soft_slack = work->soft_slack; res->iter = work->iterations;
    res->nodes = b->n_prox_qps > 0 ? b->prox_outer : 1;   // api.c:488: work->nh, the outer iterations of daqp_prox (daqp_prox.c:34,129)
}

void daqp_solve(DAQPResult *res, DAQPWorkspace *work)   // api.c:8-59
{
    if (!ws_batch(work)) { res->exitflag = DAQP_EXIT_UNSUPPORTED; set_err("workspace has not been set up"); return; }
    const double t0 = now_s();
    res->exitflag = daqp_ldp(work);
    daqp_extract_result(res, work);
    res->solve_time = now_s() - t0;
}

// api.c:161-209: QP -> LDP on a workspace (what setup_daqp_main does after its checks).  The device workspace is created
// here; returns 1 or a negative exit flag.
int setup_daqp_ldp(DAQPWorkspace *work, DAQPProblem *qp, const int init_mask)
{
    if (!work || !qp) return DAQP_EXIT_UNSUPPORTED;
    if (ws_batch(work)) { free_daqp_workspace_keep_settings(work); free_daqp_ldp(work); }
    return setup_daqp_main(qp, work, nullptr, init_mask);
}

static void free_daqp_workspace_keep_settings(DAQPWorkspace *work)
{
    if (work->timer) { daqp_batch_free(ws_batch(work)); work->timer = nullptr; }
    free(work->x); work->x = nullptr;
    free(work->lam_star); work->lam_star = nullptr;
    free(work->WS); work->WS = nullptr;
}
void free_daqp_workspace(DAQPWorkspace *work)
{
    free_daqp_workspace_keep_settings(work);
    if (work->settings != nullptr) { free(work->settings); work->settings = nullptr; }
}

void free_daqp_ldp(DAQPWorkspace *work)   // api.c:243-275
{
    if (work->sense == nullptr) return;
    free(work->sense); work->sense = nullptr;
    free(work->M); free(work->dupper); free(work->dlower); free(work->scaling); free(work->v); free(work->Rinv); free(work->RinvD);
    work->M = work->dupper = work->dlower = work->scaling = work->v = work->Rinv = work->RinvD = nullptr;
}

// api.c:61-104.  One-shot: no workspace survives the call, so none of its host mirrors are built -- a (parked) single-problem
// batch, one packed copy in, setup + solve launches, one packed copy out, one synchronisation.
void daqp_quadprog(DAQPResult *res, DAQPProblem *qp, DAQPSettings *settings)
{
    res->setup_time = 0; res->solve_time = 0;
    if (qp->problem_type != 0 || qp->nh > 1 || qp->break_points != nullptr || qp->f == nullptr) {
        set_err("AVI / hierarchical problems and problems without a linear term are outside this path");
        res->exitflag = DAQP_EXIT_UNSUPPORTED;
        return;
    }
    int ns = 0;
    if (qp->sense)
        for (int i = 0; i < qp->m; ++i) {
            if (qp->sense[i] & DAQP_SOFT) ns++;
            if (qp->sense[i] & DAQP_BINARY) { set_err("binary constraints are outside this path"); res->exitflag = DAQP_EXIT_UNSUPPORTED; return; }
        }
    const double t0 = now_s();
    DAQPBatch *b = nullptr;
    int rc = daqp_batch_create(&b, 1, qp->n, qp->m, qp->ms, ns, settings, -1);
    if (rc < 0) { res->exitflag = rc; return; }
    DAQPBatchProblem p = one_problem(qp);
    DAQPBatchResult r;
    int iters = 0, eflag = 0;
    // The lean sequence: inputs in one copy, the setup launch(es), the solve launch -- which activates a given working set itself and
    // reports a failed setup as its exit flag (iter 0) --, results written into mapped host memory by the kernel, ONE wait.  No flag
    // read-back between setup and solve.  What that cannot serve comes back with an internal code and takes the full sequence below:
    // a Hessian that needs the shift (the regularising passes and the proximal loop are host-driven), an LP.
    if (b->mapped_out && qp->H != nullptr && !getenv("DAQP_AMD_NO_LEAN")) {
        b->lean = true;
        rc = daqp_batch_setup(b, &p, DAQP_UPDATE_unconstrained | DAQP_UPDATE_eliminate);
        b->lean = false;
        res->setup_time = now_s() - t0;
        if (rc == 0) {
            const double t1 = now_s();
            memset(&r, 0, sizeof(r));
            r.x = res->x; r.lam = res->lam; r.fval = &res->fval; r.soft_slack = &res->soft_slack; r.exitflag = &eflag; r.iter = &iters;
            r.memory = DAQP_MEM_HOST;
            rc = daqp_batch_solve(b, &r);
            if (rc == 0 && eflag > DAQP_NEEDS_SHIFT) {      // (every flag of the reference is above the internal codes)
                res->exitflag = eflag;
                if (!(eflag < 0 && (iters == 0 || eflag == DAQP_EXIT_OVERDETERMINED_INITIAL))) { res->iter = iters; res->nodes = 1; res->solve_time = now_s() - t1; }   // api.c:74-77: a failed setup reports its flag, nothing else
                daqp_batch_free(b);
                return;
            }
        }
        if (rc < 0) { res->exitflag = rc; daqp_batch_free(b); return; }
        // (internal code: the full sequence, from the setup)
    }
    const double t0f = now_s();
    rc = daqp_batch_setup(b, &p, DAQP_UPDATE_unconstrained | DAQP_UPDATE_eliminate);
    int flag = 1;
    if (rc == 0) rc = daqp_batch_setup_flags(b, &flag);
    res->setup_time = now_s() - t0f;
    if (rc < 0 || flag < 0) { res->exitflag = rc < 0 ? rc : flag; daqp_batch_free(b); return; }   // api.c:74-77: no solve after a failed setup
    const double t1 = now_s();
    memset(&r, 0, sizeof(r));
    r.x = res->x; r.lam = res->lam; r.fval = &res->fval; r.soft_slack = &res->soft_slack; r.exitflag = &eflag; r.iter = &iters;
    r.memory = DAQP_MEM_HOST;
    rc = daqp_batch_solve(b, &r);
    res->exitflag = rc < 0 ? rc : eflag;
    res->iter = iters;
    res->nodes = b->n_prox_qps > 0 ? b->prox_outer : 1;
    res->solve_time = now_s() - t1;
    daqp_batch_free(b);
}

// warm-start helpers: pure host code on the caller's sense array (api.c:579-633)
void daqp_primal_init_active(DAQPProblem *qp, c_float *x)
{
    const c_float tol = 1e-9;
    for (int i = 0; i < qp->m; ++i) {
        if (qp->sense[i] & DAQP_IMMUTABLE) continue;
        c_float ax;
        if (i < qp->ms) ax = x[i];
        else {
            ax = 0;
            const c_float *row = qp->A + (size_t)(i - qp->ms) * qp->n;
            for (int j = 0; j < qp->n; ++j) ax += x[j] * row[j];
        }
        c_float slack = ax - qp->bupper[i];
        if (slack < tol && slack > -tol) { qp->sense[i] |= DAQP_ACTIVE; qp->sense[i] &= ~DAQP_LOWER; }
        else {
            slack = ax - qp->blower[i];
            if (slack < tol && slack > -tol) qp->sense[i] |= DAQP_ACTIVE + DAQP_LOWER;
        }
    }
}
void daqp_dual_init_active(DAQPProblem *qp, c_float *lam)
{
    const c_float tol = 1e-12;
    for (int i = 0; i < qp->m; ++i) {
        if (qp->sense[i] & DAQP_IMMUTABLE) continue;
        if (lam[i] > tol) { qp->sense[i] |= DAQP_ACTIVE; qp->sense[i] &= ~DAQP_LOWER; }
        else if (lam[i] < -tol) qp->sense[i] |= DAQP_ACTIVE + DAQP_LOWER;
    }
}
// api.c:636-641: the starting point of the proximal-point iterations.  This path (strictly convex QPs) never reads it --
// daqp_solve overwrites x -- but the Cython binding declares and calls the symbol (daqp.pxd:74, daqp.pyx:53,438).
void daqp_set_primal_start(DAQPWorkspace *work, c_float *x)
{
    if (!work || !x || !work->x || work->sing_ind == DAQP_UNCONSTRAINED_OPTIMAL) return;
    for (int i = 0; i < work->n; ++i) work->x[i] = x[i];
    DAQPBatch *b = ws_batch(work);
    if (b) (void)daqp_batch_set_primal_start(b, x, DAQP_MEM_HOST);   // the centre of the first proximal iteration
}

// api.c:296-371: in the reference these malloc the workspace arrays.  Here the numerical state is created on the GPU by
// setup_daqp / setup_daqp_main, so they only record the dimensions (callers that size a workspace by hand before setup --
// api.h:41-42 -- keep working; nothing is allocated that would have to be freed).
void allocate_daqp_workspace(DAQPWorkspace *work, int n, int ns) { if (work) { work->n = n; (void)ns; } }
void allocate_daqp_ldp(DAQPWorkspace *work, int n, int m, int ms, int alloc_R, int alloc_v)
{
    (void)alloc_R; (void)alloc_v;
    if (work) { work->n = n; work->m = m; work->ms = ms; }
}
// api.c:562-574 (host-only): index of the first constraint that x violates by more than tol, m if none
int daqp_first_violating(c_float *x, c_float *A, c_float *bu, c_float *bl, int n, int m, int ms, c_float tol)
{
    int i = 0;
    for (; i < ms; ++i)
        if (x[i] > bu[i] + tol || x[i] < bl[i] - tol) return i;
    for (size_t p = 0; i < m; ++i) {
        c_float ax = 0;
        for (int j = 0; j < n; ++j) ax += A[p++] * x[j];
        if (ax > bu[i] + tol || ax < bl[i] - tol) return i;
    }
    return m;
}
// api.h: daqp_minrep (redundancy removal built on repeated LDP solves, src/api.c) is outside this path.  The symbol exists
// so that the reference's Cython module (daqp.pxd:65) links against this library; it leaves is_redundant untouched and
// records the reason in daqp_amd_last_error().
void daqp_minrep(int *is_redundant, c_float *A, c_float *b, int n, int m, int ms)
{
    (void)is_redundant; (void)A; (void)b; (void)n; (void)m; (void)ms;
    set_err("daqp_minrep is outside the dense-QP path of this library");
}

} // extern "C"
