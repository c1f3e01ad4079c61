// setup_m.hip.h -- k_setup_m: the general rows of the LDP for the generic setup (n > 64), default arithmetic: M = A R^-1 on the f64
// matrix cores, row normalisation (utils.c:586-613), d (utils.c:499-544 / 664-676 + 151-159), the blocked fp64 image and the fp32
// image -- as its OWN launch, one WORKGROUP of four waves per 64 rows of A of one problem.
//
// Why (round 3's counters, profiles/r03a_pmc_summary.json): inside k_setup one wave owns a problem and walks its 38 sixteen-row
// tiles of A; every tile streamed the whole upper triangle of R^-1 past its A operand again -- 162 GB read per 10 000 problems of
// config C4 against 12.9 GB of inputs, and one exposed trip to memory per column tile (the phase was 35 % of the setup).  Here
// the four waves of a workgroup hold four A tiles in registers and SHARE each 16-column tile of R^-1 through LDS: it is read once per
// 64 rows (10 times per problem instead of 38, 16-byte coalesced loads instead of 8-byte fragments), the next tile is on its way
// while the matrix instructions of the current one run, and ten workgroups per problem instead of one wave spread a problem's
// work over the chip.
//
// k_setup (kernels.hip.h) runs first and does everything else -- checks, Cholesky, R^-1 (packed and as the zero-padded square
// image `setup_sq` that the matrix cores read), v, x_unc, the simple bounds, the record -- and leaves `qs->pad_` = what is owed:
//   bit 0: the general rows are deferred to this kernel,  bit 1: the unconstrained shortcut is in play (d from b - A x_unc),
//   bit 2: a simple bound already excludes x_unc.
// The last workgroup of a problem to finish (ticket word b.m_tick[q]) settles the record: an infeasible zero row -> DAQP_EXIT_INFEASIBLE
// (utils.c:598-606), every row feasible at x_unc -> the shortcut (utils.c:677-686).
#pragma once
#include "batch_dev.hip.h"
#include "wave_ldp_reg.hip.h"   // static_for

namespace daqp_amd {

constexpr int kSetupMRows = 64;       // rows of A per workgroup (four waves x one 16-row matrix-core tile)
constexpr int kSetupMNKT = 56;        // k steps of four the A operand holds: n <= 208 (whole blocks of eight: 52 -> 56)

struct SetupMLds { int tile, ob, rs, total_bytes; };
__host__ __device__ inline SetupMLds setup_m_lds()
{
    SetupMLds s;
    s.tile = 0;                         // [224][16] one column tile of R^-1 (rows beyond the triangle are zero in the image)
    s.ob = s.tile + 224 * 16;           // [64][16] the workgroup's results of one column tile on their way to the blocked image
    s.rs = s.ob + 64 * 16;              // [4][64][2] per-row partial sums (|row|^2, row . v) of the four thread quarters
    s.total_bytes = (s.rs + 4 * 64 * 2) * 8;
    return s;
}

__global__ void k_setup_m(BatchDev b, int nrb);      // defined once, in setup_kernel.hip (DAQP_AMD_SETUP_M_IMPL)

#ifdef DAQP_AMD_SETUP_M_IMPL
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_setup_m(BatchDev b, int nrb)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    typedef double v4d __attribute__((ext_vector_type(4)));
    // blockIdx -> (problem, row block), XCD-aware: workgroups are dealt round-robin over the eight XCDs (blockIdx mod 8), each with an L2
    // of its own, so the nrb workgroups of ONE problem are given block indices that are congruent mod 8 -- they run on one XCD, one
    // after the other, and R^-1 (every one of them streams it) comes from HBM / MALL once and from that L2 nrb - 1 times.  Dealt
    // consecutively (q = blockIdx / nrb) the ten workgroups of a C4 problem sat on eight XCDs and fetched it eight times.
    const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3);
    const int q = xcd + 8 * (slot / nrb), rb = slot % nrb;
    if (q >= b.N) return;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = b.n, m = b.m, ms = b.ms, mA = b.mA;
    QState *qs = b.qs + q;
    const int owed = __builtin_amdgcn_readfirstlane(qs->pad_);
    if (!(owed & 1) || __builtin_amdgcn_readfirstlane(qs->setup_flag) <= 0) return;
    const bool unc = (owed & 2) != 0;
    const SetupMLds o = setup_m_lds();
    double *tile = smem + o.tile, *ob = smem + o.ob, *rs = smem + o.rs;
    const DAQPSettings &st = b.st;
    const double *A = b.A + (size_t)q * mA * n, *bu = b.bu + (size_t)q * m, *bl = b.bl + (size_t)q * m;
    const double *vq = b.v + (size_t)q * n;
    const int sq_ld = round_up(n, 16);
    const double *Rsq = b.setup_sq + (size_t)q * round_up(n, 32) * sq_ld;
    double2 *Mq2 = reinterpret_cast<double2 *>(b.Mblk + (size_t)q * b.nblk * b.npair * 128);
    const int kb0 = kSetupMRows * rb;                            // first row of A of this workgroup
    const int rows_wg = (mA - kb0) < kSetupMRows ? (mA - kb0) : kSetupMRows;
    const int lr = lane & 15, lk = lane >> 4;

    // ---- the wave's A operand: lane l holds A[kb + (l & 15)][4 kt + (l >> 4)] for every k step (loaded once, stays in registers)
    const int kb = kb0 + 16 * wv;
    const bool rowok = 16 * wv + lr < rows_wg;
    const double *arow = A + (size_t)(rowok ? kb + lr : kb0) * n;      // (rows beyond the block's last: any valid row, masked below)
    double av[kSetupMNKT];
    static_for<kSetupMNKT / 8>([&](auto nb) __attribute__((always_inline)) {
        if ((n + 31) / 32 == nb + 1) {
            static_for<8 * (nb + 1)>([&](auto kt) __attribute__((always_inline)) {
                const int kk = 4 * kt + lk;
                av[kt] = arow[kk < n ? kk : 0];
            });
            static_for<8 * (nb + 1)>([&](auto kt) __attribute__((always_inline)) {
                const int kk = 4 * kt + lk;
                av[kt] = (rowok && kk < n) ? av[kt] : 0.0;
            });
            static_for<kSetupMNKT - 8 * (nb + 1)>([&](auto kt) __attribute__((always_inline)) { av[8 * (nb + 1) + kt] = 0.0; });
        }
    });

    // ---- column tiles of R^-1 through LDS.  Tile ct = columns 16 ct .. 16 ct + 15, rows 0 .. klast (upper triangular: nothing below);
    // 256 threads fetch it as 16-byte pieces: piece e -> row e / 8, column pair e % 8.  The pieces of the NEXT tile are loaded into
    // registers before the matrix instructions of the current one and stored to LDS after them.
    const int nct = (n + 15) >> 4;
    double2 pf0, pf1, pf2, pf3, pf4, pf5, pf6;                  // 224 rows x 8 pairs / 256 threads = 7 pieces per thread
    // (seven named registers, not an array: indexed through a lambda or a rolled loop the array lands in scratch memory)
#define SETUP_M_F1(ct_, np_, u, var)                                                                                        \
    { const int e = tid + 256 * (u); const int ee = e < (np_) ? e : 0;                                                        \
      var = *reinterpret_cast<const double2 *>(Rsq + (size_t)(ee >> 3) * sq_ld + 16 * (ct_) + 2 * (ee & 7)); }
#define SETUP_M_S1(np_, u, var)                                                                                             \
    { const int e = tid + 256 * (u); if (e < (np_)) *reinterpret_cast<double2 *>(tile + (size_t)(e >> 3) * 16 + 2 * (e & 7)) = var; }
#define SETUP_M_NP(ct_) ((((((n - 1 < 16 * (ct_) + 15) ? n - 1 : 16 * (ct_) + 15) >> 2) + 1) << 2) * 8)   /* whole k steps of four rows x 8 pairs */
#define SETUP_M_FETCH(ct_)                                                                                                  \
    do { const int np_ = SETUP_M_NP(ct_);                                                                                    \
         SETUP_M_F1(ct_, np_, 0, pf0) SETUP_M_F1(ct_, np_, 1, pf1) SETUP_M_F1(ct_, np_, 2, pf2) SETUP_M_F1(ct_, np_, 3, pf3)    \
         SETUP_M_F1(ct_, np_, 4, pf4) SETUP_M_F1(ct_, np_, 5, pf5) SETUP_M_F1(ct_, np_, 6, pf6) } while (0)
#define SETUP_M_STORE(ct_)                                                                                                  \
    do { const int np_ = SETUP_M_NP(ct_);                                                                                    \
         SETUP_M_S1(np_, 0, pf0) SETUP_M_S1(np_, 1, pf1) SETUP_M_S1(np_, 2, pf2) SETUP_M_S1(np_, 3, pf3)                        \
         SETUP_M_S1(np_, 4, pf4) SETUP_M_S1(np_, 5, pf5) SETUP_M_S1(np_, 6, pf6) } while (0)
    // per-row sums of this thread's part of the rows: thread t <-> row (t & 63) of the workgroup's block, column pairs (t >> 6) + 4 j
    double s2 = 0, sv = 0;
    const int prow = tid & 63, pq = tid >> 6;
    const int gi_p = ms + kb0 + (prow < rows_wg ? prow : 0);
    SETUP_M_FETCH(0);
    // (workgroup barriers that wait for LDS traffic only: __syncthreads() also waits for every outstanding global access of the wave
    //  -- here the image stores of the previous tile, a full trip to memory thirteen times per workgroup)
#define SETUP_M_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    for (int ct = 0; ct < nct; ++ct) {
        SETUP_M_LDS_BARRIER();                                   // the previous tile's readers are done with `tile` and `ob`
        SETUP_M_STORE(ct);
        SETUP_M_LDS_BARRIER();
        if (ct + 1 < nct) SETUP_M_FETCH(ct + 1);
        const int klast = (n - 1 < 16 * ct + 15) ? n - 1 : 16 * ct + 15;
        const int nblk = klast / 32 + 1;
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
        const double *bcol = tile + lk * 16 + lr;                 // lane l: R^-1[4 kt + (l >> 4)][16 ct + (l & 15)]
        static_for<kSetupMNKT / 8>([&](auto nb) __attribute__((always_inline)) {
            if (nblk == nb + 1) {
                static_for<nb + 1>([&](auto c8) __attribute__((always_inline)) {
                    double bq[8];
                    static_for<8>([&](auto u) __attribute__((always_inline)) {
                        // (rows beyond klast's k step were not stored: they belong to the zero part of the image -- the A operand's
                        //  own zero padding does not cover them, so they are read as zeros explicitly)
                        const int kr = 32 * c8 + 4 * u;
                        bq[u] = (kr <= klast) ? bcol[(size_t)kr * 16] : 0.0;
                    });
                    static_for<8>([&](auto u) __attribute__((always_inline)) {
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[8 * c8 + u], bq[u], acc, 0, 0, 0);
                    });
                });
            }
        });
        // D: row (l >> 4) + 4 r of the wave's tile, column l & 15
        static_for<4>([&](auto r) __attribute__((always_inline)) { ob[(16 * wv + lk + 4 * r) * 16 + lr] = acc[(int)r]; });
        SETUP_M_LDS_BARRIER();
        // blocked image [row/64][col/2][row%64][col%2]: thread <-> (row, column pair): 64 rows x 16 bytes contiguous per pair
        static_for<2>([&](auto h) __attribute__((always_inline)) {
            const int t = pq + 4 * h;                            // pair within the tile
            const int col = 16 * ct + 2 * t;
            if (col < n && prow < rows_wg) {
                double2 w;
                w.x = ob[prow * 16 + 2 * t];
                w.y = (col + 1 < n) ? ob[prow * 16 + 2 * t + 1] : 0.0;
                Mq2[((size_t)(gi_p >> 6) * b.npair + (col >> 1)) * 64 + (gi_p & 63)] = w;
                s2 = __builtin_fma(w.x, w.x, s2); s2 = __builtin_fma(w.y, w.y, s2);
                sv = __builtin_fma(w.x, vq[col], sv);
                if (col + 1 < n) sv = __builtin_fma(w.y, vq[col + 1], sv);
            }
        });
    }
    // ---- row norms and row . v: the four quarters of a row meet in LDS
    rs[(pq * 64 + prow) * 2] = s2; rs[(pq * 64 + prow) * 2 + 1] = sv;
    __syncthreads();
    const double sn = (rs[prow * 2] + rs[(64 + prow) * 2]) + (rs[(128 + prow) * 2] + rs[(192 + prow) * 2]);
    const double rv = (rs[prow * 2 + 1] + rs[(64 + prow) * 2 + 1]) + (rs[(128 + prow) * 2 + 1] + rs[(192 + prow) * 2 + 1]);
    int bits = 0;                                                // 1: a row infeasible at x_unc, 2: an infeasible zero row
    double scal = 1.0;
    const bool own = prow < rows_wg;
    const bool zero_row = sn < st.zero_tol;
    if (own) {
        if (!zero_row) scal = 1 / sqrt(sn);
        if (pq == 0) {
            const int gi = gi_p;
            int *sens = b.sense + (size_t)q * m;
            if (zero_row) {
                const int sg = sens[gi];
                if ((bu[gi] < -st.zero_tol || bl[gi] > st.zero_tol) && !(sg & DAQP_IMMUTABLE) && !(sg & DAQP_SOFT)) bits |= 2;
                sens[gi] = DAQP_IMMUTABLE;
            }
            b.scaling[(size_t)q * m + gi] = scal;
            double *du = b.dupper + (size_t)q * m, *dl = b.dlower + (size_t)q * m;
            if (unc) {      // A x_unc = -(A R^-1) v: the unnormalised row against v
                const double sunc = -rv;
                const double u0 = bu[gi] - sunc, l0 = bl[gi] - sunc;
                if (u0 < -st.primal_tol || l0 > st.primal_tol) bits |= 1;
                du[gi] = u0 * scal; dl[gi] = l0 * scal;
            } else {
                const double dsum = zero_row ? rv : rv * scal;
                du[gi] = bu[gi] * scal + dsum; dl[gi] = bl[gi] * scal + dsum;
            }
        }
    }
    // ---- the images: the block read back (this workgroup wrote it: L2), scaled, stored as fp64 and fp32
    // (written and read back by ONE workgroup -- one CU, one L1, one L2: a workgroup-scope fence.  A device-scope fence here, per
    //  workgroup, writes the XCD's whole dirty L2 back each time: measured 16.8 ms instead of 3 for 4 096 problems of config C4)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (own) {
        float2 *row32 = b.M32 ? reinterpret_cast<float2 *>(b.M32 + (size_t)q * b.nblk * b.nquad * 256) + (((size_t)(gi_p >> 6) * b.nquad) * 64 + (gi_p & 63)) * 2 : nullptr;
        double2 *rowp = Mq2 + ((size_t)(gi_p >> 6) * b.npair) * 64 + (gi_p & 63);
        for (int t0 = pq; t0 < b.npair; t0 += 4 * 8) {
            double2 v8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v8[u] = rowp[(size_t)((t0 + 4 * u < b.npair) ? t0 + 4 * u : pq) * 64];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + 4 * u;
                if (t < b.npair) {
                    double2 w = v8[u];
                    if (!zero_row) { w.x *= scal; w.y *= scal; rowp[(size_t)t * 64] = w; }
                    if (row32) row32[((size_t)(t >> 1) * 64) * 2 + (t & 1)] = make_float2((float)w.x, (float)w.y);
                }
            }
        }
    }
    // ---- the problem's record, by the workgroup that finishes last.  ONE relaxed atomic per workgroup carries its ticket and its
    // verdict bits (as counts: bits 0-15 workgroups done, 16-23 those with an x_unc-infeasible row, 24-31 those with an infeasible zero
    // row) -- no fence: nothing but this word travels between the workgroups of a problem
    __shared__ int wg_bits;
    if (tid == 0) wg_bits = 0;
    __syncthreads();
    if (bits) atomicOr(&wg_bits, bits);
    __syncthreads();
    if (tid == 0) {
        int *tick = b.m_tick + (size_t)q;
        const int mine = 1 + ((wg_bits & 1) ? (1 << 16) : 0) + ((wg_bits & 2) ? (1 << 24) : 0);
        const int before = atomicAdd(tick, mine);
        if ((before & 0xffff) == nrb - 1) {
            const int all = before + mine;
            const bool infeasible_xunc = ((all >> 16) & 0xff) != 0, bad_zero_row = ((all >> 24) & 0xff) != 0;
            if (bad_zero_row) { qs->setup_flag = DAQP_EXIT_INFEASIBLE; qs->exitflag = DAQP_EXIT_INFEASIBLE; qs->need_activate = 0; qs->n_prox = 0; }
            else if (unc && !infeasible_xunc && !(owed & 4)) { qs->sing_ind = DAQP_UNCONSTRAINED_OPTIMAL; qs->need_activate = 0; }
            qs->pad_ = 0;
            *tick = 0;                                            // ready for the next setup
        }
    }
}
#undef SETUP_M_FETCH
#undef SETUP_M_LDS_BARRIER
#undef SETUP_M_STORE
#undef SETUP_M_F1
#undef SETUP_M_S1
#undef SETUP_M_NP
#endif // DAQP_AMD_SETUP_M_IMPL

} // namespace daqp_amd
