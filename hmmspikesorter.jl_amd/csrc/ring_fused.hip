// Ring engine, part 4: one decode + one E-step of the SAME bound signal and model in a single
// call.  The three serial sweeps (Viterbi, forward, backward) are independent of each other, so
// they share one launch: block 3i runs the Viterbi chains of column group i, block 3i+1 the
// forward chains, block 3i+2 the backward chains.  Three times as many wavefronts are resident,
// which is what a lane-per-chain design needs when a single recording supplies only ~600 of them.
// Results are identical to hmmsort_plan_viterbi followed by hmmsort_plan_estep.
#include <type_traits>

#include "ring_chain_bodies.h"
#include "ring_common.h"

namespace hmmsort {

template <int N>
__global__ __launch_bounds__(64) void k_vfb_chain(RingGeom g, JParams<N> jp, EParams<N> ep,
                                                  const double *__restrict__ yT,
                                                  const double *__restrict__ Rf,
                                                  double *__restrict__ Pv, uint32_t *__restrict__ psi,
                                                  double *__restrict__ D0pre, double *__restrict__ D0end,
                                                  double *__restrict__ P, double *__restrict__ A0,
                                                  double *__restrict__ Q, double *__restrict__ B0,
                                                  double *__restrict__ B0h)
{
    // XCD-aware: blocks b and b+8 share an XCD (and its L2).  The three roles of column group i
    // get slots 24*(i/8) + 8*role + i%8, i.e. the same XCD, so the y / ring-score rows the Viterbi
    // and forward chains read at the same step come from HBM once.
    const int grp = blockIdx.x / 24, rem = blockIdx.x % 24;
    const int role = rem >> 3, bx = grp * 8 + (rem & 7);
    if (bx * 64 >= g.ncol) return;
    if (role == 0) vit_chain_body<N>(bx, g, jp, yT, Rf, Pv, psi, D0pre, D0end);
    else if (role == 1) fwd_chain_body<N>(bx, g, ep, yT, Rf, P, A0);
    else bwd_chain_body<N>(bx, g, ep, yT, Rf, Q, B0, B0h);
}

int ring_decode_estep_launch(RingDev *r, const double *d_y, int16_t *d_x, double *d_ll, double *d_stats,
                             hipStream_t st)
{
    const RingGeom &g = r->g;
    int rc;
    HS_HIP(hipMemsetAsync(r->diag, 0, 8 * sizeof(int64_t), st));
    if ((rc = ring_prepare(r, d_y, st))) return rc;
    const int64_t planeP = (int64_t)(g.H + g.B) * g.ncol;
    if ((rc = ring_launch_virtual(r, d_y, r->Pv, planeP, st, r->P))) return rc;  // both delay lines
    rc = dispatch_N(g.N, [&](auto n) {
        constexpr int N = decltype(n)::value;
        JParams<N> jp = make_jparams<N>(r);
        EParams<N> ep = make_eparams<N>(r);
        { PROF(r, "k_vfb_chain", st);
          hipLaunchKernelGGL((k_vfb_chain<N>), dim3(24 * ((g.ncol / 64 + 7) / 8)), dim3(64), 0, st, g, jp, ep, r->yT, r->Rf,
                             r->Pv, r->psi, r->D0pre, r->D0end, r->P, r->A0, r->Q, r->B0, r->B0h); }
        HS_HIP(hipGetLastError());
        return HMMSORT_OK;
    });
    if (rc) return rc;
    // After the sweeps the decode's post-processing (final state, backtrace, certificate, x, ll:
    // a handful of small latency-bound kernels) and the E-step's (normaliser, posteriors,
    // statistics) are independent: the former runs on the plan's internal stream beside the
    // latter and is joined back into the caller's stream before returning.
    HS_HIP(hipEventRecord(r->ev_fork, st));
    HS_HIP(hipStreamWaitEvent(r->side, r->ev_fork, 0));
    if ((rc = ring_viterbi_post(r, d_y, d_x, d_ll, r->side))) return rc;
    HS_HIP(hipEventRecord(r->ev_join, r->side));
    if ((rc = ring_estep_post(r, d_y, d_stats, st))) return rc;
    HS_HIP(hipStreamWaitEvent(st, r->ev_join, 0));
    return HMMSORT_OK;
}

}  // namespace hmmsort
