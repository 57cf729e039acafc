"""Executable specification (numpy, CPU) of the WAVE engine's arithmetic: hmmspikesorter.jl_amd/csrc/wave_*.hip
implement exactly these recurrences on the GPU (one wavefront per chain, lanes = W consecutive samples,
max-plus / linear scans across lanes).  TEST INFRASTRUCTURE: tests/test_wave_model.py checks this
model against the oracle on the CPU, so the representation (scaled linear forward/backward, ring
delay lines, per-chain normaliser) is validated without a GPU; the GPU tests then check the kernels.

Ring model (reference types.jl:94-113, no overlaps): N rings of L = K-1 states through one silent
state; junction constants c00, c0[a], cend[a], cx[a,b] ((a,L)->(b,1)), ring-interior constants
folded into the ring scores.  Frame: the per-sample emission constant A = -log2pi - log(sigma) is
dropped everywhere (reference utils.jl:1-4).
"""
import numpy as np

NEG = -np.inf
SC_FLOOR = -700.0     # entry log-probabilities below this live in the exponent, not in a coefficient


class Ring:
    def __init__(self, sm, mu, sigma):
        tr = sm.transitions if hasattr(sm, "transitions") else None
        if tr is not None:
            lp = np.asarray(tr["lp"], dtype=np.float64)
        else:
            lp = np.asarray(sm.val, dtype=np.float64)
        N, K = sm.N, sm.K
        L = K - 1
        self.N, self.L = N, L
        r = 0
        self.c00 = lp[r]; r += 1
        self.c0 = lp[r:r + N].copy(); r += N
        self.cint = np.zeros((N, L + 1))      # Cint[a, kk] = sum_{k=1}^{kk-1} lp((a,k)->(a,k+1))
        self.cend = np.zeros(N)
        self.cx = np.full((N, N), NEG)
        for a in range(N):
            acc = 0.0
            for k in range(1, L):
                acc += lp[r]; r += 1
                self.cint[a, k + 1] = acc
            self.cend[a] = lp[r]; r += 1
            for b in range(N):
                if b != a:
                    self.cx[a, b] = lp[r]; r += 1
        assert r == len(lp)
        mu = np.asarray(mu, dtype=np.float64)
        self.mean = mu[1:, :].T.copy()        # [a, k-1]
        self.mean0 = 0.0
        for l in range(N):
            self.mean0 += mu[0, l]
        self.sigma = float(sigma)
        self.den = 2.0 * (sigma * sigma)
        self.A = -0.9189385332046727 - np.log(sigma)
        # scales of the entry transitions into ring a (from silent and from the other rings' ends)
        ent = np.concatenate([self.c0[None, :], np.where(np.eye(N, dtype=bool), NEG, self.cx)], 0)
        sc = ent.max(0)
        sc = np.where(np.isfinite(sc), sc, 0.0)
        self.sc = np.maximum(sc, SC_FLOOR)                      # sc'_a
        with np.errstate(under="ignore"):
            self.CP0 = np.exp(self.c0 - self.sc)                # silent -> (a,1), relative to sc'_a
            self.CPX = np.exp(np.where(np.eye(N, dtype=bool), NEG, self.cx) - self.sc[None, :])  # [b, a]
        self.sc0 = max(self.c00, self.cend.max())
        self.P00 = np.exp(self.c00 - self.sc0)
        self.PEND = np.exp(self.cend - self.sc0)

    def q0(self, y):
        d = y - self.mean0
        return -(d * d) / self.den


def ring_scores(y, m):
    """Rf[a, t'] for t' in [0,T) (rings truncated at the end of the data) and the virtual onsets
    V[a, j], j = 1..L-1 (rings already running at the first sample); V[a, L] = -inf marker."""
    T = len(y)
    N, L = m.N, m.L
    Rf = np.zeros((N, T))
    ypad = np.concatenate([y, np.zeros(L)])
    for a in range(N):
        for k in range(1, L + 1):
            d = ypad[k - 1:k - 1 + T] - m.mean[a, k - 1]
            ok = (np.arange(T) + k - 1) < T
            Rf[a] += np.where(ok, -(d * d) / m.den, 0.0)
        kmax = np.minimum(L, T - np.arange(T))
        Rf[a] += m.cint[a, kmax]
    V = np.full((N, L + 1), NEG)
    for a in range(N):
        for j in range(1, L):
            acc = 0.0
            for k in range(1 + j, L + 1):
                d = y[k - 1 - j] - m.mean[a, k - 1]
                acc += d * d
            V[a, j] = (m.cint[a, L] - m.cint[a, 1 + j]) - acc / m.den
    return Rf, V


def _shift_up(x, d, fill):
    out = np.empty_like(x)
    out[:d] = fill
    out[d:] = x[:-d]
    return out


def scan_maxplus(a, b):
    """inclusive scan of f_j(x) = max(x + a_j, b_j): returns (A, B) with f_j o ... o f_0 (x) = max(x + A_j, B_j)"""
    A, B = a.copy(), b.copy()
    d = 1
    while d < len(a):
        Al, Bl = _shift_up(A, d, 0.0), _shift_up(B, d, NEG)
        B = np.maximum(Bl + A, B)
        A = Al + A
        d *= 2
    return A, B


def scan_linear(al, be):
    """inclusive scan of f_j(x) = al_j * x + be_j"""
    A, B = al.copy(), be.copy()
    d = 1
    while d < len(al):
        Al, Bl = _shift_up(A, d, 1.0), _shift_up(B, d, 0.0)
        B = A * Bl + B
        A = Al * A
        d *= 2
    return A, B


def chain_bounds(T, B, H, c):
    tc = c * B
    nc = min(B, T - tc)
    ts = 0 if c == 0 else tc - H
    te = min(tc + nc + H, T)
    return tc, nc, ts, te


def super_step_width(L):
    return min(L, 64)


# ------------------------------------------------------------------------------------------------
# Viterbi (viterbi.jl:44-98): one chain, super-steps of W samples, max-plus scan for delta(silent)
# ------------------------------------------------------------------------------------------------
def vit_chain(y, Rf, V, m, T, B, H, c, thr, start_state=None):
    """returns psi[t - tc, N+1] (back-pointers of the junction states, 0 = silent, b+1 = ring b's
    end), flag[t - tc, N+1] (near-tie within thr), the state before tc and the state at the end."""
    N, L = m.N, m.L
    W = super_step_width(L)
    tc, nc, ts, te = chain_bounds(T, B, H, c)
    tend = tc + nc
    P = {}                                        # delay line: onset time -> N values
    if start_state is not None:
        D0, Pin = start_state
        ts = tc
        for j in range(1, L + 1):
            P[tc - j] = Pin[:, j - 1].copy()
    elif c == 0:
        for j in range(1, L + 1):
            P[-j] = V[:, j].copy()
        D0 = -m.A                                  # T1[1,1] = 0 (viterbi.jl:63) in the A-free frame
        P[0] = Rf[:, 0].copy()
    else:
        D0 = 0.0
        P[ts] = np.full(N, NEG)
    psi = np.zeros((nc, N + 1), dtype=np.int32)
    flag = np.zeros((nc, N + 1), dtype=bool)
    pre = None
    t0 = ts + 1 if start_state is None else tc
    q0 = m.q0(y)
    while t0 < tend:
        w = min(W, tend - t0)
        tt = np.arange(t0, t0 + w)
        X = np.stack([P.get(t - L, np.full(N, NEG)) for t in tt], 1)          # [N, w]
        cand_e = X + m.cend[:, None]
        e = cand_e.max(0)
        earg = cand_e.argmax(0)                                              # first maximum
        A_, B_ = scan_maxplus(m.c00 + q0[tt], e + q0[tt])
        D = np.maximum(D0 + A_, B_)
        Dprev = _shift_up(D, 1, D0)
        best0 = Dprev + m.c00
        ps = np.zeros((w, N + 1), dtype=np.int32)
        fl = np.zeros((w, N + 1), dtype=bool)
        ps[:, 0] = np.where(e > best0, earg + 1, 0)
        srt = np.sort(np.concatenate([best0[None], cand_e], 0), 0)
        with np.errstate(invalid="ignore"):
            fl[:, 0] = (srt[-1] - srt[-2]) < thr
        Pn = np.zeros((N, w))
        for a in range(N):
            cands = [Dprev + m.c0[a]] + [X[b] + m.cx[b, a] if b != a else np.full(w, NEG) for b in range(N)]
            cands = np.stack(cands, 0)
            u = cands.max(0)
            ps[:, a + 1] = cands.argmax(0)
            s2 = np.sort(cands, 0)
            with np.errstate(invalid="ignore"):
                fl[:, a + 1] = (s2[-1] - s2[-2]) < thr
            Pn[a] = u + Rf[a, tt]
        for i, t in enumerate(tt):
            P[t] = Pn[:, i]
            if t == tc - 1:
                pre = (D[i], np.stack([P.get(tc - j, np.full(N, NEG)) for j in range(1, L + 1)], 1))
            if t >= tc:
                psi[t - tc] = ps[i]
                flag[t - tc] = fl[i]
        D0 = D[-1]
        t0 += w
    if c == 0 and start_state is None:
        pre = None
    end = (D0, np.stack([P.get(tend - j, np.full(N, NEG)) for j in range(1, L + 1)], 1))
    return psi, flag, pre, end


def vit_decode(y, m, B, H, thr=0.0):
    """whole decode with the wave engine's chain rule; returns x (1-based state ids), certificate
    spreads per boundary and the number of flagged on-path decisions"""
    T = len(y)
    N, L = m.N, m.L
    Rf, V = ring_scores(y, m)
    nch = (T + B - 1) // B
    psi = np.zeros((T, N + 1), dtype=np.int32)
    flag = np.zeros((T, N + 1), dtype=bool)
    spreads = []
    prev_end = None
    for c in range(nch):
        ps, fl, pre, end = vit_chain(y, Rf, V, m, T, B, H, c, thr)
        if c > 0:
            d = np.concatenate([[pre[0] - prev_end[0]], (pre[1] - prev_end[1]).ravel()])
            same = np.concatenate([[False], (pre[1] == prev_end[1]).ravel()])
            d = np.where(same, 0.0, d)
            spreads.append(np.nanmax(d) - np.nanmin(d) if np.all(np.isfinite(d)) else np.inf)
            if not spreads[-1] <= 1e-6:            # exact hand-off, chain redone
                ps, fl, _, end = vit_chain(y, Rf, V, m, T, B, H, c, thr, start_state=prev_end)
        tc = c * B
        psi[tc:tc + len(ps)] = ps
        flag[tc:tc + len(ps)] = fl
        prev_end = end
    # final state: first maximum over all states at T-1 (viterbi.jl:90)
    D0, Pend = prev_end
    vals = [D0] + [Pend[a, k - 1] for a in range(N) for k in range(1, L + 1)]  # P_a(T-k)
    fs = int(np.argmax(vals))
    x = np.zeros(T, dtype=np.int16)
    nflag = 0
    a, k = (-1, 0) if fs == 0 else ((fs - 1) // L, (fs - 1) % L + 1)
    for t in range(T - 1, -1, -1):
        x[t] = 1 if a < 0 else 2 + a * L + (k - 1)
        if t == 0:
            break
        if a >= 0 and k > 1:
            k -= 1
            continue
        p = psi[t, a + 1]
        nflag += int(flag[t, a + 1])
        if p == 0:
            a, k = -1, 0
        else:
            a, k = p - 1, L
    return x, spreads, nflag


# ------------------------------------------------------------------------------------------------
# forward (baumwelch.jl:25-51) in the scaled representation
#   exp(la0(t)) = x_t * exp(M_t);  onset mass of ring a at t:  lp_a(t) = fref_t + sc_a + log fv_a(t) + R_a(t)
# ------------------------------------------------------------------------------------------------
def fwd_chain(y, Rf, V, m, T, B, H, c):
    N, L = m.N, m.L
    W = super_step_width(L)
    tc, nc, ts, te = chain_bounds(T, B, H, c)
    tend = tc + nc
    q0 = m.q0(y)
    DLv, DLs = {}, {}
    la0 = {}
    fv, fref = {}, {}
    if c == 0:
        for j in range(1, L):
            DLv[-j] = np.ones(N); DLs[-j] = V[:, j].copy()
        la0[0] = q0[0]
        M, x = q0[0], 1.0
        fv[0] = np.exp(-m.sc); fref[0] = 0.0
        DLv[0] = fv[0]; DLs[0] = fref[0] + m.sc + Rf[:, 0]
    else:
        M, x = 0.0, 1.0
        la0[ts] = 0.0
        fv[ts] = np.zeros(N); fref[ts] = 0.0
        DLv[ts] = np.zeros(N); DLs[ts] = np.zeros(N)
    t0 = ts + 1
    zv, zs = np.zeros(N), np.zeros(N)
    while t0 < tend:
        w = min(W, tend - t0)
        tt = np.arange(t0, t0 + w)
        v = np.stack([DLv.get(t - L, zv) for t in tt], 1)           # [N, w]
        s = np.stack([DLs.get(t - L, zs) for t in tt], 1)
        # scale of the exits: s + exponent of v (frexp, exact power of two)
        with np.errstate(divide="ignore"):
            ex = np.where(v > 0, np.frexp(v)[1] * np.log(2.0), NEG)
        e = (s + ex + m.sc0).max(0)                                  # exits into silent, scale level
        A_, B_ = scan_maxplus(m.sc0 + q0[tt], e + q0[tt])
        Mt = np.maximum(M + A_, B_)
        Mprev = _shift_up(Mt, 1, M)
        ref = Mt - q0[tt]
        E0 = np.exp((Mprev + m.sc0) - ref)
        with np.errstate(under="ignore", over="ignore", invalid="ignore"):
            arg = np.where(v > 0, np.minimum((s + m.sc0) - ref[None, :], 700.0), NEG)   # 0 * exp(big) = NaN
            Ea = v * np.exp(arg)                                     # v_a * exp(s_a + sc0 - ref)
        al = E0 * m.P00
        be = (Ea * m.PEND[:, None]).sum(0)
        A2, B2 = scan_linear(al, be)
        xt = A2 * x + B2
        xprev = _shift_up(xt, 1, x)
        # onset masses: u_a = xprev*E0*exp(-sc0)... all entry coefficients are relative to sc_a
        base = xprev * E0 * np.exp(-m.sc0)                           # exp(la0(t-1) - ref)
        Eb = Ea * np.exp(-m.sc0)                                     # exp(X_b - ref)
        for i, t in enumerate(tt):
            u = base[i] * m.CP0 + (Eb[:, i][:, None] * m.CPX).sum(0)
            fv[t] = u; fref[t] = ref[i]
            DLv[t] = u; DLs[t] = ref[i] + m.sc + Rf[:, t]
            la0[t] = Mt[i] + np.log(xt[i])
        # carry, renormalised
        mant, ee = np.frexp(xt[-1])
        x = mant; M = Mt[-1] + ee * np.log(2.0)
        t0 += w
    return la0, fv, fref


# ------------------------------------------------------------------------------------------------
# backward (baumwelch.jl:73-98) fused with the posteriors of update() (:205-309)
#   exp(lb0(t)) = xb_t * exp(Mb_t);  beta of ring a's last state at t:  Yn_a(t) = sb_t + log vb_a(t)
# Step t (from t+1):  terms  silent: xb_{t+1} exp(Mb_{t+1} + q0(t+1) + sc0)
#                            ring a starting at t+1: vb_a exp(sw_a), sw_a = sb + R_a(t+1) + sc_a
# ------------------------------------------------------------------------------------------------
def bwd_chain(y, Rf, m, T, B, H, c, la0, fv, fref, la0_pre, last=True):
    """returns the chain's posteriors: g0[t] (silent), rho[a, t'] (onsets), xi'[a, t'] (silent ->
    ring a at t', WITHOUT the coefficient exp(c0_a - sc_a)), the normaliser z, lb0 and Yn (logs)."""
    N, L = m.N, m.L
    W = super_step_width(L)
    tc, nc, ts, te = chain_bounds(T, B, H, c)
    q0 = m.q0(y)
    ones, zs = np.ones(N), np.zeros(N)
    DLv, DLs = {te - 1: ones}, {te - 1: zs}          # beta = 0 for every state at te-1
    Mb, xb = 0.0, 1.0
    lb0 = {te - 1: 0.0}
    rec = {}                                          # per step t: (refb, wa[N], xt)
    t0 = te - 2
    lo = tc - 1
    while t0 >= lo:
        w = min(W, t0 - lo + 1)
        tt = np.arange(t0, t0 - w, -1)                # descending times, lane i <-> tt[i]
        vb = np.stack([DLv[t + L] if t + L <= te - 1 else ones for t in tt], 1)
        sb = np.stack([DLs[t + L] if t + L <= te - 1 else zs for t in tt], 1)
        R1 = Rf[:, tt + 1]
        q1 = q0[tt + 1]
        with np.errstate(divide="ignore"):
            ex = np.where(vb > 0, np.frexp(vb)[1] * np.log(2.0), NEG)
        sw = sb + R1 + m.sc[:, None]
        e = (sw + ex).max(0)
        A_, B_ = scan_maxplus(q1 + m.sc0, e)
        Mt = np.maximum(Mb + A_, B_)
        Mnext = _shift_up(Mt, 1, Mb)                  # Mb_{t+1}
        E0 = np.exp((Mnext + q1 + m.sc0) - Mt)
        with np.errstate(under="ignore", over="ignore", invalid="ignore"):
            wa = vb * np.exp(np.where(vb > 0, np.minimum(sw - Mt[None, :], 700.0), NEG))
        A2, B2 = scan_linear(E0 * m.P00, (wa * m.CP0[:, None]).sum(0))
        xt = A2 * xb + B2
        xnext = _shift_up(xt, 1, xb)
        for i, t in enumerate(tt):
            # ring a's end at t -> silent (cend_a) or ring b's first state at t+1 (cx[a,b], rel. sc_b)
            yn = xnext[i] * E0[i] * m.PEND + (m.CPX * wa[:, i][None, :]).sum(1)
            DLv[t] = yn; DLs[t] = np.full(N, Mt[i])
            lb0[t] = Mt[i] + np.log(xt[i])
            rec[t] = (Mt[i], wa[:, i].copy(), xt[i])
        mant, ee = np.frexp(xt[-1])
        xb, Mb = mant, Mt[-1] + ee * np.log(2.0)
        t0 -= w
    # normaliser at tstar = last owned sample: log sum_j alpha(j) beta(j)
    tstar = tc + nc - 1
    terms = [la0[tstar] + lb0[tstar]]
    for tp in range(tstar - L + 1, tstar + 1):
        tau = tp + L - 1
        vb_, sb_ = (DLv[tau], DLs[tau]) if tau <= te - 1 else (ones, zs)
        for a in range(N):
            if fv[tp][a] > 0:
                terms.append(fref[tp] + m.sc[a] + np.log(fv[tp][a]) + Rf[a, tp] + sb_[a] + np.log(vb_[a]))
    terms = np.array(terms)
    z = terms.max() + np.log(np.exp(terms - terms.max()).sum())
    g0, rho, xi = {}, {}, {}
    g0[te - 1] = np.exp(la0[te - 1] + lb0[te - 1] - z) if te - 1 <= tstar else None
    for t in range(min(te - 2, tstar), tc - 2, -1):
        refb, wa, xt = rec[t]
        la = la0[t] if t >= tc else la0_pre
        if t >= tc:
            g0[t] = xt * np.exp(min(la + refb - z, 700.0))
        if t + 1 <= tstar:
            with np.errstate(under="ignore"):
                rho[t + 1] = fv[t + 1] * wa * np.exp(min(fref[t + 1] + refb - z, 700.0))
                xi[t + 1] = wa * np.exp(min(la + refb - z, 700.0)) if (t >= 0) else np.zeros(N)
    Yn = {t: DLs[t][0] + np.log(DLv[t]) for t in DLv}
    return g0, rho, xi, z, lb0, Yn


def estep(y, m, B, H):
    """sufficient statistics of one Baum-Welch step with the wave engine's chain rule; returns the
    M-step results (mu[K,N], sigma, lp_new[N], pp[S]) and the per-time silent posterior"""
    T = len(y)
    N, L = m.N, m.L
    Rf, V = ring_scores(y, m)
    nch = (T + B - 1) // B
    G0 = np.zeros((N, L)); G1 = np.zeros((N, L)); G2 = np.zeros((N, L))
    Xi = np.zeros(N)
    s_all = s_m = s_y2 = 0.0
    rho_all = np.zeros((N, T))
    g0_all = np.zeros(T)
    pp = None
    for c in range(nch):
        tc, nc, ts, te = chain_bounds(T, B, H, c)
        la0, fv, fref = fwd_chain(y, Rf, V, m, T, B, H, c)
        g0, rho, xi, z, lb0, Yn = bwd_chain(y, Rf, m, T, B, H, c, la0, fv, fref,
                                            la0.get(tc - 1, 0.0))
        for t in range(tc, tc + nc):
            g0_all[t] = g0[t]
            s_all += g0[t]
            if t < T - 1:
                s_m += g0[t]
            s_y2 += g0[t] * y[t] * y[t]
            rho_all[:, t] = rho[t]
            if t >= 1:
                Xi += xi[t]
        if c == 0:
            # virtual onsets and pp = gamma[:,1] (baumwelch.jl:263)
            pp = np.full(1 + N * L, NEG)
            pp[0] = la0[0] + lb0[0] - z
            for a in range(N):
                pp[1 + a * L] = (Rf[a, 0] + Yn[L - 1][a]) - z if L - 1 in Yn else NEG
                for j in range(1, L):
                    lr = V[a, j] + Yn[L - 1 - j][a] - z
                    pp[1 + a * L + j] = lr
                    rv = np.exp(lr)
                    for k in range(j + 1, L + 1):          # phase k at sample k-1-j >= 0
                        yv = y[k - 1 - j]
                        G0[a, k - 1] += rv; G1[a, k - 1] += rv * yv; G2[a, k - 1] += rv * yv * yv
    ypad = np.concatenate([y, np.zeros(L)])
    for a in range(N):
        for k in range(1, L + 1):
            r = rho_all[a, :T - k + 1] if T - k + 1 > 0 else rho_all[a, :0]
            G0[a, k - 1] += r.sum()
            G1[a, k - 1] += (r * ypad[k - 1:k - 1 + len(r)]).sum()
            G2[a, k - 1] += (r * ypad[k - 1:k - 1 + len(r)] ** 2).sum()
    K = L + 1
    mu = np.zeros((K, N), order="F")
    mu[1:, :] = (G1 / G0).T
    x2 = (G2 - 2.0 * (G1 / G0) * G1 + (G1 / G0) ** 2 * G0).sum() + s_y2
    qq = G0.sum() + s_all
    sigma = np.sqrt(x2 / qq)
    lp_new = (m.c0 - m.sc) + np.log(Xi) - np.log(s_m)
    return mu, sigma, lp_new, pp, g0_all, rho_all
