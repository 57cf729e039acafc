"""Template post-processing between the two rounds of EM steps (reference src/baumwelch.jl:340-349,
:418-605; src/types.jl:143-166).  Host-side model selection on K x N template matrices (microseconds of
work next to an EM step), restated so that `train_model(X, N, K, ...)` runs end to end here as it does in
the reference:  condense_templates -> remove_sparse -> remove_small -> mu = mu[:, idx[idx2]].

The restatement follows the reference statement by statement, INCLUDING its quirks, because the template
count after this stage decides what the remaining EM steps see:
  * condense_templates' merge loop indexes the surviving templates with `setdiff(1:N, [i1,i2])` AFTER
    decrementing N (baumwelch.jl:458-468): the last template is dropped whenever it is not one of the merged
    pair, and the last column of the new matrix then stays zero with lp = 0;
  * the merged template is assembled with `.=+` (assignment of +0.5*mu, not `+=`; :465), so on the rows both
    alignments cover, the second template's half overwrites the first's;
  * prune_templates picks `lp[findall(in(tidx), idx)]` (types.jl:164): positions inside `idx`, not template
    numbers.
Indices in this module are 0-based; ranges are half-open Python ranges.
"""
import numpy as np


def find_best_overlap(mu, i1, i2):
    """find_best_overlap(mu, i1, i2) -> ((range1, range2), xm)   baumwelch.jl:521-541: the alignment of
    template i1 against i2 (all 2K-1 shifts, in the reference's order) with the largest inner product;
    a later shift must beat the best strictly."""
    K = mu.shape[0]
    xi = (range(0, K), range(0, K))
    xm = -np.inf
    shifts = [(range(0, s), range(K - s, K)) for s in range(1, K + 1)]
    shifts += [(range(s, K), range(0, K - s)) for s in range(1, K)]
    a, b = mu[:, i1], mu[:, i2]
    for r1, r2 in shifts:
        x = 0.0
        for k1, k2 in zip(r1, r2):
            x += a[k1] * b[k2]
        if x > xm:
            xm = x
            xi = (r1, r2)
    return xi, xm


def _chi2_sf(df, x):
    from scipy.stats import chi2          # 1 - cdf(Chisq(df), x)
    return 1.0 - chi2.cdf(x, df)


def condense_candidates(mu, sigma2, alpha=0.05):
    """condense_templates(mu, sigma2, alpha) -> (candidates, test_stat, overlap_idx)   baumwelch.jl:478-516:
    the pair of templates whose best-aligned squared distance is compatible with noise, the one with the
    LARGEST statistic among the compatible pairs first (argmax, as the reference has it)."""
    K, N = mu.shape
    cands, stats, ovl = [], [], []
    for i1 in range(N - 1):
        for i2 in range(i1 + 1, N):
            xi, _ = find_best_overlap(mu, i1, i2)
            x = 0.0
            for k1, k2 in zip(*xi):
                d = mu[k1, i1] - mu[k2, i2]
                x += d * d
            x /= sigma2
            n = len(xi[0])
            pval = 0.0 if n < 5 else _chi2_sf(n - 1, x)      # fewer than 5 matching points: no match
            if pval > alpha:
                cands.append((i1, i2)); stats.append(x); ovl.append(xi)
    if cands:
        m = int(np.argmax(stats))
        return cands[m], stats[m], ovl[m]
    return cands, stats, ovl


def condense_templates(state_matrix, mu, sigma, alpha=0.05, verbose=0):
    """condense_templates(state_matrix, mu, sigma, alpha) -> (state_matrix, mu)   baumwelch.jl:441-476"""
    from .api import StateMatrix
    from .sortdata import get_lp
    sigma2 = sigma ** 2
    lp, _ = get_lp(state_matrix)
    mu = np.array(mu, dtype=np.float64, order="F")
    K, N = mu.shape
    cand, stat, ovl = condense_candidates(mu, sigma2, alpha)
    while cand:
        i1, i2 = cand
        xi1, xi2 = ovl
        if verbose > 1:
            print("Merging templates %d and %d with Chi2 statistic %g" % (i1 + 1, i2 + 1, stat))
        N -= 1
        mu_new = np.zeros((K, N), order="F")
        lp_new = np.zeros(N)
        mu_new[list(xi1), 0] = 0.5 * mu[list(xi1), i1]
        mu_new[list(xi2), 0] = 0.5 * mu[list(xi2), i2]          # `.=+`: assignment (baumwelch.jl:465)
        lp_new[0] = np.log(0.5 * np.exp(lp[i1]) + 0.5 * np.exp(lp[i2]))
        idx = [j for j in range(N) if j not in (i1, i2)]         # setdiff(1:N, [i1,i2]) with the NEW N
        for ii, jj in enumerate(idx):
            mu_new[:, 1 + ii] = mu[:, jj]
            lp_new[1 + ii] = lp[jj]
        lp, mu = lp_new, mu_new
        cand, stat, ovl = condense_candidates(mu, sigma2, alpha)
    if N < state_matrix.N:
        return StateMatrix.create(N, K, lp, state_matrix.resolve_overlaps), mu
    return state_matrix, mu


def prune_templates(state_matrix, idx, resolve_overlaps=True):
    """prune_templates(state_matrix, idx, resolve_overlaps)   types.jl:161-166 (idx: 0-based template numbers)"""
    from .api import StateMatrix
    from .sortdata import get_lp
    if len(idx) == 0:
        # StateMatrix(0, K, Float64[]): no template left.  The reference builds a 0 x 1 state table (isempty);
        # here that is the null model (types.jl:12), which train_model returns as it is
        return StateMatrix.null()
    lp, tidx = get_lp(state_matrix)                              # tidx: 1-based neuron numbers
    tset = set(int(t) - 1 for t in tidx)
    pos = [p for p, v in enumerate(idx) if v in tset]            # findall(in(tidx), idx): POSITIONS in idx
    return StateMatrix.create(len(idx), state_matrix.K, lp[pos], resolve_overlaps)


def remove_sparse(state_matrix, lp0=-70.0):
    """remove_sparse(state_matrix, lp0) -> (state_matrix, idx)   baumwelch.jl:573-592: keep the templates
    whose silent -> first-state transition is more likely than exp(lp0)"""
    from .api import StateMatrix
    keep = []
    for src, dst, val in state_matrix.transitions:
        if src == 1 and dst != 1 and val > lp0:
            col = state_matrix.states[:, dst - 1]
            for j in range(state_matrix.states.shape[0]):
                if col[j] == 2:
                    keep.append(j)
                    break
    if not keep:
        return StateMatrix.null(), []
    return prune_templates(state_matrix, keep, state_matrix.resolve_overlaps), keep


def remove_small(state_matrix, mu, sigma, alpha=0.05):
    """remove_small(state_matrix, mu, sigma, PValue(alpha)) -> (state_matrix, idx)   baumwelch.jl:418-427:
    keep the templates whose energy is significantly different from noise (Chi2 with K-1 dof)"""
    K = mu.shape[0]
    Z = (mu ** 2).sum(0) / (sigma * sigma)
    pvals = np.array([_chi2_sf(K - 1, z) for z in Z])
    idx = [int(i) for i in np.nonzero(pvals < alpha)[0]]
    return prune_templates(state_matrix, idx, state_matrix.resolve_overlaps), idx


def match_templates(temps1, temps2):
    """match_templates(temps1, temps2) -> (mm, cc)   baumwelch.jl:546-568: for every template of temps1 the
    (1-based) template of temps2 with the smallest squared difference at the best alignment"""
    K1, N1 = temps1.shape
    K2, N2 = temps2.shape
    if K1 != K2:
        raise ValueError("The two template sets must have the same number of states")
    mm = np.zeros(N1, dtype=np.int64)
    cc = np.zeros(N1)
    for i1 in range(N1):
        m, mi = np.inf, 0
        for i2 in range(N2):
            pair = np.stack([temps1[:, i1], temps2[:, i2]], 1)
            xi, _ = find_best_overlap(pair, 0, 1)
            xm = float(((temps1[list(xi[0]), i1] - temps2[list(xi[1]), i2]) ** 2).sum())
            if xm < m:
                m, mi = xm, i2 + 1
        mm[i1], cc[i1] = mi, m
    return mm, cc


def reference_postprocess(state_matrix, mu, sigma, verbose=0):
    """the stage between the two rounds of EM steps, baumwelch.jl:340-349"""
    state_matrix, mu = condense_templates(state_matrix, mu, sigma, 0.05, verbose=verbose)
    if verbose > 0:
        print("%d templates remain after merging" % mu.shape[1])
    state_matrix, idx = remove_sparse(state_matrix)
    if verbose > 0:
        print("%d templates remain after removing sparse" % len(idx))
    if not idx:
        return state_matrix, mu[:, :0]
    state_matrix, idx2 = remove_small(state_matrix, mu[:, idx], sigma, 0.05)
    if verbose > 0:
        print("%d templates remain after removing small templates" % len(idx2))
    keep = [idx[i] for i in idx2]
    return state_matrix, np.asfortranarray(mu[:, keep])
