"""Template post-processing (reference src/baumwelch.jl:418-605) pinned by the reference's own exact
answers: test/runtests.jl:44-69 ("overlap and combine", "match templates").  Host logic only."""
import numpy as np


def test_find_best_overlap_known_answers(H):
    # test/runtests.jl:45-48
    mu = np.array([[1.0, 1.0], [2.0, 2.0], [3.0, 3.0]])
    xi, xm = H.find_best_overlap(mu, 0, 1)
    assert (list(xi[0]), list(xi[1])) == ([0, 1, 2], [0, 1, 2]) and xm == 14.0
    # :49-55  t2[5:end] = temp1[1:56]
    t1 = H.create_spike_template(60, 3.0, 0.8, 0.2)
    t2 = np.zeros_like(t1)
    t2[4:] = t1[:56]
    xi, xm = H.find_best_overlap(np.stack([t1, t2], 1), 0, 1)
    assert list(xi[0]) == list(range(0, 56)) and list(xi[1]) == list(range(4, 60))
    assert np.isclose(xm, 100.66411692920131, rtol=1e-12)


def test_condense_candidates_known_answer(H):
    # test/runtests.jl:57-60: condense_templates(cat(temp1, t2), 0.1) -> (1,2), overlap (1:56, 5:60)
    t1 = H.create_spike_template(60, 3.0, 0.8, 0.2)
    t2 = np.zeros_like(t1)
    t2[4:] = t1[:56]
    cand, stat, ovl = H.condense_candidates(np.stack([t1, t2], 1), 0.1)
    assert cand == (0, 1)
    assert list(ovl[0]) == list(range(0, 56)) and list(ovl[1]) == list(range(4, 60))
    assert abs(stat) < 1e-20          # the shifted copy matches exactly on the overlap


def test_match_templates_known_answer(H):
    # test/runtests.jl:63-69
    mu = np.array([[1.0, 1.0], [2.0, 2.0], [3.0, 3.0]])
    mu[:, 0] *= 1.3
    mm, cc = H.match_templates(mu, mu)
    assert list(mm) == [1, 2] and np.allclose(cc, [0.0, 0.0])


def test_condense_prune_pipeline(H):
    # a duplicated template is merged, an all-zero and a never-firing template are pruned
    K = 30
    t1 = H.create_spike_template(K, 3.0, 0.8, 0.2)
    t2 = H.create_spike_template(K, 4.0, 0.3, 0.2)
    mu = np.asfortranarray(np.stack([t1, t2, t1 * 1.0000001, np.zeros(K), t2 * 0.5], 1))
    mu[0, :] = 0
    lp = np.log([0.003, 0.001, 0.002, 0.001, 1e-40])
    sm = H.StateMatrix.create(5, K, lp, False)
    sm2, mu2 = H.condense_templates(sm, mu, 0.3, 0.05)
    assert mu2.shape[1] < 5 and sm2.N == mu2.shape[1]
    sm3, idx = H.remove_sparse(sm2)
    sm4, idx2 = H.remove_small(sm3, mu2[:, idx], 0.3, 0.05)
    kept = mu2[:, [idx[i] for i in idx2]]
    assert kept.shape[1] == sm4.N and kept.shape[1] >= 2
    assert np.all((kept ** 2).sum(0) > 1.0)       # only templates with real energy survive


def test_every_template_pruned_gives_the_null_model(H):
    # remove_small on templates that are all indistinguishable from noise: StateMatrix(0, K, Float64[]) in the
    # reference (a 0 x 1 state table); here the null model, and train_model hands it back
    K, N = 20, 3
    sm = H.StateMatrix.create(N, K, np.log(np.full(N, 0.01)), False)
    mu = np.zeros((K, N), order="F")
    mu[1:, :] = 1e-3
    sm2, idx = H.remove_small(sm, mu, 0.5)
    assert idx == [] and sm2.N == 0 and sm2.nstates == 1
    from hmmsort_amd.postprocess import reference_postprocess
    sm3, mu3 = reference_postprocess(sm, mu, 0.5)
    assert sm3.N == 0 and mu3.shape == (K, 0)
