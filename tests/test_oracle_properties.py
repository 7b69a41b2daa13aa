"""Property pins the reference's own tests hold for the hot path, re-created with seeds
against the oracle (tests/testthat/test-resnmtf.R:38-52 data; :98-118 and :140-184 asserts).
These are the only behavioural pins the reference owns (it has no numeric golden vectors)."""
import numpy as np
import pytest

from oracle import resnmtf_oracle as O

N = 60


def planted_pair(seed):
    rng = np.random.default_rng(seed)
    r = np.zeros((3 * N, 3)); c = np.zeros((3 * N, 3))
    for i in range(3):
        r[i * N:(i + 1) * N, i] = 1; c[i * N:(i + 1) * N, i] = 1                       # test-resnmtf.R:39-44
    data = [r @ np.diag([10.0, 10.0, 10.0]) @ c.T + 0.1 * np.abs(rng.standard_normal((3 * N, 3 * N)))
            for _ in range(2)]                                                         # :46-51
    data = [O.matrix_normalisation(O.make_non_neg(d)) for d in data]                   # check_inputs
    return data, rng


@pytest.mark.parametrize("seed", [7, 11])
def test_no_stability_no_spurious_properties(seed):
    """test-resnmtf.R:98-118."""
    data, rng = planted_pair(seed)
    f, s, g, lam, mu = O.init_mats_inner(data, [3, 3], rng)
    z = np.zeros((2, 2))
    res = O.res_nmtf_inner(data, f, s, g, z, z, z, init_lam=lam, init_mu=mu, max_iters=2000)
    np.testing.assert_allclose(res["output_f"][0].sum(0), np.ones(3), atol=1.5e-8)     # :103
    np.testing.assert_allclose(res["output_g"][0].sum(0), np.ones(3), atol=1.5e-8)     # :104
    recon = res["output_f"][0] @ res["output_s"][0] @ res["output_g"][0].T
    assert np.mean(recon.sum(0) - 1.0) < 1e-3                                          # :105-110
    assert len(res["output_f"]) == 2 and res["output_f"][0].shape == (3 * N, 3)       # :111-113
    for v in range(2):                                                                 # :114-117
        assert sorted(res["row_clusters"][v].sum(0)) == [N, N, N]
        assert sorted(res["col_clusters"][v].sum(0)) == [N, N, N]


def test_restriction_partial_overlap():
    """test-resnmtf.R:140-184: names overlap on 120 of 180, phi = psi = 1000 on the pair."""
    data, rng = planted_pair(3)
    row_names = [[f"row_{i}" for i in range(1, 181)],
                 [f"row_{i}" for i in range(1, 121)] + [f"row_{i}" for i in range(181, 241)]]
    col_names = [[f"col_{i}" for i in range(1, 181)],
                 [f"col_{i}" for i in range(1, 121)] + [f"col_{i}" for i in range(181, 241)]]
    rest = np.zeros((2, 2)); rest[0, 1] = 1000.0
    phi = O.init_rest_mats(rest, 2)
    f, s, g, lam, mu = O.init_mats_inner(data, [3, 3], rng)
    z = np.zeros((2, 2))
    res = O.res_nmtf_inner(data, f, s, g, phi, z, phi, row_names=row_names, col_names=col_names,
                           init_lam=lam, init_mu=mu, max_iters=2000)
    f1, f2 = res["output_f"]; g1, g2 = res["output_g"]
    assert np.mean(np.abs(f1[120:180] - f2[120:180])) > np.mean(np.abs(f1[:120] - f2[:120]))   # :161-169
    assert np.mean(np.abs(g1[120:180] - g2[120:180])) > np.mean(np.abs(g1[:120] - g2[:120]))   # :170-178
    for v in range(2):                                                                          # :179-182
        assert sorted(res["row_clusters"][v].sum(0)) == [N, N, N]
        assert sorted(res["col_clusters"][v].sum(0)) == [N, N, N]


def test_restriction_matrix_symmetrisation():
    """R/update_steps.r:12-24: NULL -> zeros; diagonal zeroed; M + t(M) (a symmetric input is doubled)."""
    assert np.array_equal(O.init_rest_mats(None, 3), np.zeros((3, 3)))
    m = np.array([[5.0, 2.0], [0.0, 7.0]])
    assert np.array_equal(O.init_rest_mats(m, 2), np.array([[0.0, 2.0], [2.0, 0.0]]))
    sym = np.array([[0.0, 3.0], [3.0, 0.0]])
    assert np.array_equal(O.init_rest_mats(sym, 2), 2 * sym)


def test_shared_name_maps():
    """R/utils.r:560-662: shared names per ordered pair, NA when empty."""
    names = [["a", "b", "c"], ["c", "x", "a"], ["q"]]
    sh = O.reorder_data(names)
    assert sorted(sh[0][1]) == ["a", "c"] and sorted(sh[1][0]) == ["a", "c"]
    assert sh[0][2] is None and sh[2][0] is None and sh[1][2] is None


def test_trace_form_identity():
    """(F^T X) G == (X^T F)^T G and the k x k trace form of the error."""
    rng = np.random.default_rng(0)
    x = np.abs(rng.standard_normal((50, 30))); f = np.abs(rng.standard_normal((50, 4)))
    s = np.abs(rng.standard_normal((4, 4))); g = np.abs(rng.standard_normal((30, 4)))
    lit = np.linalg.norm(x - f @ s @ g.T, "fro") ** 2
    n_ = (x.T @ f).T @ g
    tr = (x ** 2).sum() - 2 * (s * n_).sum() + (((f.T @ f) @ s @ (g.T @ g)) * s).sum()
    assert abs(lit - tr) < 1e-9 * lit
    np.testing.assert_allclose((f.T @ x) @ g, n_, rtol=1e-12)
