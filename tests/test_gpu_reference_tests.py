"""The reference's own end-to-end tests that lie on the accelerated path
(tests/testthat/test-resnmtf.R:38-118: k specified, no stability, no spurious removal), re-created
with seeded data and run through the package's apply_resnmtf mirror -> C-ABI -> HIP, including the
device-side SVD initialisation (resnmtf_init_svd)."""
import numpy as np
import pytest

import resnmtf_amd

pytestmark = pytest.mark.gpu


def planted(seed):
    """test-resnmtf.R:38-52: three 60 x 60 blocks of height 10 + 0.1 |N(0, 1)|."""
    rng = np.random.default_rng(seed)
    n_row = n_col = 60
    rc = np.zeros((3 * n_row, 3)); cc = np.zeros((3 * n_col, 3))
    for i in range(3):
        rc[i * n_row:(i + 1) * n_row, i] = 1
        cc[i * n_col:(i + 1) * n_col, i] = 1
    x = rc @ np.diag([10.0, 10.0, 10.0]) @ cc.T + 0.1 * np.abs(rng.normal(size=(3 * n_row, 3 * n_col)))
    return x, rc, cc


@pytest.mark.parametrize("host_init", [False, True])
def test_two_views_k3_recovers_planted_biclusters(host_init):
    """test-resnmtf.R:98-118 ("resnmtf runs with no stability and no spurious removal")."""
    x1, rc, cc = planted(1)
    x2, _, _ = planted(2)
    from resnmtf_amd import api, naming
    data = [x1, x2]
    if host_init:   # same entry, NumPy full-SVD initialisation (the reference's svd())
        rn, cn = naming.give_names(data, None, None, None, None)
        res = api.res_nmtf_inner(naming.check_data(data), naming.shared_names(rn), naming.shared_names(cn),
                                 k_vec=[3, 3], spurious=False, row_names=rn, col_names=cn, seed=3, host_init=True)
    else:
        res = resnmtf_amd.apply_resnmtf(data, k_val=3, spurious=False, stability=False, seed=3)
    np.testing.assert_allclose(res["output_f"][0].sum(0), np.ones(3), atol=1e-12)       # :103
    np.testing.assert_allclose(res["output_g"][0].sum(0), np.ones(3), atol=1e-12)       # :104
    recon = res["output_f"][0] @ res["output_s"][0] @ res["output_g"][0].T
    assert np.mean(recon.sum(0) - 1.0) < 1e-3                                           # :105-110
    assert len(res["output_f"]) == 2 and res["output_f"][0].shape == (180, 3)           # :111-113
    for v in range(2):                                                                  # :114-117 (setequal of column sums)
        assert sorted(res["row_clusters"][v].sum(0)) == sorted(rc.sum(0))
        assert sorted(res["col_clusters"][v].sum(0)) == sorted(cc.sum(0))


def test_negative_matrix_is_shifted_on_device():
    """test-resnmtf.R:53-58: a negative matrix is made non-negative (the reference warns); here through
    resnmtf_set_view_raw, whose flag is the warning condition."""
    from resnmtf_amd.engine import Engine
    x, _, _ = planted(4)
    e = Engine([180], [180], [3])
    assert e.set_view_raw(0, -x) is True
    d = e.init_svd(0, seed=0)
    e.set_restrictions()
    errs = e.run(30)
    e.close()
    assert np.isfinite(d).all() and np.isfinite(errs).all()


def _restriction_case(seed):
    """test-resnmtf.R:140-160: two planted views, 120 of 180 row (column) names shared, phi = psi = rest_mat with
    rest_mat[1, 2] = 1000."""
    x1, rc, cc = planted(seed)
    x2, _, _ = planted(seed + 1)
    row_names = [[f"row_{i}" for i in range(1, 181)],
                 [f"row_{i}" for i in range(1, 121)] + [f"row_{i}" for i in range(181, 241)]]
    col_names = [[f"col_{i}" for i in range(1, 181)],
                 [f"col_{i}" for i in range(1, 121)] + [f"col_{i}" for i in range(181, 241)]]
    rest = np.zeros((2, 2)); rest[0, 1] = 1000.0
    return [x1, x2], row_names, col_names, rest, rc, cc


@pytest.mark.parametrize("seed", [5, 21])
def test_restriction_matrices_partially_overlapping_names(seed):
    """test-resnmtf.R:140-184 ("resnmtf runs with restriction matrices and partially overlapping") on the HIP path,
    through the apply_resnmtf mirror (naming, init_rest_mats, check_inputs on the host; SVD init, the loop in
    convergence mode, normalisation_check and the cluster matrices on the device).  The reference's only test of
    star_prod_relevant's name matching (R/utils.r:63-78) and of the restricted branches (R/update_steps.r:156-162,
    :194-204)."""
    data, row_names, col_names, rest, rc, cc = _restriction_case(seed)
    res = resnmtf_amd.apply_resnmtf(data, k_val=3, phi=rest, psi=rest, spurious=False, stability=False,
                                    row_names=row_names, col_names=col_names, seed=seed)
    assert len(res["output_f"]) == 2 and res["output_f"][0].shape == (180, 3)              # :159-161
    f1, f2 = res["output_f"]; g1, g2 = res["output_g"]
    # :162-170  rows row_121..180 of view 1 against row_181..240 of view 2 (positions 120..179 in both) are further
    # apart than the shared rows row_1..120
    assert np.mean(np.abs(f1[120:180] - f2[120:180])) > np.mean(np.abs(f1[:120] - f2[:120]))
    assert np.mean(np.abs(g1[120:180] - g2[120:180])) > np.mean(np.abs(g1[:120] - g2[:120]))   # :171-179
    for v in range(2):                                                                     # :180-183
        assert sorted(res["row_clusters"][v].sum(0)) == sorted(rc.sum(0))
        assert sorted(res["col_clusters"][v].sum(0)) == sorted(cc.sum(0))


def test_restriction_case_matches_oracle_from_the_same_initial_factors():
    """The same problem with explicit initial factors (the oracle's init_mats_inner, i.e. the reference's full svd())
    handed to both sides: the HIP path follows the oracle's trajectory -- All_Error, returned F / S / G, and identical
    binary cluster matrices -- over the sweeps the reference's stop test would run, and both satisfy the reference's
    assertions."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import rel_fro
    from oracle import resnmtf_oracle as O
    from resnmtf_amd import api, naming
    data, row_names, col_names, rest, rc, cc = _restriction_case(9)
    data = naming.check_data(data)
    phi = naming.init_rest_mats(rest, 2)
    f0, s0, g0, lam, mu = O.init_mats_inner(data, [3, 3], np.random.default_rng(4))
    z = np.zeros((2, 2))
    ref = O.res_nmtf_inner(data, f0, s0, g0, phi, z, phi, row_names=row_names, col_names=col_names,
                           init_lam=lam, init_mu=mu, max_iters=3000)
    iters = len(ref["All_Error"])
    ref = O.res_nmtf_inner(data, f0, s0, g0, phi, z, phi, row_names=row_names, col_names=col_names,
                           init_lam=lam, init_mu=mu, n_iters=iters)
    res = api.res_nmtf_inner(data, naming.shared_names(row_names), naming.shared_names(col_names), f0, s0, g0, [3, 3],
                             phi, z, phi, n_iters=iters, spurious=False, row_names=row_names, col_names=col_names)
    np.testing.assert_allclose(res["All_Error"], ref["All_Error"], atol=2e-5)
    for v in range(2):
        assert rel_fro(res["output_f"][v], ref["output_f"][v]) < 2e-5
        assert rel_fro(res["output_g"][v], ref["output_g"][v]) < 2e-5
        assert rel_fro(res["output_s"][v], ref["output_s"][v]) < 1e-4
        assert np.array_equal(res["row_clusters"][v], ref["row_clusters"][v])
        assert np.array_equal(res["col_clusters"][v], ref["col_clusters"][v])
        assert sorted(res["row_clusters"][v].sum(0)) == sorted(rc.sum(0))
    f1, f2 = res["output_f"]
    assert np.mean(np.abs(f1[120:180] - f2[120:180])) > np.mean(np.abs(f1[:120] - f2[:120]))
