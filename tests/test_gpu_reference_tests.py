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
