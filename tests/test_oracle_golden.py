"""The oracle against the committed golden fixtures (CPU).  The fixtures are
restatement-derived (tests/golden/make_golden.py); this pins the oracle against silent edits."""
import numpy as np
import pytest

from helpers import GOLDEN_NAMES, golden_problem, load_golden, rel_fro, run_oracle


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_oracle_reproduces_golden(name):
    g = load_golden(name)
    res = run_oracle(golden_problem(g), n_iters=g["n_iters"])
    np.testing.assert_allclose(res["All_Error"], g["all_error"], rtol=1e-10, atol=1e-13)
    for v in range(g["n_views"]):
        assert rel_fro(res["output_f"][v], g["out_f"][v]) < 1e-10
        assert rel_fro(res["output_g"][v], g["out_g"][v]) < 1e-10
        assert rel_fro(res["output_s"][v], g["out_s"][v]) < 1e-10
        assert np.array_equal(res["row_clusters"][v], g["rc"][v])
        assert np.array_equal(res["col_clusters"][v], g["cc"][v])


def test_dead_component_stays_dead():
    """g4: a zeroed component gives 0/0 in all three rules; NaN -> 1 keeps it at exactly 0
    (R/update_steps.r:154,192,228) and the other components are untouched by it."""
    g = load_golden("g4_dead_component_nan")
    assert np.all(g["raw_f"][0][:, 2] == 0) and np.all(g["raw_g"][0][:, 2] == 0)
    assert np.all(g["raw_s"][0][2, :] == 0) and np.all(g["raw_s"][0][:, 2] == 0)
    assert np.all(np.isnan(g["out_f"][0][:, 2]))       # 0 / colSums == 0 -> NaN, as in R
    assert np.isfinite(np.delete(g["out_f"][0], 2, axis=1)).all()
