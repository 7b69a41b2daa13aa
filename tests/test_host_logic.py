"""CPU tests of the host side: naming / shared-name maps / index pairs, the reference-API
argument checks, and that the C-ABI library loads and exports every symbol the header
declares (no compute calls -- there is no GPU here and no CPU fallback by design)."""
import os
import re
import warnings

import numpy as np
import pytest

from helpers import ROOT
from resnmtf_amd import naming


def test_give_names_auto_and_copy():
    """R/utils.r:474-491: running counter over views; phi[i, j] > 0 copies names to view j."""
    data = [np.zeros((4, 3)), np.zeros((4, 2)), np.zeros((5, 2))]
    rn, cn = naming.give_names(data)
    assert rn[0] == ["row_1", "row_2", "row_3", "row_4"] and rn[1][0] == "row_5" and rn[2][0] == "row_9"
    assert cn[1] == ["col_4", "col_5"]
    phi = np.zeros((3, 3)); phi[0, 1] = 1.0
    rn, _ = naming.give_names(data, phi=phi)
    assert rn[1] == rn[0] and rn[2][0] == "row_5"
    phi[0, 2] = 1.0
    with pytest.raises(ValueError, match="differing number"):
        naming.give_names(data, phi=phi)
    with pytest.raises(ValueError, match="missing row names"):
        naming.give_names(data, row_names=[["a", "b", "c", "d"], None, None])


def test_shared_names_and_index_pairs_match_oracle():
    from oracle import resnmtf_oracle as O
    rng = np.random.default_rng(5)
    names = [[f"r{i}" for i in rng.permutation(30)[:20]], [f"r{i}" for i in rng.permutation(30)[:25]],
             [f"z{i}" for i in range(7)]]
    ours, ref = naming.shared_names(names), O.reorder_data(names)
    for v in range(3):
        for w in range(3):
            if v == w:
                continue
            if ref[v][w] is None:
                assert ours[v][w] is None
            else:
                assert sorted(ours[v][w]) == sorted(ref[v][w])
    iv, iw = naming.index_pairs(names[0], names[1], ours[0][1])
    assert all(names[0][a] == names[1][b] for a, b in zip(iv, iw))
    assert naming.index_pairs(names[0], names[2], None) == (None, None)


def test_init_rest_mats_and_check_data():
    m = naming.init_rest_mats(np.array([[9.0, 2.0], [0.0, 1.0]]), 2)
    assert np.array_equal(m, np.array([[0.0, 2.0], [2.0, 0.0]]))
    with pytest.raises(ValueError):
        naming.init_rest_mats(-np.ones((2, 2)), 2)
    x = np.array([[1.0, -2.0], [3.0, 4.0]])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = naming.check_data([x])[0]
    assert any("made non-negative" in str(i.message) for i in w)          # R/utils.r:24
    np.testing.assert_allclose(out.sum(0), [1.0, 1.0])
    np.testing.assert_allclose(out[:, 1], [0.0, 1.0])                       # per-COLUMN shift (R/utils.r:22)


def test_api_argument_errors():
    """Same refusals as the reference (R/utils.r:425,440-451) plus explicit out-of-scope errors."""
    import resnmtf_amd
    x = [np.abs(np.random.default_rng(0).standard_normal((12, 9)))]
    with pytest.raises(NotImplementedError, match="stability"):
        resnmtf_amd.apply_resnmtf(x, k_val=3)
    with pytest.raises(NotImplementedError, match="k sweep"):
        resnmtf_amd.apply_resnmtf(x, stability=False, spurious=False)
    with pytest.raises(ValueError, match="ranks"):
        resnmtf_amd.apply_resnmtf(x, k_val=10, stability=False, spurious=False)
    with pytest.raises(NotImplementedError, match="spurious"):
        resnmtf_amd.apply_resnmtf(x, k_val=3, stability=False)
    with pytest.raises(ValueError, match="distance"):
        resnmtf_amd.res_nmtf_inner(x, None, None, k_vec=[3], spurious=False, distance="chebyshev")


def test_library_loads_and_exports_header_symbols():
    from resnmtf_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "resnmtf_hip.h")).read()
    declared = set(re.findall(r"\b(resnmtf_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.resnmtf_abi_version() == _lib.ABI_VERSION == 2
    assert lib.resnmtf_device_count() >= 0


def test_no_cpu_fallback_without_device():
    """On a box without a GPU the product path must fail loudly, never compute on the CPU."""
    from resnmtf_amd import _lib
    from resnmtf_amd.engine import Engine, ResnmtfError
    if _lib.load().resnmtf_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ResnmtfError, match="NO_DEVICE"):
        Engine([10], [8], [3])


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "resnmtf_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".inc", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in text.replace("no CPU oracle", ""), f"{fn} mentions the oracle"
