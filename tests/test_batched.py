"""the batched driver for the repeated factorisations (k sweep R/main.r:279-321,
shuffles R/obtain_bicl.r:11-42, sub-samples R/stability_analysis.r:215-278).  CPU: the sampling
rules and the job sharding over a gloo world of 2; GPU: the jobs through the real path."""
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT
from resnmtf_amd import batched


def test_shuffle_view_permutes_all_entries():
    rng = np.random.default_rng(0)
    x = np.abs(rng.normal(size=(7, 5)))
    y = batched.shuffle_view(x, rng)
    assert y.shape == x.shape and not np.array_equal(x, y)
    np.testing.assert_array_equal(np.sort(x.ravel()), np.sort(y.ravel()))
    # a matrix that can only be arranged with an empty row or column keeps being redrawn until it is not
    z = np.zeros((3, 3)); z[0, 0] = z[1, 1] = z[2, 2] = 1.0
    w = batched.shuffle_view(z, rng)
    assert (w.sum(0) != 0).all() and (w.sum(1) != 0).all()


def test_subsample_views_follows_reference_rules():
    rng = np.random.default_rng(1)
    a = np.abs(rng.normal(size=(40, 30))) + 0.1
    b = np.abs(rng.normal(size=(40, 20))) + 0.1        # same rows as view 1, own columns
    c = np.abs(rng.normal(size=(25, 30))) + 0.1        # own rows, same columns as view 1
    rn = [[f"r{v}_{i}" for i in range(d.shape[0])] for v, d in enumerate((a, b, c))]
    cn = [[f"c{v}_{j}" for j in range(d.shape[1])] for v, d in enumerate((a, b, c))]
    new, rows, cols, nrn, ncn = batched.subsample_views([a, b, c], 0.9, rng, rn, cn)
    assert new[0].shape == (36, 27) and new[1].shape == (36, 18) and new[2].shape == (22, 27)   # floor(dim * 0.9)
    assert np.array_equal(rows[1], rows[0]) and np.array_equal(cols[2], cols[0])                 # shared draws (:114-123)
    assert len(set(rows[0])) == len(rows[0])                                                       # without replacement
    np.testing.assert_array_equal(new[1], b[np.ix_(rows[1], cols[1])])                             # not re-normalised (B11)
    assert nrn[2] == [rn[2][t] for t in rows[2]] and ncn[1] == [cn[1][t] for t in cols[1]]
    # empty columns of a sub-sample are dropped, also from the earlier views sharing the draw (:165-190)
    a2 = a.copy(); b2 = np.abs(rng.normal(size=(40, 30))) + 0.1; b2[:, 3] = 0.0
    new, rows, cols, _, _ = batched.subsample_views([a2, b2], 1.0, np.random.default_rng(2))
    assert 3 not in cols[0] and np.array_equal(cols[0], cols[1]) and new[0].shape[1] == 29


def test_job_lists():
    x = [np.ones((6, 5)), np.ones((6, 4))]
    jobs = batched.k_sweep_jobs(x, 3, 8, phi=np.array([[0, 1.0], [0, 0]]))
    assert [j.k_val for j in jobs] == [3, 4, 5, 6, 7, 8] and all(j.phi is not None for j in jobs)
    x = [np.abs(np.random.default_rng(0).normal(size=(8, 6))) + 0.1]
    sj = batched.shuffled_jobs(x, 3, num_repeats=4)
    assert len(sj) == 4 and all(j.phi is None and j.k_val == 3 for j in sj)
    assert not np.array_equal(sj[0].data[0], sj[1].data[0])
    st = batched.stability_jobs(x, 3, n_stability=3, sample_rate=0.75)
    assert len(st) == 3 and st[0].data[0].shape == (6, 4) and len(st[0].row_names[0]) == 6


def test_run_jobs_shards_over_gloo_world_2(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out = str(tmp_path / "batched.pkl")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "batched_worker.py"), "--rank", str(r), "--world", "2",
                               "--port", str(port), "--out", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=dict(os.environ, OMP_NUM_THREADS="2")) for r in range(2)]
    logs = [p.communicate(timeout=300)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = pickle.load(open(out, "rb"))
    assert [r["k"] for r in res] == [2, 3, 4, 5, 6]                    # complete, in job order, on every rank
    assert [r["rank"] for r in res] == [0, 1, 0, 1, 0]                 # round-robin placement
    assert all(r["sum"] == 78.0 for r in res)


@pytest.mark.gpu
def test_batched_factorisations_on_gpu():
    """k sweep, shuffles and sub-samples of the reference's planted 3-bicluster data (test-resnmtf.R:38-52)
    through the real path (device pre-processing of api.check_data, device SVD init, HIP loop)."""
    rng = np.random.default_rng(5)
    rc = np.kron(np.eye(3), np.ones((60, 1))); cc = np.kron(np.eye(3), np.ones((60, 1)))
    x = [rc @ (10.0 * np.eye(3)) @ cc.T + 0.1 * np.abs(rng.normal(size=(180, 180)))]
    sweep = batched.run_jobs(batched.k_sweep_jobs(x, 2, 4, n_iters=200))
    errs = [r["Error"] for r in sweep]
    assert errs[1] < 0.5 * errs[0] and errs[2] <= errs[1] * 1.05       # k = 3 explains the three planted blocks
    assert sorted(sweep[1]["row_clusters"][0].sum(0)) == [60.0, 60.0, 60.0]
    shuf = batched.run_jobs(batched.shuffled_jobs(x, 3, num_repeats=3, n_iters=200))
    for r in shuf:
        np.testing.assert_allclose(r["output_f"][0].sum(0), 1.0, atol=1e-12)
        assert r["Error"] > 10 * errs[1]                               # shuffled data has no block structure
    stab = batched.run_jobs(batched.stability_jobs(x, 3, n_stability=2, n_iters=200), pre_processed=False)
    for r in stab:
        assert r["output_f"][0].shape == (162, 3) and len(r["extras"]["row_samples"][0]) == 162
        assert sorted(r["row_clusters"][0].sum(0)) == sorted(rc[r["extras"]["row_samples"][0]].sum(0))


@pytest.mark.gpu
def test_device_resident_views_copy_shuffle_readback():
    """resnmtf_copy_view / resnmtf_shuffle_view / resnmtf_get_view: the copy factorises bitwise like a
    fresh upload; the shuffle is a permutation of the entries (normalise = 0), seeded, and with
    normalise = 1 column-normalised like the reference's shuffled input to apply_resnmtf."""
    from resnmtf_amd import synth
    from resnmtf_amd.engine import Engine
    prob = synth.make_problem([(500, 260)], 5)
    x = prob.data[0]
    base = Engine([500], [260], [5]); base.set_view(0, x)
    np.testing.assert_allclose(base.get_view(0), x, rtol=1e-6, atol=0)                    # fp32 device copy
    outs = []
    for mode in ("upload", "copy"):
        e = Engine([500], [260], [5])
        if mode == "upload":
            e.set_view(0, x)
        else:
            e.copy_view_from(0, base, 0)
        e.set_restrictions(); e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
        outs.append((e.run(25), e.finalise(0)[0])); e.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    e = Engine([500], [260], [5])
    e.shuffle_view_from(0, base, 0, seed=11, normalise=False)
    s1 = e.get_view(0)
    np.testing.assert_array_equal(np.sort(s1.ravel()), np.sort(base.get_view(0).ravel()))  # a permutation of the entries
    assert np.mean(s1 == base.get_view(0)) < 0.05                                          # ... that moves them
    e.shuffle_view_from(0, base, 0, seed=11, normalise=False)
    assert np.array_equal(e.get_view(0), s1)                                               # seeded
    e.shuffle_view_from(0, base, 0, seed=12, normalise=False)
    assert not np.array_equal(e.get_view(0), s1)
    e.shuffle_view_from(0, base, 0, seed=11, normalise=True)
    np.testing.assert_allclose(e.get_view(0).sum(0), 1.0, rtol=1e-5)
    e.close(); base.close()


@pytest.mark.gpu
def test_k_sweep_and_shuffles_from_one_upload():
    rng = np.random.default_rng(5)
    rc = np.kron(np.eye(3), np.ones((60, 1)))
    x = [rc @ (10.0 * np.eye(3)) @ rc.T + 0.1 * np.abs(rng.normal(size=(180, 180)))]
    dev = batched.DeviceData(x)
    sweep = batched.k_sweep_on_device(dev, 2, 4, n_iters=200)
    host = batched.run_jobs(batched.k_sweep_jobs(x, 2, 4, n_iters=200))
    for a, b in zip(sweep, host):                                   # same seeds, same device init: same factorisations
        assert abs(a["Error"] - b["Error"]) < 1e-6 * max(1.0, b["Error"]) + 1e-7
    assert sorted(sweep[1]["row_clusters"][0].sum(0)) == [60.0, 60.0, 60.0]
    shuf = batched.shuffles_on_device(dev, 3, num_repeats=3, n_iters=200)
    for r in shuf:
        np.testing.assert_allclose(r["output_f"][0].sum(0), 1.0, atol=1e-12)
        assert r["Error"] > 10 * sweep[1]["Error"]
    assert not np.array_equal(shuf[0]["output_f"][0], shuf[1]["output_f"][0])
    stab = batched.stability_on_device(dev, 3, n_stability=2, n_iters=200)
    for r in stab:
        rows = r["extras"]["row_samples"][0]
        assert r["output_f"][0].shape == (162, 3) and len(set(rows)) == 162
        assert sorted(r["row_clusters"][0].sum(0)) == sorted(rc[rows].sum(0))     # the planted blocks, sub-sampled
    dev.close()


@pytest.mark.gpu
def test_device_subsample_is_the_index_gather():
    from resnmtf_amd import synth
    from resnmtf_amd.engine import Engine
    x = synth.planted_view(90, 70, 3, 4)
    base = Engine([90], [70], [3]); base.set_view(0, x)
    rng = np.random.default_rng(0)
    rows = rng.choice(90, 40, replace=False); cols = rng.choice(70, 33, replace=False)
    e = Engine([40], [33], [3])
    e.subsample_view_from(0, base, 0, rows, cols)
    np.testing.assert_array_equal(e.get_view(0), base.get_view(0)[np.ix_(rows, cols)])      # not re-normalised (B11)
    from resnmtf_amd.engine import ResnmtfError
    with pytest.raises(ResnmtfError, match="out of range"):
        e.subsample_view_from(0, base, 0, rows + 60, cols)
    e.close(); base.close()


def _oracle_from(res, data, phi, xi, psi, row_names, col_names, n_iters):
    """The oracle on the data a batched job factorised, from the initial state the device built (``res["init"]``)."""
    from oracle import resnmtf_oracle as O
    init = res["init"]
    return O.res_nmtf_inner(data, [s[0] for s in init], [s[1] for s in init], [s[2] for s in init], phi, xi, psi,
                            row_names=row_names, col_names=col_names, init_lam=[s[3] for s in init],
                            init_mu=[s[4] for s in init], n_iters=n_iters)


def _close(res, ref, n_v, tol=2e-5):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import rel_fro
    np.testing.assert_allclose(res["All_Error"], ref["All_Error"], atol=2e-5)
    for v in range(n_v):
        assert rel_fro(res["output_f"][v], ref["output_f"][v]) < tol, f"F view {v}"
        assert rel_fro(res["output_g"][v], ref["output_g"][v]) < tol, f"G view {v}"
        assert rel_fro(res["output_s"][v], ref["output_s"][v]) < 1e-4, f"S view {v}"
        assert np.array_equal(res["row_clusters"][v], ref["row_clusters"][v])
        assert np.array_equal(res["col_clusters"][v], ref["col_clusters"][v])


@pytest.mark.gpu
def test_one_batched_job_of_each_kind_matches_the_oracle():
    """the repeated factorisations around the loop.  One job of each kind -- candidate k of the k sweep
    (R/main.r:279-290), a shuffled copy (R/obtain_bicl.r:11-42), a stability sub-sample (R/stability_analysis.r:215-278)
    -- run through the batched driver, host-built data and device-resident data, and compared with the oracle run on
    the SAME data (host-shuffled / host-sub-sampled, or the device's copy read back) from the SAME initial factors
    (the device's SVD initialisation, returned by the job): All_Error, F, S, G and identical cluster matrices."""
    from oracle import resnmtf_oracle as O
    from resnmtf_amd import naming
    rng = np.random.default_rng(9)
    rcm = np.kron(np.eye(3), np.ones((40, 1))); ccm = np.kron(np.eye(3), np.ones((30, 1)))
    x = [rcm @ (10.0 * np.eye(3)) @ ccm.T + 0.1 * np.abs(rng.normal(size=(120, 90))),
         rcm @ (8.0 * np.eye(3)) @ ccm.T + 0.1 * np.abs(rng.normal(size=(120, 90)))]
    phi = np.zeros((2, 2)); phi[0, 1] = 1.5                        # raw restriction matrix: apply_resnmtf symmetrises it
    iters = 40
    z = np.zeros((2, 2))
    # ---- host-built jobs (batched.run_job): data as the job holds it, pre-processed as apply_resnmtf does
    jobs = [batched.k_sweep_jobs(x, 4, 4, phi=phi, n_iters=iters)[0],
            batched.shuffled_jobs(x, 3, num_repeats=1, seed=3, n_iters=iters)[0],
            batched.stability_jobs(x, 3, n_stability=1, phi=phi, n_iters=iters, seed=5)[0]]
    for job in jobs:
        res = batched.run_job(job, return_init=True)
        data = naming.check_data([np.asarray(d, dtype=np.float64) for d in job.data])
        rn, cn = naming.give_names(data, job.phi, job.psi, job.row_names, job.col_names)
        ref = _oracle_from(res, data, naming.init_rest_mats(job.phi, 2), z, naming.init_rest_mats(job.psi, 2), rn, cn, iters)
        _close(res, ref, 2)
    # ---- the same kinds with the data resident on the device: the oracle gets the device's copy read back
    dev = batched.DeviceData(x, phi=phi)
    rows = [rng.choice(120, 100, replace=False)] * 2; cols = [rng.choice(90, 80, replace=False)] * 2
    for kw, coupled in ((dict(k=4), True), (dict(k=3, shuffle_seed=17), False), (dict(k=3, samples=(rows, cols)), True)):
        res = dev.factorise(n_iters=iters, seed=2, return_init=True, return_data=True, **kw)
        p = dev.phi if coupled else z
        rn, cn = (res["row_names"], res["col_names"]) if coupled else naming.give_names(res["data"], None, None)
        ref = _oracle_from(res, res["data"], p, z, z, rn, cn, iters)
        _close(res, ref, 2)
        if "shuffle_seed" in kw:                                   # column-normalised shuffled data (R/obtain_bicl.r:35 -> utils.r:422)
            np.testing.assert_allclose(res["data"][0].sum(0), 1.0, rtol=1e-5)
    dev.close()
    assert O is not None


@pytest.mark.gpu
def test_device_draws_with_empty_rows_and_columns_are_redrawn_or_trimmed():
    """R/obtain_bicl.r:14-18 redraws a shuffle that has an all-zero row or column; R/stability_analysis.r:165-190,
    :233-240 drop all-zero rows / columns of a sub-sample.  Sparse data: 97 % zeros."""
    from resnmtf_amd.engine import Engine
    rng = np.random.default_rng(2)
    x = np.abs(rng.normal(size=(60, 40))) * (rng.random((60, 40)) < 0.03)
    x[:, 0] = 1.0; x[0, :] = 1.0                                    # (every column / row of X itself is non-empty)
    x[5, 1:] = 0.0; x[1:, 7] = 0.0                                  # ... but row 5 / column 7 only through row 0 / column 0
    base = Engine([60], [40], [2]); base.set_view(0, x / x.sum(0)[None, :])
    e = Engine([59], [39], [2])
    rows = np.arange(1, 60); cols = np.arange(1, 40)               # sub-sample without row 0 / column 0
    e.subsample_view_from(0, base, 0, rows, cols)
    er, ec = e.empty_lines(0)
    sub = (x / x.sum(0)[None, :])[np.ix_(rows, cols)]
    assert np.array_equal(er, sub.sum(1) == 0) and np.array_equal(ec, sub.sum(0) == 0)
    assert er[4] and ec[6]                                          # row 5 / column 7 of X
    e.close()
    dev = batched.DeviceData([x])
    trimmed = dev._trim_samples(([rows], [cols]))
    assert trimmed is not None
    tr, tc = trimmed
    final = (x / x.sum(0)[None, :])[np.ix_(tr[0], tc[0])]
    assert (final.sum(0) != 0).all() and (final.sum(1) != 0).all() and 5 not in tr[0] and 7 not in tc[0]
    res = dev.factorise(2, n_iters=10, samples=([rows], [cols]))
    assert res["output_f"][0].shape[0] == len(tr[0]) and np.isfinite(res["All_Error"]).all()
    # shuffles: every accepted draw has no empty row or column (the sparse matrix makes most raw draws fail)
    s = Engine([60], [40], [2])
    accepted = 0
    for seed in range(40):
        s.shuffle_view_from(0, dev.base, 0, seed=seed, normalise=False)
        er, ec = s.empty_lines(0)
        m = s.get_view(0)
        assert np.array_equal(er, m.sum(1) == 0) and np.array_equal(ec, m.sum(0) == 0)
        accepted += int(not (er.any() or ec.any()))
    assert accepted < 40                                           # the redraw loop has work to do on such data
    s.close(); dev.close(); base.close()
