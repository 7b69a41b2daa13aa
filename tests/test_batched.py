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
