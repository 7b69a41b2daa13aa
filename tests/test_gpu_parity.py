"""Parity tests proper: the HIP path (through the C-ABI) against the fp64 oracle on the same
seeded inputs, against the committed golden fixtures, and -- at BASELINE.json's full c2 size,
where the oracle would take minutes -- through size-independent properties.

Tolerances (BASELINE.json north_star): F, G within 1e-4 relative Frobenius of the CPU
reference on the RETURNED (normalised) outputs; we test at 2e-5 and also bound S and the error
trace.  The device computes X.G / Xt.F with fp32 operands and fp32 MFMA accumulation and
everything else in fp64."""
import numpy as np
import pytest

from helpers import GOLDEN_NAMES, golden_problem, load_golden, rel_fro, run_hip, run_oracle
from resnmtf_amd import synth

pytestmark = pytest.mark.gpu

TOL_FG = 2e-5      # bar is 1e-4
TOL_S = 1e-4
TOL_ERR = 2e-5     # absolute, on All_Error (relative errors in [0, 1])


CLUSTER_EPS = 1e-4   # a binary entry may differ only where the reference's normalised F (G) is within this RELATIVE
                     # distance of the threshold 1/n (1/m): |F n - 1| < 1e-4, i.e. inside the F / G parity bar itself


def check_clusters(got_rc, got_cc, ref_f, ref_s, ref_g, ref_rc, ref_cc, got_s, view=0):
    """Binary cluster matrices (R/obtain_bicl.r:162-180): IDENTICAL to the reference's, except entries whose
    normalised factor value sits within CLUSTER_EPS (relative) of the threshold -- asserted entry by entry.
    Returns the number of entries that differ (all of them on the threshold)."""
    rel_ref = np.argmax(ref_s, axis=0)                                  # relations, obtain_bicl.r:179 (first maximum)
    assert np.array_equal(np.argmax(got_s, axis=0), rel_ref), f"view {view}: S column maxima pair differently"
    differ = 0
    for got, want, fac, src in ((got_rc, ref_rc, ref_f, rel_ref), (got_cc, ref_cc, ref_g, None)):
        assert got.shape == want.shape
        vals = fac if src is None else fac[:, src]                      # row_clusters[, relations], obtain_bicl.r:180
        on_threshold = np.abs(vals * fac.shape[0] - 1.0) < CLUSTER_EPS
        mism = got != want
        assert not (mism & ~on_threshold).any(), (
            f"view {view}: {int((mism & ~on_threshold).sum())} cluster entries differ away from the 1/n threshold")
        differ += int(mism.sum())
    return differ


def check_against(res, ref_f, ref_s, ref_g, ref_rc, ref_cc, ref_err, tol_fg=TOL_FG):
    n_v = len(ref_f)
    np.testing.assert_allclose(res["All_Error"], ref_err, atol=TOL_ERR, rtol=1e-4)
    for v in range(n_v):
        assert rel_fro(res["output_f"][v], ref_f[v]) < tol_fg, f"F view {v}"
        assert rel_fro(res["output_g"][v], ref_g[v]) < tol_fg, f"G view {v}"
        assert rel_fro(res["output_s"][v], ref_s[v]) < TOL_S, f"S view {v}"
        check_clusters(res["row_clusters"][v], res["col_clusters"][v], ref_f[v], ref_s[v], ref_g[v], ref_rc[v], ref_cc[v],
                       res["output_s"][v], view=v)


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_hip_matches_golden(name):
    g = load_golden(name)
    res = run_hip(golden_problem(g), n_iters=g["n_iters"])
    check_against(res, g["out_f"], g["out_s"], g["out_g"], g["rc"], g["cc"], g["all_error"])


def test_golden_exact_clusters_g1():
    g = load_golden("g1_single_60x40_k3")
    res = run_hip(golden_problem(g), n_iters=g["n_iters"])
    assert np.array_equal(res["row_clusters"][0], g["rc"][0])
    assert np.array_equal(res["col_clusters"][0], g["cc"][0])


@pytest.mark.parametrize("shapes,k,kw,iters", [
    ([(100, 50)], 3, {}, 60),                                           # BASELINE c1 (README toy size)
    ([(300, 200)], 5, {}, 200),
    ([(1000, 333)], 16, {}, 100),                                       # ragged m (not a multiple of 64 / 16)
    ([(257, 129)], 17, {}, 50),                                         # k = 17 -> 2 N-tiles, odd sizes
    ([(400, 320)], 48, {}, 30),                                         # 3 N-tiles
    ([(500, 384)], 64, {}, 30),                                         # k = 64 (c5's k)
    ([(600, 200), (600, 150)], 16, {"phi": 200.0}, 100),                # c3-shaped, scaled down
    ([(400, 300)] * 4, 8, {"phi": 2.0, "psi": 1.0}, 60),                # c4-shaped
    ([(320, 256)] * 3, 6, {"phi": 1.0, "psi": 1.0, "xi": 0.3}, 60),     # c5-shaped coupling
])
def test_hip_matches_oracle_seeded(shapes, k, kw, iters):
    prob = synth.make_problem(shapes, k, **kw)
    ref = run_oracle(prob, n_iters=iters)
    res = run_hip(prob, n_iters=iters)
    check_against(res, ref["output_f"], ref["output_s"], ref["output_g"], ref["row_clusters"],
                  ref["col_clusters"], ref["All_Error"])


@pytest.mark.parametrize("kk_mode", [1, 2])
@pytest.mark.parametrize("shapes,k,kw,iters", [
    ([(300, 200)], 5, {}, 80),
    ([(600, 200), (600, 150)], 16, {"phi": 200.0}, 60),
    ([(400, 320)], 48, {}, 25),
    ([(320, 256)] * 3, 6, {"phi": 1.0, "psi": 1.0, "xi": 0.3}, 40),
])
def test_both_kk_modes_match_oracle(shapes, k, kw, iters, kk_mode):
    """The k x k products can come from the update kernels' fp64 partials (mode A, job in workgroup 0
    of the pass launch) or from MFMA aux tiles (mode B, job in the last-arriving aux workgroup, i.e.
    the in-launch ticket/release/acquire hand-off); the library picks by problem size.  Force each."""
    prob = synth.make_problem(shapes, k, **kw)
    ref = run_oracle(prob, n_iters=iters)
    res = run_hip(prob, n_iters=iters, kk_mode=kk_mode)
    check_against(res, ref["output_f"], ref["output_s"], ref["output_g"], ref["row_clusters"],
                  ref["col_clusters"], ref["All_Error"])


def test_mode_b_handoff_is_deterministic():
    """Mode B's in-launch hand-off (aux slabs -> last arriver) must not depend on arrival order:
    repeated runs are bitwise identical (slabs are summed in split order, never by atomics)."""
    prob = synth.make_problem([(3000, 1200)], 16)
    a = run_hip(prob, n_iters=40, kk_mode=2)
    for _ in range(3):
        b = run_hip(prob, n_iters=40, kk_mode=2)
        assert np.array_equal(a["output_f"][0], b["output_f"][0]) and np.array_equal(a["All_Error"], b["All_Error"])


def test_500_sweeps_medium():
    """The north-star protocol (fixed 500 sweeps) at a size the oracle finishes in seconds."""
    prob = synth.make_problem([(2000, 500)], 16)
    ref = run_oracle(prob, n_iters=500)
    res = run_hip(prob, n_iters=500)
    check_against(res, ref["output_f"], ref["output_s"], ref["output_g"], ref["row_clusters"],
                  ref["col_clusters"], ref["All_Error"])


def test_graph_and_eager_agree_bitwise():
    prob = synth.make_problem([(700, 300), (700, 260)], 7, phi=3.0)
    a = run_hip(prob, n_iters=37, use_graph=True)
    b = run_hip(prob, n_iters=37, use_graph=False)
    for v in range(2):
        assert np.array_equal(a["output_f"][v], b["output_f"][v])
        assert np.array_equal(a["output_g"][v], b["output_g"][v])
    assert np.array_equal(a["All_Error"], b["All_Error"])


@pytest.mark.parametrize("shapes,k,kw", [([(700, 300)], 7, {}), ([(10000, 2000)], 16, {}), ([(3000, 900), (3000, 700)], 12, dict(phi=2.0)),
                                         ([(900, 5000)], 5, {}), ([(150, 90), (130, 90)], 3, dict(psi=1.0))])
def test_fused_update_launches_agree_bitwise(shapes, k, kw):
    """resnmtf_options.fuse_updates (opt-in), k <= 16: an uncoupled update_f / update_g (R/update_steps.r:152-155, :190-193)
    rides in the first workgroups of the streaming-pass launch that consumes it (pass_fused_kernel; the other workgroups wait
    on an arrival flag).  Same update code on the same bytes: bitwise the default sweep of separate launches, graph replay and eager, fixed sweeps
    and convergence mode, consecutive runs.  (phi-coupled views: only the G update is fused; psi-coupled: only F.)"""
    from resnmtf_amd.engine import Engine
    prob = synth.make_problem(shapes, k, **kw)
    outs = {}
    for key, opts in (("fused", dict(fuse_updates=1)), ("plain", {}), ("fused_eager", dict(fuse_updates=1, use_graph=False)),
                      ("fused_noprefetch", dict(fuse_updates=2))):
        e = _engine_for(prob, **opts)
        errs = np.concatenate([e.run(9), e.run(23)])
        conv = e.run(None, max_iters=60)
        outs[key] = (errs, conv, [e.get_factors(v) for v in range(len(shapes))])
        e.close()
    for key in ("plain", "fused_eager", "fused_noprefetch"):
        assert np.array_equal(outs["fused"][0], outs[key][0]) and np.array_equal(outs["fused"][1], outs[key][1])
        for va, vb in zip(outs["fused"][2], outs[key][2]):
            for x, y in zip(va, vb):
                assert np.array_equal(x, y)
    assert np.isfinite(outs["fused"][0]).all()


def test_convergence_mode_matches_oracle():
    """R/main.r:50-81: stop after the first sweep with |d mean err| <= 1e-6."""
    prob = synth.make_problem([(300, 200), (280, 150)], 4)
    ref = run_oracle(prob, n_iters=None, max_iters=3000)
    res = run_hip(prob, n_iters=None, max_iters=3000)
    n_ref, n_hip = len(ref["All_Error"]), len(res["All_Error"])
    assert abs(n_ref - n_hip) <= 2, (n_ref, n_hip)      # the stop test sits on rounding
    m = min(n_ref, n_hip)
    np.testing.assert_allclose(res["All_Error"][:m], ref["All_Error"][:m], atol=TOL_ERR)
    assert abs(res["Error"] - ref["Error"]) < TOL_ERR
    if n_ref == n_hip:
        for v in range(2):
            assert rel_fro(res["output_f"][v], ref["output_f"][v]) < TOL_FG


def test_resume_is_exact():
    """get_factors returns the raw state (F, S, G, lambda, mu): 20 sweeps == 12 + 8 resumed."""
    from resnmtf_amd.engine import Engine
    prob = synth.make_problem([(500, 300)], 6)

    def fresh():
        e = Engine([500], [300], [6])
        e.set_view(0, prob.data[0]); e.set_restrictions(None, None, None)
        return e
    e1 = fresh(); e1.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0]); err_a = e1.run(20)
    fa = e1.get_factors(0); e1.close()
    e2 = fresh(); e2.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0]); err_b1 = e2.run(12)
    f, s, g, lam, mu = e2.get_factors(0); e2.close()
    e3 = fresh(); e3.set_factors(0, f, s, g, lam, mu); err_b2 = e3.run(8)
    fb = e3.get_factors(0); e3.close()
    for x, y in zip(fa, fb):
        assert np.array_equal(x, y)
    assert np.array_equal(err_a, np.concatenate([err_b1, err_b2]))


def test_consecutive_runs_continue_without_a_prologue():
    """resnmtf_run directly after resnmtf_run (nothing set in between) skips the run prologue -- the device state IS what
    the prologue would recompute -- and any run length is a few launches off the graph ladder (32 / 16 / 8 / 4 / 2 / 1
    sweeps): 5 + 20 + 37 sweeps in three calls == 62 in one, bit for bit, single view and hoisted multi-view chain,
    fixed sweeps and convergence mode followed by fixed sweeps."""
    from resnmtf_amd.engine import Engine
    for shapes, k, kw in (([(500, 300)], 6, {}), ([(640, 192)] * 3, 12, {"phi": 0.7, "xi": 0.2}), ([(400, 320)], 40, {})):
        prob = synth.make_problem(shapes, k, **kw)
        e1 = _engine_for(prob); err_a = e1.run(62)
        fa = [e1.get_factors(v) for v in range(len(shapes))]; e1.close()
        e2 = _engine_for(prob)
        err_b = np.concatenate([e2.run(5), e2.run(20), e2.run(37)])
        fb = [e2.get_factors(v) for v in range(len(shapes))]
        assert np.array_equal(err_a, err_b)
        for va, vb in zip(fa, fb):
            for x, y in zip(va, vb):
                assert np.array_equal(x, y)
        # set_factors in between forces the prologue again: back to the start, same 5 sweeps
        for v in range(len(shapes)):
            e2.set_factors(v, prob.init_f[v], prob.init_s[v], prob.init_g[v])
        assert np.array_equal(e2.run(5), err_a[:5])
        e2.close()
    prob = synth.make_problem([(300, 200), (280, 150)], 4)
    e3 = _engine_for(prob); c1 = e3.run(None, max_iters=3000); more = e3.run(9); f3 = e3.get_factors(0); e3.close()
    e4 = _engine_for(prob); c2 = e4.run(len(c1) + 9); f4 = e4.get_factors(0); e4.close()
    assert np.array_equal(np.concatenate([c1, more]), c2)
    for x, y in zip(f3, f4):
        assert np.array_equal(x, y)


def test_full_size_c2_properties():
    """BASELINE c2 (10000 x 2000, k = 16) -- too large for the oracle in a test; checked through
    properties: determinism (two runs bitwise equal), non-negativity, unit column sums of the
    returned F and G (R/utils.r:182-189), error trace vs an explicit fp64 residual of the
    returned factorisation, planted clusters recovered (test-resnmtf.R:114-117 analogue)."""
    prob = synth.config("c2")
    a = run_hip(prob, n_iters=60)
    b = run_hip(prob, n_iters=60)
    assert np.array_equal(a["output_f"][0], b["output_f"][0]) and np.array_equal(a["All_Error"], b["All_Error"])
    f, s, g = a["output_f"][0], a["output_s"][0], a["output_g"][0]
    assert (f >= 0).all() and (g >= 0).all() and (s >= 0).all()
    np.testing.assert_allclose(f.sum(0), 1.0, atol=1e-12)
    np.testing.assert_allclose(g.sum(0), 1.0, atol=1e-12)
    x = prob.data[0]
    lit = np.linalg.norm(x - (f @ s) @ g.T, "fro") ** 2 / np.linalg.norm(x, "fro") ** 2
    # B4 quirk: finalise rescales S column-wise, so F S G^T is only approximately preserved;
    # compare against the raw-state error instead through a second, un-finalised read
    from resnmtf_amd.engine import Engine
    e = Engine([10000], [2000], [16]); e.set_view(0, x); e.set_restrictions()
    e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0]); errs = e.run(60)
    fr, sr, gr, _, _ = e.get_factors(0); e.close()
    lit_raw = np.linalg.norm(x - (fr @ sr) @ gr.T, "fro") ** 2 / np.linalg.norm(x, "fro") ** 2
    assert abs(errs[-1] - lit_raw) < 1e-6, (errs[-1], lit_raw)
    assert np.array_equal(errs, a["All_Error"])
    assert np.isfinite(lit)


def _engine_for(prob, **opts):
    """Engine loaded with a (possibly multi-view, coupled) problem -- the C-ABI call sequence of api.py."""
    from resnmtf_amd import naming
    from resnmtf_amd.engine import Engine
    shapes = [x.shape for x in prob.data]
    n_v = len(shapes)
    e = Engine([s[0] for s in shapes], [s[1] for s in shapes], [prob.k] * n_v, **opts)
    for v in range(n_v):
        e.set_view(v, prob.data[v])
        e.set_factors(v, prob.init_f[v], prob.init_s[v], prob.init_g[v])
    e.set_restrictions(prob.phi, prob.xi, prob.psi)
    rs, cs = naming.shared_names(prob.row_names), naming.shared_names(prob.col_names)
    for v in range(n_v):
        for w in range(n_v):
            if v != w:
                e.set_shared_rows(v, w, *naming.index_pairs(prob.row_names[v], prob.row_names[w], rs[v].get(w)))
                e.set_shared_cols(v, w, *naming.index_pairs(prob.col_names[v], prob.col_names[w], cs[v].get(w)))
    return e


@pytest.mark.parametrize("name", ["c3", "c4v1", "c4", "c5v1", "c5v2"])
def test_full_size_single_sweep_parity(name):
    """BASELINE shapes too large for a multi-sweep oracle run: ONE literal oracle sweep
    (R/update_steps.r:272-319 + R/utils.r:157-166) from the device's own state after 10 sweeps must
    reproduce the device's 11th sweep -- F, G, S, lambda, mu and the error -- at full size:
    c3 = 2 phi-coupled views 10000 x {2000, 1500} k = 16; c4 = 4 views 20000 x 4000 k = 32 phi+psi
    (c4v1: one of them, hand-off mode B); c5v1 / c5v2 = one / two views of c5 (50000 x 8000, k = 64,
    phi+psi+xi -- all eight would need 26 GB of fp64 host data for no additional code path)."""
    from oracle import resnmtf_oracle as O
    if name in ("c3", "c4"):
        prob = synth.config(name)
    elif name == "c5v2":
        prob = synth.make_problem([(50000, 8000)] * 2, 64, phi=200.0, psi=200.0, xi=200.0)
    elif name == "c4v1":
        prob = synth.make_problem([(20000, 4000)], 32)
    else:
        prob = synth.make_problem([(50000, 8000)], 64)
    n_v = len(prob.data)
    e = _engine_for(prob)
    e.run(10)
    st0 = [e.get_factors(v) for v in range(n_v)]          # (F, S, G, lambda, mu) raw
    err11 = e.run(1)
    st1 = [e.get_factors(v) for v in range(n_v)]
    e.close()
    rn, cn = prob.row_names, prob.col_names
    ri, ci = O.reorder_data(rn), O.reorder_data(cn)
    f1, s1, g1, lam1, mu1 = O.update_matrices(prob.data, [s[0] for s in st0], [s[1] for s in st0], [s[2] for s in st0],
                                              [s[3] for s in st0], [s[4] for s in st0], prob.phi, prob.xi, prob.psi,
                                              ri, ci, rn, cn)
    norms = np.array([np.linalg.norm(d, "fro") ** 2 for d in prob.data])
    err_ref = O.calculate_error(prob.data, f1, s1, g1, norms).mean()
    for v in range(n_v):
        assert rel_fro(st1[v][0], f1[v]) < TOL_FG, f"F view {v}"
        assert rel_fro(st1[v][2], g1[v]) < TOL_FG, f"G view {v}"
        assert rel_fro(st1[v][1], s1[v]) < TOL_S, f"S view {v}"
        assert rel_fro(st1[v][3], lam1[v]) < TOL_FG and rel_fro(st1[v][4], mu1[v]) < TOL_FG
    assert abs(err11[-1] - err_ref) < TOL_ERR, (err11[-1], err_ref)


@pytest.mark.parametrize("shape,k", [((300, 170), 4), ((1000, 333), 16)])
def test_raw_upload_preprocessing_matches_oracle(shape, k):
    """resnmtf_set_view_raw: make_non_neg_inner + matrix_normalisation (R/utils.r:20-27, 86-88) on the
    device vs the oracle's host pre-processing followed by the plain upload -- same factors after 30
    sweeps (the two differ only in the summation order of colSums), negative-entry flag as the
    reference's warning condition."""
    from oracle import resnmtf_oracle as O
    from resnmtf_amd.engine import Engine
    rng = np.random.default_rng(7)
    n, m = shape
    raw = synth.planted_view(n, m, k, 11) * 50.0 + rng.normal(0.0, 0.5, size=(n, m))   # some columns go negative
    raw[:, ::7] = np.abs(raw[:, ::7])                                                      # ... and some do not
    assert (raw.min(axis=0) < 0).any() and (raw.min(axis=0) >= 0).any()
    pre = O.matrix_normalisation(O.make_non_neg(raw))
    f0, s0, g0 = synth.random_init(n, m, k, 5)
    outs = []
    for mode in ("raw", "host"):
        e = Engine([n], [m], [k])
        if mode == "raw":
            assert e.set_view_raw(0, raw) is True
        else:
            e.set_view(0, pre)
        e.set_restrictions(); e.set_factors(0, f0, s0, g0)
        errs = e.run(30)
        outs.append((e.finalise(0), errs))
        e.close()
    (fa, sa, ga, rca, cca), ea = outs[0]
    (fb, sb, gb, rcb, ccb), eb = outs[1]
    assert rel_fro(fa, fb) < 1e-6 and rel_fro(ga, gb) < 1e-6 and rel_fro(sa, sb) < 1e-5
    np.testing.assert_allclose(ea, eb, atol=1e-7)
    ref = run_oracle(synth.Problem([pre], [f0], [s0], [g0], np.zeros((1, 1)), np.zeros((1, 1)), np.zeros((1, 1)), k), n_iters=30)
    assert rel_fro(fa, ref["output_f"][0]) < TOL_FG and rel_fro(ga, ref["output_g"][0]) < TOL_FG
    e = Engine([n], [m], [k])
    assert e.set_view_raw(0, np.abs(raw)) is False          # nothing negative: no warning condition
    e.close()


def _distinct_blocks(n, m, k, seed):
    """Planted blocks of unequal size and strength (distinct singular values), column-normalised."""
    rng = np.random.default_rng(seed)
    rb = np.sort(rng.choice(np.arange(1, n), size=k - 1, replace=False)); cb = np.sort(rng.choice(np.arange(1, m), size=k - 1, replace=False))
    rb = np.concatenate([[0], rb, [n]]); cb = np.concatenate([[0], cb, [m]])
    x = 0.05 * np.abs(rng.normal(size=(n, m)))
    for j in range(k):
        x[rb[j]:rb[j + 1], cb[j]:cb[j + 1]] += 10.0 / (1.0 + 0.35 * j)
    return x / x.sum(axis=0)[None, :]


@pytest.mark.parametrize("shape,k", [((600, 400), 5), ((3000, 1100), 16), ((2000, 900), 30)])
def test_device_svd_init_matches_oracle(shape, k):
    """resnmtf_init_svd (randomized subspace iteration on the pass kernels) vs the oracle's
    init_mats_inner (full SVD, R/update_steps.r:78-125) with the noise switched off: leading singular
    values, F0, G0, S0, lambda, mu; with noise on: S0 only gains non-negative terms of the right size."""
    from oracle import resnmtf_oracle as O
    from resnmtf_amd.engine import Engine
    n, m = shape
    x = _distinct_blocks(n, m, k, 3)
    rf, rs, rg, rlam, rmu = O.init_mats_inner([x], [k], np.random.default_rng(0), sigma=0.0)
    d_ref = np.linalg.svd(x, compute_uv=False)[:k]
    e = Engine([n], [m], [k]); e.set_view(0, x); e.set_restrictions()
    d = e.init_svd(0, seed=1, sigma=0.0)
    f0, s0, g0, lam, mu = e.get_factors(0)
    np.testing.assert_allclose(d, d_ref, rtol=2e-5)
    # singular VECTORS move by (perturbation / gap): the f32 streaming passes perturb X by ~1e-7, the
    # trailing triplets of the k = 30 case sit ~1e-3 apart in relative terms
    tol = 1e-4 if k <= 16 else 2e-3
    assert rel_fro(f0, rf[0]) < tol and rel_fro(g0, rg[0]) < tol and rel_fro(s0, rs[0]) < 1e-4
    np.testing.assert_allclose(lam, rlam[0], rtol=1e-10); np.testing.assert_allclose(mu, rmu[0], rtol=1e-10)
    e.init_svd(0, seed=2, sigma=0.05)
    _, s1, _, _, _ = e.get_factors(0)
    noise = (s1 - s0) / (f0.sum(0) * 0 + 1.0)            # columns were scaled by cF cG before normalisation
    scale = (np.abs(np.linalg.svd(x, full_matrices=False)[0][:, :k]).sum(0) * np.abs(np.linalg.svd(x, full_matrices=False)[2].T[:, :k]).sum(0))
    noise = noise / scale[None, :]
    assert (noise >= -1e-12).all()
    assert abs(noise.mean() - np.sqrt(2 * 0.05 / np.pi)) < 0.08                     # E|N(0, 0.05)| = 0.178
    # the initialised engine runs: error decreases from the first sweep on
    errs = e.run(20)
    e.close()
    assert np.isfinite(errs).all() and errs[-1] <= errs[0]


@pytest.mark.parametrize("shapes,k,kw", [
    ([(400, 300)], 48, {}),
    ([(1000, 700)], 64, {}),
    ([(300, 200), (300, 150)], 40, dict(phi=50.0, psi=0.0, xi=20.0)),
    ([(700, 500)], 24, {}),
])
def test_bf16_split_option_stays_inside_the_bar(shapes, k, kw):
    """resnmtf_options.bf16_split for k > 16.  0 (default): three bf16 pieces per operand (exact truncation split), six
    products on the K = 32 bf16 MFMA in wide workgroups, f32-grade -- held to the same 2e-5 as everything else.  2: the
    plain f32 MFMA, one tile per workgroup -- same tolerance.  (1, the former two-piece form, is retired and REFUSED: a
    caller that asked for its speed / precision trade must not silently get another one.)"""
    from resnmtf_amd._lib import ResnmtfError
    prob = synth.make_problem(shapes, k, **kw)
    ref = run_oracle(prob, n_iters=30)
    outs = {}
    for mode in (0, 2):
        res = run_hip(prob, n_iters=30, bf16_split=mode)
        check_against(res, ref["output_f"], ref["output_s"], ref["output_g"], ref["row_clusters"], ref["col_clusters"],
                      ref["All_Error"])
        outs[mode] = res["output_f"][0]
    assert not np.array_equal(outs[0], outs[2])        # two distinct arithmetic forms
    with pytest.raises(ResnmtfError, match="retired"):
        run_hip(prob, n_iters=2, bf16_split=1)


@pytest.mark.parametrize("shape,k", [((5, 4), 2), ((17, 3), 3), ((63, 65), 2), ((64, 64), 16), ((3, 70), 2), ((200, 2), 2)])
def test_tiny_and_ragged_shapes(shape, k):
    """Edge sizes: far below one 64-wide tile, one past it, exactly one tile, three rows / two columns
    (k >= 2: the reference itself fails for k = 1, Appendix B13)."""
    prob = synth.make_problem([shape], k)
    ref = run_oracle(prob, n_iters=25)
    res = run_hip(prob, n_iters=25)
    check_against(res, ref["output_f"], ref["output_s"], ref["output_g"], ref["row_clusters"], ref["col_clusters"],
                  ref["All_Error"])


def test_coupled_views_without_shared_names():
    """Appendix B2: views coupled through phi / psi whose name sets are disjoint (the reference's NA maps,
    R/utils.r:70): star_prod_relevant contributes nothing to the numerators, yet phi * F (psi * G) still
    enters the denominators (R/update_steps.r:158, :200)."""
    prob = synth.make_problem([(120, 90), (100, 90)], 4)
    off = 1.0 - np.eye(2)
    prob.phi, prob.psi = 5.0 * off, 3.0 * off
    prob.row_names = [[f"a{i}" for i in range(120)], [f"b{i}" for i in range(100)]]      # no common row name
    prob.col_names = [[f"c{j}" for j in range(90)], [f"d{j}" for j in range(90)]]        # no common column name
    ref = run_oracle(prob, n_iters=40)
    res = run_hip(prob, n_iters=40)
    check_against(res, ref["output_f"], ref["output_s"], ref["output_g"], ref["row_clusters"], ref["col_clusters"],
                  ref["All_Error"])
    uncoupled = synth.make_problem([(120, 90), (100, 90)], 4)
    free = run_hip(uncoupled, n_iters=40)
    assert rel_fro(res["output_f"][0], free["output_f"][0]) > 1e-3        # the denominators did change the result


def test_zero_rows_and_columns_in_x():
    """A view with all-zero rows (legal: only the columns are normalised): the rows of F they feed go to
    zero exactly as in the reference; the NaN -> 1 guard of the unrestricted branch keeps 0/0 quotients."""
    prob = synth.make_problem([(150, 80)], 3)
    x = prob.data[0].copy(); x[10:14, :] = 0.0
    x = x / x.sum(axis=0)[None, :]
    prob.data[0] = x
    ref = run_oracle(prob, n_iters=40)
    res = run_hip(prob, n_iters=40)
    check_against(res, ref["output_f"], ref["output_s"], ref["output_g"], ref["row_clusters"], ref["col_clusters"],
                  ref["All_Error"])
    assert (res["output_f"][0][10:14] == 0).all() or np.allclose(res["output_f"][0][10:14], ref["output_f"][0][10:14], atol=1e-300)


@pytest.mark.parametrize("shape,k", [((5, 4), 2), ((17, 3), 3), ((30, 20), 10), ((200, 12), 12), ((9, 500), 4)])
def test_device_svd_init_thin_views(shape, k):
    """Views whose short side is smaller than the random sketch take the exact route (Gram of the short
    side on the device, Jacobi on the host): same outputs as the oracle's full-SVD init_mats_inner."""
    from oracle import resnmtf_oracle as O
    from resnmtf_amd.engine import Engine
    n, m = shape
    x = _distinct_blocks(n, m, min(k, 3), 9) if min(n, m) >= 6 else synth.planted_view(n, m, k, 7)
    rf, rs, rg, rlam, rmu = O.init_mats_inner([x], [k], np.random.default_rng(0), sigma=0.0)
    d_ref = np.linalg.svd(x, compute_uv=False)[:k]
    e = Engine([n], [m], [k]); e.set_view(0, x); e.set_restrictions()
    d = e.init_svd(0, seed=1, sigma=0.0)
    f0, s0, g0, lam, mu = e.get_factors(0)
    e.close()
    # X lives on the device in fp32 and the Gram squares the condition number: the smallest of the k
    # triplets of a full-rank thin view is the least accurate
    np.testing.assert_allclose(d, d_ref, rtol=1e-5, atol=1e-7 * d_ref[0])
    strong = d_ref > 1e-3 * d_ref[0]
    assert rel_fro(f0[:, strong], rf[0][:, strong]) < 1e-4 and rel_fro(g0[:, strong], rg[0][:, strong]) < 1e-4


@pytest.mark.parametrize("shapes,k,kw", [
    ([(600, 200), (600, 150)], 16, {"phi": 200.0}),                     # 2 views, X.G in 4 splits
    ([(640, 128)] * 3, 7, {"phi": 1.5, "xi": 0.2}),                     # 3 views -> the 4-view instantiation
    ([(512, 192)] * 5, 12, {"phi": 0.7}),                               # 5 views -> the 8-view instantiation, fall-back emit (> 4 owned)
    ([(300, 128), (300, 128)], 5, {}),                                  # uncoupled views: NaN -> 1 branch inside the chain
])
def test_hoisted_fused_f_chain_equals_per_view_launches(shapes, k, kw):
    """resnmtf_run hoists the F updates of a sweep into one launch (f_chain_kernel) when the views share their rows
    in the same order and k <= 16.  Same arithmetic as one factor_update_kernel launch per view in the
    reference's order: results must agree to rounding of nothing (bitwise), fixed sweeps and convergence mode."""
    prob = synth.make_problem(shapes, k, **kw)
    a = run_hip(prob, n_iters=40)
    b = run_hip(prob, n_iters=40, no_f_chain=True)
    assert np.array_equal(a["All_Error"], b["All_Error"])
    for v in range(len(shapes)):
        assert np.array_equal(a["output_f"][v], b["output_f"][v])
        assert np.array_equal(a["output_g"][v], b["output_g"][v])
        assert np.array_equal(a["output_s"][v], b["output_s"][v])
    ref = run_oracle(prob, n_iters=40)
    check_against(a, ref["output_f"], ref["output_s"], ref["output_g"], ref["row_clusters"], ref["col_clusters"],
                  ref["All_Error"])
    c = run_hip(prob, n_iters=None, max_iters=400)
    d = run_hip(prob, n_iters=None, max_iters=400, no_f_chain=True)
    assert len(c["All_Error"]) == len(d["All_Error"]) and np.array_equal(c["All_Error"], d["All_Error"])


@pytest.mark.parametrize("shapes,k,kw,iters", [
    ([(1000, 333)], 16, {}, 100),
    ([(600, 200), (600, 150)], 16, {"phi": 200.0}, 60),
])
def test_experimental_fp16_image_passes(shapes, k, kw, iters):
    """resnmtf_options.x_half (EXPERIMENTAL, off by default, NOT a parity mode): the passes stream an fp16 image
    of X.  11 bits of X are not enough for the 1e-4 bar on every problem (measured 2e-5 ... 6e-4,
    tools/half_parity.py), so this only pins the mechanics: layouts, scales, ragged splits -- an indexing
    error would show as O(1), not O(1e-3)."""
    prob = synth.make_problem(shapes, k, **kw)
    ref = run_oracle(prob, n_iters=iters)
    res = run_hip(prob, n_iters=iters, x_half=1)
    for v in range(len(shapes)):
        assert rel_fro(res["output_f"][v], ref["output_f"][v]) < 3e-3
        assert rel_fro(res["output_g"][v], ref["output_g"][v]) < 3e-3
    np.testing.assert_allclose(res["All_Error"], ref["All_Error"], atol=5e-3)


@pytest.mark.parametrize("shapes,k,kw,iters", [
    ([(300, 200)], 5, {}, 200),
    ([(1000, 333)], 16, {}, 100),                                       # ragged m
    ([(600, 200), (600, 150)], 16, {"phi": 200.0}, 100),                # c3-shaped, fused F chain + K-packed operand copies
    ([(400, 300)] * 4, 8, {"phi": 2.0, "psi": 1.0}, 60),
    ([(320, 256)] * 3, 6, {"phi": 1.0, "psi": 1.0, "xi": 0.3}, 60),
])
def test_uniform_16bit_image_passes_inside_the_bar(shapes, k, kw, iters):
    """resnmtf_options.x_half = 2 (opt-in): the passes stream X as uniform 16-bit integers (one power-of-two step per
    view), widened exactly to f32.  F / G measured 1e-6 ... 3e-5 from the fp64 oracle (tools/half_parity.py, same
    values as the CPU study tools/quant_study.py predicts); asserted at 5e-5, the bar is 1e-4."""
    prob = synth.make_problem(shapes, k, **kw)
    ref = run_oracle(prob, n_iters=iters)
    res = run_hip(prob, n_iters=iters, x_half=2)
    for v in range(len(shapes)):
        assert rel_fro(res["output_f"][v], ref["output_f"][v]) < 5e-5
        assert rel_fro(res["output_g"][v], ref["output_g"][v]) < 5e-5
        assert rel_fro(res["output_s"][v], ref["output_s"][v]) < 1e-4
        assert np.array_equal(res["row_clusters"][v], ref["row_clusters"][v])
        assert np.array_equal(res["col_clusters"][v], ref["col_clusters"][v])
    np.testing.assert_allclose(res["All_Error"], ref["All_Error"], atol=5e-4)


def test_guarded_16bit_image_falls_back_on_outliers():
    """x_half = 3: the 16-bit integer image is used per view only when its relative quantisation error, measured at
    upload, is at most 3e-5.  Planted blocks pass the guard; the same data with a few x100 outliers does not and runs on
    the f32 images (f32-grade parity) -- factors uploaded BEFORE the data (operand copies rewritten) and after."""
    from resnmtf_amd.engine import Engine
    from resnmtf_amd.synth import Problem
    n, m, k = 600, 300, 8
    clean = synth.make_problem([(n, m)], k)
    rng = np.random.default_rng(5)
    raw = synth.planted_view(n, m, k, 77, normalise=False)
    raw = raw + 100.0 * raw.max() * (rng.random((n, m)) < 20.0 / (n * m))
    dirty = Problem([raw / raw.sum(axis=0)[None, :]], clean.init_f, clean.init_s, clean.init_g, clean.phi, clean.xi, clean.psi, k)
    for prob, want_kind in ((clean, 2), (dirty, 0)):
        ref = run_oracle(prob, n_iters=60)
        for factors_first in (False, True):
            with Engine([n], [m], [k], x_half=3) as e:
                if factors_first:
                    e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
                e.set_view(0, prob.data[0])
                if not factors_first:
                    e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
                e.set_restrictions(None, None, None)
                kind, rel = e.view_image_info(0)
                assert kind == want_kind, (kind, rel)
                assert (rel <= 3e-5) == (want_kind == 2)
                errs = e.run(60)
                f, s, g, _, _ = e.finalise(0)
            tol = 5e-5 if want_kind == 2 else TOL_FG
            assert rel_fro(f, ref["output_f"][0]) < tol and rel_fro(g, ref["output_g"][0]) < tol
            np.testing.assert_allclose(errs, ref["All_Error"], atol=5e-4 if want_kind == 2 else TOL_ERR)
