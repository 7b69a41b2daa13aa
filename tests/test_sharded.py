"""Multi-rank path: world_size 2 over gloo.

* CPU (not gpu): the product's ShardedSweep driver with a stand-in engine whose phases are the
  oracle's update rules.  Checks that the ordered-broadcast schedule reproduces the reference's
  Gauss-Seidel sweep EXACTLY (same arithmetic, so agreement to rounding of nothing: bitwise up to
  the error formula) against the single-process oracle, plus the exchange-plan logic.
* GPU (gpu): the same driver with the real HIP engine, two ranks sharing the one GPU of the box.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT, rel_fro


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(mode, tmp_path, world=2, sweeps=12, timeout=600, xi=0.4, k=5, extra=()):
    port = free_port()
    out = str(tmp_path / f"sharded_{mode}.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), "--rank", str(r),
                               "--world", str(world), "--port", str(port), "--mode", mode, "--sweeps", str(sweeps),
                               "--xi", str(xi), "--k", str(k), "--out", out, *[str(x) for x in extra]], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return np.load(out)


def oracle_reference(sweeps=12):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker
    from oracle import resnmtf_oracle as O
    prob = dist_worker.build_problem()
    return O.res_nmtf_inner(prob.data, prob.init_f, prob.init_s, prob.init_g, prob.phi, prob.xi, prob.psi,
                            row_names=prob.row_names, col_names=prob.col_names, n_iters=sweeps)


def oracle_reference_one_view_per_rank(world=2, sweeps=12, identity=False, xi=0.4, k=5):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker
    from oracle import resnmtf_oracle as O
    prob = dist_worker.build_problem_one_view_per_rank(world, identity=identity, xi=xi, k=k)
    return O.res_nmtf_inner(prob.data, prob.init_f, prob.init_s, prob.init_g, prob.phi, prob.xi, prob.psi,
                            row_names=prob.row_names, col_names=prob.col_names, n_iters=sweeps)


def test_exchange_plan():
    from resnmtf_amd import naming
    from resnmtf_amd.sharded import exchange_plan
    names = [["a", "b", "c"], ["c", "a", "z"], ["q", "r", "s"]]
    sh = naming.shared_names(names)
    phi = np.array([[0, 2.0, 2.0], [2.0, 0, 0], [2.0, 0, 0]])
    z = np.zeros((3, 3))
    # views 0,1 on different ranks and sharing rows -> F of both travels; view 2 shares nothing (NA)
    plan = exchange_plan(3, [0, 1, 0], phi, z, z, sh, sh)
    assert plan[0]["F"] and plan[1]["F"] and not plan[2]["F"]
    assert not any(p["G"] or p["S"] for p in plan)
    # everything on one rank -> nothing travels
    plan = exchange_plan(3, [0, 0, 0], phi, phi, phi, sh, sh)
    assert not any(any(p.values()) for p in plan)
    # xi couples S regardless of names
    plan = exchange_plan(3, [0, 1, 1], z, phi, z, sh, sh)
    assert plan[0]["S"] and plan[1]["S"] and plan[2]["S"]


def test_sharded_schedule_matches_oracle_gloo_cpu(tmp_path):
    got = launch("cpu", tmp_path)
    ref = oracle_reference()
    np.testing.assert_allclose(got["all_error"], ref["All_Error"], rtol=1e-12, atol=1e-14)
    for v in range(3):
        assert rel_fro(got[f"output_f{v}"], ref["output_f"][v]) < 1e-13
        assert rel_fro(got[f"output_g{v}"], ref["output_g"][v]) < 1e-13
        assert rel_fro(got[f"output_s{v}"], ref["output_s"][v]) < 1e-13
        assert np.array_equal(got[f"row_clusters{v}"], ref["row_clusters"][v])
        assert np.array_equal(got[f"col_clusters{v}"], ref["col_clusters"][v])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["gpu", "gpu_norep"])
def test_sharded_hip_two_ranks_one_gpu(tmp_path, mode):
    """mode "gpu": replicated F chain (every rank runs the F update of every coupled view from the
    broadcast exchange blocks); "gpu_norep": ordered F broadcasts.  Same results, and every rank's copy of
    every F is bitwise the owner's."""
    got = launch(mode, tmp_path)
    assert bool(got["mirrors_ok"])
    ref = oracle_reference()
    np.testing.assert_allclose(got["all_error"], ref["All_Error"], atol=2e-5, rtol=1e-4)
    for v in range(3):
        assert rel_fro(got[f"output_f{v}"], ref["output_f"][v]) < 2e-5
        assert rel_fro(got[f"output_g{v}"], ref["output_g"][v]) < 2e-5
        assert rel_fro(got[f"output_s{v}"], ref["output_s"][v]) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_hip_allgather_layout(tmp_path, world):
    """One view per rank, equal exchange blocks: the blocks travel once per sweep over the library's block
    arena (one all-gather on RCCL; gloo, used here because the ranks share one GPU, moves the same bytes by
    one broadcast per block).  Same results as the sequential oracle, every F mirror bitwise the owner's."""
    got = launch("gpu_allgather", tmp_path, world=world)
    assert bool(got["mirrors_ok"])
    ref = oracle_reference_one_view_per_rank(world)
    np.testing.assert_allclose(got["all_error"], ref["All_Error"], atol=2e-5, rtol=1e-4)
    for v in range(world):
        assert rel_fro(got[f"output_f{v}"], ref["output_f"][v]) < 2e-5
        assert rel_fro(got[f"output_g{v}"], ref["output_g"][v]) < 2e-5
        assert rel_fro(got[f"output_s{v}"], ref["output_s"][v]) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("world,xi", [(2, 0.4), (3, 0.4), (4, 0.4), (2, 0.0), (4, 0.0)])
def test_sharded_hip_fused_f_chain(tmp_path, world, xi):
    """Views sharing their rows in the same order: RESNMTF_PHASE_F_ALL is one launch (f_chain_kernel, the 2- and
    4-view instantiations; the 8-view one runs in test_f_chain_eight_view_instantiation_one_process -- a box allows six
    processes on its GPU, pytest included).  Against the sequential oracle, and against the same run with one launch per view.
    xi = 0: only the F blocks cross ranks, the rank's share of a sweep is one RESNMTF_PHASE_LOCAL_SWEEP call."""
    got = launch("gpu_chain", tmp_path, world=world, xi=xi)
    assert bool(got["mirrors_ok"])
    ref = oracle_reference_one_view_per_rank(world, identity=True, xi=xi)
    np.testing.assert_allclose(got["all_error"], ref["All_Error"], atol=2e-5, rtol=1e-4)
    off = launch("gpu_chain_off", tmp_path, world=world, xi=xi)
    np.testing.assert_allclose(got["all_error"], off["all_error"], atol=1e-12, rtol=1e-10)
    for v in range(world):
        assert rel_fro(got[f"output_f{v}"], ref["output_f"][v]) < 2e-5
        assert rel_fro(got[f"output_g{v}"], ref["output_g"][v]) < 2e-5
        assert rel_fro(got[f"output_s{v}"], ref["output_s"][v]) < 1e-4
        assert rel_fro(got[f"output_f{v}"], off[f"output_f{v}"]) < 1e-10
        assert np.array_equal(got[f"row_clusters{v}"], ref["row_clusters"][v])


def test_allgather_layout_eligibility():
    """Host logic only: the blocks travel by one all-gather iff every rank owns exactly one view (view v on rank v),
    every view is replicated and the blocks tile one arena evenly."""
    import torch
    from resnmtf_amd import sharded

    class FakeEngine:
        def __init__(self, sizes):
            self.arena = torch.zeros(sum(sizes), dtype=torch.float64)
            self.off = np.concatenate([[0], np.cumsum(sizes)]).astype(int)

        def factor_tensor(self, v, which):
            return self.arena if which == "FBLOCK_ALL" else self.arena[self.off[v]:self.off[v + 1]]

    prob = sharded.local_problem(2, (64, 48), 4, phi=1.0, owned=[])
    mk = lambda sizes, owner_of, world, **kw: sharded.ShardedSweep(prob, owner_of, 0, world, engine=FakeEngine(sizes),
                                                                  replicate_f=True, **kw)
    assert mk([32, 32], [0, 1], 2).allgather_layout
    assert not mk([32, 40], [0, 1], 2).allgather_layout            # unequal blocks
    assert not mk([32, 32], [1, 0], 2).allgather_layout            # view v not on rank v
    assert not mk([32, 32], [0, 1], 2, allgather_blocks=False).allgather_layout
    uncoupled = sharded.local_problem(2, (64, 48), 4, phi=0.0, owned=[])
    drv = sharded.ShardedSweep(uncoupled, [0, 1], 0, 2, engine=FakeEngine([32, 32]), replicate_f=True)
    assert not any(drv.replicated) and not drv.allgather_layout    # nothing to exchange at all


@pytest.mark.gpu
def test_sharded_graph_chunk_replay_one_rank_rccl(tmp_path):
    """Opt-in graph_chunk: sweeps of the sharded loop (library launches + the in-place RCCL all-gather) captured in one
    graph and replayed must give bitwise the eager loop's results (one rank: RCCL refuses two ranks on one GPU)."""
    got = launch("gpu_graph1", tmp_path, world=1, sweeps=19)
    assert bool(got["same"])
    assert len(got["all_error"]) == 19 and np.isfinite(got["all_error"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("views", [6, 8])
def test_f_chain_eight_view_instantiation_one_process(views):
    """One rank's share of a `views`-way sharded run in ONE process: view 0 owned, the others F replicas whose exchange
    blocks are copies of view 0's.  The fused launch (f_chain_kernel<8, 1>) against one launch per view: bitwise."""
    import torch
    from resnmtf_amd import _lib, sharded
    from resnmtf_amd.engine import Engine
    n, m, k = 700, 192, 11
    prob = sharded.local_problem(views, (n, m), k, phi=3.0, owned=[0])
    out = {}
    for off in (False, True):
        st = torch.cuda.Stream()
        eng = Engine([n] * views, [m] * views, [k] * views, owned=[v == 0 for v in range(views)], stream=st.cuda_stream,
                     replicate_f=True, no_f_chain=off)
        eng.set_view(0, prob.data[0])
        for v in range(views):
            eng.set_factors(v, prob.init_f[v], prob.init_s[v], prob.init_g[v])
        eng.set_restrictions(prob.phi, prob.xi, prob.psi)
        idx = np.arange(n, dtype=np.int32)
        for v in range(views):
            for w in range(views):
                if v != w:
                    eng.set_shared_rows(v, w, idx, idx)
                    eng.set_shared_cols(v, w, None, None)
        eng.reserve_sweeps(64)
        eng.prepare()
        eng.synchronize()
        ad = sharded.HipEngineAdapter(eng)
        blk0 = ad.factor_tensor(0, "FBLOCK")
        for v in range(1, views):
            ad.factor_tensor(v, "FBLOCK").copy_(blk0)
        torch.cuda.synchronize()
        for _ in range(3):
            eng.phase(0, _lib.PHASE_F_ALL, 0)
        eng.synchronize()
        out[off] = [ad.factor_tensor(v, "F").cpu().numpy().copy() for v in range(views)]
        ad._views.clear()
        eng.close()
    for v in range(views):
        assert np.isfinite(out[False][v]).all() and out[False][v].max() > 0
        assert np.array_equal(out[False][v], out[True][v]), f"view {v}"


@pytest.mark.gpu
@pytest.mark.parametrize("views,k,n,m", [(3, 32, 450, 130), (5, 64, 300, 70)])
def test_wide_chain_f_only_one_process_bitwise(views, k, n, m):
    """The same for a phi-only layout (replicated F chain, G and S stay with the owner): PHASE_F_ALL through
    wide_chain_kernel against one launch per view, then the owner's PHASE_G on the result -- F of every view, G and S of
    the owned one, bitwise."""
    import torch
    from resnmtf_amd import _lib, sharded
    from resnmtf_amd.engine import Engine
    prob = sharded.local_problem(views, (n, m), k, phi=3.0, owned=[0])
    out = {}
    for off in (False, True):
        st = torch.cuda.Stream()
        eng = Engine([n] * views, [m] * views, [k] * views, owned=[v == 0 for v in range(views)], stream=st.cuda_stream,
                     replicate_f=True, no_f_chain=off)
        eng.set_view(0, prob.data[0])
        for v in range(views):
            eng.set_factors(v, prob.init_f[v], prob.init_s[v], prob.init_g[v])
        eng.set_restrictions(prob.phi, prob.xi, prob.psi)
        ri = np.arange(n, dtype=np.int32)
        for v in range(views):
            for w in range(views):
                if v != w:
                    eng.set_shared_rows(v, w, ri, ri)
                    eng.set_shared_cols(v, w, None, None)
        eng.reserve_sweeps(64)
        eng.prepare()
        eng.synchronize()
        ad = sharded.HipEngineAdapter(eng)
        for t in range(3):
            blk0 = ad.factor_tensor(0, "FBLOCK")
            for v in range(1, views):
                ad.factor_tensor(v, "FBLOCK").copy_(blk0)
            torch.cuda.synchronize()
            eng.phase(0, _lib.PHASE_F_ALL, t)
            eng.phase(0, _lib.PHASE_G, t)
            eng.synchronize()
        res = [ad.factor_tensor(v, "F").cpu().numpy().copy() for v in range(views)]
        res += [ad.factor_tensor(0, "G").cpu().numpy().copy(), ad.factor_tensor(0, "S").cpu().numpy().copy()]
        out[off] = res
        ad._views.clear()
        eng.close()
    for x, y in zip(out[False], out[True]):
        assert np.isfinite(x).all() and np.abs(x).max() > 0
        assert np.array_equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("views,k,n,m", [(3, 24, 700, 200), (6, 32, 333, 161), (4, 64, 1000, 330), (8, 64, 517, 96), (8, 57, 2100, 64), (5, 40, 410, 75),
                                         (4, 11, 700, 192), (8, 16, 517, 96), (2, 5, 96, 72)])
def test_wide_chain_one_process_bitwise(views, k, n, m):
    """One rank's share of a `views`-way sharded run with replicated F / G / S chains, in ONE process (k > 16: wide_chain_kernel,
    F and G form; k <= 16: f_chain_kernel, F form and -- round 3 -- its G form with the cross product T^T G' among the partials): view 0
    owned, the others replicas whose exchange blocks are copies of view 0's.  The fused launches (wide_chain_kernel, F and
    G form) against one factor_update_kernel launch per view: every view's F and G bitwise, and the owned view's operand
    copies too (seen through the next pass: the T and U blocks it produces)."""
    import torch
    from resnmtf_amd import _lib, sharded
    from resnmtf_amd.engine import Engine
    prob = sharded.local_problem(views, (n, m), k, phi=3.0, xi=2.0, psi=1.5, owned=[0])
    out = {}
    for off in (False, True):
        st = torch.cuda.Stream()
        eng = Engine([n] * views, [m] * views, [k] * views, owned=[v == 0 for v in range(views)], stream=st.cuda_stream,
                     replicate_f=True, replicate_gs=True, no_f_chain=off)
        eng.set_view(0, prob.data[0])
        for v in range(views):
            eng.set_factors(v, prob.init_f[v], prob.init_s[v], prob.init_g[v])
        eng.set_restrictions(prob.phi, prob.xi, prob.psi)
        ri, ci = np.arange(n, dtype=np.int32), np.arange(m, dtype=np.int32)
        for v in range(views):
            for w in range(views):
                if v != w:
                    eng.set_shared_rows(v, w, ri, ri)
                    eng.set_shared_cols(v, w, ci, ci)
        eng.reserve_sweeps(64)
        eng.prepare()
        eng.synchronize()
        ad = sharded.HipEngineAdapter(eng)
        res = []
        steps = (("FBLOCK", (_lib.PHASE_F_ALL, _lib.PHASE_XTF)), ("GBLOCK", (_lib.PHASE_G_ALL, _lib.PHASE_XG)),
                 ("SBLOCK", (_lib.PHASE_S_ALL,)))
        for t in range(2):
            for kind, phases in steps:                 # stand-in for the exchange: every view's block = a copy of view 0's
                blk0 = ad.factor_tensor(0, kind)
                for v in range(1, views):
                    ad.factor_tensor(v, kind).copy_(blk0)
                torch.cuda.synchronize()
                for ph in phases:
                    eng.phase(0, ph, t)
                eng.synchronize()
            res.append([ad.factor_tensor(v, "F").cpu().numpy().copy() for v in range(views)])
            res.append([ad.factor_tensor(v, "G").cpu().numpy().copy() for v in range(views)])
            res.append([ad.factor_tensor(0, kind).cpu().numpy().copy() for kind in ("FBLOCK", "GBLOCK", "SBLOCK")])
        out[off] = res
        ad._views.clear()
        eng.close()
    for a, b in zip(out[False], out[True]):
        for x, y in zip(a, b):
            assert np.isfinite(x).all() and np.abs(x).max() > 0
            assert np.array_equal(x, y)


@pytest.mark.gpu
def test_sharded_hip_replicated_chains_unequal_blocks(tmp_path, monkeypatch):
    """Views whose F exchange blocks differ in size (row counts in different paddings): the S blocks keep their own
    arena and travel on their own (three exchanges per sweep, by broadcasts) -- same results against the oracle."""
    monkeypatch.setenv("RESNMTF_TEST_UNEVEN", "1")
    got = launch("gpu_gs", tmp_path, world=3, k=24)
    assert bool(got["mirrors_ok"])
    ref = oracle_reference_gs(3, k=24)
    np.testing.assert_allclose(got["all_error"], ref["All_Error"], atol=2e-5, rtol=1e-4)
    for v in range(3):
        assert rel_fro(got[f"output_f{v}"], ref["output_f"][v]) < 2e-5
        assert rel_fro(got[f"output_g{v}"], ref["output_g"][v]) < 2e-5
        assert rel_fro(got[f"output_s{v}"], ref["output_s"][v]) < 1e-4


def oracle_reference_gs(world, sweeps=12, k=5):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker
    from oracle import resnmtf_oracle as O
    prob = dist_worker.build_problem_gs(world, k)
    return O.res_nmtf_inner(prob.data, prob.init_f, prob.init_s, prob.init_g, prob.phi, prob.xi, prob.psi,
                            row_names=prob.row_names, col_names=prob.col_names, n_iters=sweeps)


@pytest.mark.parametrize("world", [2, 3])
def test_replicated_chains_schedule_matches_oracle_gloo_cpu(tmp_path, world):
    """phi + psi + xi coupling across ranks, one view per rank: the driver's replicated-chains sweep (F chain, own Xt.F,
    [T blocks], G chain, own X.G, [S blocks], S chain, [U blocks]) with a stand-in engine in exact fp64 reproduces the
    sequential oracle (R/update_steps.r:282-314) to rounding of the re-associated products only."""
    got = launch("cpu_gs", tmp_path, world=world)
    ref = oracle_reference_gs(world)
    np.testing.assert_allclose(got["all_error"], ref["All_Error"], rtol=1e-10, atol=1e-12)
    for v in range(world):
        assert rel_fro(got[f"output_f{v}"], ref["output_f"][v]) < 1e-11
        assert rel_fro(got[f"output_g{v}"], ref["output_g"][v]) < 1e-11
        assert rel_fro(got[f"output_s{v}"], ref["output_s"][v]) < 1e-11
        assert np.array_equal(got[f"row_clusters{v}"], ref["row_clusters"][v])
        assert np.array_equal(got[f"col_clusters{v}"], ref["col_clusters"][v])


@pytest.mark.gpu
@pytest.mark.parametrize("world,k", [(2, 5), (3, 5), (4, 5), (2, 24), (4, 32), (3, 40), (2, 64), (3, 57)])
def test_sharded_hip_replicated_g_and_s_chains(tmp_path, world, k):
    """The same layout with the real HIP engine (ranks share the one GPU of the box, gloo): results against the oracle,
    and every rank's copy of every F, G and S bitwise the owner's.  k <= 16: hand-off mode A; k = 24 / 40 / 64: mode B
    (the k x k job of the last-arriving aux workgroup publishes the S block), the wide bf16-piece passes and -- k = 24 / 32 / 57 / 64 --
    the fused chain launches (wide_chain_kernel; k = 40: its KP = 48 instantiation)."""
    got = launch("gpu_gs", tmp_path, world=world, k=k)
    assert bool(got["mirrors_ok"])
    ref = oracle_reference_gs(world, k=k)
    np.testing.assert_allclose(got["all_error"], ref["All_Error"], atol=2e-5, rtol=1e-4)
    for v in range(world):
        assert rel_fro(got[f"output_f{v}"], ref["output_f"][v]) < 2e-5
        assert rel_fro(got[f"output_g{v}"], ref["output_g"][v]) < 2e-5
        assert rel_fro(got[f"output_s{v}"], ref["output_s"][v]) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("k", [7, 40])
def test_replicated_chains_one_rank_on_rccl(tmp_path, k):
    """The replicated-chains layout with RCCL as the backend (one rank: RCCL refuses two ranks on one GPU): the in-place
    all_gather_into_tensor calls over the library's F / G / S block arenas, the block phases in between, results against
    the oracle."""
    got = launch("gpu_gs_rccl1", tmp_path, world=1, k=k)
    assert bool(got["same"])


# ---------------------------------------------------------------------------------------------------------------------
# Row-sliced chains (resnmtf_options.slice_chains): rank r walks the F (G) chain of every view on row (column) slice r
# ---------------------------------------------------------------------------------------------------------------------
def oracle_reference_slice(world, sweeps=12, k=5, n=96, m=72, max_iters=None, tol=1e-6):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker
    from oracle import resnmtf_oracle as O
    prob = dist_worker.build_problem_slice(world, k, n, m)
    return O.res_nmtf_inner(prob.data, prob.init_f, prob.init_s, prob.init_g, prob.phi, prob.xi, prob.psi,
                            row_names=prob.row_names, col_names=prob.col_names, n_iters=None if max_iters else sweeps,
                            max_iters=max_iters, tol=tol)


def test_sliceable_layouts():
    """Host logic: which layouts can be row-sliced (one view per rank, equal shapes, identical names among coupled views)."""
    from resnmtf_amd import sharded
    prob = sharded.local_problem(3, (64, 48), 4, phi=1.0, psi=1.0, owned=[])
    assert sharded.sliceable(prob, [0, 1, 2], 3)
    assert not sharded.sliceable(prob, [0, 1, 1], 3) and not sharded.sliceable(prob, [0, 1, 2], 4)
    prob.row_names[1] = list(reversed(prob.row_names[1]))            # same names, other order: rows of different slices couple
    assert not sharded.sliceable(prob, [0, 1, 2], 3)
    uneq = sharded.local_problem(2, [(64, 48), (64, 40)], 4, phi=1.0, owned=[])
    assert not sharded.sliceable(uneq, [0, 1], 2)


@pytest.mark.parametrize("world,n,m", [(2, 96, 72), (3, 96, 72), (4, 26, 50), (8, 200, 72)])
def test_sliced_chains_schedule_matches_oracle_gloo_cpu(tmp_path, world, n, m):
    """The driver's sliced sweep (F chain on my rows, [new F rows], own Xt.F, [T slices], G chain on my columns, [new G
    rows], own X.G, [S blocks] S chain || [U slices]; one all-to-all per exchange, a final collect of the fp64 slices) with
    a stand-in engine in exact fp64 reproduces the sequential oracle (R/update_steps.r:282-314) to rounding of the
    re-associated products only.  (4 ranks on 26 rows: the last rank's slice is EMPTY.)"""
    got = launch("cpu_slice", tmp_path, world=world, extra=("--n", n, "--m", m))
    ref = oracle_reference_slice(world, n=n, m=m)
    np.testing.assert_allclose(got["all_error"], ref["All_Error"], rtol=1e-10, atol=1e-12)
    for v in range(world):
        assert rel_fro(got[f"output_f{v}"], ref["output_f"][v]) < 1e-11
        assert rel_fro(got[f"output_g{v}"], ref["output_g"][v]) < 1e-11
        assert rel_fro(got[f"output_s{v}"], ref["output_s"][v]) < 1e-11
        assert np.array_equal(got[f"row_clusters{v}"], ref["row_clusters"][v])
        assert np.array_equal(got[f"col_clusters{v}"], ref["col_clusters"][v])


def test_sliced_convergence_mode_gloo_cpu(tmp_path):
    """The reference's default loop (R/main.r:50-81) in the view-sharded path: every rank evaluates the stop test on the
    full per-view error table it holds, so all ranks stop on the same sweep without a collective -- here with the stand-in
    engine (trace-form errors for the test, as the library), against the oracle's own convergence run."""
    got = launch("cpu_slice_conv", tmp_path, world=3, sweeps=400, extra=("--tol", 1e-5))
    ref = oracle_reference_slice(3, max_iters=400, tol=1e-5)
    assert bool(got["same_stop"])
    assert abs(int(got["sweeps_done"]) - len(ref["All_Error"])) <= 1 and int(got["sweeps_done"]) < 400
    nn = min(len(got["all_error"]), len(ref["All_Error"]))
    np.testing.assert_allclose(got["all_error"][:nn], ref["All_Error"][:nn], rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("world,k,n,m", [(2, 5, 333, 161), (3, 16, 700, 200), (4, 32, 450, 130), (3, 40, 301, 97), (2, 64, 1000, 330),
                                         (4, 57, 90, 64)])
def test_sharded_hip_sliced_chains(tmp_path, world, k, n, m):
    """Row-sliced chains with the real HIP engine (ranks share the one GPU of the box, gloo; all-to-alls staged through the
    host): results against the oracle, and every owner's F, G, S and the error trace BITWISE those of the replicated-chains
    layout in the same hand-off mode -- the slice walks the same instruction sequence on the same bytes.  Ragged row counts
    (slices of different length; (4, 57, 90, 64): rank 3's row slice is EMPTY), k = 5 ... 64 (all four KP instantiations)."""
    got = launch("gpu_slice", tmp_path, world=world, k=k, extra=("--n", n, "--m", m))
    assert bool(got["bitwise_vs_replicated"])
    ref = oracle_reference_slice(world, k=k, n=n, m=m)
    np.testing.assert_allclose(got["all_error"], ref["All_Error"], atol=2e-5, rtol=1e-4)
    for v in range(world):
        assert rel_fro(got[f"output_f{v}"], ref["output_f"][v]) < 2e-5
        assert rel_fro(got[f"output_g{v}"], ref["output_g"][v]) < 2e-5
        assert rel_fro(got[f"output_s{v}"], ref["output_s"][v]) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("world,k,n,m,force", [(3, 16, 700, 200, True), (3, 40, 301, 97, True), (2, 64, 1000, 330, True), (4, 57, 90, 64, True),
                                               (5, 64, 400, 150, False), (5, 48, 333, 97, False)])
def test_sharded_hip_sliced_chains_two_launch_form(tmp_path, world, k, n, m, force, monkeypatch):
    """The sliced chains as slice_products_kernel (every (16-row group, pair of views) a workgroup of its own) +
    slice_walk_kernel (the element-wise chain): chosen by the library for more than four views at k > 32 (the last two cases:
    five ranks on the one GPU -- with the test runner's own process the six the box allows), forced here for the smaller layouts (RESNMTF_SLICE_FUSED=0).  Bitwise the replicated
    chains, like the one-launch form; odd view counts, ragged and empty slices."""
    if force:
        monkeypatch.setenv("RESNMTF_SLICE_FUSED", "0")
    got = launch("gpu_slice", tmp_path, world=world, k=k, sweeps=10, extra=("--n", n, "--m", m), timeout=900)
    assert bool(got["bitwise_vs_replicated"])
    ref = oracle_reference_slice(world, sweeps=10, k=k, n=n, m=m)
    np.testing.assert_allclose(got["all_error"], ref["All_Error"], atol=2e-5, rtol=1e-4)
    for v in range(world):
        assert rel_fro(got[f"output_f{v}"], ref["output_f"][v]) < 2e-5
        assert rel_fro(got[f"output_g{v}"], ref["output_g"][v]) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("mode,world,k", [("gpu_slice_conv", 3, 24), ("gpu_slice_conv", 2, 64), ("gpu_gs_conv", 3, 7)])
def test_sharded_hip_convergence_mode(tmp_path, mode, world, k):
    """Convergence mode (R/main.r:50-81) in the view-sharded layouts with a replicated S chain: the stop test runs on the
    device in s_chain_kernel, every rank stops on the same sweep, within +-2 sweeps of the oracle (the plateau of the
    error difference around 1e-6 is flat: DESIGN.md section 5)."""
    got = launch(mode, tmp_path, world=world, k=k, sweeps=600, extra=("--n", 333, "--m", 161, "--tol", 1e-6))
    ref = oracle_reference_slice(world, k=k, n=333, m=161, max_iters=600)
    assert bool(got["same_stop"])
    done = int(got["sweeps_done"])
    assert done < 600 and abs(done - len(ref["All_Error"])) <= 2, (done, len(ref["All_Error"]))
    assert len(got["all_error"]) == done
    nn = min(done, len(ref["All_Error"]))
    np.testing.assert_allclose(got["all_error"][:nn], ref["All_Error"][:nn], atol=2e-5, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [7, 40])
def test_sliced_chains_one_rank_on_rccl(tmp_path, k):
    """The sliced layout with RCCL as the backend (one rank: RCCL refuses two ranks on one GPU): all_to_all_single over the
    library's exchange buffers, the U slices on the second communicator and stream beside the S chain, the final collect."""
    got = launch("gpu_slice_rccl1", tmp_path, world=1, k=k, extra=("--n", 512, "--m", 192))
    assert bool(got["same"])


@pytest.mark.gpu
def test_replicated_gs_graph_chunk_one_rank_rccl(tmp_path):
    """graph_chunk with the replicated G / S chains: the S chain takes the sweep index from its device counters, so a
    replayed capture writes the errors of the sweep it is in -- bitwise the eager loop's error trace and results."""
    got = launch("gpu_gs_graph1", tmp_path, world=1, sweeps=19, k=7)
    assert bool(got["same"])
    assert len(got["all_error"]) == 19 and np.isfinite(got["all_error"]).all() and got["all_error"][-1] != got["all_error"][0]


@pytest.mark.gpu
@pytest.mark.parametrize("world,k,n,m", [(2, 16, 700, 200), (3, 40, 301, 97), (4, 64, 1000, 330), (4, 57, 90, 64), (4, 32, 20000, 4000)])
def test_sharded_hip_sliced_chains_peer_stores(tmp_path, world, k, n, m):
    """resnmtf_options.slice_p2p: the four exchanges of a sliced sweep as PEER STORES into the receiving rank's buffers (mapped
    through hipIpc -- here between processes on one GPU; xGMI peer access across GPUs) ordered by arrival counters the consumer's
    stream waits on (hipStreamWaitValue32): no collective in the sweep.  Results against the oracle and BITWISE those of the
    all-to-all exchange; single receive buffers, ragged and empty slices; the last case is BASELINE's c4 at full size (4 views
    20000 x 4000, k = 32, phi + psi + xi), four ranks on the one GPU."""
    sweeps = 6 if n >= 10000 else 14
    got = launch("gpu_slice_p2p", tmp_path, world=world, k=k, sweeps=sweeps, extra=("--n", n, "--m", m), timeout=900)
    assert bool(got["bitwise_vs_replicated"])
    ref = oracle_reference_slice(world, sweeps=sweeps, k=k, n=n, m=m)
    np.testing.assert_allclose(got["all_error"], ref["All_Error"], atol=2e-5, rtol=1e-4)
    for v in range(world):
        assert rel_fro(got[f"output_f{v}"], ref["output_f"][v]) < 2e-5
        assert rel_fro(got[f"output_g{v}"], ref["output_g"][v]) < 2e-5
        assert rel_fro(got[f"output_s{v}"], ref["output_s"][v]) < 1e-4


@pytest.mark.gpu
def test_sharded_hip_peer_stores_convergence_mode(tmp_path):
    """Convergence mode with the peer-store exchange: the arrival counters keep counting through the sweeps enqueued after
    the stop test fired (their kernels return at once, their signals still go out), all ranks stop on the same sweep."""
    got = launch("gpu_slice_p2p_conv", tmp_path, world=3, k=24, sweeps=600, extra=("--n", 333, "--m", 161, "--tol", 1e-6))
    ref = oracle_reference_slice(3, k=24, n=333, m=161, max_iters=600)
    assert bool(got["same_stop"])
    done = int(got["sweeps_done"])
    assert done < 600 and abs(done - len(ref["All_Error"])) <= 2, (done, len(ref["All_Error"]))


# ---------------------------------------------------------------------------------------------------------------------
# slice_p2p without slice_chains: the exchange blocks of the REPLICATED layouts by peer stores + stream-waited arrival counters
@pytest.mark.gpu
@pytest.mark.parametrize("world,k,uneven", [(2, 5, False), (3, 24, False), (4, 32, False), (3, 57, True), (2, 64, False)])
def test_sharded_hip_replicated_chains_peer_stores(tmp_path, world, k, uneven, monkeypatch):
    """F, G and S chains replicated (phi + psi + xi across ranks, rows / columns shared in part and at different positions --
    a layout the sliced chains do not cover): the T blocks behind the Xt.F pass and the U rows + S block behind the X.G pass
    are stored straight into every peer's arena, G_ALL / S_ALL wait for their V arrivals.  Bitwise the all-gather form, every
    rank's copies bitwise the owner's, results against the oracle.  uneven: F blocks of different sizes (S blocks apart)."""
    if uneven:
        monkeypatch.setenv("RESNMTF_TEST_UNEVEN", "1")
    got = launch("gpu_block_p2p", tmp_path, world=world, k=k, sweeps=14)
    assert bool(got["bitwise_vs_collectives"]) and bool(got["mirrors_ok"])
    ref = oracle_reference_gs(world, sweeps=14, k=k)
    np.testing.assert_allclose(got["all_error"], ref["All_Error"], atol=2e-5, rtol=1e-4)
    for v in range(world):
        assert rel_fro(got[f"output_f{v}"], ref["output_f"][v]) < 2e-5
        assert rel_fro(got[f"output_g{v}"], ref["output_g"][v]) < 2e-5
        assert rel_fro(got[f"output_s{v}"], ref["output_s"][v]) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("world,k", [(2, 5), (3, 6), (4, 16)])
def test_sharded_hip_replicated_f_chain_peer_stores(tmp_path, world, k):
    """phi only, different column counts (BASELINE c3's layout): one exchange per sweep -- the own F block (U rows, F
    coefficients, lambda) stored to the peers behind the X.G pass, after every rank has acknowledged reading the previous
    one.  Bitwise the all-gather form; against the oracle.  Odd k: rows in the same order (fused F chain), even: permuted."""
    got = launch("gpu_block_p2p_f", tmp_path, world=world, k=k, sweeps=15, xi=0.0)
    assert bool(got["bitwise_vs_collectives"]) and bool(got["mirrors_ok"])
    ref = oracle_reference_one_view_per_rank(world, sweeps=15, identity=(k % 2 == 1), xi=0.0, k=k)
    np.testing.assert_allclose(got["all_error"], ref["All_Error"], atol=2e-5, rtol=1e-4)
    for v in range(world):
        assert rel_fro(got[f"output_f{v}"], ref["output_f"][v]) < 2e-5
        assert rel_fro(got[f"output_g{v}"], ref["output_g"][v]) < 2e-5


@pytest.mark.gpu
def test_sharded_hip_replicated_chains_peer_stores_convergence(tmp_path):
    got = launch("gpu_block_p2p_conv", tmp_path, world=3, k=7, sweeps=600, extra=("--tol", 1e-6))
    assert bool(got["same_stop"]) and bool(got["mirrors_ok"])
    assert int(got["sweeps_done"]) < 600


@pytest.mark.gpu
@pytest.mark.parametrize("k", [6, 7])
def test_sharded_peer_stores_auto_falls_back_on_every_rank(tmp_path, k):
    """ShardedSweep.create(slice_p2p="auto"): when mapping a peer's buffers (k = 6) or the library's self-test (k = 7) fails on
    ONE rank (injected), every rank closes its peer-store engine and builds the same layout with its collectives -- same
    decision everywhere, nobody left waiting, results unchanged."""
    got = launch("gpu_block_p2p_fallback", tmp_path, world=3, k=k, sweeps=9)
    assert bool(got["bitwise_vs_collectives"]) and bool(got["mirrors_ok"])


@pytest.mark.gpu
@pytest.mark.parametrize("mode,world,k,extra", [("gpu_block_p2p_f", 2, 5, ()), ("gpu_block_p2p", 3, 24, ()),
                                                ("gpu_slice_p2p", 3, 40, ("--n", 301, "--m", 97))])
def test_sharded_peer_stores_replayed_from_graphs(tmp_path, mode, world, k, extra):
    """resnmtf_options.slice_p2p = 2: the waits of the phases are one-wave kernels (device counters number them), so the sweeps --
    stores, signals and waits included -- are captured once and replayed (3-sweep graphs + eager remainders here).  Bitwise
    the eager collective form, in each of the three peer-store layouts."""
    got = launch(mode, tmp_path, world=world, k=k, sweeps=14, xi=(0.0 if mode.endswith("_f") else 0.4), extra=("--graph", 3, *extra))
    key = "bitwise_vs_replicated" if mode == "gpu_slice_p2p" else "bitwise_vs_collectives"
    assert bool(got[key])


# ---------------------------------------------------------------------------------------------------------------------
# sharded.res_nmtf_inner: the reference's entry point (R/main.r:32-140) over the ranks of a process group
def _oracle_inner(prob, **kw):
    from oracle import resnmtf_oracle as O
    return O.res_nmtf_inner(prob.data, prob.init_f, prob.init_s, prob.init_g, prob.phi, prob.xi, prob.psi,
                            row_names=prob.row_names, col_names=prob.col_names, **kw)


@pytest.mark.parametrize("mode", ["cpu_api", "cpu_api_conv"])
def test_sharded_res_nmtf_inner_gloo_cpu(tmp_path, mode):
    """Fixed sweeps and the convergence loop (here sweep by sweep: the layout has no replicated S chain) with the exact-fp64
    stand-in engine, two ranks: the reference's return value -- factors after normalisation_check, cluster matrices, All_Error,
    Error (last value / mean of the last ten, R/main.r:127-129) -- equals the sequential oracle's."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker
    conv = mode.endswith("_conv")
    got = launch(mode, tmp_path, world=2, sweeps=(400 if conv else 12), extra=("--tol", 1e-5))
    ref = _oracle_inner(dist_worker.build_problem(), **({"tol": 1e-5, "max_iters": 400} if conv else {"n_iters": 12}))
    assert len(got["all_error"]) == len(ref["All_Error"]) and (not conv or len(ref["All_Error"]) < 400)
    np.testing.assert_allclose(got["all_error"], ref["All_Error"], rtol=1e-10, atol=1e-12)
    want_error = float(np.mean(ref["All_Error"][-10:])) if conv else float(ref["All_Error"][-1])
    assert abs(float(got["Error"]) - want_error) < 1e-12
    for v in range(3):
        assert rel_fro(got[f"output_f{v}"], ref["output_f"][v]) < 1e-11
        assert rel_fro(got[f"output_g{v}"], ref["output_g"][v]) < 1e-11
        assert np.array_equal(got[f"row_clusters{v}"], ref["row_clusters"][v])
        assert np.array_equal(got[f"col_clusters{v}"], ref["col_clusters"][v])


@pytest.mark.gpu
@pytest.mark.parametrize("mode,world,k", [("gpu_api", 3, 7), ("gpu_api_conv", 2, 24)])
def test_sharded_res_nmtf_inner_hip(tmp_path, mode, world, k):
    """The same entry point with the HIP engines (ranks share the one GPU; exchange form chosen by the self-test): against the
    oracle, fixed sweeps and the device-side convergence loop."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker
    conv = mode.endswith("_conv")
    got = launch(mode, tmp_path, world=world, k=k, sweeps=(600 if conv else 12), extra=("--tol", 1e-6))
    ref = _oracle_inner(dist_worker.build_problem_gs(world, k), **({"tol": 1e-6, "max_iters": 600} if conv else {"n_iters": 12}))
    if conv:
        assert abs(len(got["all_error"]) - len(ref["All_Error"])) <= 2 and len(got["all_error"]) < 600
    else:
        np.testing.assert_allclose(got["all_error"], ref["All_Error"], atol=2e-5, rtol=1e-4)
        for v in range(world):
            assert rel_fro(got[f"output_f{v}"], ref["output_f"][v]) < 2e-5
            assert rel_fro(got[f"output_g{v}"], ref["output_g"][v]) < 2e-5
