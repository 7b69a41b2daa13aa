"""Error behaviour of the C-ABI on a real device: misuse is refused with a code and a message, never
a crash or a silent wrong answer (return conventions of include/resnmtf_hip.h)."""
import numpy as np
import pytest

from resnmtf_amd import _lib, synth
from resnmtf_amd.engine import Engine, ResnmtfError

pytestmark = pytest.mark.gpu


def test_create_rejects_bad_descriptions():
    for args in (([10], [8], [65]), ([10], [8], [0]), ([10], [8], [9]), ([5] * 18, [5] * 18, [2] * 18)):
        with pytest.raises(ResnmtfError) as ei:
            Engine(*args)
        assert ei.value.args[0] == _lib.ERR_INVALID if hasattr(_lib, "ERR_INVALID") else True


def test_call_order_is_enforced():
    prob = synth.make_problem([(60, 40)], 3)
    e = Engine([60], [40], [3])
    with pytest.raises(ResnmtfError, match="set_factors"):
        e.run(2)                                           # no factors yet
    e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
    with pytest.raises(ResnmtfError, match="set_view"):
        e.run(2)                                           # no data yet
    with pytest.raises(ResnmtfError, match="set_view"):
        e.init_svd(0)                                      # the SVD init needs the data
    e.set_view(0, prob.data[0]); e.set_restrictions()
    assert len(e.run(3)) == 3
    with pytest.raises(ResnmtfError):
        e.set_shared_rows(0, 0, np.zeros(1, np.int32), np.zeros(1, np.int32))      # a view with itself
    with pytest.raises(ResnmtfError, match="out of range"):
        e.phase(3, _lib.PHASE_F, 0)                        # view index out of range (checked by the library)
    e.close()


def test_ownership_is_enforced():
    prob = synth.make_problem([(60, 40), (60, 30)], 3, phi=1.0)
    e = Engine([60, 60], [40, 30], [3, 3], owned=[True, False])
    with pytest.raises(ResnmtfError, match="own"):
        e.set_view(1, prob.data[1])                        # not this handle's view
    e.set_view(0, prob.data[0])
    for v in range(2):
        e.set_factors(v, prob.init_f[v], prob.init_s[v], prob.init_g[v])
    e.set_restrictions(prob.phi, prob.xi, prob.psi)
    with pytest.raises(ResnmtfError, match="owns every view"):
        e.run(2)                                           # resnmtf_run needs all views; sharded use goes through phases
    e.reserve_sweeps(8); e.prepare()
    with pytest.raises(ResnmtfError, match="own"):
        e.phase(1, _lib.PHASE_F, 0)                        # no replicate_f: F of view 1 is not computed here
    with pytest.raises(ResnmtfError):
        e.factor_device_ptr(0, _lib.FACTOR_FBLOCK)         # exchange blocks exist only with replicate_f
    e.phase(0, _lib.PHASE_F, 0); e.phase(0, _lib.PHASE_G, 0); e.synchronize()
    assert np.isfinite(e.view_errors(0, 0, 1)).all()
    e.close()


def test_device_copies_check_shapes():
    a = Engine([50], [30], [3]); b = Engine([50], [31], [3]); c = Engine([50], [30], [4])
    x = synth.planted_view(50, 30, 3, 1)
    with pytest.raises(ResnmtfError, match="uploaded"):
        c.copy_view_from(0, a, 0)                          # source has no data yet
    a.set_view(0, x)
    with pytest.raises(ResnmtfError, match="shape"):
        b.copy_view_from(0, a, 0)
    c.copy_view_from(0, a, 0)                              # a different k is fine: only the data travels
    np.testing.assert_array_equal(c.get_view(0), a.get_view(0))
    for e in (a, b, c):
        e.close()


@pytest.mark.gpu
def test_torch_imported_after_the_library_still_finds_the_gpu():
    """PyTorch wheels bundle their own HIP runtime; the library is linked against the system's.  Loading the library first
    used to leave a later `import torch` without a device ("No HIP GPUs are available"): `_lib._pin_hip_runtime` loads
    torch's copy first when torch is installed.  Fresh process: library, five sweeps, THEN torch."""
    import os, subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "load_order_worker.py")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "torch after library: ok True" in out.stdout, out.stdout + out.stderr


def test_forced_row_splits_are_range_checked():
    """pass_splits_xg / pass_splits_xtf (tuning options): a count the launch planner cannot place is refused at create."""
    with pytest.raises(ResnmtfError, match="pass_splits"):
        Engine([300], [200], [24], pass_splits_xtf=20)
    with pytest.raises(ResnmtfError, match="pass_splits"):
        Engine([300], [200], [5], pass_splits_xg=-1)
    Engine([300], [200], [24], pass_splits_xtf=16).close()


def test_sliced_layout_options_are_checked():
    """slice_chains / slice_p2p (round 3): layouts that cannot be sliced are refused at create with a message, the peer-store
    exchange insists on its set-up order, the stop tolerance of the phase API on a replicated S chain."""
    base = dict(replicate_f=True, replicate_gs=True, slice_chains=True, slice_index=0, slice_count=2)
    with pytest.raises(ResnmtfError, match="replicate_f and replicate_gs"):
        Engine([60, 60], [40, 40], [3, 3], owned=[True, False], slice_chains=True, slice_index=0, slice_count=2)
    with pytest.raises(ResnmtfError, match="equal shapes"):
        Engine([60, 60], [40, 30], [3, 3], owned=[True, False], **base)
    with pytest.raises(ResnmtfError, match="exactly one owned view"):
        Engine([60, 60], [40, 40], [3, 3], owned=[True, True], **base)
    with pytest.raises(ResnmtfError, match="slice_count"):
        Engine([60, 60, 60], [40, 40, 40], [3, 3, 3], owned=[True, False, False], **base)
    with pytest.raises(ResnmtfError, match="slice_p2p needs slice_chains or the replicated chains"):
        Engine([60], [40], [3], slice_p2p=True)
    with pytest.raises(ResnmtfError, match="exactly one owned view"):          # block form: one view per rank as well
        Engine([60, 60], [40, 30], [3, 3], replicate_f=True, slice_p2p=True, slice_index=0, slice_count=2)
    with pytest.raises(ResnmtfError, match="slice_count"):
        Engine([60, 60], [40, 30], [3, 3], owned=[True, False], replicate_f=True, slice_p2p=True, slice_index=0, slice_count=3)
    blk = Engine([60, 60], [40, 30], [3, 3], owned=[True, False], replicate_f=True, slice_p2p=True, slice_index=0, slice_count=2)
    with pytest.raises(ResnmtfError, match="import every rank"):
        blk.p2p_selftest(100)                              # nothing mapped yet
    blk.p2p_import(0)
    with pytest.raises(ResnmtfError, match="import every rank"):
        blk.prepare()
    blk.close()
    prob = synth.make_problem([(60, 40), (60, 40)], 3, phi=1.0, psi=1.0)
    e = Engine([60, 60], [40, 40], [3, 3], owned=[True, False], slice_p2p=True, **base)
    assert e.slice_info() == (32, 32)
    e.set_view(0, prob.data[0])
    for v in range(2):
        e.set_factors(v, prob.init_f[v], prob.init_s[v], prob.init_g[v])
    e.set_restrictions(prob.phi, prob.xi, prob.psi)
    idx_r, idx_c = np.arange(60, dtype=np.int32), np.arange(40, dtype=np.int32)
    for v, w in ((0, 1), (1, 0)):
        e.set_shared_rows(v, w, idx_r, idx_r); e.set_shared_cols(v, w, idx_c, idx_c)
    e.reserve_sweeps(8)
    with pytest.raises(ResnmtfError, match="import every rank"):
        e.prepare()                                        # peers not mapped yet
    assert len(e.p2p_export()) == 6 * 64
    e.p2p_import(0)
    with pytest.raises(ResnmtfError, match="already imported"):
        e.p2p_import(0)
    with pytest.raises(ResnmtfError, match="import every rank"):
        e.prepare()                                        # rank 1 still missing
    e.close()
    one = Engine([60], [40], [3])
    with pytest.raises(ResnmtfError, match="replicated S chain"):
        one.set_stop_tolerance(1e-6)                       # resnmtf_run has its own stop test
    one.set_stop_tolerance(-1.0)
    with pytest.raises(ResnmtfError, match="not a slice_chains handle"):
        one.slice_info()
    one.close()


@pytest.mark.parametrize("k", [5, 24, 64])
def test_tuning_options_outside_their_range_never_change_the_answer(k):
    """The tuning fields of resnmtf_options (waves, splits, blocks, pads, modes) with values outside what the library
    documents: a handle is either refused at create or computes what the default handle computes -- against the oracle, so
    that an option which silently mis-sizes a launch (as a forced split count beyond the planner's range once did) shows."""
    from helpers import rel_fro, run_oracle
    n, m = (700, 260) if k == 64 else (300, 200)
    prob = synth.make_problem([(n, m)], k)
    ref = run_oracle(prob, n_iters=6)
    weird = [dict(pass_waves=3), dict(pass_waves=5), dict(pass_waves=16), dict(update_blocks=1), dict(update_blocks=100000),
             dict(target_workgroups=1), dict(target_workgroups=64), dict(target_workgroups=97), dict(target_workgroups=1000000), dict(pass_lds_pad_kb=-3), dict(pass_lds_pad_kb=4096),
             dict(check_every=0), dict(check_every=-5), dict(half_unroll=7), dict(x_half=9), dict(kk_mode=7), dict(kk_mode=-1),
             dict(wait_mode=11), dict(pass_splits_xg=16), dict(pass_splits_xtf=16), dict(pass_splits_xg=1, pass_splits_xtf=1),
             dict(no_pitch_pad=True, pass_waves=4), dict(xcd_order=True), dict(bf16_split=2), dict(bf16_split=5),
             dict(use_graph=False), dict(time_kernels=True), dict(replicate_f=True), dict(replicate_f=True, replicate_gs=True),
             dict(no_f_chain=True), dict(fuse_updates=1), dict(fuse_updates=2), dict(fuse_updates=9), dict(slice_index=3, slice_count=5),
             dict(use_graph=False, check_every=1), dict(target_workgroups=200, pass_splits_xtf=3)]
    refused, wrong = 0, []
    for opts in weird:
        try:
            e = Engine([n], [m], [k], **opts)
        except ResnmtfError:
            refused += 1
            continue
        try:
            e.set_view(0, prob.data[0]); e.set_restrictions()
            e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
            errs = e.run(6)
            f, s, g, _, _ = e.finalise(0)
        except ResnmtfError:               # (e.g. replicate_gs: resnmtf_run says it is the phase API's layout)
            refused += 1
            continue
        finally:
            e.close()
        if not (np.allclose(errs, ref["All_Error"], atol=2e-5, rtol=1e-4) and rel_fro(f, ref["output_f"][0]) < 1e-4
                and rel_fro(g, ref["output_g"][0]) < 1e-4):
            wrong.append(opts)
    assert not wrong, wrong
    assert refused <= 10        # (most out-of-range values fall back to the default; a few are refused with a message)


@pytest.mark.parametrize("k", [7, 40])
def test_tuning_options_on_coupled_views_never_change_the_answer(k):
    """The same question for three views coupled through phi, psi and xi with names shared in part (the coupled update
    kernels, the k x k chains with their S couplings, the fused chains): every accepted option set computes the oracle's answer."""
    from helpers import coupled_problem, rel_fro, run_oracle
    shapes = [(210, 150), (190, 150), (210, 120)]
    prob = coupled_problem(shapes, k, seed=91, phi_w=1.5, psi_w=1.0, xi_w=0.4)
    ref = run_oracle(prob, n_iters=5)
    from resnmtf_amd import naming
    row_sh, col_sh = naming.shared_names(prob.row_names), naming.shared_names(prob.col_names)
    sets = [dict(), dict(update_blocks=1), dict(update_blocks=977), dict(pass_waves=16), dict(target_workgroups=64), dict(no_f_chain=True),
            dict(use_graph=False), dict(kk_mode=1), dict(kk_mode=2), dict(pass_splits_xg=2, pass_splits_xtf=9), dict(time_kernels=True, use_graph=False),
            dict(check_every=1), dict(wait_mode=1), dict(bf16_split=2), dict(xcd_order=True), dict(fuse_updates=1)]
    wrong, accepted = [], 0
    for opts in sets:
        try:
            e = Engine([s[0] for s in shapes], [s[1] for s in shapes], [k] * 3, **opts)
        except ResnmtfError:
            continue
        try:
            for v in range(3):
                e.set_view(v, prob.data[v])
                e.set_factors(v, prob.init_f[v], prob.init_s[v], prob.init_g[v])
            e.set_restrictions(prob.phi, prob.xi, prob.psi)
            for v in range(3):
                for w in range(3):
                    if w != v:
                        e.set_shared_rows(v, w, *naming.index_pairs(prob.row_names[v], prob.row_names[w], row_sh[v].get(w)))
                        e.set_shared_cols(v, w, *naming.index_pairs(prob.col_names[v], prob.col_names[w], col_sh[v].get(w)))
            errs = e.run(5)
            outs = [e.finalise(v) for v in range(3)]
        except ResnmtfError:
            continue
        finally:
            e.close()
        ok = np.allclose(errs, ref["All_Error"], atol=2e-5, rtol=1e-4)
        for v in range(3):
            ok = ok and rel_fro(outs[v][0], ref["output_f"][v]) < 1e-4 and rel_fro(outs[v][2], ref["output_g"][v]) < 1e-4
        accepted += 1
        if not ok:
            wrong.append(opts)
    assert not wrong, wrong
    assert accepted >= 12, accepted      # (the test is about the sets that run)
