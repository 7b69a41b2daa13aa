#!/usr/bin/env python3
"""bench.py -- multiplicative-update iterations/sec of the ResNMTF inner loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one full sweep of update_matrices + calculate_error (R/main.r:84-108) over all
views.  N = 1: BASELINE.json configs[1] (c2: one view 10000 x 2000, k = 16, synthetic planted
blocks).  N = 2 / 4 / 8: BASELINE's c3 / c4 / c5, one view per GPU, one process per
GPU over torch.distributed (RCCL), the reference's Gauss-Seidel order kept exactly.  Started either
by ``python -m torch.distributed.run`` (RANK / WORLD_SIZE in the environment) or plainly as
``python bench.py --gpus N``: the ranks are then started here as CHILD processes, before anything in
this process touches the GPU, and rank 0's JSON line is relayed.

``value`` = view-updates per second = steps x n_views / wall (inputs resident in HBM before the
timed region).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
METRIC = "multiplicative-update iters/sec (all views)"


def cpu_baseline(prob, warm: int = 3, timed: int = 20, budget_s: float = 0.0, note: str = "the same workload") -> dict:
    """The oracle (literal fp64 restatement: four passes over X per sweep, materialised
    residual -- the BLAS call sequence R would issue) on the host cores; NumPy/OpenBLAS.
    ``budget_s`` > 0: the number of timed sweeps is cut so that the sample takes about that long (large views)."""
    from oracle import resnmtf_oracle as O
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    data = prob.data
    cur_f, cur_s, cur_g = list(prob.init_f), list(prob.init_s), list(prob.init_g)
    lam, mu = O.explicit_init_lm(cur_f, cur_g)
    rn, cn = prob.row_names, prob.col_names
    ri, ci = O.reorder_data(rn), O.reorder_data(cn)
    norms = np.array([np.linalg.norm(d, "fro") ** 2 for d in data])

    def sweep():
        nonlocal cur_f, cur_s, cur_g, lam, mu
        cur_f, cur_s, cur_g, lam, mu = O.update_matrices(data, cur_f, cur_s, cur_g, lam, mu, prob.phi, prob.xi,
                                                         prob.psi, ri, ci, rn, cn)
        return O.calculate_error(data, cur_f, cur_s, cur_g, norms)
    t_w = time.perf_counter()
    for _ in range(warm):
        sweep()
    per = (time.perf_counter() - t_w) / max(warm, 1)
    if budget_s > 0.0:
        timed = int(max(2, min(timed, budget_s / max(per, 1e-9))))
    t0 = time.perf_counter()
    for _ in range(timed):
        sweep()
    dt = time.perf_counter() - t0
    return {"value": timed * len(data) / dt, "unit": "view-updates/s", "cores": int(threads), "kind": "port",
            "sample": f"{timed} sweeps (after {warm} warm-up) of {note}, NumPy+OpenBLAS fp64, "
                      f"{threads} threads; R unavailable on the box"}


def pass_roofline(make_engine, steps: int, warmup: int) -> dict:
    """Roofline leg: `steps` sweeps of one c2-shaped view, eager, every streaming-pass launch timed by HIP events
    attached to the dispatch on the library's stream (dominant kernels: the X.G and Xt.F streaming passes)."""
    eng2, _ = make_engine(time_kernels=True)
    if warmup > 0:
        eng2.run(min(warmup, 20))
    eng2.pass_timings(reset=True)
    eng2.run(steps)
    t = eng2.pass_timings()
    eng2.close()
    launches = t["xg_launches"] + t["xtf_launches"]
    pass_ms = (t["xg_ms_total"] + t["xtf_ms_total"]) / max(launches, 1)
    bytes_per_launch = (t["xg_bytes"] * t["xg_launches"] + t["xtf_bytes"] * t["xtf_launches"]) / max(launches, 1)
    achieved = bytes_per_launch / (pass_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": "pass_kernel<NT,NW,UNROLL,IS_XG,MODE_A,SPLIT> (X.G and Xt.F streaming passes: pass_body at k <= 16, "
                                          "pass_body_wide -- three bf16 pieces on the K = 32 MFMA -- above)", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "avg_launch_us": round(pass_ms * 1e3, 3), "launches": int(launches),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "xg_avg_us": round(t["xg_ms_total"] / max(t["xg_launches"], 1) * 1e3, 3),
                "xtf_avg_us": round(t["xtf_ms_total"] / max(t["xtf_launches"], 1) * 1e3, 3),
                "mfma_tflops": round((t["xg_flops"] + t["xtf_flops"]) / 2 / (pass_ms * 1e-3) / 1e12, 2)}
    # `traffic` cannot be measured inside this run (PMC passes need their own rocprofv3 runs): it is the figure of the
    # committed PMC session, labelled as such (profiles/README.md; tools/profile_round.sh regenerates it)
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file):
        try:
            t = json.load(open(traffic_file))
            if abs(float(t.get("algorithmic_bytes_per_launch", 0.0)) - bytes_per_launch) < 1.0:      # same workload only
                roofline["traffic"] = t.get("atb_pass_kernel_bytes_per_launch")
                roofline["traffic_source"] = "static: profiles/traffic.json (" + str(t.get("source", "PMC session")) + ")"
        except Exception:
            pass
    return roofline


def run_single(args) -> dict:
    import torch
    from resnmtf_amd import synth
    from resnmtf_amd.engine import Engine

    prob = synth.config("c2")
    n, m = prob.data[0].shape
    k = prob.k

    def make_engine(**kw):
        if os.environ.get("RESNMTF_NO_GRAPH") == "1":      # profiling runs: plain launches, one row per dispatch
            kw.setdefault("use_graph", False)
        if os.environ.get("RESNMTF_FUSE_UPDATES"):         # A/B (opt-in, measured slower): 1 = updates fused into the consuming pass launch, 2 = without prefetch
            kw.setdefault("fuse_updates", int(os.environ["RESNMTF_FUSE_UPDATES"]))
        e = Engine([n], [m], [k], device_id=0, **kw)
        t0 = time.perf_counter()
        e.set_view(0, prob.data[0])
        up = time.perf_counter() - t0
        e.set_restrictions(None, None, None)
        e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
        return e, up

    eng, upload_s = make_engine()
    t_cold = time.perf_counter()
    # one-off costs of a new handle stay outside the timed region, as a compile cache would: the library captures a
    # hipGraph per run length at the first run of that length (resnmtf_run) -- run the timed length once, put the
    # initial factors back, then the W warm-up sweeps and the K timed ones start from the initial state as always
    eng.run(args.steps)
    torch.cuda.synchronize()
    cold_s = time.perf_counter() - t_cold
    eng.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
    if args.warmup > 0:
        eng.run(args.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    errs = eng.run(args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert len(errs) == args.steps and np.isfinite(errs).all()
    eng.close()

    roofline = pass_roofline(make_engine, args.steps, args.warmup)

    # informational only (never `value`): the same workload with the opt-in, guarded 16-bit integer image of X
    # (resnmtf_options.x_half = 3: half the pass bytes when the image's relative quantisation error is <= 3e-5;
    # F / G then 1e-6 ... 3e-5 from the oracle instead of ~1e-7)
    alt = None
    try:
        eng3, _ = make_engine(x_half=3)
        kind, rel = eng3.view_image_info(0)
        if args.warmup > 0:
            eng3.run(args.warmup)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        errs3 = eng3.run(args.steps)
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t1
        eng3.close()
        alt = {"view_updates_per_s": round(args.steps / dt3, 2), "final_error": float(errs3[-1]),
               "image": {0: "f32 (guard refused)", 1: "fp16", 2: "u16"}.get(kind, str(kind)), "x_rel_quantisation_error": rel,
               "note": "opt-in x_half=3 (guarded uniform 16-bit image of X, f32 MFMA); not the default path, not `value`"}
    except Exception as exc:
        print(f"[bench] x_half=2 leg failed: {exc}", file=sys.stderr)
    cpu = cpu_baseline(prob) if not args.no_cpu_baseline else None
    value = args.steps * 1 / dt
    return {
        "metric": METRIC, "value": round(value, 2), "unit": "view-updates/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "c2: 1 view 10000x2000, k=16, fixed sweeps (BASELINE.json configs[1])",
                   "n_views": 1, "rows": n, "cols": m, "k": k, "final_error": float(errs[-1]),
                   "arithmetic": "fp32 X + f32 MFMA accumulate for X.G / Xt.F; fp64 factors and epilogues",
                   "upload_s_pcie_inclusive": round(upload_s, 4), "opt_in_x_u16": alt,
                   # the timed call replays a hipGraph of exactly `steps` sweeps that an untimed call of the same length
                   # captured before the warm-up (as a compile cache would); a cold first call of that length took:
                   "precaptured_graph": True, "first_run_incl_capture_s": round(cold_s, 5)},
        "roofline": roofline, "cpu_baseline": cpu,
    }


def _all_gather_ints(dist, value: int, world: int):
    out = [None] * world
    dist.all_gather_object(out, int(value))
    return out


def _make_driver(sharded, prob, n_views, rank, world, local_rank, **extra):
    if "slice_chains" in extra and extra["slice_chains"] is None:
        extra.pop("slice_chains")
    return sharded.ShardedSweep.create(prob, owner_of=list(range(n_views)), rank=rank, world=world, device_index=local_rank,
                                replicate_f=("force" if os.environ.get("RESNMTF_FORCE_REPLICATE") == "1" else
                                             os.environ.get("RESNMTF_NO_REPLICATE") != "1"),
                                allgather_blocks=(os.environ.get("RESNMTF_NO_ALLGATHER") != "1"),
                                **({"replicate_gs": False} if os.environ.get("RESNMTF_NO_REPLICATE_GS") == "1" else {}),
                                **({"slice_chains": os.environ["RESNMTF_SLICE_CHAINS"] == "1"}
                                   if ("RESNMTF_SLICE_CHAINS" in os.environ and "slice_chains" not in extra) else {}),
                                **({"overlap_u": False} if os.environ.get("RESNMTF_NO_OVERLAP") == "1" else {}),
                                # the exchanges as peer stores + stream-ordered flags (resnmtf_options.slice_p2p): the caller decides
                                # (run_sharded: after the library's self-test and a bitwise cross-check on this node)
                                **({"slice_p2p": False} if "slice_p2p" not in extra else {}),
                                **({"serial_exchange": os.environ["RESNMTF_SERIAL_EXCHANGE"] == "1"}
                                   if "RESNMTF_SERIAL_EXCHANGE" in os.environ else {}), **extra)


def run_sharded(args) -> dict:
    import torch
    import torch.distributed as dist
    from resnmtf_amd import sharded, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # rehearsal on a one-GPU box: RESNMTF_BENCH_DEVICE pins every rank to one device and
    # RESNMTF_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU); never set by the driver
    if os.environ.get("RESNMTF_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["RESNMTF_BENCH_DEVICE"])
    backend = os.environ.get("RESNMTF_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if backend == "nccl":
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    n_views = world
    # BASELINE.json configs[2..4]: c3 on 2 GPUs, c4 on 4, c5 on 8 (one view per GPU); any other N -- or
    # RESNMTF_BENCH_SAME_SHAPE=1 -- N phi-coupled c2-shaped views (per-GPU work as at N = 1).  RESNMTF_BENCH_SHAPE=n,m,k:
    # N fully coupled views of that shape (rehearsals at small sizes)
    table = {2: ("c3", [(10000, 2000), (10000, 1500)], 16, dict(phi=200.0)),
             4: ("c4", [(20000, 4000)] * 4, 32, dict(phi=200.0, psi=200.0)),
             8: ("c5", [(50000, 8000)] * 8, 64, dict(phi=200.0, xi=200.0, psi=200.0))}
    if os.environ.get("RESNMTF_BENCH_SHAPE"):
        sn, sm, sk = (int(x) for x in os.environ["RESNMTF_BENCH_SHAPE"].split(","))
        cfg_name, shapes, k, coupling = f"{world} x {sn}x{sm}", [(sn, sm)] * world, sk, dict(phi=200.0, xi=200.0, psi=200.0)
    elif world in table and os.environ.get("RESNMTF_BENCH_SAME_SHAPE") != "1":
        cfg_name, shapes, k, coupling = table[world]
    else:
        cfg_name, shapes, k, coupling = f"{world} x c2", [(10000, 2000)] * world, 16, dict(phi=200.0)
    n, m = shapes[rank]
    # every rank builds only the view it owns (same seeds as synth.make_problem) + all initial factors
    prob = sharded.local_problem(n_views, shapes, k, owned=[rank], **coupling)
    # Exchange form.  RESNMTF_P2P=1 / 0 forces peer stores / collectives; default ("auto"), N > 1: peer stores when (a) the
    # layout has them, (b) the library's self-test passes on every rank (probe stores, arrivals, stream wait -- host-side
    # deadlines, nothing can hang) and (c) five sweeps by peer stores give BITWISE the per-view errors of five sweeps of the
    # same layout with its collectives, here, on this node; otherwise the collectives.  The decision is part of the set-up,
    # outside the timed region, and is reported in config.
    want = os.environ.get("RESNMTF_P2P", "auto")
    use_p2p, p2p_note, force_slice = False, "collectives (RESNMTF_P2P=0)", None
    tab_ref = None                                 # per-view errors of the first sweeps by the collective exchange
    # RESNMTF_P2P_GRAPH=1 (opt-in): the waits as one-wave kernels, the sweeps replayed from a captured graph.  Measured with one
    # rank (tools/round3/p2p_one_rank.sh): 52.8 us per sweep replayed against 52.4 us with plain launches and stream waits -- the
    # sweep is not bound by the host's launch sequence, so the default keeps the command-processor waits
    p2p_graph = os.environ.get("RESNMTF_P2P_GRAPH", "0") == "1"
    p2p_chunk = max(c for c in range(1, 33) if args.steps % c == 0) if p2p_graph else 0
    if want == "1":
        use_p2p, p2p_note = True, "peer stores (RESNMTF_P2P=1)"
    elif want == "auto" and world > 1:
        probe = _make_driver(sharded, prob, n_views, rank, world, local_rank, slice_p2p="auto", p2p_graph=p2p_graph)
        if not probe.p2p:
            p2p_note = "collectives (no peer-store form for this layout, or the self-test failed: see stderr)"
            probe.close()
        else:
            force_slice = probe.sliced
            probe.reserve(16); probe.run(5, graph_chunk=(5 if p2p_graph else 0))      # (the form the timed run uses)
            tab_p = probe.view_error_table()
            probe.close()
            ref = _make_driver(sharded, prob, n_views, rank, world, local_rank, slice_p2p=False, slice_chains=force_slice)
            ref.reserve(16); ref.run(5)
            tab_c = tab_ref = ref.view_error_table()
            ref.close()
            use_p2p = bool(np.array_equal(tab_p, tab_c) and np.isfinite(tab_p).all())
            p2p_note = ("peer stores (self-test passed; 5 sweeps bitwise equal to the collective exchange on this node)" if use_p2p else
                        "collectives (peer stores disagreed with the collective exchange on this node)")
    drv = _make_driver(sharded, prob, n_views, rank, world, local_rank, slice_p2p=(True if want == "1" else "auto" if use_p2p else False), slice_chains=force_slice,
                       **({"p2p_graph": True} if (use_p2p and p2p_graph) else {}))
    if use_p2p and not drv.p2p:      # (the self-test runs again for every engine: same outcome on every rank)
        use_p2p, p2p_note = False, "collectives (the self-test failed when the timed driver was built)"
    if os.environ.get("RESNMTF_FORCE_BCAST") == "1":      # rehearsal: issue the F broadcast even with one rank
        drv.plan[0]["F"] = True
    drv.reserve(args.warmup + args.steps + 8)      # per-sweep error slots for both run() calls
    # RESNMTF_SHARDED_GRAPH=K (opt-in, RCCL only): K sweeps incl. their collectives replayed from one captured graph
    chunk = int(os.environ.get("RESNMTF_SHARDED_GRAPH", "0")) if backend == "nccl" else 0
    if use_p2p and p2p_graph:
        chunk = p2p_chunk
    captured = drv.precapture(chunk) if chunk > 0 else False      # (set-up: the timed run only replays)
    drv.run(args.warmup, graph_chunk=chunk)
    drv.collect()
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    drv.run(args.steps, graph_chunk=chunk)
    drv.collect()                                  # (sliced chains: the owners' fp64 F / G whole again -- part of the job)
    dist.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    errs = drv.mean_errors()
    # the timed run started from the same factors as the collective reference of the set-up: its first sweeps must be the same bits
    verified = None
    if use_p2p and tab_ref is not None:
        tab_run = drv.view_error_table()
        verified = bool(len(tab_run) >= len(tab_ref) and np.array_equal(tab_run[:len(tab_ref)], tab_ref))
    replicated = any(drv.replicated)
    allgather = drv.allgather_layout
    replicated_gs = drv.replicate_gs
    sliced = drv.sliced
    per_sweep = drv.collectives_per_sweep
    p2p = drv.p2p
    overlapped = getattr(drv, "_group_u", None) is not None
    n_devices = len({int(d) for d in _all_gather_ints(dist, local_rank, world)})
    drv.close()
    # ---- roofline leg: the same sharded sweep, every rank, eager, every kernel of the library timed by HIP events attached
    # to its dispatch (resnmtf_kernel_timings): the streaming passes (dominant) AND the chain / pack kernels of the layout
    roofline = None
    kt = None
    t_steps = max(2, min(args.steps, 30))
    try:
        drv_t = _make_driver(sharded, prob, n_views, rank, world, local_rank, time_kernels=True, slice_p2p=(True if want == "1" else "auto" if use_p2p else False), slice_chains=force_slice)
        drv_t.reserve(t_steps + 8)
        drv_t.run(2)
        torch.cuda.synchronize()
        drv_t.engine.kernel_timings(reset=True); drv_t.engine.e.pass_timings(reset=True)
        drv_t.run(t_steps)
        torch.cuda.synchronize()
        kt = drv_t.engine.kernel_timings()
        pt = drv_t.engine.e.pass_timings()
        drv_t.close()
    except Exception as exc:          # never lose the bench line to the extra leg
        print(f"[bench] rank {rank}: timed leg failed: {exc}", file=sys.stderr)
    single = None
    if rank == 0 and kt is not None:
        launches = pt["xg_launches"] + pt["xtf_launches"]
        pass_ms = (pt["xg_ms_total"] + pt["xtf_ms_total"]) / max(launches, 1)
        bytes_per_launch = (pt["xg_bytes"] * pt["xg_launches"] + pt["xtf_bytes"] * pt["xtf_launches"]) / max(launches, 1)
        achieved = bytes_per_launch / (pass_ms * 1e-3) / 1e9
        per_kind = {name: round(ms / t_steps * 1e3, 2) for name, (ms, cnt) in kt.items() if cnt}
        roofline = {"bound": "hbm", "kernel": "pass_kernel (X.G and Xt.F streaming passes of rank 0's view inside the SHARDED sweep)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": None, "avg_launch_us": round(pass_ms * 1e3, 3), "launches": int(launches),
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                    "rank0_kernel_us_per_sweep": per_kind, "rank0_kernel_us_per_sweep_total": round(sum(per_kind.values()), 2),
                    "note": "HIP events on every library launch of rank 0 during a separate eager leg of the same sharded sweep "
                            f"({t_steps} sweeps); f_chain / g_chain / s_chain / pack = the layout's chain, fold / slice kernels"}
    dist.barrier()
    if rank == 0:      # the one-view rate of rank 0's shape (the denominator of a retention figure), others wait
        try:
            from resnmtf_amd.engine import Engine
            e1 = Engine([n], [m], [k], device_id=local_rank)
            e1.set_view(0, prob.data[0]); e1.set_restrictions(None, None, None)
            e1.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
            e1.run(min(args.steps, 50)); e1.run(3)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            e1.run(min(args.steps, 50))
            torch.cuda.synchronize()
            single = round(min(args.steps, 50) / (time.perf_counter() - t1), 2)
            e1.close()
            if roofline is not None:
                roofline["single_view_updates_per_s"] = single
        except Exception as exc:
            print(f"[bench] single-view leg failed: {exc}", file=sys.stderr)
    dist.barrier()
    dist.destroy_process_group()
    if rank != 0:
        return {}
    # CPU baseline at N > 1: ONE view of the workload (rank 0's, uncoupled) on the host cores, a bounded sample;
    # the other ranks have left, every core is free
    cpu = None
    if not args.no_cpu_baseline:
        try:
            one = synth.Problem([prob.data[0]], [prob.init_f[0]], [prob.init_s[0]], [prob.init_g[0]], np.zeros((1, 1)), np.zeros((1, 1)),
                                np.zeros((1, 1)), k, "one view")
            one.row_names = [prob.row_names[0]]; one.col_names = [prob.col_names[0]]
            cpu = cpu_baseline(one, warm=1, timed=20, budget_s=15.0,
                               note=f"ONE of the {n_views} views ({n}x{m}, k={k}; coupling terms left out: O(n k) beside the four X passes)")
        except Exception as exc:
            print(f"[bench] cpu baseline failed: {exc}", file=sys.stderr)
    layout = (("F and G chains ROW-SLICED over the ranks, S chain replicated; exchange by peer stores into the receivers' buffers (hipIpc / xGMI) "
               "ordered by stream-waited arrival counters, no collective in the sweep")
              if (sliced and per_sweep == 0) else
              (("F, G and S chains replicated on every rank" if replicated_gs else "F chain replicated on every rank") +
               "; the exchange blocks stored straight into every peer's arena (hipIpc / xGMI), ordered by stream-waited arrival counters, "
               "no collective in the sweep")
              if (per_sweep == 0 and p2p) else
              ("F and G chains ROW-SLICED over the ranks (all-to-all of row slices), S chain replicated; "
               f"{per_sweep} collectives per sweep between dependent steps" + (", U slices on a second communicator beside the S chain" if overlapped else ""))
              if sliced else
              ("F, G and S chains replicated on every rank, " + ("two" if per_sweep == 2 else "three") + " block all-gathers per sweep") if replicated_gs else
              ("F chain replicated on every rank (its inputs: one all-gather per sweep)" if allgather else
               "F chain replicated on every rank (its inputs broadcast once per sweep)") if replicated else
              "F exchanged by ordered broadcasts")
    return {
        "metric": METRIC, "value": round(args.steps * n_views / dt, 2), "unit": "view-updates/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{cfg_name}: {n_views} views " + ",".join(f"{a}x{b}" for a, b in sorted(set(shapes), reverse=True))
                               + f", k={k}, " + "+".join(f"{key}={val:g}" for key, val in coupling.items())
                               + " (all rows / columns of coupled views shared), one view per GPU, Gauss-Seidel order kept exactly; " + layout,
                   "n_views": n_views, "shapes": [list(sh) for sh in shapes], "k": k,
                   "backend": ("rccl" if backend == "nccl" else backend), "world_size": world, "distinct_devices": n_devices,
                   "exchange": p2p_note, "timed_run_first_sweeps_equal_collective_reference": verified, "sweeps_per_graph_replay": (chunk if captured else 0),
                   "final_error": float(errs[-1]) if len(errs) else None,
                   "scaling_note": "BASELINE.json prescribes a different workload per GPU count (c3 / c4 / c5): compare value / n_gpus "
                                   "with the one-view rate of the same shape (roofline.single_view_updates_per_s), not across N"},
        "roofline": roofline, "cpu_baseline": cpu,
    }


def spawn_ranks(args) -> int:
    """``python bench.py --gpus N`` without a launcher: start the N ranks as child processes (one per GPU, rendezvous on
    127.0.0.1) BEFORE this process has touched the GPU -- it never does -- and relay rank 0's JSON line.  No exec."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    try:
        out, _ = procs[0].communicate(timeout=float(os.environ.get("RESNMTF_BENCH_DEADLINE_S", "1500")))
    except subprocess.TimeoutExpired:      # a rank that never comes back must not hold the caller for ever
        print("[bench] rank 0 did not finish in time: stopping the ranks this process started", file=sys.stderr)
        for p in procs:
            p.kill()
        out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        try:
            p.wait(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()                     # (the exact child this process started)
            p.wait()
        rc = rc or p.returncode
    line = None
    for ln in (out or b"").decode(errors="replace").splitlines():
        if ln.startswith("{"):
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if line:
        print(line, flush=True)
    return rc if rc else (0 if line else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    # RESNMTF_FORCE_SHARDED=1 exercises the multi-rank code path (nccl init, ordered exchange driver,
    # barrier/max timing) with a single rank -- a rehearsal on the one-GPU box
    sharded_path = args.gpus > 1 or os.environ.get("RESNMTF_FORCE_SHARDED") == "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # plain `python bench.py --gpus N`: this process only starts the ranks
        raise SystemExit(spawn_ranks(args))
    out = run_sharded(args) if sharded_path else run_single(args)
    if out:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
