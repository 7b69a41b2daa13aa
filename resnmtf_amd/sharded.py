"""View-sharded sweep: one process per GPU, views placed on ranks, coupled factor blocks
exchanged through ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU node, "gloo"
in the CPU tests) in the reference's Gauss-Seidel order.

The reference updates views strictly in index order inside a sweep and every coupling term reads
the RUNNING lists (``R/update_steps.r:282-314``): view v sees this sweep's F/G/S of views w < v and
the previous sweep's of views w > v.  A one-shot all-gather would change that (Jacobi), so the
exchange is a sequence of ordered broadcasts, one per (view, factor) that some other rank reads:

    for v in views:                      # every rank walks the same list
        owner(v): PHASE_F(v)             # update_f
        broadcast F_v        if another rank's view is phi-coupled to v through shared rows
        owner(v): PHASE_G(v)             # Xt.F pass, update_g, X.G pass (+ update_s, update_lm, error)
        broadcast G_v        if psi-coupled through shared columns
        broadcast S_v        if xi-coupled

Only the small factor blocks (n x k, m x k, k x k) travel; the streaming passes over X -- all
of the HBM traffic -- touch own-view data only.  Uncoupled views need no exchange at all.

Streams (GPU path): the engine's kernels run on a dedicated compute stream, the broadcasts on a
second (exchange) stream, ordered against each other by HIP events only where the data demands it:

    broadcast of F_v      waits for the latest PHASE_F enqueued on this rank  (root: F_v is complete;
                          receiver: nobody still reads the old mirror of F_v)
    broadcast of G_v/S_v  waits for the latest PHASE_G/PHASE_S enqueued on this rank  (same two reasons)
    PHASE_F / PHASE_G     wait for the latest broadcast enqueued before them

so a rank's two streaming passes (PHASE_G, all of the HBM traffic) overlap the other ranks' F
updates and broadcasts of the same sweep; only the F chain itself -- which the reference's
Gauss-Seidel order makes serial -- stays on the critical path.

Replicated F chain (``replicate_f``, default with the HIP engine).  With phi coupling the F updates
of a sweep form a chain (F_v' reads F_w' of every coupled w < v), so broadcasting each F_v' puts N
serial broadcast latencies on the critical path of every sweep.  Instead every rank keeps, for
every view of a phi-coupled component that spans several ranks, the INPUTS of its F update (the
X.G slabs, the k x k coefficient matrices, lambda: one contiguous exchange block) and runs that
update itself -- the same kernel on the same bytes, hence bitwise the same F everywhere.  The block
of view v is broadcast once per sweep after the owner's PHASE_G(v), off the chain; what remains
on the critical path of a rank is N short F-update kernels plus its own two passes.  The block is
compact: X.G folded into one f32 slab + two k x k matrices + lambda (0.64 MB at 10000 x 16).  When every
rank owns exactly one view, all views are replicated and their blocks have one size (the weak-scaling
layout of bench.py), the N broadcasts of a sweep become ONE in-place all-gather over the library's
block arena (``RESNMTF_FACTOR_FBLOCK_ALL``) at the end of the sweep: one collective launch and one host
call instead of N, and every xGMI link carries a block at the same time.  In that layout the F updates of a sweep are also hoisted
to its start as one ``RESNMTF_PHASE_F_ALL`` (F_w' reads neither G nor S of the same sweep): one kernel launch
walks the whole chain when the views share their rows in the same order.

The engine is injected (``engine`` argument) so that the identical driver code runs in the CPU
tests on a stand-in engine (no streams); in production it is ``resnmtf_amd.engine.Engine``.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from . import naming
from ._lib import (FACTOR_F, FACTOR_FBLOCK, FACTOR_FBLOCK_ALL, FACTOR_FNEW_RECV, FACTOR_FNEW_SEND, FACTOR_G, FACTOR_GBLOCK,
                   FACTOR_GBLOCK_ALL, FACTOR_GNEW_RECV, FACTOR_GNEW_SEND, FACTOR_S, FACTOR_SBLOCK, FACTOR_SBLOCK_ALL,
                   FACTOR_T_RECV, FACTOR_T_SEND, FACTOR_U_RECV, FACTOR_U_SEND, PHASE_F, PHASE_F_ALL, PHASE_G, PHASE_G_ALL,
                   PHASE_LOCAL_SWEEP, PHASE_S, PHASE_S_ALL, PHASE_SLICE_F, PHASE_SLICE_G, PHASE_SLICE_XG, PHASE_SLICE_XTF,
                   PHASE_XG, PHASE_XTF)
from .synth import Problem, planted_view, random_init

_WHICH = {"F": FACTOR_F, "G": FACTOR_G, "S": FACTOR_S, "FBLOCK": FACTOR_FBLOCK, "FBLOCK_ALL": FACTOR_FBLOCK_ALL,
          "GBLOCK": FACTOR_GBLOCK, "GBLOCK_ALL": FACTOR_GBLOCK_ALL, "SBLOCK": FACTOR_SBLOCK, "SBLOCK_ALL": FACTOR_SBLOCK_ALL,
          "U_SEND": FACTOR_U_SEND, "U_RECV": FACTOR_U_RECV, "FNEW_SEND": FACTOR_FNEW_SEND, "FNEW_RECV": FACTOR_FNEW_RECV,
          "T_SEND": FACTOR_T_SEND, "T_RECV": FACTOR_T_RECV, "GNEW_SEND": FACTOR_GNEW_SEND, "GNEW_RECV": FACTOR_GNEW_RECV}


def sliceable(prob: Problem, owner_of: Sequence[int], world: int) -> bool:
    """Can the F / G chains of ``prob`` be ROW-SLICED over the ranks (``resnmtf_options.slice_chains``)?  One view per rank
    (view v on rank v, at most 8), equal shapes and k, and every phi- (psi-) coupled pair of views shares ALL its rows
    (columns) in the same order -- the reference's auto-naming (``R/utils.r:482-491``) and every BASELINE configuration:
    ``star_prod_relevant`` (``R/utils.r:63-78``) then couples row r of view v with row r of view i and nothing else."""
    n_v = len(prob.init_f)
    if n_v != world or n_v > 8 or list(owner_of) != list(range(world)):
        return False
    shapes = {(f.shape[0], g.shape[0], f.shape[1]) for f, g in zip(prob.init_f, prob.init_g)}
    if len(shapes) != 1:
        return False
    phi, psi = np.asarray(prob.phi), np.asarray(prob.psi)
    for v in range(n_v):
        for w in range(n_v):
            if v == w:
                continue
            if phi[v, w] != 0 and list(prob.row_names[v]) != list(prob.row_names[w]):
                return False
            if psi[v, w] != 0 and list(prob.col_names[v]) != list(prob.col_names[w]):
                return False
    return True


def exchange_plan(n_views: int, owner_of: Sequence[int], phi, xi, psi, row_shared, col_shared) -> List[Dict[str, bool]]:
    """For every view, which of its factors some OTHER rank reads (identical on every rank).

    F_v is read by view w when phi[v, w] != 0 and w has a shared-row map towards v that is not NA
    (``R/utils.r:66-71``); likewise G_v through psi / shared columns; S_v when xi[v, w] != 0
    (``R/utils.r:42``).  A factor travels only if such a w lives on a rank other than v's owner."""
    phi = np.asarray(phi); xi = np.asarray(xi); psi = np.asarray(psi)
    plan = []
    for v in range(n_views):
        need = {"F": False, "G": False, "S": False}
        for w in range(n_views):
            if w == v or owner_of[w] == owner_of[v]:
                continue
            if phi[v, w] != 0 and row_shared[w].get(v) is not None:
                need["F"] = True
            if psi[v, w] != 0 and col_shared[w].get(v) is not None:
                need["G"] = True
            if xi[v, w] != 0:
                need["S"] = True
        plan.append(need)
    return plan


def replicated_views(n_views: int, owner_of: Sequence[int], phi, row_shared) -> List[bool]:
    """Views whose F update every rank runs: those in a phi-coupling component (edges: phi != 0 and a
    shared-row map that is not NA, either direction) whose views live on more than one rank."""
    phi = np.asarray(phi)
    parent = list(range(n_views))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for v in range(n_views):
        for w in range(n_views):
            if v != w and phi[v, w] != 0 and (row_shared[w].get(v) is not None or row_shared[v].get(w) is not None):
                parent[find(v)] = find(w)
    owners: Dict[int, set] = {}
    for v in range(n_views):
        owners.setdefault(find(v), set()).add(owner_of[v])
    return [len(owners[find(v)]) > 1 for v in range(n_views)]


class _CudaBlob:
    """Exposes a raw device range through ``__cuda_array_interface__`` so that torch can alias it."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


class HipEngineAdapter:
    """What the driver needs from an engine, on top of ``resnmtf_amd.engine.Engine``."""
    supports_replicated_gs = True
    supports_sliced = True

    def slice_info(self):
        return self.e.slice_info()

    def set_stop_tolerance(self, tol):
        self.e.set_stop_tolerance(tol)

    def loop_state(self):
        return self.e.loop_state()

    def kernel_timings(self, reset=False):
        return self.e.kernel_timings(reset)

    def p2p_export(self):
        return self.e.p2p_export()

    def p2p_import(self, rank, handles=b""):
        self.e.p2p_import(rank, handles)

    def p2p_selftest(self, timeout_ms=10000):
        self.e.p2p_selftest(timeout_ms)

    def __init__(self, engine):
        self.e = engine
        self._views: Dict[tuple, object] = {}

    def reserve_sweeps(self, n):
        self.e.reserve_sweeps(n)

    def prepare(self):
        self.e.prepare()

    def phase(self, v, ph, sweep):
        self.e.phase(v, ph, sweep)

    def factor_tensor(self, v: int, which: str):
        import torch
        key = (v, which)
        if key not in self._views:
            ptr, nbytes = self.e.factor_device_ptr(v, _WHICH[which])
            self._views[key] = torch.as_tensor(_CudaBlob(ptr, nbytes // 8), device="cuda")
        return self._views[key]

    @property
    def sblock_in_fblock(self) -> bool:
        """True when the library keeps the S exchange block of a view at the end of its F exchange block (replicated
        G / S chains, equal-shaped views): the all-gather of the F blocks then moves the S blocks too."""
        try:
            fptr, fbytes = self.e.factor_device_ptr(0, _WHICH["FBLOCK"])
            sptr, sbytes = self.e.factor_device_ptr(0, _WHICH["SBLOCK"])
        except Exception:
            return False
        return fptr <= sptr and sptr + sbytes <= fptr + fbytes

    def synchronize(self):
        self.e.synchronize()

    def view_errors(self, v, first, count):
        return self.e.view_errors(v, first, count)

    def finalise(self, v):
        return self.e.finalise(v)

    def get_factors(self, v):
        return self.e.get_factors(v)

    def close(self):
        self._views.clear()
        self.e.close()


def make_hip_engine(prob: Problem, owned: Sequence[bool], device_index: int, stream: int,
                    replicate_f: bool = False, replicate_gs: bool = False, **engine_opts) -> HipEngineAdapter:
    """Engine for this rank: data only for owned views, factor mirrors for the others, every kernel
    on the given HIP stream -- the torch stream the driver makes current around its broadcasts, so
    that torch.distributed orders them against the kernels.  (The legacy NULL stream must not be
    used: the library would fall back to a private non-blocking stream the broadcasts never see.)"""
    from .engine import Engine

    if not stream:
        raise ValueError("a non-default HIP stream is required")
    n_v = len(prob.init_f)
    shapes = prob.extras["shapes"]
    eng = Engine([s[0] for s in shapes], [s[1] for s in shapes], [prob.k] * n_v, owned=list(owned),
                 device_id=device_index, stream=stream, replicate_f=replicate_f, replicate_gs=replicate_gs, **engine_opts)
    for v in range(n_v):
        if owned[v]:
            eng.set_view(v, prob.data[v])
        eng.set_factors(v, prob.init_f[v], prob.init_s[v], prob.init_g[v])
    eng.set_restrictions(prob.phi, prob.xi, prob.psi)
    row_sh, col_sh = naming.shared_names(prob.row_names), naming.shared_names(prob.col_names)
    for v in range(n_v):
        for w in range(n_v):
            if w == v:
                continue
            iv, iw = naming.index_pairs(prob.row_names[v], prob.row_names[w], row_sh[v].get(w))
            eng.set_shared_rows(v, w, iv, iw)
            iv, iw = naming.index_pairs(prob.col_names[v], prob.col_names[w], col_sh[v].get(w))
            eng.set_shared_cols(v, w, iv, iw)
    return HipEngineAdapter(eng)


def local_problem(n_views: int, shape, k: int, phi: float = 0.0, xi: float = 0.0, psi: float = 0.0,
                  owned: Optional[Sequence[int]] = None) -> Problem:
    """``synth.make_problem`` generating the (large) data matrix only for the views in ``owned``; the (small) initial
    factors of every view are generated everywhere, with the same seeds as ``synth.make_problem`` (1000+v data,
    2000+v factors).  ``shape`` = one (n, m) for all views or a list of them (phi-coupled views share n, psi-coupled
    views share m: the reference's auto-naming semantics, ``R/utils.r:482-491``)."""
    shapes = [tuple(shape)] * n_views if isinstance(shape[0], (int, np.integer)) else [tuple(sh) for sh in shape]
    owned = list(range(n_views)) if owned is None else list(owned)
    data, f0, s0, g0 = [], [], [], []
    for v, (n, m) in enumerate(shapes):
        data.append(planted_view(n, m, k, 1000 + v) if v in owned else None)
        f, s, g = random_init(n, m, k, 2000 + v)
        f0.append(f); s0.append(s); g0.append(g)
    off = 1.0 - np.eye(n_views)
    prob = Problem(data, f0, s0, g0, phi * off, xi * off, psi * off, k, f"{n_views} views " + ",".join(f"{n}x{m}" for n, m in shapes))
    rn, cn, rb, cb = [], [], 1, 1
    for v, (n, m) in enumerate(shapes):
        if phi != 0.0:
            rn.append([f"row_{t}" for t in range(1, n + 1)])
        else:
            rn.append([f"row_{t}" for t in range(rb, rb + n)]); rb += n
        if psi != 0.0:
            cn.append([f"col_{t}" for t in range(1, m + 1)])
        else:
            cn.append([f"col_{t}" for t in range(cb, cb + m)]); cb += m
    prob.row_names, prob.col_names = rn, cn
    prob.extras["shapes"] = shapes
    return prob


_ROLE_STREAMS: Dict[tuple, object] = {}


def _role_stream(device_index: int, role: str):
    """One stream per (device, role) and process, shared by the drivers created one after the other (a bench builds several):
    every new stream is another hardware queue for the scheduler to map, and queues left behind by closed drivers were
    measured to halve the rate of three processes sharing one GPU (runlist oversubscription).  Drivers alive at the same
    time on one device merely share the queue."""
    import torch
    key = (int(device_index), role)
    st = _ROLE_STREAMS.get(key)
    if st is None:
        st = _ROLE_STREAMS[key] = torch.cuda.Stream(device=device_index)
    return st


class P2PUnavailable(RuntimeError):
    """The peer-store exchange cannot be used (layout or node); raised on every rank alike."""


class ShardedSweep:
    """Runs sweeps of a problem whose views are spread over the ranks of a process group."""

    @classmethod
    def create(cls, prob, owner_of, rank, world, slice_p2p="auto", **kw):
        """``slice_p2p="auto"``: the peer-store exchange when the layout has one and the node passes the library's self-test
        (resnmtf_p2p_selftest: probe stores, arrivals and the stream wait under host-side deadlines, the same outcome on every
        rank), else the same layout with its collectives.  True / False: as the constructor."""
        if slice_p2p != "auto":
            return cls(prob, owner_of, rank, world, slice_p2p=bool(slice_p2p), **kw)
        if world > 1 and kw.get("engine") is None and kw.get("engine_factory") is None:
            try:
                return cls(prob, owner_of, rank, world, slice_p2p=True, **kw)
            except (P2PUnavailable, ValueError) as exc:      # (both deterministic across ranks)
                if rank == 0:
                    import sys
                    print(f"[resnmtf_amd.sharded] peer-store exchange not used: {exc}", file=sys.stderr)
        return cls(prob, owner_of, rank, world, slice_p2p=False, **kw)

    def __init__(self, prob: Problem, owner_of: Sequence[int], rank: int, world: int,
                 device_index: int = 0, group=None, engine=None,
                 engine_factory: Optional[Callable] = None, replicate_f: Optional[bool] = None, **engine_opts):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank, self.world = rank, world
        self.n_views = len(prob.init_f)
        self._k = int(prob.init_f[0].shape[1])
        self.owner_of = list(owner_of)
        if len(self.owner_of) != self.n_views or any(o < 0 or o >= world for o in self.owner_of):
            raise ValueError("owner_of must give a valid rank for every view")
        self.owned = [o == rank for o in self.owner_of]
        if "shapes" not in prob.extras:
            prob.extras["shapes"] = [(f.shape[0], g.shape[0]) for f, g in zip(prob.init_f, prob.init_g)]
        row_sh, col_sh = naming.shared_names(prob.row_names), naming.shared_names(prob.col_names)
        self.plan = exchange_plan(self.n_views, self.owner_of, prob.phi, prob.xi, prob.psi, row_sh, col_sh)
        rep = replicated_views(self.n_views, self.owner_of, prob.phi, row_sh)
        if replicate_f is None:       # default: on with the HIP engine, off with an injected (stand-in) engine
            replicate_f = engine is None and engine_factory is None
        if replicate_f == "force":    # rehearsal of the replicated path with fewer ranks than it needs (bench.py, one GPU)
            rep = [True] * self.n_views
        self.replicated = rep if (replicate_f and any(rep)) else [False] * self.n_views
        # Replicated G and S chains (psi / xi coupling across ranks, one view per rank, equal k): every rank runs the G
        # and S updates of EVERY view from all-gathered inputs and streams only its own X -- the two passes of all ranks
        # then run at the same time, where ordered broadcasts of G_v / S_v would serialise whole ranks behind each other
        # (G_v' needs G_w' of every w < v, R/update_steps.r:195-204; S likewise, :231-237)
        want_gs = engine_opts.pop("replicate_gs", None)
        want_slice = engine_opts.pop("slice_chains", None)
        self.p2p = bool(engine_opts.pop("slice_p2p", False))
        # peer stores with the waits as one-wave kernels (resnmtf_options.slice_p2p = 2): whole sweeps can then be captured in a
        # graph and replayed (run(graph_chunk=...) / precapture) -- no host launch sequence per sweep
        self._p2p_graph = bool(engine_opts.pop("p2p_graph", False)) and self.p2p
        self._overlap_u = bool(engine_opts.pop("overlap_u", True))
        gs_coupled = any(p["G"] or p["S"] for p in self.plan)
        k_all = [f.shape[1] for f in prob.init_f]
        can_gs = (bool(replicate_f) and self.owner_of == list(range(world)) and len(set(k_all)) == 1 and self.n_views == world)
        self.replicate_gs = bool(can_gs and (gs_coupled if want_gs is None else want_gs))
        # Row-sliced chains instead of replicated ones (one view's worth of chain work per rank instead of V views'): needs the
        # layout of `sliceable`.  Chosen by default when the chain work it takes off a rank outweighs its three extra
        # exchanges per sweep (measured on MI355X, tools/time_replica_updates.py: the replicated F + G chains cost ~12 ps per
        # element and view -- c5 x 8: 321 + 57 us, c4 x 4: 22 + 10 us -- the sliced ones 1 / V of that plus ~30 us of pack /
        # unpack and latency floors; an exchange ~25 us): c5 x 8 saves ~280 us per sweep and is sliced, c4 x 4 would save
        # nothing and keeps the replicated chains with their two all-gathers, the small k <= 16 layouts likewise
        can_slice = bool(self.replicate_gs and sliceable(prob, self.owner_of, world) and
                         (engine is None or getattr(engine, "supports_sliced", False)))
        n0, m0 = prob.init_f[0].shape[0], prob.init_g[0].shape[0]
        saved_us = 12.0e-6 * self.n_views * (n0 + m0) * k_all[0] * (1.0 - 1.0 / max(self.n_views, 1)) - 30.0
        # (with the peer-store exchange an extra exchange costs a signal and a stream wait, not a collective launch: slice whenever possible)
        self.sliced = bool(can_slice and ((saved_us > 80.0 or self.p2p) if want_slice is None else want_slice))
        if want_slice and not self.sliced:
            raise ValueError("slice_chains needs one view per rank (<= 8), equal shapes and k, coupled views sharing all their "
                             "rows / columns in the same order, and coupling that calls for the replicated-chains layout")
        if self.replicate_gs:
            self.replicated = [True] * self.n_views
            for v in range(self.n_views):
                self.plan[v]["G"] = self.plan[v]["S"] = False
        if any(self.replicated):      # the F blocks of replicated views travel instead of their F
            for v in range(self.n_views):
                if self.replicated[v]:
                    self.plan[v]["F"] = False
        self._tstream = None      # compute stream (GPU path only)
        self._xstream = None      # exchange stream
        self._ev_f = self._ev_g = None
        self._ev_recv: Dict[tuple, object] = {}      # (view, factor) -> event after its latest broadcast
        serial_opt = engine_opts.pop("serial_exchange", None)   # None: one stream in the all-gather layout, two otherwise
        self._serial = bool(serial_opt)
        engine_opts_allgather = bool(engine_opts.pop("allgather_blocks", True))
        if engine is not None:
            self.engine = engine
        elif engine_factory is not None:
            self.engine = engine_factory(prob, self.owned)
        else:
            import torch
            torch.cuda.set_device(device_index)
            self._tstream = _role_stream(device_index, "compute")
            self._xstream = self._tstream if self._serial else _role_stream(device_index, "exchange")
            if self.p2p and not self.sliced:
                # block form: the exchange blocks of the replicated layouts by peer stores (one view per rank, every view replicated)
                if not (all(self.replicated) and self.owner_of == list(range(world)) and self.n_views == world and world <= 8):
                    raise ValueError("slice_p2p needs the sliced or the replicated-chains layout with one view per rank (<= 8)")
                engine_opts = dict(engine_opts, slice_index=rank, slice_count=world, slice_p2p=(2 if self._p2p_graph else 1))
            if self.sliced:
                engine_opts = dict(engine_opts, slice_chains=True, slice_index=rank, slice_count=world,
                                   slice_p2p=((2 if self._p2p_graph else 1) if self.p2p else 0))
            self.engine = make_hip_engine(prob, self.owned, device_index, self._tstream.cuda_stream,
                                          replicate_f=any(self.replicated), replicate_gs=self.replicate_gs, **engine_opts)
        self.sweeps_done = 0
        self._prepared = False
        if self.replicate_gs and not getattr(self.engine, "supports_replicated_gs", engine is None and engine_factory is None):
            raise ValueError("replicate_gs needs an engine with the block phases (PHASE_XTF / G_ALL / XG / S_ALL)")
        self._allgather_blocks = self._can_allgather(engine_opts_allgather)
        if self.replicate_gs and self._tstream is not None:       # collectives between dependent steps: one stream
            self._serial = True
            self._xstream = self._tstream
        if self.sliced and engine_factory is not None and not getattr(self.engine, "supports_sliced", False):
            raise ValueError("slice_chains needs an engine with the slice phases")
        # sliced chains: the U slices of a sweep travel beside the S-block gather and the S chain (nothing but the NEXT sweep's
        # F chain reads them) -- on a second stream and a second communicator, ordered by events (RCCL only; gloo is blocking)
        self._group_u = None
        self._ustream = None
        self._ev_u = self._ev_xg = None
        self._collected = True
        if self.p2p:
            # peer stores + stream-ordered flags instead of collectives (resnmtf_options.slice_p2p): every rank maps every
            # rank's receive buffers (hipIpc; the handles travel as objects over the process group, once), then a barrier --
            # nobody signals before everybody has mapped and zeroed
            if self._tstream is None or not hasattr(self.engine, "p2p_export"):
                raise ValueError("slice_p2p needs the HIP engine")
            # Every step that can fail on one rank only (export, mapping, the probes) is followed by a gather of every rank's
            # outcome before anybody acts on it: all ranks leave together, with the same exception, and nothing is hung on.
            def agree(problem: Optional[str]):
                outcomes: List[Optional[str]] = [None] * world
                dist.all_gather_object(outcomes, problem, group=group)
                failed = [o for o in outcomes if o]
                if failed:
                    dist.barrier(group=group)      # nobody unmaps while a peer may still be probing
                    self.engine.close()
                    raise P2PUnavailable("; ".join(failed))

            problem, mine = None, b""
            try:
                mine = self.engine.p2p_export()
            except Exception as exc:
                problem = f"rank {rank}: {exc}"
            everyone: List[Optional[bytes]] = [None] * world
            dist.all_gather_object(everyone, mine, group=group)
            if problem is None and all(everyone):
                try:
                    for r in range(world):
                        self.engine.p2p_import(r, b"" if r == rank else everyone[r])
                except Exception as exc:
                    problem = f"rank {rank}: {exc}"
            agree(problem)
            dist.barrier(group=group)              # everybody has mapped everybody
            # probe stores / arrivals / the stream wait under host-side deadlines (resnmtf_p2p_selftest)
            problem = None
            if not self.sliced and not self.replicate_gs and not (self._allgather_blocks and not any(p["G"] or p["S"] for p in self.plan)):
                problem = ("slice_p2p with the replicated F chain alone needs the one-exchange layout (equal F blocks, "
                           "no G / S coupling across ranks)")
            else:
                try:
                    self.engine.p2p_selftest(10000)
                except Exception as exc:      # (ResnmtfError: reported, never hung on)
                    problem = f"rank {rank}: {exc}"
            agree(problem)
        elif self.sliced and self._tstream is not None and self._overlap_u and dist.get_backend(group) == "nccl":
            import torch
            self._group_u = dist.new_group(ranks=list(range(world)), backend="nccl")      # collective: every rank gets here
            self._ustream = _role_stream(device_index, "u_exchange")
            self._ev_u, self._ev_xg = torch.cuda.Event(), torch.cuda.Event()
        # S blocks inside the F blocks (the HIP library with equal-shaped views): two collectives per sweep instead of three
        self._s_in_f = bool(self.replicate_gs and getattr(self.engine, "sblock_in_fblock", False))
        # all-gather layout: the exchange sits between two dependent steps of the sweep (nothing to overlap it
        # with), so it is issued on the compute stream itself -- measured with one rank on RCCL: 54.6 us per
        # sweep against 90.0 us through a second stream and its event edges
        if self._allgather_blocks and serial_opt is None and self._tstream is not None:
            self._serial = True
            self._xstream = self._tstream
        if self.p2p and self._tstream is not None:       # the stream waits of the phases order everything: one stream
            self._serial = True
            self._xstream = self._tstream

    def _can_allgather(self, wanted: bool) -> bool:
        """One in-place all-gather per sweep instead of one block broadcast per view: every rank owns exactly
        one view (view v on rank v), every view is replicated, and the blocks tile the arena evenly."""
        if not wanted or not all(self.replicated) or self.owner_of != list(range(self.world)):
            return False
        if not hasattr(self.engine, "factor_tensor"):
            return False
        try:
            arena = self.engine.factor_tensor(0, "FBLOCK_ALL")
            blocks = [self.engine.factor_tensor(v, "FBLOCK") for v in range(self.n_views)]
        except Exception:
            return False
        size = blocks[0].numel()
        esz = arena.element_size()
        return (arena.numel() == size * self.n_views and
                all(b.numel() == size and b.data_ptr() == arena.data_ptr() + v * size * esz for v, b in enumerate(blocks)))

    @property
    def collectives_per_sweep(self) -> int:
        """Collectives between dependent steps of one sweep in the replicated-chains layouts (0: ordered broadcasts)."""
        if self.p2p:
            return 0             # peer stores + stream-ordered flags
        if self.sliced:          # F rows back, T slices, G rows back, S blocks (+ the U slices beside the S chain when overlapped)
            return 4 if self._group_u is not None else 5
        if self.replicate_gs:
            return 2 if self._s_in_f else 3
        return 1 if self._allgather_blocks else 0

    @property
    def allgather_layout(self) -> bool:
        """True when the F exchange blocks travel by one all-gather per sweep (one view per rank, equal blocks)."""
        return self._allgather_blocks

    # ------------------------------------------------------------------
    def _allgather(self):
        """Every rank's F exchange block to every rank (end of a sweep / after the run prologue)."""
        arena = self.engine.factor_tensor(0, "FBLOCK_ALL")
        two = self._tstream is not None and self._xstream is not self._tstream
        if two:      # after the latest PHASE_F (last reader of the old blocks) and PHASE_G (writer of the own block)
            for after in (self._ev_f, self._ev_g):
                if after is not None:
                    self._xstream.wait_event(after)
        if self.dist.get_backend(self.group) == "nccl":
            mine = self.engine.factor_tensor(self.rank, "FBLOCK")      # = arena[rank]: in place
            self.dist.all_gather_into_tensor(arena, mine, group=self.group)
        else:        # gloo has no device all-gather: same data movement as one broadcast per block (rehearsals, tests)
            for v in range(self.n_views):
                self.dist.broadcast(self.engine.factor_tensor(v, "FBLOCK"), src=v, group=self.group)
        if two:
            import torch
            ev = self._ev_recv.get((0, "FBLOCK"))
            if ev is None:
                ev = torch.cuda.Event()
                for v in range(self.n_views):
                    self._ev_recv[(v, "FBLOCK")] = ev
            ev.record(self._xstream)

    def _gather_blocks(self, kind: str):
        """Every rank's exchange block of one kind ("FBLOCK", "GBLOCK", "SBLOCK") to every rank: one in-place all-gather
        over the library's arena when the blocks have one size (RCCL), else -- unequal views, or gloo in the tests and
        rehearsals -- the same bytes by one broadcast per block."""
        blocks = [self.engine.factor_tensor(v, kind) for v in range(self.n_views)]
        if self._tstream is not None and self.dist.get_backend(self.group) == "nccl":
            arena = self.engine.factor_tensor(0, kind + "_ALL")
            size = blocks[0].numel()
            if arena.numel() == size * self.n_views and all(b.numel() == size for b in blocks):
                self.dist.all_gather_into_tensor(arena, blocks[self.rank], group=self.group)
                return
        for v in range(self.n_views):
            self.dist.broadcast(blocks[v], src=self.owner_of[v], group=self.group)

    def _sweep_replicated_gs(self, t: int):
        """One sweep with all three chains replicated (R/update_steps.r:282-314 in the reference's order):
            F chain (every view, every rank)  ->  Xt.F pass of the own view  ->  [T blocks]  ->  G chain (every view)
            ->  X.G' pass of the own view + first half of its k x k job  ->  [U blocks with the S blocks inside]  ->  S chain,
            lambda, mu, error, F coefficients (every view); engines that keep the S blocks apart: [S blocks] before the
            S chain, [U blocks] after it"""
        r = self.rank
        if self.p2p:         # the phases store their blocks to the peers and wait for their own arrivals in stream order
            for ph in (PHASE_F_ALL, PHASE_XTF, PHASE_G_ALL, PHASE_XG, PHASE_S_ALL):
                self.engine.phase(r, ph, t)
            return
        self.engine.phase(r, PHASE_F_ALL, t)
        self.engine.phase(r, PHASE_XTF, t)
        self._gather_blocks("GBLOCK")
        self.engine.phase(r, PHASE_G_ALL, t)
        self.engine.phase(r, PHASE_XG, t)
        if self._s_in_f:
            # the F blocks carry the S blocks: one gather; S_ALL then writes the F coefficients of every view into the local
            # copies of the blocks (every rank computes them for every view, what arrived in those slots is overwritten)
            self._gather_blocks("FBLOCK")
            self.engine.phase(r, PHASE_S_ALL, t)
        else:
            self._gather_blocks("SBLOCK")
            self.engine.phase(r, PHASE_S_ALL, t)
            self._gather_blocks("FBLOCK")

    def _all_to_all(self, recv, send, group=None):
        """Chunk c of ``send`` to rank c, chunk r of ``recv`` from rank r (equal chunks): RCCL all-to-all on device tensors;
        CPU tensors (stand-in engine) go through gloo as they are; device tensors over gloo (ranks sharing one GPU in the
        tests and rehearsals) are staged through the host."""
        group = self.group if group is None else group
        if not recv.is_cuda or self.dist.get_backend(group) == "nccl":
            self.dist.all_to_all_single(recv, send, group=group)
            return
        host_in = send.cpu()                      # (blocking copies on the current = compute stream)
        host_out = host_in.new_empty(host_in.shape)
        self.dist.all_to_all_single(host_out, host_in, group=group)
        recv.copy_(host_out)

    def _exchange(self, kind: str, group=None):
        self._all_to_all(self.engine.factor_tensor(0, kind + "_RECV"), self.engine.factor_tensor(0, kind + "_SEND"), group)

    def _exchange_u(self):
        """The U slices of the X.G' passes just enqueued -> the slice holders.  Only the next sweep's F chain reads them, so
        with RCCL they travel on their own stream and communicator beside the S-block gather and the S chain."""
        if self._group_u is None:
            self._exchange("U")
            return
        import torch
        self._ev_xg.record(self._tstream)
        self._ustream.wait_event(self._ev_xg)                 # the pack of the pass that produced them
        with torch.cuda.stream(self._ustream):
            self._exchange("U", self._group_u)
        self._ev_u.record(self._ustream)

    def _sweep_sliced(self, t: int):
        """One sweep with ROW-SLICED F and G chains and the replicated S chain (R/update_steps.r:282-314 in the reference's
        order; star_prod_relevant is row-local, R/utils.r:67-73):
            F chain of every view on my row slice -> [new F rows to their owners] -> operand copies, Xt.F pass of the own view
            -> [T column slices + G coefficients] -> G chain of every view on my column slice -> [new G rows to their owners]
            -> operand copies, X.G' pass of the own view + first half of its k x k job
            -> [S blocks] -> S chain, lambda, mu, error, F coefficients of every view    ||   [U row slices]  (beside it)"""
        r, e = self.rank, self.engine
        if self.p2p:         # the phases store to the peers and wait for their own arrivals in stream order: nothing to do here
            for ph in (PHASE_SLICE_F, PHASE_SLICE_XTF, PHASE_SLICE_G, PHASE_SLICE_XG, PHASE_S_ALL):
                e.phase(r, ph, t)
            return
        if self._group_u is not None:
            self._tstream.wait_event(self._ev_u)             # this sweep's F chain reads the U slices
        e.phase(r, PHASE_SLICE_F, t)
        self._exchange("FNEW")
        e.phase(r, PHASE_SLICE_XTF, t)
        self._exchange("T")
        e.phase(r, PHASE_SLICE_G, t)
        self._exchange("GNEW")
        e.phase(r, PHASE_SLICE_XG, t)
        self._exchange_u()
        self._gather_blocks("SBLOCK")
        e.phase(r, PHASE_S_ALL, t)

    def collect(self):
        """Sliced chains: during the sweeps rank r keeps rows (columns) slice r of EVERY view's fp64 F (G) current and nothing
        else; this hands every slice to the rank that owns the view (one all-to-all per factor), after which the owner's
        F and G are whole.  Called by the result accessors; cheap (2 x n k doubles per rank)."""
        if not self.sliced or self._collected:
            return
        import torch
        if self._tstream is not None:
            # on the ENGINE's stream: the slices are written by its kernels and the whole factors are read by its finalise /
            # get_factors, which synchronise that stream only (outside run() torch's current stream is another one: the
            # slices were once read before the last sweeps had finished)
            with torch.cuda.stream(self._tstream):
                self._collect_on_current_stream()
            self._tstream.synchronize()
        else:
            self._collect_on_current_stream()
        self._collected = True

    def _collect_on_current_stream(self):
        import torch
        per = dict(zip(("F", "G"), self.engine.slice_info()))
        k = self._k
        for kind in ("F", "G"):
            full = [self.engine.factor_tensor(v, kind) for v in range(self.n_views)]
            n_el, chunk = full[0].numel(), per[kind] * k
            send = torch.zeros(self.n_views * chunk, dtype=full[0].dtype, device=full[0].device)
            lo, hi = min(self.rank * chunk, n_el), min((self.rank + 1) * chunk, n_el)
            for v in range(self.n_views):
                send[v * chunk:v * chunk + (hi - lo)] = full[v][lo:hi]
            recv = torch.empty_like(send)
            self._all_to_all(recv, send)
            full[self.rank].copy_(recv[:n_el])

    def _bcast(self, v: int, which: str):
        t = self.engine.factor_tensor(v, which)
        if self._tstream is None:                      # CPU stand-in engine: plain blocking broadcast
            self.dist.broadcast(t, src=self.owner_of[v], group=self.group)
            return
        import torch
        # F: after the latest PHASE_F; G / S: after the latest PHASE_G; an F exchange block is complete after
        # the owner's PHASE_G and must not land while a local PHASE_F still reads the previous one: both
        if self._xstream is not self._tstream:
            for after in {"F": (self._ev_f,), "FBLOCK": (self._ev_f, self._ev_g)}.get(which, (self._ev_g,)):
                if after is not None:
                    self._xstream.wait_event(after)
        # torch.distributed orders the collective on the CURRENT stream: run() has made the exchange stream
        # current for the whole call (the engine enqueues on its own stream handle regardless)
        self.dist.broadcast(t, src=self.owner_of[v], group=self.group)
        if self._xstream is not self._tstream:
            ev = self._ev_recv.get((v, which))           # one dedicated event per (view, factor), re-recorded
            if ev is None:
                ev = self._ev_recv[(v, which)] = torch.cuda.Event()
            ev.record(self._xstream)

    def _next_event(self):
        """Events are recycled round-robin: a wait captures the record that precedes it, so an event
        may be re-recorded while earlier waits on it are still queued."""
        import torch
        if not hasattr(self, "_ev_pool"):
            self._ev_pool = [torch.cuda.Event() for _ in range(64)]
            self._ev_next = 0
        ev = self._ev_pool[self._ev_next]
        self._ev_next = (self._ev_next + 1) % len(self._ev_pool)
        return ev

    def _phase(self, v: int, phases, sweep: int, reads_mirrors: bool):
        """Enqueue phases of an owned view on the compute stream, after the broadcasts they read."""
        two = self._tstream is not None and self._xstream is not self._tstream
        if two and reads_mirrors:
            # only what the phase reads: PHASE_F(v) its own exchange block and the broadcast F mirrors (the
            # replicated ones are computed on this very stream); PHASE_G / PHASE_S the G and S mirrors --
            # NOT the latest broadcast of any kind, which would turn every sweep into a barrier
            if phases[0] == PHASE_F_ALL:     # every block (all-gather layout: one shared event)
                waits = list({id(ev): ev for (w, which), ev in self._ev_recv.items() if which in ("F", "FBLOCK")}.values())
            elif phases[0] == PHASE_F:
                waits = [ev for (w, which), ev in self._ev_recv.items() if which == "F" or (which == "FBLOCK" and w == v)]
            else:
                waits = [ev for (w, which), ev in self._ev_recv.items() if which in ("G", "S")]
            for ev in waits:
                self._tstream.wait_event(ev)
        for ph in phases:
            self.engine.phase(v, ph, sweep)
        if two:
            ev = self._next_event()
            ev.record(self._tstream)
            if phases[0] in (PHASE_F, PHASE_F_ALL):
                self._ev_f = ev
            else:
                self._ev_g = ev

    def reserve(self, total_sweeps: int):
        """Room for the per-sweep errors of ``total_sweeps`` sweeps over all run() calls; call before the first run()
        when more than max(1024, first n_sweeps) sweeps will follow."""
        if self._prepared:
            raise RuntimeError("reserve() must precede the first run()")
        self._want_reserved = int(total_sweeps)

    def run(self, n_sweeps: Optional[int], graph_chunk: int = 0, tol: float = 1.0e-6, max_iters: int = 100000,
            check_every: int = 16) -> int:
        """``n_sweeps`` more sweeps (fixed-iteration mode, R/main.r:83-108), or -- ``n_sweeps=None``, layouts with the
        replicated S chain -- the reference's default loop (R/main.r:50-81): sweeps until ``|mean_err_t - mean_err_{t-1}| <=
        tol`` (``max_iters`` is a guard the reference lacks).  The S chain of every rank holds the full per-view error table
        and evaluates the test on the device, on identical bytes: every rank stops on the same sweep without a collective; the
        host looks at the flag every ``check_every`` sweeps (the sweeps enqueued after it fired return at once).  Returns the
        number of sweeps executed by this call.

        ``graph_chunk`` > 0 (one-stream layouts on RCCL only, OFF by default): ``graph_chunk`` sweeps -- the library's
        launches and the collectives between them -- are captured once in a ``torch.cuda.CUDAGraph`` and replayed.
        Measured with one rank: 46.5 instead of 54.1 us per sweep; not yet validated on several GPUs, hence opt-in."""
        if n_sweeps is None:
            return self._run_to_convergence(float(tol), int(max_iters), max(1, int(check_every)))
        if n_sweeps <= 0:
            return 0
        if self._tstream is not None:
            import torch
            with torch.cuda.stream(self._xstream):      # one context switch per call, not one per broadcast
                left = n_sweeps
                if graph_chunk > 0 and self._xstream is self._tstream and n_sweeps >= graph_chunk and self._graphable():
                    left = self._run_graphed(n_sweeps, int(graph_chunk))
                if left > 0:
                    self._run(left)
            return n_sweeps
        self._run(n_sweeps)
        return n_sweeps

    def _graphable(self) -> bool:
        """Sweeps that can be captured: the one-stream collective layouts other than the sliced one (RCCL), and every
        peer-store layout whose waits are kernels (p2p_graph)."""
        return self._p2p_graph if self.p2p else not self.sliced

    def precapture(self, graph_chunk: int):
        """Run prologue + capture of a ``graph_chunk``-sweep graph now (set-up), so that a later run() only replays."""
        if graph_chunk <= 0 or self._tstream is None or self._xstream is not self._tstream or not self._graphable():
            return False
        import torch
        with torch.cuda.stream(self._xstream):
            self._run_graphed(0, int(graph_chunk))
        return True

    def _run_to_convergence(self, tol: float, max_iters: int, check_every: int) -> int:
        if not self.replicate_gs:
            raise ValueError("convergence mode of the view-sharded path needs the replicated S chain (replicate_gs / slice_chains)")
        if not self._prepared:
            self.reserve(max(getattr(self, "_want_reserved", 0), max_iters))
        self.engine.set_stop_tolerance(tol)
        start = self.sweeps_done
        try:
            while self.sweeps_done - start < max_iters:
                todo = min(check_every, max_iters - (self.sweeps_done - start))
                before = self.sweeps_done
                self.run(todo)
                _, done, stop_sweep = self.engine.loop_state()      # synchronises; identical on every rank
                if done:
                    self.sweeps_done = max(before, min(self.sweeps_done, stop_sweep))
                    break
        finally:
            self.engine.set_stop_tolerance(-1.0)
        return self.sweeps_done - start

    def _run_graphed(self, n_sweeps: int, chunk: int) -> int:
        """Replays of a captured chunk; returns the sweeps left for the eager loop."""
        import torch
        self._run(0)                                    # run prologue and first exchange: outside the capture
        if self.sweeps_done + n_sweeps > self._reserved:
            raise RuntimeError("reserve more sweeps before the first run() (errors are kept per sweep)")
        graphs = self.__dict__.setdefault("_graphs", {})
        g = graphs.get(chunk)
        if g is None:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            before = self.sweeps_done
            # thread-local capture mode: the process group's watchdog thread polls the events of earlier (eager) collectives
            # while this thread captures -- in the default (global) mode such a call from ANY thread invalidates the capture
            with torch.cuda.graph(g, stream=self._tstream, capture_error_mode="thread_local"):
                self._run(chunk)
            self.sweeps_done = before                   # capturing enqueued nothing
            graphs[chunk] = g
        while n_sweeps >= chunk:
            g.replay()
            self.sweeps_done += chunk
            n_sweeps -= chunk
        return n_sweeps

    def _run(self, n_sweeps: int):
        if not self._prepared:
            self._reserved = max(1024, n_sweeps, getattr(self, "_want_reserved", 0))
            self.engine.reserve_sweeps(self._reserved)
            self.engine.prepare()
            self._prepared = True
            self._gs_exchanged = any(p["G"] or p["S"] for p in self.plan)
            if self.replicate_gs:         # F blocks (U, coefficients, lambda) and G blocks (mu of every view)
                self._gather_blocks("FBLOCK")
                self._gather_blocks("GBLOCK")
                if self.sliced and not self.p2p:      # ... and every view's U rows of my slice (the run prologue packed them;
                    self._exchange_u()                 #     slice_p2p: it stored them to the peers itself)
            elif any(self.replicated):    # the run prologue filled the owners' blocks: hand them round once
                if self._tstream is not None and self._xstream is not self._tstream:
                    self._ev_g = self._next_event()
                    self._ev_g.record(self._tstream)
                if self._allgather_blocks:
                    self._allgather()
                else:
                    for v in range(self.n_views):
                        if self.replicated[v]:
                            self._bcast(v, "FBLOCK")
            if self.p2p and not self.sliced:
                # block form: the first blocks travelled by collectives into the very arenas the sweeps store to -- nobody
                # stores before every rank's copies have landed (once per run; the sliced form stores to buffers of its own)
                if self._tstream is not None:
                    self.engine.synchronize()
                self.dist.barrier(group=self.group)
        if self.sweeps_done + n_sweeps > self._reserved:
            raise RuntimeError("reserve more sweeps before the first run() (errors are kept per sweep)")
        two_streams = self._tstream is not None and self._xstream is not self._tstream
        for _ in range(n_sweeps):
            t = self.sweeps_done
            if self.sliced:
                self._sweep_sliced(t)
                self._collected = False
                self.sweeps_done += 1
                continue
            if self.replicate_gs:
                self._sweep_replicated_gs(t)
                self.sweeps_done += 1
                continue
            # all-gather layout: every F update of the sweep first, as ONE phase (one launch when the views share
            # their rows in the same order).  Legal hoist: F_w' reads neither G nor S of this sweep.
            hoist = self._allgather_blocks
            if hoist and not self._gs_exchanged and not two_streams:
                # nothing crosses ranks inside the sweep: the rank's whole share in one library call
                self.engine.phase(self.rank, PHASE_LOCAL_SWEEP, t)
                if not self.p2p:      # (slice_p2p: the phase stored the own block to the peers itself)
                    self._allgather()
                self.sweeps_done += 1
                continue
            if hoist:
                self._phase(self.rank, (PHASE_F_ALL,), t, True)
            for v in range(self.n_views):
                mine = self.owned[v]
                if (mine or self.replicated[v]) and not hoist:
                    self._phase(v, (PHASE_F,), t, True)                      # reads the mirrors of coupled F_w (and v's block)
                if self.plan[v]["F"]:
                    self._bcast(v, "F")
                if mine:
                    self._phase(v, (PHASE_G, PHASE_S), t, self._gs_exchanged)  # reads G_w / S_w mirrors if exchanged
                if self.replicated[v] and not self._allgather_blocks:
                    self._bcast(v, "FBLOCK")                                 # next sweep's F update inputs, off the chain
                if self.plan[v]["G"]:
                    self._bcast(v, "G")
                if self.plan[v]["S"]:
                    self._bcast(v, "S")
            if self._allgather_blocks:
                self._allgather()                                            # next sweep's F update inputs of every view
            self.sweeps_done += 1

    # ------------------------------------------------------------------
    def view_error_table(self) -> np.ndarray:
        """[sweeps][views] relative errors, identical on every rank."""
        mine = {v: np.asarray(self.engine.view_errors(v, 0, self.sweeps_done)) for v in range(self.n_views) if self.owned[v]}
        gathered: List[Optional[dict]] = [None] * self.world
        self.dist.all_gather_object(gathered, mine, group=self.group)
        table = np.zeros((self.sweeps_done, self.n_views))
        for part in gathered:
            for v, e in part.items():
                table[:, v] = e
        return table

    def mean_errors(self) -> np.ndarray:
        """All_Error: mean over views per sweep (R/main.r:77-78,104-107)."""
        return self.view_error_table().mean(axis=1)

    def gather_results(self, dst: int = 0):
        """normalisation_check + binary clusters of every view, collected on rank ``dst``."""
        self.collect()
        mine = {v: self.engine.finalise(v) for v in range(self.n_views) if self.owned[v]}
        gathered: List[Optional[dict]] = [None] * self.world
        self.dist.all_gather_object(gathered, mine, group=self.group)
        if self.rank != dst:
            return None
        out: Dict[int, tuple] = {}
        for part in gathered:
            out.update(part)
        keys = ("output_f", "output_s", "output_g", "row_clusters", "col_clusters")
        return {k: [out[v][i] for v in range(self.n_views)] for i, k in enumerate(keys)}

    def close(self):
        if self.p2p:             # peers may still be storing into this rank's buffers / reading its flags
            if self._tstream is not None:
                self.engine.synchronize()
            self.dist.barrier(group=self.group)
        self.engine.close()


def res_nmtf_inner(data, row_indices=None, column_indices=None, init_f=None, init_s=None, init_g=None,
                   k_vec=None, phi=None, xi=None, psi=None, n_iters=None, *, rank: int, world: int,
                   owner_of: Optional[Sequence[int]] = None, row_names=None, col_names=None, device_index: Optional[int] = None,
                   group=None, tol: float = 1.0e-6, max_iters: int = 100000, no_clusts: bool = False, dst: int = 0,
                   slice_p2p="auto", **driver_opts):
    """``res_nmtf_inner`` (``R/main.r:32-140``, the explicit-init entry) over the ranks of a process group, one call per rank:
    the view-sharded counterpart of ``resnmtf_amd.api.res_nmtf_inner`` with the same arguments and the same return value
    (on rank ``dst``; ``None`` on the others).

    ``data[v]`` is needed only on the rank that owns view v (``owner_of[v]``, default ``v % world``) and may be ``None``
    elsewhere; the initial factors of EVERY view are needed on every rank (they are small, and the replicated chains start
    from them).  ``n_iters=None`` runs the reference's convergence loop (``R/main.r:50-81``): on the device when the layout
    has the replicated S chain (every rank stops on the same sweep), otherwise sweep by sweep with the error gathered after
    each.  ``row_indices`` / ``column_indices`` are accepted for signature parity; the shared-name maps are rebuilt from
    ``row_names`` / ``col_names`` (``naming.shared_names``), as ``api.res_nmtf_inner`` does when they are ``None``."""
    n_v = len(init_f) if init_f is not None else 0
    if not n_v or init_s is None or init_g is None or len(init_s) != n_v or len(init_g) != n_v:
        raise ValueError("the view-sharded entry needs explicit initial factors of every view on every rank")
    data = list(data)
    if len(data) != n_v:
        raise ValueError("data must have one entry per view (None for views this rank does not own)")
    k_all = [int(np.asarray(f).shape[1]) for f in init_f]
    if k_vec is not None and [int(k) for k in np.atleast_1d(k_vec)] != k_all:
        raise ValueError("k_vec does not match the initial factors")
    if len(set(k_all)) != 1:
        raise ValueError("the view-sharded path needs the same k in every view")
    owner_of = [v % world for v in range(n_v)] if owner_of is None else list(owner_of)
    for v in range(n_v):
        if owner_of[v] == rank and data[v] is None:
            raise ValueError(f"rank {rank} owns view {v} but was not given its data")
    shapes = [(np.asarray(f).shape[0], np.asarray(g).shape[0]) for f, g in zip(init_f, init_g)]
    zeros = np.zeros((n_v, n_v))
    prob = Problem([None if d is None else np.asarray(d, dtype=np.float64) for d in data],
                   [np.asarray(x, dtype=np.float64) for x in init_f], [np.asarray(x, dtype=np.float64) for x in init_s],
                   [np.asarray(x, dtype=np.float64) for x in init_g],
                   zeros if phi is None else np.asarray(phi, dtype=np.float64), zeros if xi is None else np.asarray(xi, dtype=np.float64),
                   zeros if psi is None else np.asarray(psi, dtype=np.float64), k_all[0], "res_nmtf_inner (view-sharded)")
    if row_names is None or col_names is None:      # the reference's auto-naming (R/utils.r:482-491) from the shapes alone
        stand_in = [np.broadcast_to(0.0, sh) for sh in shapes]      # (shape only: no memory behind it)
        rn, cn = naming.give_names(stand_in, prob.phi if prob.phi.any() else None, prob.psi if prob.psi.any() else None)
        row_names, col_names = row_names or rn, col_names or cn
    prob.row_names, prob.col_names = [list(x) for x in row_names], [list(x) for x in col_names]
    prob.extras["shapes"] = shapes
    if "engine_factory" in driver_opts or "engine" in driver_opts:
        drv = ShardedSweep(prob, owner_of, rank, world, group=group, **driver_opts)
    else:
        drv = ShardedSweep.create(prob, owner_of, rank, world, slice_p2p=slice_p2p, group=group,
                                  device_index=(rank if device_index is None else device_index), **driver_opts)
    try:
        if n_iters is not None:
            drv.reserve(int(n_iters) + 8)
            drv.run(int(n_iters))
            errs = drv.mean_errors()
        elif drv.replicate_gs:
            drv.run(None, tol=tol, max_iters=max_iters)
            errs = drv.mean_errors()
        else:      # no replicated S chain to hold the stop test: one sweep, one look (R/main.r:55,77-80)
            drv.reserve(max_iters + 8)
            prev = None
            while drv.sweeps_done < max_iters:
                drv.run(1)
                errs = drv.mean_errors()
                if prev is not None and not (abs(errs[-1] - prev) > tol):
                    break
                prev = errs[-1]
        res = drv.gather_results(dst)
        lm = {v: drv.engine.get_factors(v)[3:5] for v in range(n_v) if drv.owned[v]} if hasattr(drv.engine, "get_factors") else {}
        everyone: List[Optional[dict]] = [None] * world
        drv.dist.all_gather_object(everyone, lm, group=group)
    finally:
        drv.close()
    if rank != dst:
        return None
    if no_clusts:                                                                                 # main.r:115-120
        return {k: res[k] for k in ("output_f", "output_s", "output_g")}
    lam_mu = {}
    for part in everyone:
        lam_mu.update(part or {})
    error = float(np.mean(errs[-10:])) if n_iters is None else float(errs[-1])                    # main.r:127 / :129
    return {"output_f": res["output_f"], "output_s": res["output_s"], "output_g": res["output_g"],
            "Error": error, "All_Error": np.asarray(errs), "bisil": None,
            "row_clusters": res["row_clusters"], "col_clusters": res["col_clusters"],
            "lambda": [lam_mu.get(v, (None, None))[0] for v in range(n_v)], "mu": [lam_mu.get(v, (None, None))[1] for v in range(n_v)]}
