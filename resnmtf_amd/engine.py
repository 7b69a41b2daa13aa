"""Thin object wrapper over the C-ABI (include/resnmtf_hip.h): one ``Engine`` = one
``resnmtf_handle`` = the device-resident state of the multiplicative-update loop on ONE GPU.

Everything numeric happens in the HIP library; this module only marshals NumPy arrays
(fp64, column-major, as R would hand them over) and raises on any error code.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import ResnmtfError


def _f64_colmajor(a, shape=None) -> np.ndarray:
    arr = np.asfortranarray(np.asarray(a, dtype=np.float64))
    if shape is not None and tuple(arr.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(arr.shape)}")
    return arr


def _dp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


def device_count() -> int:
    return int(_lib.load().resnmtf_device_count())


class Engine:
    def __init__(self, n_rows: Sequence[int], n_cols: Sequence[int], k: Sequence[int],
                 owned: Optional[Sequence[bool]] = None, device_id: int = 0, stream: int = 0,
                 use_graph: bool = True, check_every: int = 32, target_workgroups: int = 0,
                 time_kernels: bool = False, pass_waves: int = 0, pass_splits_xg: int = 0,
                 pass_splits_xtf: int = 0, pass_lds_pad_kb: int = 0, update_blocks: int = 0, no_pitch_pad: bool = False,
                 kk_mode: int = 0, bf16_split: int = 0, replicate_f: bool = False, no_f_chain: bool = False,
                 x_half: int = 0, half_unroll: int = 0, replicate_gs: bool = False, wait_mode: int = 0,
                 slice_chains: bool = False, slice_index: int = 0, slice_count: int = 0, fuse_updates: int = 0, slice_p2p: bool = False, xcd_order: bool = False):
        self._lib = _lib.load()
        self.n_views = len(n_rows)
        self.n_rows = [int(x) for x in n_rows]
        self.n_cols = [int(x) for x in n_cols]
        self.k = [int(x) for x in k]
        self.owned = [True] * self.n_views if owned is None else [bool(x) for x in owned]
        opts = _lib.Options()
        self._lib.resnmtf_default_options(C.byref(opts))
        opts.device_id = int(device_id)
        opts.stream = C.c_void_p(int(stream)) if stream else None
        opts.use_graph = 1 if use_graph else 0
        opts.check_every = int(check_every)
        opts.target_workgroups = int(target_workgroups)
        opts.time_kernels = 1 if time_kernels else 0
        opts.pass_waves = int(pass_waves)
        opts.pass_splits_xg = int(pass_splits_xg)
        opts.pass_splits_xtf = int(pass_splits_xtf)
        opts.pass_lds_pad_kb = int(pass_lds_pad_kb)
        opts.update_blocks = int(update_blocks)
        opts.no_pitch_pad = 1 if no_pitch_pad else 0
        opts.kk_mode = int(kk_mode)
        opts.bf16_split = int(bf16_split)
        opts.replicate_f = 1 if replicate_f else 0
        opts.no_f_chain = 1 if no_f_chain else 0
        opts.x_half = int(x_half)
        opts.half_unroll = int(half_unroll)
        opts.replicate_gs = 1 if replicate_gs else 0
        opts.wait_mode = int(wait_mode)
        opts.slice_chains = 1 if slice_chains else 0
        opts.slice_index = int(slice_index)
        opts.slice_count = int(slice_count)
        opts.fuse_updates = int(fuse_updates)
        opts.slice_p2p = int(slice_p2p)          # (True = 1: stream waits; 2: wait kernels, graph-capturable)
        opts.xcd_order = 1 if xcd_order else 0
        nr = np.asarray(self.n_rows, dtype=np.int32)
        nc = np.asarray(self.n_cols, dtype=np.int32)
        kk = np.asarray(self.k, dtype=np.int32)
        ow = np.asarray([1 if o else 0 for o in self.owned], dtype=np.int32)
        self._h = C.c_void_p()
        rc = self._lib.resnmtf_create(self.n_views, _ip(nr), _ip(nc), _ip(kk), _ip(ow), C.byref(opts),
                                      C.byref(self._h))
        if rc != _lib.OK:
            text = self._lib.resnmtf_last_error(None)
            self._h = None
            raise ResnmtfError(rc, text.decode() if text else "")

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc: int):
        if rc != _lib.OK:
            text = self._lib.resnmtf_last_error(self._h)
            raise ResnmtfError(rc, text.decode() if text else "")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.resnmtf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ inputs
    def set_view(self, v: int, x):
        x = _f64_colmajor(x, (self.n_rows[v], self.n_cols[v]))
        self._check(self._lib.resnmtf_set_view(self._h, v, _dp(x)))

    def set_view_raw(self, v: int, x_raw) -> bool:
        """Upload a raw view; the non-negativity shift and the column normalisation of
        ``check_inputs`` (``R/utils.r:416,422``) run on the device.  Returns True when an entry was
        negative (the reference warns, ``R/utils.r:23-25``)."""
        import ctypes as C
        x = _f64_colmajor(x_raw, (self.n_rows[v], self.n_cols[v]))
        neg = C.c_int(0)
        self._check(self._lib.resnmtf_set_view_raw(self._h, v, _dp(x), C.byref(neg)))
        return bool(neg.value)

    def copy_view_from(self, v: int, other: "Engine", v_src: int = 0):
        """Device copy of a view another engine (same GPU, same shape) has uploaded."""
        self._check(self._lib.resnmtf_copy_view(self._h, v, other._h, v_src))

    def shuffle_view_from(self, v: int, other: "Engine", v_src: int = 0, seed: int = 0, normalise: bool = True):
        """``shuffle_view`` (``R/obtain_bicl.r:11-22``) of another engine's view, drawn on the device."""
        self._check(self._lib.resnmtf_shuffle_view(self._h, v, other._h, v_src, int(seed), 1 if normalise else 0))

    def subsample_view_from(self, v: int, other: "Engine", v_src: int, rows, cols):
        """The sub-sample ``X[rows, cols]`` of another engine's view (``R/stability_analysis.r:230-249``), gathered
        on the device; this engine's view v must have the shape ``(len(rows), len(cols))``."""
        rows = np.ascontiguousarray(rows, dtype=np.int32); cols = np.ascontiguousarray(cols, dtype=np.int32)
        if len(rows) != self.n_rows[v] or len(cols) != self.n_cols[v]:
            raise ValueError("index counts must equal the view's shape")
        self._check(self._lib.resnmtf_subsample_view(self._h, v, other._h, v_src, _ip(rows), _ip(cols)))

    def empty_lines(self, v: int):
        """(row_mask, col_mask) of view ``v``'s latest device-drawn data (shuffle / sub-sample): True where a row / column
        sums to exactly zero -- the condition of the reference's redraw (``R/obtain_bicl.r:14-18``) and of its trimming
        of sub-samples (``R/stability_analysis.r:165-190``)."""
        nr, nc = C.c_int(0), C.c_int(0)
        rm = np.zeros(self.n_rows[v], dtype=np.uint8); cm = np.zeros(self.n_cols[v], dtype=np.uint8)
        self._check(self._lib.resnmtf_view_empty_lines(self._h, v, C.byref(nr), C.byref(nc),
                                                       rm.ctypes.data_as(C.POINTER(C.c_ubyte)), cm.ctypes.data_as(C.POINTER(C.c_ubyte))))
        return rm.astype(bool), cm.astype(bool)

    def get_view(self, v: int) -> np.ndarray:
        """The device copy of the view's data (fp32 precision) as an fp64 matrix."""
        x = np.zeros((self.n_rows[v], self.n_cols[v]), order="F")
        self._check(self._lib.resnmtf_get_view(self._h, v, _dp(x)))
        return x

    def init_svd(self, v: int, seed: int = 0, sigma: float = 0.05, n_power: int = 0) -> np.ndarray:
        """``init_mats_inner`` (``R/update_steps.r:78-125``) on the device for view ``v`` (randomized
        top-k SVD on the streaming-pass kernels); returns the k leading singular values."""
        d = np.zeros(self.k[v])
        self._check(self._lib.resnmtf_init_svd(self._h, v, int(seed), float(sigma), int(n_power), _dp(d)))
        return d

    def set_factors(self, v: int, f, s, g, lam=None, mu=None):
        k = self.k[v]
        f = _f64_colmajor(f, (self.n_rows[v], k))
        s = _f64_colmajor(s, (k, k))
        g = _f64_colmajor(g, (self.n_cols[v], k))
        lam = None if lam is None else np.ascontiguousarray(lam, dtype=np.float64)
        mu = None if mu is None else np.ascontiguousarray(mu, dtype=np.float64)
        self._check(self._lib.resnmtf_set_factors(self._h, v, _dp(f), _dp(s), _dp(g), _dp(lam), _dp(mu)))

    def set_restrictions(self, phi=None, xi=None, psi=None):
        mats = []
        for m in (phi, xi, psi):
            mats.append(None if m is None else _f64_colmajor(m, (self.n_views, self.n_views)))
        self._check(self._lib.resnmtf_set_restrictions(self._h, _dp(mats[0]), _dp(mats[1]), _dp(mats[2])))

    def _set_shared(self, fn, v, w, idx_v, idx_w):
        if idx_v is None:           # NA
            self._check(fn(self._h, v, w, -1, None, None))
            return
        iv = np.ascontiguousarray(idx_v, dtype=np.int32)
        iw = np.ascontiguousarray(idx_w, dtype=np.int32)
        if iv.shape != iw.shape:
            raise ValueError("index arrays differ in length")
        self._check(fn(self._h, v, w, int(iv.size), _ip(iv), _ip(iw)))

    def set_shared_rows(self, v: int, w: int, idx_v, idx_w):
        self._set_shared(self._lib.resnmtf_set_shared_rows, v, w, idx_v, idx_w)

    def set_shared_cols(self, v: int, w: int, idx_v, idx_w):
        self._set_shared(self._lib.resnmtf_set_shared_cols, v, w, idx_v, idx_w)

    # ------------------------------------------------------------------ loop
    def run(self, n_iters: Optional[int] = None, tol: float = 1.0e-6, max_iters: int = 100000):
        """Fixed ``n_iters`` sweeps, or (``n_iters=None``) until |d mean err| <= tol.
        Returns the All_Error vector of the sweeps executed."""
        if n_iters is not None and n_iters <= 0:
            raise ValueError("n_iters must be positive (None = run to convergence)")
        cap = int(n_iters) if n_iters else int(max_iters)
        buf = self.__dict__.get("_err_buf")
        if buf is None or buf[0].size < cap:            # (kept across calls: a short run is a few hundred microseconds)
            arr = np.empty(max(cap, 1024), dtype=np.float64)
            buf = self._err_buf = (arr, arr.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(0))
        errs, ptr, done = buf
        rc = self._lib.resnmtf_run(self._h, int(n_iters or 0), float(tol), int(max_iters), ptr, cap, C.byref(done))
        if rc != _lib.OK:
            self._check(rc)
        return errs[:done.value].copy()

    def reserve_sweeps(self, sweeps: int):
        self._check(self._lib.resnmtf_reserve_sweeps(self._h, int(sweeps)))

    def prepare(self):
        self._check(self._lib.resnmtf_prepare(self._h))

    def phase(self, v: int, phase: int, sweep: int):
        self._check(self._lib.resnmtf_phase(self._h, v, phase, sweep))

    def synchronize(self):
        self._check(self._lib.resnmtf_synchronize(self._h))

    def factor_device_ptr(self, v: int, which: int):
        ptr = C.c_void_p()
        nbytes = C.c_size_t()
        self._check(self._lib.resnmtf_factor_device_ptr(self._h, v, which, C.byref(ptr), C.byref(nbytes)))
        return int(ptr.value), int(nbytes.value)

    def view_errors(self, v: int, first: int, count: int) -> np.ndarray:
        out = np.zeros(count, dtype=np.float64)
        self._check(self._lib.resnmtf_view_errors(self._h, v, first, count, _dp(out)))
        return out

    # ------------------------------------------------------------------ outputs
    def get_factors(self, v: int, with_lm: bool = True):
        n, m, k = self.n_rows[v], self.n_cols[v], self.k[v]
        f = np.zeros((n, k), order="F"); s = np.zeros((k, k), order="F"); g = np.zeros((m, k), order="F")
        lam = np.zeros(k) if with_lm else None
        mu = np.zeros(k) if with_lm else None
        self._check(self._lib.resnmtf_get_factors(self._h, v, _dp(f), _dp(s), _dp(g), _dp(lam), _dp(mu)))
        return f, s, g, lam, mu

    def finalise(self, v: int):
        n, m, k = self.n_rows[v], self.n_cols[v], self.k[v]
        f = np.zeros((n, k), order="F"); s = np.zeros((k, k), order="F"); g = np.zeros((m, k), order="F")
        rc = np.zeros((n, k), order="F"); cc = np.zeros((m, k), order="F")
        self._check(self._lib.resnmtf_finalise(self._h, v, _dp(f), _dp(s), _dp(g), _dp(rc), _dp(cc)))
        return f, s, g, rc, cc

    def view_image_info(self, v: int):
        """(kind, rel_error): kind 0 = f32 images, 1 = fp16, 2 = uniform 16-bit integers; the relative quantisation
        error of the 2-byte image of view ``v`` measured at upload (``x_half``)."""
        kind = C.c_int(0)
        rel = C.c_double(0.0)
        self._check(self._lib.resnmtf_view_image_info(self._h, v, C.byref(kind), C.byref(rel)))
        return int(kind.value), float(rel.value)

    def set_stop_tolerance(self, tol: float):
        """Phase API: ``tol >= 0`` = convergence mode (``R/main.r:50-81``) for the phases enqueued from now on."""
        self._check(self._lib.resnmtf_set_stop_tolerance(self._h, float(tol)))

    def loop_state(self):
        """(sweeps closed since prepare, stop flag, sweep count at which it fired) -- synchronises."""
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self._lib.resnmtf_loop_state(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), bool(b.value), int(c.value)

    def slice_info(self):
        a, b = C.c_int(0), C.c_int(0)
        self._check(self._lib.resnmtf_slice_info(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def p2p_export(self) -> bytes:
        """slice_p2p: the IPC handles of this engine's receive buffers and arrival counters (opaque bytes for the peers)."""
        buf = C.create_string_buffer(1024)
        n = C.c_size_t(0)
        self._check(self._lib.resnmtf_p2p_export(self._h, buf, 1024, C.byref(n)))
        return buf.raw[:n.value]

    def p2p_import(self, rank: int, handles: bytes = b""):
        buf = C.create_string_buffer(handles, max(len(handles), 1))
        self._check(self._lib.resnmtf_p2p_import(self._h, int(rank), buf, len(handles)))

    def p2p_selftest(self, timeout_ms: int = 10000):
        """slice_p2p: probe stores, arrivals and the stream wait with host-side deadlines (every rank, after the imports and a
        barrier, before prepare); raises ResnmtfError when the node cannot run the peer-store exchange."""
        self._check(self._lib.resnmtf_p2p_selftest(self._h, int(timeout_ms)))

    def kernel_timings(self, reset: bool = False) -> dict:
        """time_kernels: {kind: (ms_total, launches)} for the kernels of a view-sharded sweep."""
        n = len(_lib.TIMED_KINDS)
        ms = (C.c_double * n)(); cnt = (C.c_longlong * n)()
        self._check(self._lib.resnmtf_kernel_timings(self._h, ms, cnt, 1 if reset else 0))
        return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(_lib.TIMED_KINDS)}

    def pass_timings(self, reset: bool = False) -> dict:
        t = _lib.PassTiming()
        self._check(self._lib.resnmtf_pass_timings(self._h, C.byref(t), 1 if reset else 0))
        return {name: getattr(t, name) for name, _ in _lib.PassTiming._fields_}
