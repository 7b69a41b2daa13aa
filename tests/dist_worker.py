"""Worker process of the multi-rank tests (launched by tests/test_sharded_*.py, one per rank).

mode "cpu": the ShardedSweep driver over gloo with a stand-in engine that evaluates each phase
with the oracle's update rules on CPU tensors -- this exercises the product's exchange schedule
(ordered broadcasts, running-list semantics, result gathering) without a GPU.
mode "gpu": the same driver with the real HIP engine (both ranks may share one GPU), gloo backend.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleEngine:
    """Stand-in engine (TEST ONLY): same interface as resnmtf_amd.sharded.HipEngineAdapter, phases
    evaluated by oracle/resnmtf_oracle.py on CPU tensors.  Mirrors are plain tensors that the
    driver's broadcasts overwrite in place."""

    def __init__(self, prob, owned):
        import torch
        from oracle import resnmtf_oracle as O
        self.O, self.torch = O, torch
        self.prob, self.owned = prob, list(owned)
        self.n_v = len(prob.init_f)
        self.F = [torch.tensor(np.ascontiguousarray(f), dtype=torch.float64) for f in prob.init_f]
        self.S = [torch.tensor(np.ascontiguousarray(s), dtype=torch.float64) for s in prob.init_s]
        self.G = [torch.tensor(np.ascontiguousarray(g), dtype=torch.float64) for g in prob.init_g]
        self.lam = [f.sum(0) for f in prob.init_f]
        self.mu = [g.sum(0) for g in prob.init_g]
        self.row_idx = O.reorder_data(prob.row_names)
        self.col_idx = O.reorder_data(prob.col_names)
        self.norms = [None if d is None else np.linalg.norm(d, "fro") ** 2 for d in prob.data]
        self.errs = {v: [] for v in range(self.n_v) if self.owned[v]}

    def reserve_sweeps(self, n): pass
    def prepare(self): pass
    def synchronize(self): pass
    def close(self): pass

    def _lists(self):
        return [t.numpy() for t in self.F], [t.numpy() for t in self.S], [t.numpy() for t in self.G]

    def phase(self, v, ph, sweep):
        from resnmtf_amd._lib import PHASE_F, PHASE_G, PHASE_S
        O, p = self.O, self.prob
        fl, sl, gl = self._lists()
        x = p.data[v]
        if ph == PHASE_F:
            new = O.update_f(x, fl, sl[v], gl[v], self.lam[v], p.phi, v, self.row_idx[v], p.row_names[v], p.row_names)
            self.F[v].copy_(self.torch.from_numpy(new))
        elif ph == PHASE_G:
            newg = O.update_g(x, fl[v], sl[v], gl, self.mu[v], p.psi, v, self.col_idx[v], p.col_names[v], p.col_names)
            self.G[v].copy_(self.torch.from_numpy(newg))
            fl, sl, gl = self._lists()
            news = O.update_s(x, fl[v], sl, gl[v], p.xi, v)
            self.S[v].copy_(self.torch.from_numpy(news))
            self.lam[v] = O.update_lm(self.lam[v], fl[v])
            self.mu[v] = O.update_lm(self.mu[v], gl[v])
            x_hat = (fl[v] @ news) @ gl[v].T
            self.errs[v].append(np.linalg.norm(x - x_hat, "fro") ** 2 / self.norms[v])
        elif ph == PHASE_S:
            pass

    def factor_tensor(self, v, which):
        return {"F": self.F, "G": self.G, "S": self.S}[which][v].view(-1)

    def view_errors(self, v, first, count):
        return np.array(self.errs[v][first:first + count])

    def finalise(self, v):
        O = self.O
        f, g, s = O.normalisation_check([self.F[v].numpy()], [self.G[v].numpy()], [self.S[v].numpy()])
        rc, cc = O.binary_clusters(f, g, s)
        return f[0], s[0], g[0], rc[0], cc[0]


class OracleBlockEngine(OracleEngine):
    """Stand-in for the replicated-chains layout (replicate_gs): the exchange blocks are CPU tensors holding what the
    library's blocks hold in exact fp64 -- F block [U = X G | S | G^T G | lambda], G block [T = X^T F' | S | F'^T F' | mu],
    S block [S old | F'^T X G' | F'^T F' | G'^T G'] -- and the _ALL phases evaluate the reference's rules for EVERY view
    from them, with the reference's association order (R/update_steps.r:141-251 restated on the products instead of X:
    x %*% g -> U, crossprod(x, f) -> T, crossprod(f, x) %*% g -> T^T g)."""
    supports_replicated_gs = True

    def __init__(self, prob, owned):
        super().__init__(prob, owned)
        t = self.torch
        k = prob.k
        self.nm = [(f.shape[0], g.shape[0]) for f, g in zip(prob.init_f, prob.init_g)]
        self.fblk = [t.zeros(n * k + 2 * k * k + k, dtype=t.float64) for n, _ in self.nm]
        self.gblk = [t.zeros(m * k + 2 * k * k + k, dtype=t.float64) for _, m in self.nm]
        self.sblk = [t.zeros(4 * k * k, dtype=t.float64) for _ in self.nm]
        self.lam_all = [t.tensor(f.sum(0)) for f in prob.init_f]          # every rank tracks lambda / mu of every view
        self.mu_all = [t.tensor(g.sum(0)) for g in prob.init_g]
        self.k = k

    def prepare(self):                                                     # run prologue of the owned views: U = X G
        for v in range(self.n_v):
            if self.owned[v]:
                self._fill_fblock(v)

    def _fill_fblock(self, v):
        k, (n, _) = self.k, self.nm[v]
        g, sm = self.G[v].numpy(), self.S[v].numpy()
        blk = self.fblk[v].numpy()
        blk[:n * k] = (self.prob.data[v] @ g).ravel()
        blk[n * k:n * k + k * k] = sm.ravel()
        blk[n * k + k * k:n * k + 2 * k * k] = (g.T @ g).ravel()
        blk[n * k + 2 * k * k:] = self.lam_all[v].numpy()

    def factor_tensor(self, v, which):
        if which == "FBLOCK":
            return self.fblk[v]
        if which == "GBLOCK":
            return self.gblk[v]
        if which == "SBLOCK":
            return self.sblk[v]
        if which.endswith("_ALL"):
            raise RuntimeError("no arena in the stand-in")
        return super().factor_tensor(v, which)

    def phase(self, v, ph, sweep):
        from resnmtf_amd._lib import PHASE_F_ALL, PHASE_G_ALL, PHASE_S_ALL, PHASE_XG, PHASE_XTF
        O, p, k = self.O, self.prob, self.k
        if ph == PHASE_F_ALL:                                               # update_f of every view, R/update_steps.r:141-165
            for w in range(self.n_v):
                n = self.nm[w][0]
                blk = self.fblk[w].numpy()
                u = blk[:n * k].reshape(n, k); sm = blk[n * k:n * k + k * k].reshape(k, k)
                gtg = blk[n * k + k * k:n * k + 2 * k * k].reshape(k, k); lam = blk[n * k + 2 * k * k:]
                fl = [t.numpy() for t in self.F]
                cur = fl[w]
                numerator = u @ sm.T                                         # :146
                denominator = (cur @ sm) @ (gtg @ sm.T)                      # :147-148
                lam_mat = 0.5 * np.tile(lam, (n, 1))                         # :151
                phi_vec = p.phi[:, w]
                if phi_vec.sum() == 0:                                       # :152
                    with np.errstate(divide="ignore", invalid="ignore"):
                        ratio = numerator / (denominator + lam_mat)
                    ratio[np.isnan(ratio)] = 1.0
                    new = cur * ratio
                else:
                    num_mat_prod = O.star_prod_relevant(phi_vec, fl, cur, self.row_idx[w], p.row_names[w], p.row_names)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        new = cur * ((numerator + num_mat_prod) / (denominator + phi_vec.sum() * cur + lam_mat))
                self.F[w].copy_(self.torch.from_numpy(np.abs(new)))
        elif ph == PHASE_XTF:                                               # T = X^T F', F'^T F' of the own view
            m = self.nm[v][1]
            f = self.F[v].numpy()
            blk = self.gblk[v].numpy()
            blk[:m * k] = (p.data[v].T @ f).ravel()
            blk[m * k:m * k + k * k] = self.S[v].numpy().ravel()
            blk[m * k + k * k:m * k + 2 * k * k] = (f.T @ f).ravel()
            blk[m * k + 2 * k * k:] = self.mu_all[v].numpy()
        elif ph == PHASE_G_ALL:                                             # update_g of every view, R/update_steps.r:180-207
            for w in range(self.n_v):
                m = self.nm[w][1]
                blk = self.gblk[w].numpy()
                tt = blk[:m * k].reshape(m, k); sm = blk[m * k:m * k + k * k].reshape(k, k)
                ftf = blk[m * k + k * k:m * k + 2 * k * k].reshape(k, k); mu = blk[m * k + 2 * k * k:]
                gl = [t.numpy() for t in self.G]
                cur = gl[w]
                numerator = tt @ sm                                          # :185
                denominator = (cur @ sm.T) @ (ftf @ sm)                      # :186-187
                mu_mat = 0.5 * np.tile(mu, (m, 1))                           # :188
                if p.psi.sum() == 0:                                         # :190 (whole matrix)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        ratio = numerator / (denominator + mu_mat)
                    ratio[np.isnan(ratio)] = 1.0
                    new = cur * ratio
                else:
                    psi_vec = p.psi[:, w]
                    num_mat_prod = O.star_prod_relevant(psi_vec, gl, cur, self.col_idx[w], p.col_names[w], p.col_names)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        new = cur * ((numerator + num_mat_prod) / (denominator + psi_vec.sum() * cur + mu_mat))
                self.G[w].copy_(self.torch.from_numpy(np.abs(new)))
        elif ph == PHASE_XG:                                                # S block + the U of the next F update
            f, g = self.F[v].numpy(), self.G[v].numpy()
            m = self.nm[v][1]
            tt = self.gblk[v].numpy()[:m * k].reshape(m, k)
            blk = self.sblk[v].numpy()
            blk[:k * k] = self.S[v].numpy().ravel()
            blk[k * k:2 * k * k] = ((f.T @ p.data[v]) @ g).ravel()           # :223
            blk[2 * k * k:3 * k * k] = (f.T @ f).ravel()
            blk[3 * k * k:] = (g.T @ g).ravel()
            assert tt.shape == (m, k)
            n = self.nm[v][0]
            self.fblk[v].numpy()[:n * k] = (p.data[v] @ g).ravel()
        elif ph == PHASE_S_ALL:                                             # update_s, update_lm of every view; error of the own
            olds = [self.sblk[w].numpy()[:k * k].reshape(k, k).copy() for w in range(self.n_v)]
            run = list(olds)
            for w in range(self.n_v):
                blk = self.sblk[w].numpy()
                nn = blk[k * k:2 * k * k].reshape(k, k); ftf = blk[2 * k * k:3 * k * k].reshape(k, k); gtg = blk[3 * k * k:].reshape(k, k)
                cur = run[w]
                denominator = (ftf @ cur) @ gtg                              # :224
                if p.xi.sum() == 0:                                          # :226
                    with np.errstate(divide="ignore", invalid="ignore"):
                        ratio = nn / denominator
                    ratio[np.isnan(ratio)] = 1.0
                    new = cur * ratio
                else:
                    xi_vec = p.xi[:, w]
                    with np.errstate(divide="ignore", invalid="ignore"):
                        new = cur * ((nn + O.star_prod(xi_vec, run)) / (denominator + xi_vec.sum() * cur))
                run[w] = np.abs(new)
                self.S[w].copy_(self.torch.from_numpy(run[w]))
                self.lam_all[w] = self.lam_all[w] * self.F[w].sum(0)        # :249-251, :312-313
                self.mu_all[w] = self.mu_all[w] * self.G[w].sum(0)
                n = self.nm[w][0]
                fb = self.fblk[w].numpy()
                fb[n * k:n * k + k * k] = run[w].ravel()
                fb[n * k + k * k:n * k + 2 * k * k] = gtg.ravel()
                fb[n * k + 2 * k * k:] = self.lam_all[w].numpy()
                if self.owned[w]:
                    x = p.data[w]
                    x_hat = (self.F[w].numpy() @ run[w]) @ self.G[w].numpy().T
                    self.errs[w].append(np.linalg.norm(x - x_hat, "fro") ** 2 / self.norms[w])
        else:
            raise ValueError(f"phase {ph} is not part of the replicated-chains layout")


class OracleSliceEngine(OracleBlockEngine):
    """Stand-in for the ROW-SLICED chains layout (slice_chains): rank r keeps rows (columns) slice r of EVERY view's F (G)
    current and walks the F (G) chain of all views on it; the exchange buffers are CPU tensors holding what the library's
    hold, in exact fp64 -- U / T slices (T chunks followed by the own view's S and F'^T F': what the G rule needs besides
    the rows), new F / G rows on their way back to the owners, S blocks [S old | F'^T X G' | F'^T F' | G'^T G' | colSums(G')
    | colSums(F') | ||X||^2].  Rules and association order as OracleBlockEngine (R/update_steps.r:141-251 on the products)."""
    supports_sliced = True

    def __init__(self, prob, owned, rank, world):
        super().__init__(prob, owned)
        t, k = self.torch, self.k
        self.rank, self.world = rank, world
        n, m = self.nm[0]
        up = lambda a, b: (a + b - 1) // b * b
        self.per_r, self.per_c = up((n + world - 1) // world, 8), up((m + world - 1) // world, 8)
        z = lambda count: t.zeros(count, dtype=t.float64)
        self.buf = {"U_SEND": z(world * self.per_r * k), "U_RECV": z(world * self.per_r * k),
                    "FNEW_SEND": z(world * self.per_r * k), "FNEW_RECV": z(world * self.per_r * k),
                    "T_SEND": z(world * (self.per_c * k + 2 * k * k)), "T_RECV": z(world * (self.per_c * k + 2 * k * k)),
                    "GNEW_SEND": z(world * self.per_c * k), "GNEW_RECV": z(world * self.per_c * k)}
        self.sblk = [z(4 * k * k + 2 * k + 1) for _ in self.nm]
        self.tol, self.done, self.stop_sweep, self.prev_mean, self.closed = -1.0, False, 0, 0.0, 0

    def slice_info(self):
        return self.per_r, self.per_c

    def set_stop_tolerance(self, tol):
        self.tol = tol

    def loop_state(self):
        return self.closed, self.done, self.stop_sweep

    def _rows(self, per, full):
        lo = min(self.rank * per, full)
        return lo, min(lo + per, full)

    def _pack(self, key, mat, per, tail=None):
        k, chunk = self.k, per * self.k + (0 if tail is None else tail.size)
        buf = self.buf[key].numpy()
        for c in range(self.world):
            rows = mat[c * per:(c + 1) * per]
            buf[c * chunk:c * chunk + rows.size] = rows.ravel()
            if tail is not None:
                buf[c * chunk + per * k:(c + 1) * chunk] = tail

    def prepare(self):
        super().prepare()
        v = self.rank
        n = self.nm[v][0]
        self._pack("U_SEND", self.fblk[v].numpy()[:n * self.k].reshape(n, self.k), self.per_r)

    def factor_tensor(self, v, which):
        if which in self.buf:
            return self.buf[which]
        return super().factor_tensor(v, which)

    def _chain(self, is_g):
        """update_f / update_g of every view, in view order, on my slice (R/update_steps.r:141-165 / :180-207)."""
        p, k, V = self.prob, self.k, self.n_v
        full = self.nm[0][1] if is_g else self.nm[0][0]
        per = self.per_c if is_g else self.per_r
        lo, hi = self._rows(per, full)
        recv = self.buf["T_RECV" if is_g else "U_RECV"].numpy()
        chunk = per * k + (2 * k * k if is_g else 0)
        out = self.buf["GNEW_SEND" if is_g else "FNEW_SEND"].numpy()
        state = self.G if is_g else self.F
        rest = p.psi if is_g else p.phi
        run = [state[c].numpy()[lo:hi].copy() for c in range(V)]                 # running values of the slice
        for w in range(V):
            prod = recv[w * chunk:w * chunk + (hi - lo) * k].reshape(hi - lo, k)
            if is_g:
                tail = recv[w * chunk + per * k:(w + 1) * chunk]
                sm, gram = tail[:k * k].reshape(k, k), tail[k * k:].reshape(k, k)
                lm = self.mu_all[w].numpy()
                numerator = prod @ sm                                              # :185
                denominator = (run[w] @ sm.T) @ (gram @ sm)                        # :186-187
                unrestricted = rest.sum() == 0                                     # :190 (whole matrix)
            else:
                nrow = self.nm[w][0]
                blk = self.fblk[w].numpy()
                sm, gram = blk[nrow * k:nrow * k + k * k].reshape(k, k), blk[nrow * k + k * k:nrow * k + 2 * k * k].reshape(k, k)
                lm = blk[nrow * k + 2 * k * k:]
                numerator = prod @ sm.T                                            # :146
                denominator = (run[w] @ sm) @ (gram @ sm.T)                        # :147-148
                unrestricted = rest[:, w].sum() == 0                               # :152
            lm_mat = 0.5 * np.tile(lm, (hi - lo, 1))                               # :151 / :188
            if unrestricted:
                with np.errstate(divide="ignore", invalid="ignore"):
                    ratio = numerator / (denominator + lm_mat)
                ratio[np.isnan(ratio)] = 1.0
                new = run[w] * ratio
            else:
                vec = rest[:, w]
                acc = 0.0                                                          # star_prod_relevant with every row shared in
                for i in range(V):                                                 # the same order (R/utils.r:63-78)
                    if vec[i] != 0:
                        acc = acc + vec[i] * run[i] * full                         # :73  (masked = the coupled view's rows)
                num_prod = acc / full                                              # :77
                with np.errstate(divide="ignore", invalid="ignore"):
                    new = run[w] * ((numerator + num_prod) / (denominator + vec.sum() * run[w] + lm_mat))
            run[w] = np.abs(new)
            state[w].numpy()[lo:hi] = run[w]
            out[w * per * k:w * per * k + (hi - lo) * k] = run[w].ravel()

    def phase(self, v, ph, sweep):
        from resnmtf_amd._lib import PHASE_S_ALL, PHASE_SLICE_F, PHASE_SLICE_G, PHASE_SLICE_XG, PHASE_SLICE_XTF
        if self.tol >= 0 and self.done:
            return
        O, p, k = self.O, self.prob, self.k
        n, m = self.nm[v]
        if ph == PHASE_SLICE_F:
            self._chain(False)
        elif ph == PHASE_SLICE_G:
            self._chain(True)
        elif ph == PHASE_SLICE_XTF:                                          # received rows -> whole F' of the own view; T, F'^T F'
            f = self.buf["FNEW_RECV"].numpy()[:n * k].reshape(n, k)
            self.F[v].copy_(self.torch.from_numpy(f.copy()))
            tail = np.concatenate([self.S[v].numpy().ravel(), (f.T @ f).ravel()])
            tt = p.data[v].T @ f
            self._tt = tt
            self._pack("T_SEND", tt, self.per_c, tail)
        elif ph == PHASE_SLICE_XG:
            g = self.buf["GNEW_RECV"].numpy()[:m * k].reshape(m, k)
            self.G[v].copy_(self.torch.from_numpy(g.copy()))
            f = self.F[v].numpy()
            blk = self.sblk[v].numpy()
            blk[:k * k] = self.S[v].numpy().ravel()
            blk[k * k:2 * k * k] = ((f.T @ p.data[v]) @ g).ravel()           # :223
            blk[2 * k * k:3 * k * k] = (f.T @ f).ravel()
            blk[3 * k * k:4 * k * k] = (g.T @ g).ravel()
            blk[4 * k * k:4 * k * k + k] = g.sum(0)
            blk[4 * k * k + k:4 * k * k + 2 * k] = f.sum(0)
            blk[4 * k * k + 2 * k] = self.norms[v]
            self._pack("U_SEND", p.data[v] @ g, self.per_r)
        elif ph == PHASE_S_ALL:                                              # update_s chain, update_lm, error of every view
            kk = k * k
            run = [self.sblk[w].numpy()[:kk].reshape(k, k).copy() for w in range(self.n_v)]
            errs = []
            for w in range(self.n_v):
                blk = self.sblk[w].numpy()
                nn, ftf, gtg = blk[kk:2 * kk].reshape(k, k), blk[2 * kk:3 * kk].reshape(k, k), blk[3 * kk:4 * kk].reshape(k, k)
                cur = run[w]
                denominator = (ftf @ cur) @ gtg                              # :224
                if p.xi.sum() == 0:                                          # :226
                    with np.errstate(divide="ignore", invalid="ignore"):
                        ratio = nn / denominator
                    ratio[np.isnan(ratio)] = 1.0
                    new = cur * ratio
                else:
                    xi_vec = p.xi[:, w]
                    with np.errstate(divide="ignore", invalid="ignore"):
                        new = cur * ((nn + O.star_prod(xi_vec, run)) / (denominator + xi_vec.sum() * cur))
                run[w] = np.abs(new)
                self.S[w].copy_(self.torch.from_numpy(run[w]))
                self.lam_all[w] = self.lam_all[w] * self.torch.from_numpy(blk[4 * kk + k:4 * kk + 2 * k].copy())   # :249-251, :312-313
                self.mu_all[w] = self.mu_all[w] * self.torch.from_numpy(blk[4 * kk:4 * kk + k].copy())
                nrow = self.nm[w][0]
                fb = self.fblk[w].numpy()
                fb[nrow * k:nrow * k + kk] = run[w].ravel()
                fb[nrow * k + kk:nrow * k + 2 * kk] = gtg.ravel()
                fb[nrow * k + 2 * kk:] = self.lam_all[w].numpy()
                xn = blk[4 * kk + 2 * k]                                     # trace form (the library's, R/utils.r:157-166)
                errs.append((xn - 2.0 * np.sum(run[w] * nn) + np.sum(((ftf @ run[w]) @ gtg) * run[w])) / xn)
                if self.owned[w]:
                    x = p.data[w]
                    x_hat = (self.F[w].numpy() @ run[w]) @ self.G[w].numpy().T
                    self.errs[w].append(np.linalg.norm(x - x_hat, "fro") ** 2 / self.norms[w])
            self.closed += 1
            if self.tol >= 0:                                                # R/main.r:55,77-80
                mean = float(np.sum(errs) / self.n_v)
                if not (abs(mean - self.prev_mean) > self.tol):
                    self.done, self.stop_sweep = True, self.closed
                self.prev_mean = mean
        else:
            raise ValueError(f"phase {ph} is not part of the sliced-chains layout")


def build_problem():
    """3 views, phi + psi + xi coupled, rows/columns partially shared at different positions."""
    from resnmtf_amd.synth import Problem, planted_view, random_init
    rng = np.random.default_rng(77)
    shapes = [(96, 72), (80, 72), (96, 64)]
    k = 5
    data = [planted_view(n, m, k, 500 + v) for v, (n, m) in enumerate(shapes)]
    inits = [random_init(n, m, k, 600 + v) for v, (n, m) in enumerate(shapes)]
    rown = [[f"r{i}" for i in range(96)], [f"r{i}" for i in rng.permutation(np.arange(10, 90))],
            [f"r{i}" for i in rng.permutation(96)]]
    coln = [[f"c{i}" for i in range(72)], [f"c{i}" for i in rng.permutation(72)], [f"q{i}" for i in range(64)]]
    up = np.triu(np.ones((3, 3)), 1)
    sym = lambda a: a + a.T
    prob = Problem(data, [i[0] for i in inits], [i[1] for i in inits], [i[2] for i in inits],
                   sym(1.5 * up), sym(0.4 * up), sym(1.0 * up), k, row_names=rown, col_names=coln)
    prob.extras["shapes"] = shapes
    return prob


def build_problem_one_view_per_rank(world=2, identity=False, xi=0.4, k=5):
    """One view per rank with equal row counts (equal F exchange blocks): the layout in which the
    blocks travel by one all-gather per sweep.  phi + xi coupled, rows shared at permuted positions."""
    from resnmtf_amd.synth import Problem, planted_view, random_init
    rng = np.random.default_rng(78)
    shapes = [(96, 72 - 8 * v) for v in range(world)]
    data = [planted_view(n, m, k, 700 + v) for v, (n, m) in enumerate(shapes)]
    inits = [random_init(n, m, k, 800 + v) for v, (n, m) in enumerate(shapes)]
    # identity: every view lists the shared rows in the same order (what auto-naming gives): the F updates
    # of a sweep then run as one fused launch (f_chain_kernel)
    rown = [[f"r{i}" for i in (range(96) if (v == 0 or identity) else rng.permutation(96))] for v in range(world)]
    coln = [[f"c{v}_{i}" for i in range(m)] for v, (_, m) in enumerate(shapes)]
    off = 1.0 - np.eye(world)
    prob = Problem(data, [i[0] for i in inits], [i[1] for i in inits], [i[2] for i in inits],
                   1.5 * off, xi * off, 0.0 * off, k, row_names=rown, col_names=coln)
    prob.extras["shapes"] = shapes
    return prob


def build_problem_gs(world, k=5):
    """One view per rank, phi + psi + xi all coupling across ranks, rows AND columns shared in part and at different
    positions, a different weight per pair (tests/helpers.py coupled_problem): the replicated-chains layout."""
    from helpers import coupled_problem
    shapes = [(96 + 8 * v, 72 - 8 * (v % 2)) for v in range(world)]
    if os.environ.get("RESNMTF_TEST_UNEVEN") == "1":      # row counts in different 64-row paddings: F blocks of different sizes
        shapes = [(96 + 90 * v, 72 - 8 * (v % 2)) for v in range(world)]
    prob = coupled_problem(shapes, k, seed=31 + world, phi_w=1.5, psi_w=1.0, xi_w=0.4)
    prob.extras["shapes"] = shapes
    return prob


def build_problem_slice(world, k=5, n=96, m=72):
    """One view per rank, equal shapes, phi + psi + xi coupling, every row / column shared in the same order (the
    reference's auto-naming, R/utils.r:482-491): the layout the row-sliced chains need."""
    from resnmtf_amd import sharded
    prob = sharded.local_problem(world, (n, m), k, phi=1.5, xi=0.4, psi=1.0)
    return prob


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--mode", choices=["cpu", "cpu_gs", "gpu", "gpu_gs", "gpu_norep", "gpu_allgather", "gpu_chain", "gpu_chain_off", "gpu_graph1", "gpu_gs_rccl1",
                                       "cpu_slice", "cpu_slice_conv", "gpu_slice", "gpu_slice_conv", "gpu_gs_conv", "gpu_slice_rccl1", "gpu_gs_graph1",
                                       "gpu_slice_p2p", "gpu_slice_p2p_conv", "gpu_block_p2p", "gpu_block_p2p_f", "gpu_block_p2p_conv", "gpu_block_p2p_fallback", "cpu_api", "cpu_api_conv", "gpu_api", "gpu_api_conv"], required=True)
    ap.add_argument("--n", type=int, default=96)
    ap.add_argument("--m", type=int, default=72)
    ap.add_argument("--tol", type=float, default=1e-6)
    ap.add_argument("--sweeps", type=int, default=12)
    ap.add_argument("--xi", type=float, default=0.4)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--graph", type=int, default=0)      # peer-store modes: sweeps replayed from captured graphs of this many
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(a.port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from resnmtf_amd import sharded
    if a.mode == "gpu_graph1":          # one rank on RCCL: the captured-chunk replay against the eager loop
        import torch
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        prob = sharded.local_problem(1, (512, 192), 7, phi=1.0, owned=[0])
        outs = []
        for chunk in (0, 4):
            drv = sharded.ShardedSweep(prob, [0], 0, 1, device_index=0, replicate_f="force")
            assert drv.allgather_layout
            drv.run(5, graph_chunk=chunk)
            drv.run(a.sweeps - 5, graph_chunk=chunk)          # 4-sweep replays + eager remainder
            torch.cuda.synchronize()
            outs.append((drv.mean_errors(), drv.gather_results(0)))
            drv.close()
        same = np.array_equal(outs[0][0], outs[1][0]) and all(
            np.array_equal(x, y) for key in outs[0][1] for x, y in zip(outs[0][1][key], outs[1][1][key]))
        np.savez(a.out, same=np.array(same), all_error=outs[1][0])
        dist.destroy_process_group()
        return
    if a.mode == "gpu_gs_graph1":       # one rank on RCCL, replicated G / S chains: captured chunks against the eager loop
        import torch
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        prob = sharded.local_problem(1, (512, 192), a.k, owned=[0])
        outs = []
        for chunk in (0, 4):
            drv = sharded.ShardedSweep(prob, [0], 0, 1, device_index=0, replicate_f="force", replicate_gs=True)
            assert drv.replicate_gs and not drv.sliced
            drv.run(5, graph_chunk=chunk)
            drv.run(a.sweeps - 5, graph_chunk=chunk)
            torch.cuda.synchronize()
            outs.append((drv.mean_errors(), drv.gather_results(0)))
            drv.close()
        same = np.array_equal(outs[0][0], outs[1][0]) and all(
            np.array_equal(x, y) for key in outs[0][1] for x, y in zip(outs[0][1][key], outs[1][1][key]))
        np.savez(a.out, same=np.array(same), all_error=outs[1][0])
        dist.destroy_process_group()
        return
    if a.mode == "gpu_gs_rccl1":        # one rank on RCCL: the replicated-chains layout with the in-place all-gathers over the arenas
        import torch
        from helpers import rel_fro, run_oracle
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        prob = sharded.local_problem(1, (512, 192), a.k, owned=[0])
        drv = sharded.ShardedSweep(prob, [0], 0, 1, device_index=0, replicate_f="force", replicate_gs=True)
        assert drv.replicate_gs
        drv.run(a.sweeps)
        torch.cuda.synchronize()
        errs = drv.mean_errors(); res = drv.gather_results(0)
        drv.close()
        ref = run_oracle(prob, n_iters=a.sweeps)
        ok = (np.allclose(errs, ref["All_Error"], atol=2e-5) and rel_fro(res["output_f"][0], ref["output_f"][0]) < 2e-5 and
              rel_fro(res["output_g"][0], ref["output_g"][0]) < 2e-5 and rel_fro(res["output_s"][0], ref["output_s"][0]) < 1e-4)
        np.savez(a.out, same=np.array(ok), all_error=errs)
        dist.destroy_process_group()
        return
    if a.mode == "gpu_slice_rccl1":     # one rank on RCCL: the sliced layout's all-to-alls, second communicator and stream
        import torch
        from helpers import rel_fro, run_oracle
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        prob = sharded.local_problem(1, (a.n, a.m), a.k, owned=[0])
        drv = sharded.ShardedSweep(prob, [0], 0, 1, device_index=0, replicate_f="force", replicate_gs=True, slice_chains=True)
        assert drv.sliced and drv._group_u is not None
        drv.run(a.sweeps // 2); drv.run(a.sweeps - a.sweeps // 2)
        torch.cuda.synchronize()
        errs = drv.mean_errors(); res = drv.gather_results(0)
        drv.close()
        ref = run_oracle(prob, n_iters=a.sweeps)
        ok = (np.allclose(errs, ref["All_Error"], atol=2e-5) and rel_fro(res["output_f"][0], ref["output_f"][0]) < 2e-5 and
              rel_fro(res["output_g"][0], ref["output_g"][0]) < 2e-5 and rel_fro(res["output_s"][0], ref["output_s"][0]) < 1e-4)
        np.savez(a.out, same=np.array(ok), all_error=errs)
        dist.destroy_process_group()
        return
    dist.init_process_group("gloo", rank=a.rank, world_size=a.world)
    if a.mode in ("cpu_slice", "cpu_slice_conv", "gpu_slice", "gpu_slice_conv", "gpu_gs_conv", "gpu_slice_p2p", "gpu_slice_p2p_conv"):
        slice_main(a, dist, sharded)
        return
    if a.mode.startswith("gpu_block_p2p"):
        block_main(a, dist, sharded)
        return
    if "_api" in a.mode:
        api_main(a, dist, sharded)
        return
    one_per_rank = a.mode in ("gpu_allgather", "gpu_chain", "gpu_chain_off")
    gs = a.mode in ("cpu_gs", "gpu_gs")
    prob = (build_problem_gs(a.world, a.k) if gs else
            build_problem_one_view_per_rank(a.world, identity=a.mode != "gpu_allgather", xi=a.xi) if one_per_rank else build_problem())
    n_v = len(prob.init_f)
    owner_of = [v % a.world for v in range(n_v)]
    if a.mode == "cpu":
        drv = sharded.ShardedSweep(prob, owner_of, a.rank, a.world, engine_factory=lambda p, owned: OracleEngine(p, owned))
    elif a.mode == "cpu_gs":
        drv = sharded.ShardedSweep(prob, owner_of, a.rank, a.world, engine_factory=lambda p, owned: OracleBlockEngine(p, owned),
                                   replicate_f=True)
        assert drv.replicate_gs
    elif a.mode == "gpu_gs":
        drv = sharded.ShardedSweep(prob, owner_of, a.rank, a.world, device_index=0)
        assert drv.replicate_gs and all(drv.replicated)
    else:
        opts = {"no_f_chain": True} if a.mode == "gpu_chain_off" else {}
        drv = sharded.ShardedSweep(prob, owner_of, a.rank, a.world, device_index=0, replicate_f=(a.mode != "gpu_norep"), **opts)
        assert any(drv.replicated) == (a.mode != "gpu_norep")
        assert drv.allgather_layout == one_per_rank
    drv.run(a.sweeps // 2)
    drv.run(a.sweeps - a.sweeps // 2)          # two calls: state carries over
    mirrors_ok = True
    if a.mode not in ("cpu", "cpu_gs"):          # every rank's copy of every F (G, S: replicated chains) must be bitwise the owner's
        import torch
        drv.engine.synchronize(); torch.cuda.synchronize()
        kinds = ("F", "G", "S") if gs else ("F",)
        mine = [b"".join(drv.engine.factor_tensor(v, kd).cpu().numpy().tobytes() for kd in kinds) for v in range(n_v)]
        allf = [None] * a.world
        dist.all_gather_object(allf, mine)
        mirrors_ok = all(allf[r][v] == allf[owner_of[v]][v] for r in range(a.world) for v in range(n_v))
    errs = drv.mean_errors()
    res = drv.gather_results(0)
    drv.close()
    if a.rank == 0:
        out = {"all_error": errs, "mirrors_ok": np.array(mirrors_ok)}
        for key, lst in res.items():
            for v, arr in enumerate(lst):
                out[f"{key}{v}"] = arr
        np.savez(a.out, **out)
    dist.barrier()
    dist.destroy_process_group()


def api_main(a, dist, sharded):
    """sharded.res_nmtf_inner -- the reference's entry point over the ranks of a group: stand-in engine on CPU (three views
    with names shared in part, two of them on one rank when world = 2) / HIP engines sharing the one GPU (one view per rank)."""
    conv = a.mode.endswith("_conv")
    if a.mode.startswith("cpu"):
        prob = build_problem()
        opts = {"engine_factory": lambda p, owned: OracleEngine(p, owned)}
    else:
        prob = build_problem_gs(a.world, a.k)
        opts = {"device_index": 0}
    n_v = len(prob.init_f)
    owner_of = [v % a.world for v in range(n_v)]
    data = [d if owner_of[v] == a.rank else None for v, d in enumerate(prob.data)]      # a rank holds only its own views
    res = sharded.res_nmtf_inner(data, None, None, prob.init_f, prob.init_s, prob.init_g, None, prob.phi, prob.xi, prob.psi,
                                 None if conv else a.sweeps, rank=a.rank, world=a.world, owner_of=owner_of,
                                 row_names=prob.row_names, col_names=prob.col_names, tol=a.tol, max_iters=a.sweeps, **opts)
    assert (res is None) == (a.rank != 0)
    if a.rank == 0:
        out = {"all_error": res["All_Error"], "Error": np.array(res["Error"])}
        for key in ("output_f", "output_s", "output_g", "row_clusters", "col_clusters"):
            for v, arr in enumerate(res[key]):
                out[f"{key}{v}"] = arr
        np.savez(a.out, **out)
    dist.barrier()
    dist.destroy_process_group()


def block_main(a, dist, sharded):
    """The exchange blocks of the replicated layouts by peer stores (slice_p2p without slice_chains), ranks sharing one GPU:
    gpu_block_p2p -- F, G and S chains replicated (phi + psi + xi across ranks, rows / columns shared in part);
    gpu_block_p2p_f -- the F chain alone (phi only, different column counts: BASELINE c3's layout), one exchange per sweep;
    each against the same layout with its collectives (gloo here): bitwise.  _conv: the convergence loop."""
    import torch
    f_only = a.mode == "gpu_block_p2p_f"
    prob = build_problem_one_view_per_rank(a.world, identity=(a.k % 2 == 1), xi=0.0, k=a.k) if f_only else build_problem_gs(a.world, a.k)
    n_v = a.world
    owner_of = list(range(n_v))
    conv = a.mode.endswith("_conv")
    extra = {}

    fallback = a.mode.endswith("_fallback")
    if fallback and a.rank == a.world - 1:      # ONE rank's self-test (odd k) / mapping (even k) fails: every rank must end up on the collectives
        def broken(self, *args, **kw):
            raise RuntimeError("failure injected by the test")
        if a.k % 2:
            sharded.HipEngineAdapter.p2p_selftest = broken
        else:
            sharded.HipEngineAdapter.p2p_import = broken

    def run(p2p):
        if fallback and p2p:
            drv = sharded.ShardedSweep.create(prob, owner_of, a.rank, a.world, device_index=0, slice_p2p="auto")
            p2p = False
        else:
            drv = sharded.ShardedSweep.create(prob, owner_of, a.rank, a.world, device_index=0, slice_p2p=("auto" if p2p else False),
                                              **({"p2p_graph": True} if (p2p and a.graph) else {}))
        chunk = a.graph if p2p else 0
        assert drv.p2p == p2p and not drv.sliced and all(drv.replicated) and drv.replicate_gs == (not f_only)
        assert drv.collectives_per_sweep == (0 if p2p else 1 if f_only else (2 if drv._s_in_f else 3))
        if conv:
            done = drv.run(None, tol=a.tol, max_iters=a.sweeps, check_every=7)
        else:
            drv.run(a.sweeps // 2, graph_chunk=chunk)
            drv.run(a.sweeps - a.sweeps // 2, graph_chunk=chunk)          # two calls: state carries over
            done = a.sweeps
            assert not chunk or len(getattr(drv, "_graphs", {})) == 1
        drv.engine.synchronize(); torch.cuda.synchronize()
        kinds = ("F",) if f_only else ("F", "G", "S")
        raw = [b"".join(drv.engine.factor_tensor(v, kd).cpu().numpy().tobytes() for kd in kinds) for v in range(n_v)]
        errs = drv.mean_errors()
        res = drv.gather_results(0)
        drv.close()
        return done, raw, errs, res

    done, raw, errs, res = run(True)
    allr = [None] * a.world
    dist.all_gather_object(allr, raw)
    extra["mirrors_ok"] = np.array(all(allr[r][v] == allr[v][v] for r in range(a.world) for v in range(n_v)))
    if conv:
        alld = [None] * a.world
        dist.all_gather_object(alld, int(done))
        extra["sweeps_done"] = np.array(done)
        extra["same_stop"] = np.array(len(set(alld)) == 1)
    else:
        _, raw2, errs2, _ = run(False)
        same = [None] * a.world
        dist.all_gather_object(same, bool(raw == raw2 and np.array_equal(errs, errs2)))
        extra["bitwise_vs_collectives"] = np.array(all(same))
    if a.rank == 0:
        out = {"all_error": errs, **extra}
        for key, lst in res.items():
            for v, arr in enumerate(lst):
                out[f"{key}{v}"] = arr
        np.savez(a.out, **out)
    dist.barrier()
    dist.destroy_process_group()


def slice_main(a, dist, sharded):
    """Row-sliced chains (and the convergence mode of the replicated-chains layouts): stand-in engine on CPU or the HIP
    engine with the ranks sharing one GPU."""
    prob = build_problem_slice(a.world, a.k, a.n, a.m)
    n_v = a.world
    owner_of = list(range(n_v))
    conv = a.mode.endswith("_conv")
    extra = {}

    def make(sliced, **opts):
        if a.mode.startswith("cpu"):
            return sharded.ShardedSweep(prob, owner_of, a.rank, a.world, replicate_f=True, slice_chains=True,
                                        engine_factory=lambda p, owned: OracleSliceEngine(p, owned, a.rank, a.world))
        return sharded.ShardedSweep(prob, owner_of, a.rank, a.world, device_index=0, slice_chains=sliced, **opts)

    def results(drv):
        errs = drv.mean_errors()
        res = drv.gather_results(0)
        raw = None
        if not a.mode.startswith("cpu"):
            import torch
            drv.engine.synchronize(); torch.cuda.synchronize()
            raw = b"".join(drv.engine.factor_tensor(a.rank, kd).cpu().numpy().tobytes() for kd in ("F", "G", "S"))
        return errs, res, raw

    p2p = "p2p" in a.mode
    drv = make(a.mode != "gpu_gs_conv", kk_mode=2, **({"slice_p2p": True, "p2p_graph": bool(a.graph)} if p2p else {}))
    chunk = a.graph if p2p else 0
    assert drv.sliced == (a.mode != "gpu_gs_conv") and drv.replicate_gs and drv.p2p == p2p
    assert not p2p or drv.collectives_per_sweep == 0
    if conv:
        done = drv.run(None, tol=a.tol, max_iters=a.sweeps, check_every=7)
        extra["sweeps_done"] = np.array(done)
        alld = [None] * a.world
        dist.all_gather_object(alld, int(done))
        extra["same_stop"] = np.array(len(set(alld)) == 1)
    else:
        drv.run(a.sweeps // 2, graph_chunk=chunk)
        drv.run(a.sweeps - a.sweeps // 2, graph_chunk=chunk)          # two calls: state carries over
        assert not chunk or len(getattr(drv, "_graphs", {})) == 1
    errs, res, raw = results(drv)
    drv.close()
    if a.mode in ("gpu_slice", "gpu_slice_p2p"):   # the same run with the chains REPLICATED (same hand-off mode) / p2p: with the
        drv2 = make(a.mode == "gpu_slice_p2p", kk_mode=2)      # all-to-all exchange (gloo here): bitwise
        assert drv2.sliced == (a.mode == "gpu_slice_p2p") and drv2.replicate_gs and not drv2.p2p
        drv2.run(a.sweeps // 2)
        drv2.run(a.sweeps - a.sweeps // 2)
        errs2, res2, raw2 = results(drv2)
        drv2.close()
        same = [None] * a.world
        dist.all_gather_object(same, bool(raw == raw2 and np.array_equal(errs, errs2)))
        extra["bitwise_vs_replicated"] = np.array(all(same))
    if a.rank == 0:
        out = {"all_error": errs, **extra}
        for key, lst in res.items():
            for v, arr in enumerate(lst):
                out[f"{key}{v}"] = arr
        np.savez(a.out, **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
