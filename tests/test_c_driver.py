"""The C-ABI from plain C: examples/c_driver.c is compiled with gcc -std=c99 -pedantic against
include/resnmtf_hip.h and libresnmtf_hip.so (what the R-side shim r/shim.c does, minus R).
CPU: it compiles and links.  GPU: it reproduces a golden fixture and the oracle's coupled run."""
import os
import struct
import subprocess

import numpy as np
import pytest

from helpers import ROOT, golden_problem, load_golden, rel_fro, run_oracle
from resnmtf_amd import naming
from test_gpu_parity import check_clusters


def build_driver(tmp_path):
    exe = str(tmp_path / "c_driver")
    libdir = os.path.join(ROOT, "resnmtf_amd")
    if not os.path.exists(os.path.join(libdir, "libresnmtf_hip.so")):
        pytest.skip("libresnmtf_hip.so not built")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_driver.c"), "-L" + libdir, "-lresnmtf_hip", "-Wl,-rpath," + libdir, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def write_problem(path, prob, n_iters):
    n_v = len(prob.data)
    rs, cs = naming.shared_names(prob.row_names), naming.shared_names(prob.col_names)
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", n_v, n_iters))
        for v in range(n_v):
            n, m = prob.data[v].shape
            f.write(struct.pack("<iii", n, m, prob.init_f[v].shape[1]))
        for mat in (prob.phi, prob.xi, prob.psi):
            f.write(np.asfortranarray(mat, dtype=np.float64).tobytes(order="F"))
        for v in range(n_v):
            for mat in (prob.data[v], prob.init_f[v], prob.init_s[v], prob.init_g[v]):
                f.write(np.asarray(mat, dtype=np.float64).tobytes(order="F"))
        for v in range(n_v):
            for w in range(n_v):
                if v == w:
                    continue
                for names, shared in ((prob.row_names, rs), (prob.col_names, cs)):
                    iv, iw = naming.index_pairs(names[v], names[w], shared[v].get(w))
                    if iv is None:
                        f.write(struct.pack("<i", -1))
                    else:
                        f.write(struct.pack("<i", len(iv)))
                        f.write(np.asarray(iv, dtype=np.int32).tobytes()); f.write(np.asarray(iw, dtype=np.int32).tobytes())


def read_result(path, prob, n_iters):
    raw = np.fromfile(path, dtype=np.float64)
    out = {"All_Error": raw[:n_iters], "f": [], "s": [], "g": [], "rc": [], "cc": []}
    pos = n_iters
    for v in range(len(prob.data)):
        n, m = prob.data[v].shape
        k = prob.init_f[v].shape[1]
        for key, shape in (("f", (n, k)), ("s", (k, k)), ("g", (m, k)), ("rc", (n, k)), ("cc", (m, k))):
            cnt = shape[0] * shape[1]
            out[key].append(raw[pos:pos + cnt].reshape(shape, order="F")); pos += cnt
    assert pos == raw.size
    return out


def test_c_driver_compiles_as_strict_c99(tmp_path):
    build_driver(tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["g1_single_60x40_k3", "g3_three_views_phi_psi_xi"])
def test_c_driver_matches_golden(tmp_path, name):
    exe = build_driver(tmp_path)
    g = load_golden(name)
    prob = golden_problem(g)
    pin, pout = str(tmp_path / "p.bin"), str(tmp_path / "r.bin")
    write_problem(pin, prob, g["n_iters"])
    r = subprocess.run([exe, pin, pout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    res = read_result(pout, prob, g["n_iters"])
    np.testing.assert_allclose(res["All_Error"], g["all_error"], atol=2e-5, rtol=1e-4)
    for v in range(g["n_views"]):
        assert rel_fro(res["f"][v], g["out_f"][v]) < 2e-5
        assert rel_fro(res["g"][v], g["out_g"][v]) < 2e-5
        assert rel_fro(res["s"][v], g["out_s"][v]) < 1e-4
        # binary cluster matrices through the C-ABI caller: identical, a difference allowed only on the 1/n threshold
        check_clusters(res["rc"][v], res["cc"][v], g["out_f"][v], g["out_s"][v], g["out_g"][v], g["rc"][v], g["cc"][v], res["s"][v], view=v)
