import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from resnmtf_amd import _lib, synth
T = sys.argv[1]
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libresnmtf_hip_ut{T}.so")
from resnmtf_amd.engine import Engine
prob = synth.config("c2"); n, m = prob.data[0].shape
e = Engine([n], [m], [prob.k]); e.set_view(0, prob.data[0]); e.set_restrictions(); e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
e.run(50); t0 = time.perf_counter(); e.run(500); dt = time.perf_counter() - t0
print(f"UPDATE_THREADS {T}: {500/dt:.0f} sweeps/s ({dt/500*1e6:.1f} us)")
