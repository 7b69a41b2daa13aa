import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resnmtf_amd.engine import Engine
from resnmtf_amd import synth
prob = synth.make_problem([(300, 200)], 5)
e = Engine([300], [200], [5]); e.set_view(0, prob.data[0]); e.set_restrictions(); e.set_factors(0, prob.init_f[0], prob.init_s[0], prob.init_g[0])
print("errs", e.run(5)[-1])
import torch
st = torch.cuda.Stream()
torch.cuda.synchronize()
print("torch after library: ok", torch.cuda.is_available(), torch.zeros(4, device="cuda").sum().item())
e.close()
