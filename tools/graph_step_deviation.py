"""Relative deviation over the first 16 learner steps: graph replay vs eager, eager vs eager, graph vs graph (tests/test_graph_step_gpu.py _series).  usage: python tools/graph_step_deviation.py [dqn|iqn]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
if os.environ.get("DET") == "1":
    torch.backends.cudnn.deterministic = True
if os.environ.get("DET") == "2":            # this library's weight-gradient kernels (fixed summation order) at every frame count
    from rltime_amd.models.torch import fused
    fused._CONV_WRW_MIN_WORK = 0
from test_graph_step_gpu import _series
kind = sys.argv[1] if len(sys.argv) > 1 else "dqn"
a = _series(kind, True); b = _series(kind, "no-capture"); b2 = _series(kind, "no-capture"); a2 = _series(kind, True)
for k in ("qloss", "grad_norm"):
    for name, x, y in (("graph-eager", a, b), ("eager-eager", b, b2), ("graph-graph", a, a2)):
        d = np.abs(x[k][:24] - y[k][:24]) / np.abs(y[k][:24])
        print(kind, k, name, " ".join("%.0e" % v for v in d))

for name, x, y in (("graph-eager", a, b), ("eager-eager", b, b2), ("graph-graph", a, a2)):
    n = len(x["qloss"])
    first = next((i for i in range(n) if x["qloss"][i] != y["qloss"][i] or x["grad_norm"][i] != y["grad_norm"][i]), n)
    same_params = all(torch.equal(p, q) for p, q in zip(x["params"], y["params"]))
    print(kind, name, "steps", n, "first differing step", first, "final parameters identical", same_params)
