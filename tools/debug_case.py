import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from conftest import load_golden, make_table
from smudgeplot_amd import engine
import brute
for name in sys.argv[1:]:
    g = load_golden(name)
    want = brute.hetmers_plot(g["packed"], g["counts"], g["k"])
    for mode in ("hash", "exact", "none"):
        plot, st = engine.hetmers_run(make_table(g), symcheck=mode)
        d = plot - want
        s, m = np.nonzero(d)
        print(name, mode, "path", st["path"], "nreq", st["nrequests"], "npairs", st["npairs"], "want", int(want.sum()),
              "ndiff", len(s), [(int(a), int(b), int(d[a, b])) for a, b in list(zip(s, m))[:6]])
