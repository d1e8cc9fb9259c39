"""PMV_BA_CHECK=1: one BA problem through the multi-kernel and the single-workgroup solver, differences on stderr."""
import sys, os, importlib, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["PMV_BA_CHECK"] = "1"
import scenes
pmv = importlib.import_module("practical-multi-view_amd")
ctx = pmv.Context(64, 64, n_slots=1)
for nc, npts, iters in ((5, 300, 1), (5, 300, 2), (5, 300, 5), (10, 1000, 5)):
    P = scenes.ba_problem(1, nc=nc, npts=npts)
    _, _, s = ctx.ba_solve(P["cams"], P["pts"], P["obs"], P["cam_idx"], P["pt_idx"], scenes.K, 1.0, iters)
    print(nc, npts, iters, "->", s.initial_cost, s.final_cost, s.iterations, s.successful_steps, s.termination)
