#!/usr/bin/env python3
"""Data-term energy after n conjugate-gradient steps (307 200 points, 2000 nodes): how many steps the warp solve needs."""
import numpy as np, torch, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicfusion_amd import WarpField, synth
cfg = synth.CONFIGS["512"]
pos, sigma = synth.make_nodes(cfg)
rng = np.random.default_rng(5); N = cfg.cols * cfg.rows
src = (pos[rng.integers(0, len(pos), N)] + rng.normal(0, 0.03, (N, 3))).astype(np.float32)
dst = (src + 0.01 * np.sin(5 * src)).astype(np.float32)
for it in (5, 10, 15, 20, 30, 40, 80, 160):
    wf = WarpField(k=8); wf.init(pos, sigma=sigma)
    dq, en = wf.energy_data(torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda(), iters=it)
    torch.cuda.synchronize()
    e = en.cpu().numpy()
    print(it, "E0 %.6g  E %.6g  ratio %.6f" % (e[0], e[1], e[1] / e[0]))
