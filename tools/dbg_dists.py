import sys, os, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_lib as O
from dynamicfusion_amd import Intr, compute_dists, synth, upload_u16, download_u16
cfg = synth.Config(64, 1.0, nodes=0)
depth = synth.depth_frame(cfg, 0); intr = np.array(cfg.intr, np.float32)
ref = O.compute_dists(depth, intr)
g = download_u16(compute_dists(upload_u16(depth), Intr(*cfg.intr))); torch.cuda.synchronize()
bad = np.argwhere(g != ref); print("mismatches", len(bad), "of", g.size)
for y, x in bad[:10]:
    xl = (np.float32(x) - intr[2]) * (np.float32(1) / intr[0]); yl = (np.float32(y) - intr[3]) * (np.float32(1) / intr[1])
    lam = np.sqrt(np.float32(xl * xl + yl * yl + np.float32(1)))
    v = np.float32(depth[y, x]) * lam; w = np.float32(v * np.float32(0.001))
    print(y, x, depth[y, x], hex(g[y, x]), hex(ref[y, x]), "f32 product", float(w), "exact", float(np.float64(v) * np.float64(np.float32(0.001))), np.float16(w).view(np.uint16))
