import sys, os, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from dynamicfusion_amd import Intr, TsdfVolume, WarpField, sharded, synth, upload_u16
from scene import Scene
from test_gpu_fullsize import setup
cfg = synth.CONFIGS["256"]; intr = Intr(*cfg.intr); sc = Scene(cfg, n_frames=2); X, Y, Z = cfg.dims
a = setup(cfg); d = upload_u16(sc.dists[0]); a.integrate(d, sc.cam_poses[0], intr); a.integrate(upload_u16(sc.dists[1]), sc.cam_poses[1], intr)
fp = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda"); fn = torch.empty_like(fp); fk = torch.empty((cfg.rows, cfg.cols), dtype=torch.int32, device="cuda")
a.raycast(sc.cam_poses[1], intr, fp, fn, keys=fk)
world = 8; halo = sharded.halo_planes(sc.trunc, cfg.raycast_step_factor, cfg.gradient_delta_factor, float(sc.vs[2])); print("halo", halo)
for r in range(world):
    zs, zn = sharded.slab_range(Z, r, world)
    v = setup(cfg, slab=(zs, zn, halo)); v.data().copy_(a.data()[v.z_store0:v.z_store0 + v.z_store_n])
    p, n = torch.empty_like(fp), torch.empty_like(fp); k32 = torch.empty_like(fk)
    v.raycast(sc.cam_poses[1], intr, p, n, keys=k32); torch.cuda.synchronize()
    mine = (k32 == fk) & (fk != -1)
    bad = mine & (torch.isnan(p[..., 0]) != torch.isnan(fp[..., 0]))
    diff = mine & ~torch.isnan(fp[..., 0]) & ((p != fp).any(-1) | (n != fn).any(-1))
    print("rank", r, "slab", zs, zn, "store", v.z_store0, v.z_store_n, "mine", int(mine.sum()), "nanmask bad", int(bad.sum()), "value diff", int(diff.sum()))
    if int(bad.sum()):
        ys, xs = torch.nonzero(bad, as_tuple=True); y, x = int(ys[0]), int(xs[0])
        print("  pixel", y, x, "key", int(fk[y, x]) & 0xffffffff, "full p", fp[y, x].tolist(), "slab p", p[y, x].tolist(), "full n", fn[y, x].tolist(), "slab n", n[y, x].tolist())
