#!/usr/bin/env python3
"""One-process timing sweep of the hot-path kernels (HIP events), used to pick launch variants."""
import os, sys, json, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from dynamicfusion_amd import Intr, TsdfVolume, WarpField, capi, compute_dists, synth, upload_u16

def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "512"
    cfg = synth.CONFIGS[name]; intr = Intr(*cfg.intr)
    st = torch.cuda.current_stream().cuda_stream
    print("config", cfg.name, torch.cuda.get_device_name(0))
    src = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
    st = torch.cuda.current_stream().cuda_stream
    ms = timeit(lambda: capi.check(capi.lib().dfusion_copy_bandwidth_probe(dst.data_ptr(), src.data_ptr(), 1 << 30, st)))
    print("copy probe: %.1f GB/s (r+w)" % (2 * (1 << 30) / ms / 1e6))
    ms = timeit(lambda: dst.copy_(src)); print("torch copy: %.1f GB/s" % (2 * (1 << 30) / ms / 1e6))
    sink = torch.zeros(4, dtype=torch.int32, device="cuda")
    ms = timeit(lambda: capi.check(capi.lib().dfusion_read_bandwidth_probe(src.data_ptr(), 1 << 30, sink.data_ptr(), st)))
    print("read probe: %.1f GB/s" % ((1 << 30) / ms / 1e6))
    del src, dst
    depth = upload_u16(synth.depth_frame(cfg, 0)); dists = compute_dists(depth, intr)
    cam = synth.camera_pose(cfg, 1)
    vol = TsdfVolume(cfg.dims); vol.setSize([cfg.size]*3); vol.setTruncDist(cfg.trunc_dist); vol.setMaxWeight(cfg.max_weight); vol.setPose(cfg.volume_pose)
    vol.setRaycastStepFactor(cfg.raycast_step_factor); vol.setGradientDeltaFactor(cfg.gradient_delta_factor)
    nvox = np.prod(cfg.dims)
    ms = timeit(lambda: vol.clear()); print("clear: %.3f ms  %.1f GB/s" % (ms, 4 * nvox / ms / 1e6))
    n = torch.zeros(1, dtype=torch.int64, device="cuda")
    vol.integrate(dists, cam, intr, n_updated=n); nupd = int(n.item()); print("rigid n_upd", nupd, nupd / nvox)
    ms = timeit(lambda: vol.integrate(dists, cam, intr, sync=False))
    print("rigid : %.3f ms  alg %.0f GB/s  sweep %.0f GB/s" % (ms, 8 * nupd / ms / 1e6, 8 * nvox / ms / 1e6))
    if cfg.nodes:
        pos, sigma = synth.make_nodes(cfg); dq = synth.node_transforms(cfg, 1)
        wf = WarpField(k=cfg.k); wf.init(pos, sigma=sigma, transforms=dq)
        t0 = time.time(); wf.ensure_index(vol, cfg.k); torch.cuda.synchronize(); print("index build %.3f s" % (time.time() - t0))
        vol.clear(); n.zero_()
        vol.integrate_warped(dists, cam, intr, wf, n_updated=n); nw = int(n.item()); print("warped n_upd", nw, nw / nvox)
        for tab, wts in ((True, True), (True, False), (False, False)):
            for cull in (True, False):
                ms = timeit(lambda: vol.integrate_warped(dists, cam, intr, wf, cull=cull, use_table=tab, use_weights=wts, sync=False), iters=5, warm=1)
                print("warped knn_table=%s w_table=%s cull=%s : %.3f ms  alg %.0f GB/s" % (tab, wts, cull, ms, 8 * nw / ms / 1e6))
        for kw in (dict(pipelined=False), dict(use_lds=False)):
            ms = timeit(lambda: vol.integrate_warped(dists, cam, intr, wf, sync=False, **kw), iters=5, warm=1)
            print("warped tables %s : %.3f ms" % (kw, ms))
    buf = torch.empty((1 << 24, 4), dtype=torch.float32, device="cuda")
    ms = timeit(lambda: vol.fetchCloud(buf)); npts = vol.last_cloud_count_
    print("fetchCloud (incl. count readback): %.3f ms, %d points, scan %.0f GB/s" % (ms, npts, 4 * nvox / ms / 1e6))
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    from dynamicfusion_amd.synth import aff12
    ms = timeit(lambda: capi.check(capi.lib().dfusion_extract_cloud(vol.c_volume(), None, capi.floats(aff12(vol.getPose())), buf.data_ptr(), buf.shape[0], cnt.data_ptr(), st)), iters=20)
    print("extract kernel only: %.3f ms, scan %.0f GB/s" % (ms, 4 * nvox / ms / 1e6))
    cl = vol.fetchCloud(buf); ms = timeit(lambda: vol.fetchNormals(cl)); print("fetchNormals: %.3f ms" % ms)
    pts = torch.empty((cfg.rows, cfg.cols, 4), dtype=torch.float32, device="cuda"); nrm = torch.empty_like(pts)
    ms = timeit(lambda: vol.raycast(cam, intr, pts, nrm)); print("raycast: %.3f ms, hits %.3f" % (ms, float(torch.isfinite(pts[..., 0]).float().mean())))

if __name__ == "__main__":
    main()
