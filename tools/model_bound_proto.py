#!/usr/bin/env python3
"""Which part of the block model's bound is loose (round 5): the model against exact translation ranges, exact rotation ranges, the exact
optimum under sum lambda = 1 (fractional knapsack) and T = (sum w) * sum lambda_i t_i.  Derived from tools/model_variants_proto.py:

CPU (numpy) model of the warped sweep's launch-plan verdicts, on the headline scene -- what dfusion_warp_blocks.h was designed from.

For a sample of 8x8x8 blocks it compares, against the exact answer (does the oracle update any voxel of the block?):
  ball   df_tile_culled: a ball around the UNWARPED block centre sized for the largest motion any node could cause
  exact  the exact bounding box of the block's warped voxels through the box tests (the floor for any box-based test)
  model  the per-block blend model: intervals of every union node's normalised and raw weight over the block's voxels (rounded to
         halves as stored), the frame's node transforms, interval arithmetic through the Gibbs-vector form of the blend
Round 3 on the 512^3 headline config (every 4th block of every 4th layer): ideal 27.6 %, exact 28.7 %, model 34.9 %, ball 56.6 % alive;
measured on the GPU afterwards: swept / updated voxels 2.42 -> 1.53 (= 34.9 % / 22.8 % updated voxels).  Runs the CPU oracle (tests/):
test infrastructure, not product code.  Usage: tools/block_model_proto.py [layer,layer,...] [stride]      (about 5 s per layer)"""
import os, sys, time, collections
import numpy as np
from scipy.spatial import cKDTree
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_lib as O                                       # noqa: E402
from dynamicfusion_amd import synth                          # noqa: E402
from scene import Scene                                      # noqa: E402

cfg = synth.CONFIGS["512"]
sc = Scene(cfg, n_frames=2); f = 1
X, Y, Z = cfg.dims; vs = sc.vs.astype(np.float64)
pose = sc.pose.astype(np.float64); w2c = sc.world2cam(f).astype(np.float64)
dq = sc.dqs[f].astype(np.float64); r = dq[:, :4]; d = dq[:, 4:]
NU_MAX = 16


def qmul(a, b):
    w = a[..., 0] * b[..., 0] - (a[..., 1:] * b[..., 1:]).sum(-1)
    v = a[..., :1] * b[..., 1:] + b[..., :1] * a[..., 1:] + np.cross(a[..., 1:], b[..., 1:])
    return np.concatenate([w[..., None], v], -1)


rc = r.copy(); rc[:, 1:] *= -1
tn = qmul(2 * d, rc / np.linalg.norm(r, axis=1, keepdims=True))[:, 1:]     # node translations (dual_quaternion.hpp:120-125)
sig = sc.sigma.astype(np.float64)
tree = cKDTree(sc.pos.astype(np.float64))
fx, fy, cx, cy = [float(v) for v in cfg.intr]
dists = sc.dists[f].view(np.float16).astype(np.float64)
trunc, K = sc.trunc, cfg.k
max_t = np.linalg.norm(tn, axis=1).max(); sin_half = (np.linalg.norm(r[:, 1:], axis=1) / np.linalg.norm(r, axis=1)).max()
origin_cam = np.linalg.norm(w2c[:3, 3]); tile_r = 0.5 * np.linalg.norm(7 * vs)

print("oracle: one warped frame at 512^3 ...", flush=True)
vol = sc.new_volume()
O.integrate_warped(sc.dists[f], vol, sc.ovol(vol), synth.aff12(sc.pose), synth.aff12(sc.world2cam(f)), sc.intr, sc.pos, sc.dqs[f], sc.sigma, cfg.k)
upd = (vol >> 16) != 0
print("updated voxels: %.1f %%" % (100 * upd.mean()))


def footprint_dead(C):
    """camera-frame box C = [(lo, hi)] * 3 through the tests of df_block_box_dead"""
    if C[2][1] <= 0: return True
    if C[2][0] > 0.05:
        zl, zh = C[2]
        ulo = fx * min(C[0][0] / zl, C[0][0] / zh) + cx - 2; uhi = fx * max(C[0][1] / zl, C[0][1] / zh) + cx + 2
        vlo = fy * min(C[1][0] / zl, C[1][0] / zh) + cy - 2; vhi = fy * max(C[1][1] / zl, C[1][1] / zh) + cy + 2
        if uhi < 0 or vhi < 0 or ulo > cfg.cols - 1 or vlo > cfg.rows - 1: return True
        m = dists[int(max(vlo, 0)):int(min(vhi, cfg.rows - 1)) + 1, int(max(ulo, 0)):int(min(uhi, cfg.cols - 1)) + 1].max()
        if m == 0: return True
        rmin = np.sqrt(sum(0 if (lo <= 0 <= hi) else min(lo * lo, hi * hi) for lo, hi in C))
        if rmin > m * 1.002 + trunc: return True
    return False


def to_cam(P):
    R, t = w2c[:3, :3], w2c[:3, 3]
    C = []
    for i in range(3):
        lo = hi = t[i]
        for j in range(3):
            a, b = R[i, j] * P[j][0], R[i, j] * P[j][1]
            lo += min(a, b); hi += max(a, b)
        C.append((lo - 1e-3, hi + 1e-3))
    return C


def imul(a, b):
    p = [a[0] * b[0], a[0] * b[1], a[1] * b[0], a[1] * b[1]]; return (min(p), max(p))


def box_dead(qlo, qhi, ulo, uhi, Tlo, Thi):
    """interval arithmetic through R(u) q = q + 2 / (1 + |u|^2) (u x q + u x (u x q)) + T"""
    U = [(ulo[i], uhi[i]) for i in range(3)]; Q = [(qlo[i], qhi[i]) for i in range(3)]
    sub = lambda a, b: (a[0] - b[1], a[1] - b[0]); add = lambda a, b: (a[0] + b[0], a[1] + b[1])
    cross = lambda A, B: [sub(imul(A[1], B[2]), imul(A[2], B[1])), sub(imul(A[2], B[0]), imul(A[0], B[2])), sub(imul(A[0], B[1]), imul(A[1], B[0]))]
    c1 = cross(U, Q); c2 = cross(U, c1)
    u2lo = sum(0 if (a <= 0 <= b) else min(a * a, b * b) for a, b in U); u2hi = sum(max(a * a, b * b) for a, b in U)
    S = (2 / (1 + u2hi), 2 / (1 + u2lo))
    return footprint_dead(to_cam([add(add(Q[i], imul(S, add(c1[i], c2[i]))), (Tlo[i], Thi[i])) for i in range(3)]))


def ball_dead(c, wk):
    cn = np.linalg.norm(c)
    rho = (tile_r + 2 * sin_half * (cn + tile_r) + wk * max_t) * 1.002 + 1e-3
    rho_r = min(rho, (tile_r + 2 * sin_half * origin_cam + wk * max_t) * 1.002 + 1e-3)
    cc = w2c[:3, :3] @ c + w2c[:3, 3]
    rmin = np.linalg.norm(cc) - rho_r
    if cc[2] + rho <= 0: return True
    if (fx * cc[0] + cx * cc[2]) / np.hypot(fx, cx) < -rho or (-fx * cc[0] + (cfg.cols - cx) * cc[2]) / np.hypot(fx, cfg.cols - cx) < -rho: return True
    if (fy * cc[1] + cy * cc[2]) / np.hypot(fy, cy) < -rho or (-fy * cc[1] + (cfg.rows - cy) * cc[2]) / np.hypot(fy, cfg.rows - cy) < -rho: return True
    if cc[2] - rho > 0.05:
        C = [(cc[i] - rho, cc[i] + rho) for i in range(3)]
        zl, zh = C[2]
        ulo = fx * min(C[0][0] / zl, C[0][0] / zh) + cx - 2; uhi = fx * max(C[0][1] / zl, C[0][1] / zh) + cx + 2
        vlo = fy * min(C[1][0] / zl, C[1][0] / zh) + cy - 2; vhi = fy * max(C[1][1] / zl, C[1][1] / zh) + cy + 2
        if uhi < 0 or vhi < 0 or ulo > cfg.cols - 1 or vlo > cfg.rows - 1: return True
        m = dists[int(max(vlo, 0)):int(min(vhi, cfg.rows - 1)) + 1, int(max(ulo, 0)):int(min(uhi, cfg.cols - 1)) + 1].max()
        return m == 0 or rmin > m * 1.002 + trunc
    return rmin > dists.max() * 1.002 + trunc



def model_iv(bi, bw, un=None):
    """lambda / w intervals (half-rounded) of node set `un` over the voxels (bi, bw)"""
    if un is None: un = np.unique(bi)
    wr = np.zeros((bi.shape[0], len(un)))
    for k_, node in enumerate(un): wr[:, k_] = np.where(bi == node, bw, 0).max(1)
    lam = wr / wr.sum(1, keepdims=True)
    lmid = (0.5 * (lam.min(0) + lam.max(0))).astype(np.float16).astype(np.float64); lhw = np.maximum(lam.max(0) - lmid, lmid - lam.min(0)) * 1.004 + 1e-7
    wmid = (0.5 * (wr.min(0) + wr.max(0))).astype(np.float16).astype(np.float64); whw = np.maximum(wr.max(0) - wmid, wmid - wr.min(0)) * 1.004 + 1e-7
    return un, lmid, lhw, wmid, whw


def uT(un, lmid, lhw, wmid, whw):
    e0 = np.argmax(lhw); smin, smax = r[un, 0].min(), r[un, 0].max()
    ulo = np.zeros(3); uhi = np.zeros(3); Tlo = np.zeros(3); Thi = np.zeros(3)
    for c in range(3):
        v = r[un, 1 + c]; mid = (lmid * v).sum() + (1 - lmid.sum()) * v[e0]; e = (lhw * np.abs(v - v[e0])).sum()
        ulo[c] = min((mid - e) / smin, (mid - e) / smax); uhi[c] = max((mid + e) / smin, (mid + e) / smax)
        tv = tn[un, c]; m_ = (wmid * tv).sum(); e_ = (whw * np.abs(tv)).sum(); Tlo[c], Thi[c] = m_ - e_ - 1e-3, m_ + e_ + 1e-3
    return ulo, uhi, Tlo, Thi





layers = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(2, 64, 4))
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = collections.Counter()
def lp_range(mid, hw, a, total=1.0):
    """exact range of sum_i lam_i a_i subject to lam_i in [mid_i - hw_i, mid_i + hw_i], sum lam_i = total (fractional knapsack)"""
    lo_b = mid - hw; hi_b = mid + hw
    out = []
    for sign in (1.0, -1.0):
        order = np.argsort(-sign * a)              # maximise sign * sum lam a: give the most to the largest sign*a
        lam = lo_b.copy(); rest = total - lam.sum()
        if rest < 0: rest = 0.0                    # (infeasible by rounding: fall back to the box bound below)
        for i in order:
            g = min(hi_b[i] - lo_b[i], rest); lam[i] += g; rest -= g
            if rest <= 0: break
        out.append((lam * a).sum())
    return out[1], out[0]
for bz in layers:
    t0 = time.time()
    zz, yy, xx = np.meshgrid(np.arange(bz * 8, bz * 8 + 8), np.arange(Y), np.arange(X), indexing="ij")
    p = np.stack([xx * vs[0], yy * vs[1], zz * vs[2]], -1).reshape(-1, 3) @ pose[:3, :3].T + pose[:3, 3]
    dd, ii = tree.query(p, k=K)
    w = np.exp(-(dd ** 2) / (2 * sig[ii] ** 2))
    ii = ii.reshape(8, Y, X, K); w = w.reshape(8, Y, X, K); p = p.reshape(8, Y, X, 3)
    for by in range(0, Y // 8, stride):
        for bx in range(0, X // 8, stride):
            sl = (slice(None), slice(by * 8, by * 8 + 8), slice(bx * 8, bx * 8 + 8))
            bi = ii[sl]; bw = w[sl]; bp = p[sl]
            u8 = upd[bz * 8:bz * 8 + 8, by * 8:by * 8 + 8, bx * 8:bx * 8 + 8]
            fb = lambda a: a.reshape(512, -1)
            ball = not ball_dead(0.5 * (fb(bp).min(0) + fb(bp).max(0)), min(K, fb(bw).sum(1).max() * 1.0001))
            un = np.unique(bi); have = len(un) <= NU_MAX
            n["blocks"] += 1; n["ideal"] += u8.any(); n["ball"] += ball
            if not ball: continue
            keys = ("model", "exactNT", "mu_eT", "eu_mT", "mu_sT", "lpu_sT", "lpu_lpsT", "lpu_eT", "pca1", "pca2", "pca3", "pca4")
            if not have:
                for k_ in keys: n[k_] += 1
                continue
            M = model_iv(fb(bi), fb(bw)); un_, lmid, lhw, wmid, whw = M
            ulo, uhi, Tlo, Thi = uT(*M)
            Bi, Bw, Bp = fb(bi).reshape(512, K), fb(bw).reshape(512, K), fb(bp)
            qlo, qhi = Bp.min(0), Bp.max(0)
            mq = (Bw[:, :, None] * r[Bi]).sum(1); uu = mq[:, 1:] / mq[:, :1]
            T = (Bw[:, :, None] * tn[Bi]).sum(1)
            eu = (uu.min(0) - 1e-6, uu.max(0) + 1e-6); eT = (T.min(0) - 1e-3, T.max(0) + 1e-3)
            n["model"] += not box_dead(qlo, qhi, ulo, uhi, Tlo, Thi)
            n["exactNT"] += not box_dead(qlo, qhi, eu[0], eu[1], eT[0], eT[1])
            n["mu_eT"] += not box_dead(qlo, qhi, ulo, uhi, eT[0], eT[1])
            n["eu_mT"] += not box_dead(qlo, qhi, eu[0], eu[1], Tlo, Thi)
            # T = s * sum lam_i t_i, s = sum w in [slo, shi] (stored as halves, rounded outward)
            s = Bw.sum(1); slo = float(np.float16(s.min() * 0.999)); shi = float(np.float16(s.max() * 1.001)) ;
            if slo > s.min(): slo = s.min() * 0.998
            if shi < s.max(): shi = s.max() * 1.002
            e0 = np.argmax(lhw)
            def sT(lp):
                lo = np.zeros(3); hi = np.zeros(3)
                for c in range(3):
                    tv = tn[un_, c]
                    if lp: a_lo, a_hi = lp_range(lmid, lhw, tv)
                    else:
                        mid = (lmid * tv).sum() + (1 - lmid.sum()) * tv[e0]; e = (lhw * np.abs(tv - tv[e0])).sum(); a_lo, a_hi = mid - e, mid + e
                    cands = [slo * a_lo, slo * a_hi, shi * a_lo, shi * a_hi]
                    lo[c], hi[c] = min(cands) - 1e-3, max(cands) + 1e-3
                return lo, hi
            def lpu():
                lo = np.zeros(3); hi = np.zeros(3)
                smin, smax = r[un_, 0].min(), r[un_, 0].max()
                d_lo, d_hi = lp_range(lmid, lhw, r[un_, 0]); d_lo = max(d_lo, smin); d_hi = min(d_hi, smax)
                for c in range(3):
                    n_lo, n_hi = lp_range(lmid, lhw, r[un_, 1 + c])
                    cands = [n_lo / d_lo, n_lo / d_hi, n_hi / d_lo, n_hi / d_hi]
                    lo[c], hi[c] = min(cands), max(cands)
                return lo, hi
            sT0 = sT(False); sT1 = sT(True); lu = lpu()
            # PCA model of lambda over the block's voxels: lambda(q) = m + sum_j c_j(q) E_j + R(q); ranges of c_j, max |R_i|
            wr = np.zeros((512, len(un_)))
            for k_, node in enumerate(un_): wr[:, k_] = np.where(Bi == node, Bw, 0).max(1)
            L = wr / wr.sum(1, keepdims=True); m_ = L.mean(0); Uu, Ss, Vt = np.linalg.svd(L - m_, full_matrices=False)
            for rr in (1, 2, 3, 4):
                E = Vt[:rr].astype(np.float16).astype(np.float64); E -= E.mean(1, keepdims=True)           # (rows sum to 0, kept exactly after rounding)
                mm = m_.astype(np.float16).astype(np.float64); mm += (1 - mm.sum()) / len(mm)
                Cc = (L - mm) @ np.linalg.pinv(E); Rr = (L - mm) - Cc @ E
                clo, chi = Cc.min(0) - 1e-6, Cc.max(0) + 1e-6; res = np.abs(Rr).max(0) * 1.004 + 1e-7
                e0r = np.argmax(res)
                def rng(a):
                    base = (mm * a).sum(); lo = hi = base
                    for j in range(rr):
                        g = (E[j] * a).sum(); x, y = clo[j] * g, chi[j] * g; lo += min(x, y); hi += max(x, y)
                    e = (res * np.abs(a - a[e0r])).sum()
                    return lo - e, hi + e
                d_lo, d_hi = rng(r[un_, 0]); d_lo = max(d_lo, r[un_, 0].min()); d_hi = min(d_hi, r[un_, 0].max())
                plo = np.zeros(3); phi = np.zeros(3)
                for c in range(3):
                    n_lo, n_hi = rng(r[un_, 1 + c]); cands = [n_lo / d_lo, n_lo / d_hi, n_hi / d_lo, n_hi / d_hi]; plo[c], phi[c] = min(cands), max(cands)
                n["pca%d" % rr] += not box_dead(qlo, qhi, plo, phi, Tlo, Thi)
            n["mu_sT"] += not box_dead(qlo, qhi, ulo, uhi, sT0[0], sT0[1])
            n["lpu_sT"] += not box_dead(qlo, qhi, lu[0], lu[1], sT0[0], sT0[1])
            n["lpu_lpsT"] += not box_dead(qlo, qhi, lu[0], lu[1], sT1[0], sT1[1])
            n["lpu_eT"] += not box_dead(qlo, qhi, lu[0], lu[1], eT[0], eT[1])
    b = n["blocks"]
    print("layer %2d (%.0f s) %5d blocks alive: ideal %.3f exactNT %.3f | model %.3f | model-u+exact-T %.3f exact-u+model-T %.3f | model-u + s*lam T %.3f | LP-u + s*lam T %.3f | LP-u + s*LP T %.3f | LP-u + exact T %.3f | PCA-u r=1..4 + model T %.3f %.3f %.3f %.3f | ball %.3f" % (
        bz, time.time() - t0, b, n["ideal"] / b, n["exactNT"] / b, n["model"] / b, n["mu_eT"] / b, n["eu_mT"] / b, n["mu_sT"] / b, n["lpu_sT"] / b, n["lpu_lpsT"] / b, n["lpu_eT"] / b, n["pca1"] / b, n["pca2"] / b, n["pca3"] / b, n["pca4"] / b, n["ball"] / b), flush=True)
