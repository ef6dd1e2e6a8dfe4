"""Deterministic synthetic inputs for the hot path: analytic depth frames, camera
poses, warp nodes and node transforms.  Pure numpy, no files, no oracle.

Scene (metres, world == canonical frame; camera frame 0 == world): the volume is the cube
[-s/2, s/2] x [-s/2, s/2] x [0.5, 0.5+s] (volume_pose = translate(-s/2, -s/2, 0.5),
/root/reference/kfusion/src/kinfu.cpp:27,67); a sphere of radius 0.2 s sits at the volume centre in
front of a back plane at z = 0.5 + 0.85 s.  Frame f looks from a camera rotated by 0.25 deg * f
about the vertical axis through the volume centre.
"""
import numpy as np

F32 = np.float32


class Config:
    """One BASELINE.json config: volume, image and warp-field sizes."""

    def __init__(self, dims, size, cols=640, rows=480, nodes=0, k=8, name=None):
        self.dims = (int(dims),) * 3 if np.isscalar(dims) else tuple(int(d) for d in dims)
        self.size = float(size)
        self.cols, self.rows = int(cols), int(rows)
        scale = self.cols / 640.0
        # kinfu.cpp:23 default_params_dynamicfusion intrinsics (x2 at 1280x960)
        self.intr = (570.342 * scale, 570.342 * scale, 320.0 * scale, 240.0 * scale)
        self.nodes, self.k = int(nodes), int(k)
        self.trunc_dist, self.max_weight = 0.04, 64                  # kinfu.cpp:39-40
        self.raycast_step_factor, self.gradient_delta_factor = 0.75, 0.5   # kinfu.cpp:42-43
        self.name = name or "%d^3" % self.dims[0]

    @property
    def volume_pose(self):
        return translation(-self.size / 2, -self.size / 2, 0.5)


CONFIGS = {
    "cpu128": Config(128, 1.0, nodes=8, k=8, name="128^3 rigid (identity warp)"),
    "256": Config(256, 1.0, nodes=500, k=4, name="256^3, ~500 nodes, k=4"),
    "512": Config(512, 3.0, nodes=2000, k=8, name="512^3, ~2000 nodes, k=8 (headline)"),
    "1024": Config(1024, 3.0, cols=1280, rows=960, nodes=5000, k=8, name="1024^3, 1280x960, ~5000 nodes"),
}


# ------------------------------------------------------------------ affine helpers (4x4 float32, row-major)
def identity():
    return np.eye(4, dtype=F32)


def translation(x, y, z):
    m = np.eye(4, dtype=F32)
    m[:3, 3] = (x, y, z)
    return m


def rot_y_about(angle_rad, centre):
    c, s = np.cos(angle_rad), np.sin(angle_rad)
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)
    m = np.eye(4, dtype=np.float64)
    m[:3, :3] = R
    m[:3, 3] = np.asarray(centre, np.float64) - R @ np.asarray(centre, np.float64)
    return m.astype(F32)


def affine_inv(m):
    """Affine3f::inv() stand-in (float64 general inverse rounded to float32)."""
    return np.linalg.inv(m.astype(np.float64)).astype(F32)


def affine_mul(a, b):
    return (a.astype(np.float64) @ b.astype(np.float64)).astype(F32)


def aff12(m):
    """4x4 -> the C-ABI's 12 floats: R row-major then t (device::Aff3f)."""
    return np.concatenate([m[:3, :3].reshape(-1), m[:3, 3]]).astype(F32)


def camera_pose(cfg, frame):
    centre = (0.0, 0.0, 0.5 + cfg.size / 2)
    return rot_y_about(np.deg2rad(0.25 * frame), centre)


# ------------------------------------------------------------------ analytic depth
def _hit_points(cfg, pose):
    """Nearest hit of every pixel ray with the sphere / back plane; returns camera-frame depth (z) and
    the world-frame hit points."""
    fx, fy, cx, cy = cfg.intr
    s = cfg.size
    xs, ys = np.meshgrid(np.arange(cfg.cols, dtype=np.float64), np.arange(cfg.rows, dtype=np.float64))
    d_cam = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones_like(xs)], -1)
    R, o = pose[:3, :3].astype(np.float64), pose[:3, 3].astype(np.float64)
    d = d_cam @ R.T
    t = np.full(xs.shape, np.inf)
    # back plane z = 0.5 + 0.85 s
    zp = 0.5 + 0.85 * s
    with np.errstate(divide="ignore", invalid="ignore"):
        tp = (zp - o[2]) / d[..., 2]
    ok = (tp > 0) & np.isfinite(tp)
    t = np.where(ok, tp, t)
    # sphere
    c = np.array([0.0, 0.0, 0.5 + s / 2])
    r = 0.2 * s
    oc = o - c
    a = (d * d).sum(-1)
    b = 2 * (d @ oc)
    cc = oc @ oc - r * r
    disc = b * b - 4 * a * cc
    with np.errstate(invalid="ignore"):
        ts = (-b - np.sqrt(disc)) / (2 * a)
    ok = (disc > 0) & (ts > 0)
    t = np.where(ok & (ts < t), ts, t)
    pts = o + d * t[..., None]
    return t, pts           # camera-frame depth == t because d_cam.z == 1


def depth_frame(cfg, frame, holes=0.02, seed=1):
    """uint16 millimetres, 0 = invalid (2 % random holes; hits outside [0.4 m, 0.5 m + size] zeroed)."""
    t, _ = _hit_points(cfg, camera_pose(cfg, frame))
    valid = np.isfinite(t) & (t >= 0.4) & (t <= 0.5 + cfg.size)
    mm = np.where(valid, np.rint(np.where(valid, t, 0.0) * 1000.0), 0.0)
    mm = np.clip(mm, 0, 65535).astype(np.uint16)
    rng = np.random.RandomState(seed + 7919 * frame)       # MT19937
    mm[rng.random_sample(mm.shape) < holes] = 0
    return mm


# ------------------------------------------------------------------ warp nodes
def make_nodes(cfg, seed=2):
    """M node positions on the visible surface of frame 0 (min-distance rejection), dg_w = 2 x mean
    nearest-neighbour spacing."""
    M = cfg.nodes
    t, pts = _hit_points(cfg, camera_pose(cfg, 0))
    valid = np.isfinite(t) & (t >= 0.4) & (t <= 0.5 + cfg.size)
    cand = pts[valid]
    # keep only points inside the volume cube
    s = cfg.size
    inside = (np.abs(cand[:, 0]) < s / 2) & (np.abs(cand[:, 1]) < s / 2) & (cand[:, 2] > 0.5) & (cand[:, 2] < 0.5 + s)
    cand = cand[inside]
    rng = np.random.RandomState(seed)
    order = rng.permutation(len(cand))
    area = 4.0 * s * s                                        # rough visible surface area
    min_d = 0.5 * np.sqrt(area / max(M, 1)) * 0.5
    chosen = []
    cell = {}
    inv = 1.0 / min_d
    for i in order:
        p = cand[i]
        key = tuple(np.floor(p * inv).astype(int))
        ok = True
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    for q in cell.get((key[0] + dx, key[1] + dy, key[2] + dz), ()):
                        if ((p - q) ** 2).sum() < min_d * min_d:
                            ok = False
        if ok:
            cell.setdefault(key, []).append(p)
            chosen.append(p)
            if len(chosen) == M:
                break
    if len(chosen) < M:
        raise RuntimeError("could only place %d of %d nodes" % (len(chosen), M))
    pos = np.asarray(chosen, dtype=F32)
    # mean nearest-neighbour spacing
    d2 = ((pos[:, None, :].astype(np.float64) - pos[None, :, :]) ** 2).sum(-1) if M <= 2500 else None
    if d2 is not None:
        np.fill_diagonal(d2, np.inf)
        nn = np.sqrt(d2.min(1)).mean()
    else:
        nn = 0.0
        for a in range(0, M, 500):
            dd = ((pos[a:a + 500, None, :].astype(np.float64) - pos[None, :, :]) ** 2).sum(-1)
            dd[np.arange(dd.shape[0]), np.arange(a, a + dd.shape[0])] = np.inf
            nn += np.sqrt(dd.min(1)).sum()
        nn /= M
    sigma = np.full(M, 2.0 * nn, dtype=F32)
    return pos, sigma


def dq_from_twist(rvec, tvec):
    """DualQuaternion::from_twist (dual_quaternion.hpp:212-229), vectorised, float32:
    returns [M, 8] = {rotation (w,x,y,z), translation_/dual (w,x,y,z)}."""
    r = np.asarray(rvec, F32)
    t = np.asarray(tvec, F32)
    norm = np.sqrt((r * r).sum(-1, dtype=F32)).astype(F32)
    small = norm <= F32(1e-6)
    safe = np.where(small, F32(1), norm)
    cosn = np.cos(safe).astype(F32)
    sign = np.sign(cosn).astype(F32)
    s_over = (sign * np.sin(safe) / safe).astype(F32)
    rot = np.concatenate([(cosn * sign)[:, None], r * s_over[:, None]], -1).astype(F32)
    rot[small] = (1, 0, 0, 0)
    half = np.concatenate([np.zeros((len(t), 1), F32), F32(0.5) * t], -1)
    dual = quat_mul(half, rot)
    return np.concatenate([rot, dual], -1).astype(F32)


def quat_mul(a, b):
    aw, ax, ay, az = [a[..., i] for i in range(4)]
    bw, bx, by, bz = [b[..., i] for i in range(4)]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw], -1).astype(F32)


def identity_dq(M):
    """M default-constructed DualQuaternion<float> (both quaternions (1,0,0,0), dual_quaternion.hpp:25-29)."""
    dq = np.zeros((M, 8), F32)
    dq[:, 0] = 1
    dq[:, 4] = 1
    return dq


def node_transforms(cfg, frame, seed=3, rot_amp=0.05, trans_amp=0.01):
    """Per-frame node transforms: twist amplitudes ~U(-a, a) (seeded), smoothly scaled per frame."""
    M = cfg.nodes
    rng = np.random.RandomState(seed)
    rv = rng.uniform(-rot_amp, rot_amp, (M, 3))
    tv = rng.uniform(-trans_amp, trans_amp, (M, 3))
    s = np.sin(frame / 10.0 + 0.5)
    return dq_from_twist((rv * s).astype(F32), (tv * s).astype(F32))
