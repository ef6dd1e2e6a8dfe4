"""Pins the oracle (oracle/dfusion_oracle.c) BEFORE it is trusted:

  1. against the known-answer vectors of the reference's own tests
     (/root/reference/tests/utils/test_quaternion.cc, test_dual_quaternion.cc, nanoflann_test.cpp);
  2. against tests/golden/*.npz, generated from the reference's own headers compiled unmodified
     (tests/golden/make_golden.py -> oracle/_ref/libdfref.so);
  3. where /root/reference is present (the build container), live against oracle/_ref.

integrate / raycast / compute_dists / clear / extract / front-end / ICP: pinned against the reference's own CUDA kernels
compiled for the host -- tests/test_oracle_refcu.py.
"""
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F32 = np.float32


def bits(a):
    return np.ascontiguousarray(a, F32).view(np.uint32)


# ------------------------------------------------------------------ reference tests' known answers
def test_quaternion_encode_rotation_kat():      # test_quaternion.cc:6-15
    out = np.zeros(4, F32)
    O.lib().orc_quat_encode_rotation(F32(np.pi / 4), 0, 0, 1, out)
    assert np.allclose(out, [0.9238795, 0, 0, 0.38268346], rtol=4 * 1.2e-7, atol=0)   # ASSERT_FLOAT_EQ = 4 ulp


def test_quaternion_rotate_kat():               # test_quaternion.cc:17-25
    v = np.array([0, 0, 1], F32)
    O.lib().orc_quat_rotate_xyz(np.array([0, 0, 1, 1], F32), v)
    assert v.tolist() == [0, 2, 0]


def test_quaternion_product_kat():              # test_quaternion.cc:27-36
    out = np.zeros(4, F32)
    O.lib().orc_quat_mul(np.array([1, 1, 2, 2], F32), np.array([0, 0, 1, 1], F32), out)
    assert out.tolist() == [-4, 0, 0, 2]


def test_quaternion_normalize_kat():            # test_quaternion.cc:45-50
    out = np.zeros(4, F32)
    O.lib().orc_quat_normalize(np.full(4, 10, F32), out)
    assert out.tolist() == [0.5, 0.5, 0.5, 0.5]


def test_dual_quaternion_ctor_kat_from_reference_headers():
    """test_dual_quaternion.cc:6-21: DQ(1,2,3, roll 1, pitch 2, yaw 3).  The expected rotation there is
    (0.435953, -0.718287, 0.310622, 0.454649) +-0.01; the reference's own header gives z = 0.444435
    (|delta| = 0.0102, i.e. the reference test fails its own tolerance on z); translation ~ (1,2,3) +-0.1."""
    g = np.load(os.path.join(GOLD, "quaternion_kat.npz"))
    assert np.allclose(g["dq_euler_rot"][:3], [0.435953, -0.718287, 0.310622], atol=0.01)
    assert abs(g["dq_euler_rot"][3] - 0.454649) < 0.011
    assert np.allclose(g["dq_euler_trans"], [0, 1, 2, 3], atol=0.1)


# ------------------------------------------------------------------ golden vectors from the reference headers
def test_quaternion_golden_bit_exact():
    g = np.load(os.path.join(GOLD, "quaternion_kat.npz"))
    L = O.lib()
    out = np.zeros(4, F32)
    for i in range(len(g["rand_a"])):
        L.orc_quat_mul(g["rand_a"][i].copy(), g["rand_b"][i].copy(), out)
        assert np.array_equal(bits(out), bits(g["rand_mul"][i]))
        L.orc_quat_normalize(g["rand_a"][i].copy(), out)
        assert np.array_equal(bits(out), bits(g["rand_normalize"][i]))
    dq = np.zeros(8, F32)
    for i in range(len(g["twist_r"])):
        L.orc_dq_from_twist(g["twist_r"][i].copy(), g["twist_t"][i].copy(), dq)
        assert np.array_equal(bits(dq), bits(g["twist_dq"][i]))
        L.orc_node_translation(g["twist_dq"][i].copy(), out)
        assert np.array_equal(bits(out), bits(g["twist_get_translation"][i]))
    assert np.array_equal(bits(g["encode_pi4_z"]), bits(np.array([0.9238795, 0, 0, 0.38268346], F32)))
    assert g["rotate_001_by_0011"].tolist() == [0, 2, 0]
    assert g["mul_1122_0011"].tolist() == [-4, 0, 0, 2]
    assert g["dot_1122_0011"].tolist() == [4]                   # test_quaternion.cc:38-43
    assert g["normalize_10"].tolist() == [0.5, 0.5, 0.5, 0.5]


def test_knn_golden():
    g = np.load(os.path.join(GOLD, "knn.npz"))
    for k, ki, kd in ((4, "idx4", "d2_4"), (8, "idx8", "d2_8")):
        idx, d2 = O.knn(g["pos"], g["queries"], k)
        assert np.array_equal(idx, g[ki])
        assert np.array_equal(bits(d2), bits(g[kd]))
    # nanoflann_test.cpp fixture: 8 cube corners, k = 8 -> every query returns all 8 nodes, equidistant ones in nanoflann's
    # tree order (the oracle restates the tree, so the index lists are identical, not just the sets)
    idx, d2 = O.knn(g["corners"], g["corner_queries"], 8)
    assert np.array_equal(bits(d2), bits(g["corner_d2"]))
    assert np.array_equal(idx, g["corner_idx"])


@pytest.mark.parametrize("name", ["grid", "surface", "dups"])
def test_knn_tie_order_golden(name):
    """Exact distance ties resolve as in the reference's nanoflann (tests/golden/make_golden_ties.py): gridded nodes and
    queries, the bench's pixel-grid surface construction, duplicated nodes."""
    g = np.load(os.path.join(GOLD, "knn_ties.npz"))
    pos, q = g[name + "_pos"], g[name + "_q"]
    for k in (4, 8):
        idx, d2 = O.knn(pos, q, k)
        assert np.array_equal(idx, g["%s_idx%d" % (name, k)].astype(np.int32))
        assert np.array_equal(bits(d2), bits(g["%s_d2_%d" % (name, k)]))
        bi, bd = O.knn(pos, q, k, brute=True)                  # the sets really are tie-heavy: index order disagrees
        assert np.array_equal(bits(bd), bits(d2)) and (bi != idx).any(1).sum() > 100
    if name == "surface":
        wp, _ = O.warp_points(pos, g["surface_dq"], g["surface_sigma"], q, None, 8)
        assert np.array_equal(bits(wp), bits(g["surface_warp8"]))


@pytest.mark.parametrize("tag", ["s3", "s015"])
@pytest.mark.parametrize("k", [4, 8])
def test_dqb_and_warp_golden_bit_exact(tag, k):
    g = np.load(os.path.join(GOLD, "dqb_warp.npz"))
    sigma = g["sigma_" + tag]
    out = O.dqb(g["pos"], g["dq"], sigma, g["points"], k)
    assert np.array_equal(bits(out), bits(g["dqb_%s_k%d" % (tag, k)]))
    p, n = O.warp_points(g["pos"], g["dq"], sigma, g["points"], g["normals"], k)
    assert np.array_equal(bits(p), bits(g["warp_p_%s_k%d" % (tag, k)]))
    assert np.array_equal(bits(n), bits(g["warp_n_%s_k%d" % (tag, k)]))


# ------------------------------------------------------------------ live against the reference headers (build container)
@pytest.mark.skipif(not os.path.isdir("/root/reference/kfusion/src/utils"), reason="reference checkout absent")
def test_live_against_reference_headers():
    assert O.ref().ref_nanoflann_version() == 0x123          # nanoflann.hpp:73
    rng = np.random.RandomState(11)
    pos = rng.uniform(-1, 1, (300, 3)).astype(F32)
    q = rng.uniform(-1.2, 1.2, (3000, 3)).astype(F32)
    dq = np.zeros((300, 8), F32)
    for j in range(300):
        O.ref().ref_dq_from_twist(rng.uniform(-0.1, 0.1, 3).astype(F32), rng.uniform(-0.05, 0.05, 3).astype(F32), dq[j])
    sigma = rng.uniform(0.1, 3.0, 300).astype(F32)
    for k in (1, 4, 8):
        i1, d1 = O.knn(pos, q, k)
        i2, d2 = O.knn(pos, q, k, use_ref=True)
        assert np.array_equal(i1, i2) and np.array_equal(bits(d1), bits(d2))
        assert np.array_equal(bits(O.dqb(pos, dq, sigma, q, k)), bits(O.dqb(pos, dq, sigma, q, k, use_ref=True)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/kfusion/src/utils"), reason="reference checkout absent")
@pytest.mark.parametrize("M,n", [(11, 6), (100, 12), (1000, 24), (4913, 40)])
def test_live_tie_order_against_reference_nanoflann(M, n):
    """Gridded node sets of several sizes (tree depths 1..10), gridded + random queries inside and outside the bounding box."""
    rng = np.random.RandomState(M)
    side = int(np.ceil(M ** (1 / 3)))
    g = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(F32) * F32(0.5)
    pos = g[rng.permutation(len(g))[:M]]
    q = np.stack(np.meshgrid(*[np.arange(-2, n)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(F32) * F32(0.25 * side / (n - 4) * 2)
    q = np.concatenate([q, rng.uniform(-1, side * 0.5 + 1, (2000, 3)).astype(F32)])
    for k in (1, 3, 8):
        if M < k:
            continue
        i1, d1 = O.knn(pos, q, k)
        i2, d2 = O.knn(pos, q, k, use_ref=True)
        assert np.array_equal(i1, i2) and np.array_equal(bits(d1), bits(d2))


@pytest.mark.skipif(not os.path.isdir("/root/reference/kfusion/src/utils"), reason="reference checkout absent")
def test_live_warped_integrate_through_reference_classes():
    """The north-star composition (per-voxel DQB o TSDF update) driven by the reference's own nanoflann / DQB / transform classes
    (ref_integrate_warped, what bench.py times as cpu_baseline kind "reference") gives the oracle's volume bit for bit."""
    from dynamicfusion_amd import synth
    from scene import Scene
    cfg = synth.Config(64, 1.0, cols=160, rows=120, nodes=100, k=8)
    sc = Scene(cfg, n_frames=2)
    a, b = sc.new_volume(), sc.new_volume()
    for f in range(2):
        n1 = O.integrate_warped(sc.dists[f], a, sc.ovol(a), synth.aff12(sc.pose), synth.aff12(sc.world2cam(f)), sc.intr, sc.pos,
                                sc.dqs[f], sc.sigma, cfg.k)
        n2, used = O.ref_integrate_warped(sc.dists[f], b, cfg.dims, sc.vs, sc.trunc, cfg.max_weight, synth.aff12(sc.pose),
                                          synth.aff12(sc.world2cam(f)), sc.intr, sc.pos, sc.dqs[f], sc.sigma, cfg.k, 0, 0, cfg.dims[2])
        assert n1 == n2 and used >= 1
    assert np.array_equal(a, b) and (a >> 16).max() == 2
