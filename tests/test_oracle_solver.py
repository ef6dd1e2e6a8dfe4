"""The warp-field data term: oracle restatement of the GPU conjugate-gradient solve, pinned by the
reference's own solver tests (tests/ceres_warp_test.cpp: after energy_data + warp the source vertices sit on the targets
within 1e-3) and by a float64 least-squares solution."""
import numpy as np

import oracle_lib as O
from dynamicfusion_amd import synth

F32 = np.float32
CUBE = np.array([[1, 1, 1], [1, 1, -1], [1, -1, 1], [1, -1, -1], [-1, 1, 1], [-1, 1, -1], [-1, -1, 1], [-1, -1, -1]], F32)


def solve_and_warp(src, dst, iters=100):
    """WarpField::init(cube corners) (sigma = 3, identity transforms, warp_field.cpp:68-88), energy_data, warp."""
    sigma = np.full(8, 3.0, F32)
    dq = synth.identity_dq(8)
    out, en = O.solve_data_term(CUBE, dq, sigma, src, dst, 8, iters)
    warped, _ = O.warp_points(CUBE, out, sigma, src, np.tile(F32([0, 0, 1]), (len(src), 1)), 8)
    return out, en, warped


def test_reference_EnergyDataSingleVertexTest():             # tests/ceres_warp_test.cpp:6-50
    src = F32([[0, 0, 0]]); dst = F32([[0.05, 0.05, 0.05]])
    _, en, warped = solve_and_warp(src, dst)
    assert np.abs(warped - dst).max() < 1e-3 and en[1] < 1e-8 < en[0]


def test_reference_EnergyDataRigidTest():                    # tests/ceres_warp_test.cpp:53-117
    src = F32([[2, 2, 2], [3, 3, 3]]); dst = F32([[2.05, 2.05, 2.05], [3.05, 3.05, 3.05]])
    _, en, warped = solve_and_warp(src, dst)
    assert np.abs(warped - dst).max() < 1e-3 and en[1] < 1e-8 < en[0]


def test_matches_float64_least_squares_and_skips_nan_points():
    rng = np.random.default_rng(3)
    M, N, k = 40, 600, 4
    pos = rng.uniform(-1, 1, (M, 3)).astype(F32); sigma = np.full(M, 0.6, F32)
    dq = synth.dq_from_twist(np.zeros((M, 3), F32), rng.uniform(-0.02, 0.02, (M, 3)).astype(F32))     # non-zero starting translations
    src = rng.uniform(-1, 1, (N, 3)).astype(F32)
    true_t = 0.05 * np.sin(3 * pos)                                                                   # a smooth translation field
    idx, d2 = O.knn(pos, src, k)
    w = np.exp(-d2.astype(np.float64) / (2 * 0.6 ** 2))
    dst = (src + np.einsum("vk,vkc->vc", w, true_t[idx])).astype(F32)
    src[5] = np.nan; dst[9, 1] = np.nan                                                               # skipped (warp_field.cpp:130-136)
    out, en = O.solve_data_term(pos, dq, sigma, src, dst, k, 400)
    assert en[1] < 1e-6 * en[0]
    # float64 normal equations on the valid points: the fitted field reproduces the targets like the CG solution does
    ok = ~(np.isnan(src).any(1) | np.isnan(dst).any(1))
    W = np.zeros((ok.sum(), M)); rows = np.arange(ok.sum())
    for j in range(k):
        np.add.at(W, (rows, idx[ok][:, j]), w[ok][:, j])
    T64 = np.linalg.lstsq(W, (dst[ok] - src[ok]).astype(np.float64), rcond=None)[0]
    t_out = np.zeros((M, 4), F32)
    for n in range(M):
        O.lib().orc_node_translation(out[n], t_out[n])
    assert np.abs(W @ t_out[:, 1:].astype(np.float64) - W @ T64).max() < 2e-4
    # rotations untouched, identity here
    assert np.array_equal(out[:, :4], dq[:, :4])
