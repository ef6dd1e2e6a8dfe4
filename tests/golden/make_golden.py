#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE'S OWN code: oracle/_ref/libdfref.so, i.e.
/root/reference/kfusion/include/nanoflann/nanoflann.hpp, kfusion/src/utils/{quaternion,dual_quaternion,
knn_point_cloud}.hpp compiled unmodified (oracle/ref_glue.cpp, oracle/Makefile).  Run in the build
container only (needs /root/reference); the .npz files are committed so that the GPU box -- where
/root/reference does not exist -- can still pin the oracle and the HIP path to reference outputs.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402

F32 = np.float32


def main():
    assert os.path.isdir("/root/reference/kfusion/src/utils"), "needs the reference checkout"
    R = O.ref()
    rng = np.random.RandomState(20260925)

    # ---- quaternion / dual-quaternion known answers (inputs of tests/utils/test_quaternion.cc,
    # test_dual_quaternion.cc plus random ones), outputs from the reference headers
    q = {}
    out = np.zeros(4, F32)
    R.ref_quat_encode_rotation(F32(np.pi / 4), 0, 0, 1, out); q["encode_pi4_z"] = out.copy()
    v = np.array([0, 0, 1], F32); R.ref_quat_rotate_xyz(np.array([0, 0, 1, 1], F32), v); q["rotate_001_by_0011"] = v.copy()
    R.ref_quat_mul(np.array([1, 1, 2, 2], F32), np.array([0, 0, 1, 1], F32), out); q["mul_1122_0011"] = out.copy()
    q["dot_1122_0011"] = np.array([R.ref_quat_dot(np.array([1, 1, 2, 2], F32), np.array([0, 0, 1, 1], F32))], F32)
    R.ref_quat_normalize(np.full(4, 10, F32), out); q["normalize_10"] = out.copy()
    rot, tr = np.zeros(4, F32), np.zeros(4, F32)
    R.ref_dq_euler(1, 2, 3, 1, 2, 3, rot, tr); q["dq_euler_rot"] = rot.copy(); q["dq_euler_trans"] = tr.copy()
    qa = rng.normal(size=(256, 4)).astype(F32); qb = rng.normal(size=(256, 4)).astype(F32)
    prod = np.zeros((256, 4), F32); nrm = np.zeros((256, 4), F32)
    for i in range(256):
        R.ref_quat_mul(qa[i], qb[i], prod[i]); R.ref_quat_normalize(qa[i], nrm[i])
    q.update(rand_a=qa, rand_b=qb, rand_mul=prod, rand_normalize=nrm)
    rv = rng.uniform(-0.3, 0.3, (256, 3)).astype(F32); tv = rng.uniform(-0.2, 0.2, (256, 3)).astype(F32)
    rv[0] = 0                                              # the |r| <= epsilon branch of from_twist
    dq = np.zeros((256, 8), F32); gt = np.zeros((256, 4), F32); tp = rng.normal(size=(256, 3)).astype(F32); tp_out = tp.copy()
    for i in range(256):
        R.ref_dq_from_twist(rv[i], tv[i], dq[i]); R.ref_dq_get_translation(dq[i], gt[i]); R.ref_dq_transform(dq[i], tp_out[i])
    q.update(twist_r=rv, twist_t=tv, twist_dq=dq, twist_get_translation=gt, transform_in=tp, transform_out=tp_out)
    np.savez_compressed(os.path.join(HERE, "quaternion_kat.npz"), **q)

    # ---- k-NN (nanoflann kd-tree) : the nanoflann_test.cpp fixture + a seeded cloud
    corners = np.array([[1, 1, 1], [1, 1, -1], [1, -1, 1], [1, -1, -1], [-1, 1, 1], [-1, 1, -1], [-1, -1, 1], [-1, -1, -1]], F32)
    cq = np.array([[-1, -1, -1], [0, 0, 0], [1, 1, 1], [2, 2, 2], [3, 3, 3]], F32)
    ci, cd = O.knn(corners, cq, 8, use_ref=True)
    M = 700
    pos = (rng.uniform(-0.5, 0.5, (M, 3)) + np.array([0, 0, 1.0])).astype(F32)
    qs = (rng.uniform(-0.7, 0.7, (4096, 3)) + np.array([0, 0, 1.0])).astype(F32)
    i4, d4 = O.knn(pos, qs, 4, use_ref=True)
    i8, d8 = O.knn(pos, qs, 8, use_ref=True)
    np.savez_compressed(os.path.join(HERE, "knn.npz"), corners=corners, corner_queries=cq, corner_idx=ci, corner_d2=cd,
                        pos=pos, queries=qs, idx4=i4, d2_4=d4, idx8=i8, d2_8=d8)

    # ---- DQB + warp through the reference classes
    tw_r = rng.uniform(-0.05, 0.05, (M, 3)).astype(F32); tw_t = rng.uniform(-0.01, 0.01, (M, 3)).astype(F32)
    ndq = np.zeros((M, 8), F32)
    for i in range(M):
        R.ref_dq_from_twist(tw_r[i], tw_t[i], ndq[i])
    g = {"pos": pos, "dq": ndq, "points": qs[:2048]}
    nrm_in = rng.normal(size=(2048, 3)).astype(F32)
    g["normals"] = nrm_in
    for tag, sig in (("s3", 3.0), ("s015", 0.15)):
        sigma = np.full(M, sig, F32)
        g["sigma_" + tag] = sigma
        for k in (4, 8):
            g["dqb_%s_k%d" % (tag, k)] = O.dqb(pos, ndq, sigma, qs[:2048], k, use_ref=True)
            wp, wn = O.warp_points(pos, ndq, sigma, qs[:2048], nrm_in, k, use_ref=True)
            g["warp_p_%s_k%d" % (tag, k)] = wp
            g["warp_n_%s_k%d" % (tag, k)] = wn
    np.savez_compressed(os.path.join(HERE, "dqb_warp.npz"), **g)
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
