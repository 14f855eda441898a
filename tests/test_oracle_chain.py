"""Pins oracle/chain_oracle.py the way the reference pins its originals (test/transform_chains_test.cc:12-232): every analytic
derivative of the chain against a numerical one (rotations perturbed on the right, differences taken through the log map), the
composed poses against plain pose multiplication, all 64 activity masks of a link, and mixed masks along a chain.  The SO(3) helpers of
the absent geometry_utils dependency are restated in chain_oracle.py, so these checks pin those restatements as well."""
import itertools

import numpy as np

from oracle import chain_oracle as CH

LINKS = [(np.array([-0.5, 0.5, 0.3]), np.array([1.0, 0.5, 2.0])), (np.array([0.8, 0.5, 1.2]), np.array([0.5, 0.75, -0.5])),
         (np.array([1.5, -0.2, 0.0]), np.array([1.2, -0.5, 0.1])), (np.array([0.2, -0.1, 0.3]), np.array([0.1, -0.1, 0.2]))]   # transform_chains_test.cc:15-20


def links():
    return [CH.Pose(CH.so3_exp(w), t.copy()) for w, t in LINKS]


def numerical_jacobian(x0, fn, manifold=False, h=1e-3):
    """Fourth-order central differences; for rotation-valued fn the difference is log(fn(x0)^T fn(x0 + d)) (right tangent)."""
    x0 = np.asarray(x0, float)
    f0 = fn(x0)
    cols = []
    for i in range(x0.shape[0]):
        def at(s):
            x = x0.copy(); x[i] += s
            v = fn(x)
            return CH.so3_log(f0.T @ v) if manifold else v
        cols.append((-at(2 * h) + 8 * at(h) - 8 * at(-h) + at(-2 * h)) / (12 * h))
    return np.stack(cols, axis=1)


def test_so3_helpers():
    rng = np.random.default_rng(0)
    for _ in range(20):
        w = rng.uniform(-1.5, 1.5, 3)
        R = CH.so3_exp(w)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-14)
        np.testing.assert_allclose(CH.so3_log(R), w, atol=1e-12)
        ang = rng.uniform(-1.2, 1.2, 3)
        R, D = CH.so3_from_euler_xyz(ang)
        np.testing.assert_allclose(CH.euler_xyz_from_rotation(R), ang, atol=1e-13)
        Dn = numerical_jacobian(ang, lambda a: CH.so3_from_euler_xyz(a)[0], manifold=True)
        np.testing.assert_allclose(D, Dn, atol=1e-9)
    for x in (-7.0, -np.pi, -0.3, 0.0, 3.0, np.pi, 9.5):
        y = CH.mod_pi(x)
        assert -np.pi <= y < np.pi and abs(np.sin(y) - np.sin(x)) < 1e-12 and abs(np.cos(y) - np.cos(x)) < 1e-12


def test_compute_chain():
    """ChainComputationBufferTest.TestComputeChain, transform_chains_test.cc:12-102 (tolerance 1e-9 as there)."""
    base = links()
    N = len(base)

    def perturbed_rot(angles):
        return [CH.Pose(l.rotation @ CH.so3_exp(angles[3 * i:3 * i + 3]), l.translation) for i, l in enumerate(base)]

    def perturbed_trans(tr):
        return [CH.Pose(l.rotation, l.translation + tr[3 * i:3 * i + 3]) for i, l in enumerate(base)]

    c = CH.compute_chain(base)
    z = np.zeros(3 * N)
    np.testing.assert_allclose(numerical_jacobian(z, lambda a: CH.compute_chain(perturbed_rot(a)).i_t_end[:, 0]), c.translation_D_rotation, atol=1e-9)
    np.testing.assert_allclose(numerical_jacobian(z, lambda a: CH.compute_chain(perturbed_rot(a)).i_R_end[0], manifold=True), c.rotation_D_rotation, atol=1e-9)
    np.testing.assert_allclose(numerical_jacobian(z, lambda t: CH.compute_chain(perturbed_trans(t)).i_t_end[:, 0]), c.translation_D_translation, atol=1e-9)
    assert len(c.i_R_end) == N + 1 and c.i_t_end.shape[1] == N + 1
    cur = CH.Pose()
    for i, p in enumerate(CH.compute_all_poses(c)):           # :86-101
        np.testing.assert_allclose(cur.translation, p.translation, atol=1e-9)
        np.testing.assert_allclose(cur.rotation, p.rotation, atol=1e-9)
        if i < N:
            cur = cur * base[i]
    assert CH.compute_chain([]).i_t_end.shape == (3, 0)       # empty chain clears the buffer, transform_chains.cc:24-32


def test_actuator_link_compute_pose_all_masks():
    """ActuatorLinkTest.TestComputePose, transform_chains_test.cc:117-181: all 64 masks."""
    pose = CH.Pose(CH.so3_exp(np.array([-0.3, 0.5, 0.4])), np.array([0.4, -0.2, 1.2]))
    inp = np.array([0.2, 0.1, 0.35, -0.2, 0.5, 0.6])
    off = 3
    for mask in itertools.product((0, 1), repeat=6):
        link = CH.ActuatorLink(pose, mask)
        fixed = np.concatenate([CH.euler_xyz_from_rotation(pose.rotation), pose.translation])
        combined = np.where(np.array(mask) > 0, inp, fixed)
        params = np.full(10, np.nan)
        pos = off
        for i in range(6):
            if mask[i]:
                params[pos] = combined[i]; pos += 1
        got, J = link.compute(params, off)
        np.testing.assert_allclose(got.translation, combined[3:], atol=1e-12)
        np.testing.assert_allclose(got.rotation, CH.so3_from_euler_xyz(combined[:3])[0], atol=1e-12)
        assert J.shape == (3, link.active_rotation_count())
        if link.active_rotation_count():
            live = [off + j for j in range(link.active_count())]

            def rot(p_live):
                p = params.copy(); p[live] = p_live
                return link.compute(p, off)[0].rotation
            Jn = numerical_jacobian(params[live], rot, manifold=True)
            np.testing.assert_allclose(Jn[:, :link.active_rotation_count()], J, atol=1e-9)
            np.testing.assert_allclose(Jn[:, link.active_rotation_count():], 0, atol=1e-12)   # translations do not turn the link


def test_actuator_chain_effector_derivatives():
    """ActuatorChainTest.TestComputeEffector, transform_chains_test.cc:183-232: different masks on different links."""
    masks = list(itertools.product((0, 1), repeat=6))
    base = links()
    idx = 0
    while idx <= len(masks) - len(base):
        chain = CH.ActuatorChain([CH.ActuatorLink(p, masks[idx + j]) for j, p in enumerate(base)])
        idx += len(base)
        total = chain.total_active()
        assert 0 < total < 24
        params = np.array([(i * 0.112) if (i % 2) else (-i * 0.0421) for i in range(total)])
        chain.update(params)
        tD, rD = chain.translation_D_params.copy(), chain.rotation_D_params.copy()

        def trans(p):
            chain.update(p); return chain.translation()

        def rot(p):
            chain.update(p); return chain.rotation()
        np.testing.assert_allclose(numerical_jacobian(params, trans), tD, atol=1e-9)
        np.testing.assert_allclose(numerical_jacobian(params, rot, manifold=True), rD, atol=1e-9)
