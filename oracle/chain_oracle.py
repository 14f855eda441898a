"""CPU restatement of the reference's kinematic-chain test robots (test/transform_chains.{hpp,cc}) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product (mini_opt_amd) never does.
Plain numpy, rotations as 3x3 matrices (the reference carries quaternions; the group operations are the same).  The SO(3) pieces the
reference takes from its un-vendored `geometry_utils` dependency (math::QuaternionExp, SO3FromEulerAngles, EulerAnglesFromSO3, Skew3,
ModPi -- absent from /root/reference: dependencies/geometry_utils is an empty submodule) are restated here in closed form from their
use sites; tests/test_oracle_chain.py pins them the way test/transform_chains_test.cc:12-232 pins the originals: every analytic
derivative against a numerical one.

Conventions (transform_chains.hpp:61-72): link i is the pose of frame i+1 in frame i; derivatives of a rotation are taken in the RIGHT
tangent space of SO(3): R(theta + d) ~= R(theta) Exp(J d).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np


def skew3(v):
    """math::Skew3: [v]_x."""
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def so3_exp(w):
    """Rodrigues' formula (math::QuaternionExp as a matrix)."""
    w = np.asarray(w, float)
    th = float(np.linalg.norm(w))
    K = skew3(w)
    if th < 1e-12:
        return np.eye(3) + K + 0.5 * K @ K
    return np.eye(3) + (math.sin(th) / th) * K + ((1.0 - math.cos(th)) / (th * th)) * K @ K


def so3_log(R):
    """Inverse of so3_exp for rotations away from pi (used by the numerical derivatives only)."""
    c = max(-1.0, min(1.0, 0.5 * (np.trace(R) - 1.0)))
    th = math.acos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-9:
        return 0.5 * v
    return v * (th / (2.0 * math.sin(th)))


def _rx(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]])


def _ry(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])


def _rz(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def so3_from_euler_xyz(angles):
    """math::SO3FromEulerAngles(angles, CompositionOrder::XYZ) as used at transform_chains.cc:110-111, 151-152:
    R = Rx(x) Ry(y) Rz(z), and rotation_D_angles (3x3): column a is the right-tangent derivative of R wrt angle a.
    Perturbing z multiplies R on the right by Rz(d): column z = e_z; y: R Exp(d Rz^T e_y); x: R Exp(d (Ry Rz)^T e_x)."""
    x, y, z = (float(a) for a in angles)
    Rx, Ry, Rz = _rx(x), _ry(y), _rz(z)
    R = Rx @ Ry @ Rz
    D = np.zeros((3, 3))
    D[:, 0] = (Ry @ Rz).T @ np.array([1.0, 0.0, 0.0])
    D[:, 1] = Rz.T @ np.array([0.0, 1.0, 0.0])
    D[:, 2] = np.array([0.0, 0.0, 1.0])
    return R, D


def euler_xyz_from_rotation(R):
    """The angles (x, y, z) with R = Rx(x) Ry(y) Rz(z) -- what transform_chains.cc:108 obtains as -EulerAnglesFromSO3(R^-1) (the ZYX
    decomposition of the inverse, negated)."""
    y = math.asin(max(-1.0, min(1.0, R[0, 2])))
    x = math.atan2(-R[1, 2], R[2, 2])
    z = math.atan2(-R[0, 1], R[0, 0])
    return np.array([x, y, z])


def mod_pi(x):
    """math::ModPi: wrap an angle into [-pi, pi)."""
    return x - 2.0 * math.pi * math.floor((x + math.pi) / (2.0 * math.pi))


@dataclass
class Pose:                                                   # transform_chains.hpp:24-58
    rotation: np.ndarray = field(default_factory=lambda: np.eye(3))
    translation: np.ndarray = field(default_factory=lambda: np.zeros(3))

    def __mul__(self, other: "Pose") -> "Pose":              # :40-42
        return Pose(self.rotation @ other.rotation, self.translation + self.rotation @ other.translation)

    def inverse(self) -> "Pose":                              # :45-48
        return Pose(self.rotation.T, self.rotation.T @ -self.translation)


@dataclass
class ChainComputationBuffer:                                 # transform_chains.hpp:74-98
    rotation_D_rotation: np.ndarray = None
    translation_D_rotation: np.ndarray = None
    translation_D_translation: np.ndarray = None
    i_R_end: List[np.ndarray] = None
    i_t_end: np.ndarray = None


def compute_chain(links: Sequence[Pose]) -> ChainComputationBuffer:
    """ComputeChain, transform_chains.cc:23-82."""
    c = ChainComputationBuffer()
    N = len(links)
    if N == 0:                                                # :24-32
        c.rotation_D_rotation = np.zeros((3, 0)); c.translation_D_rotation = np.zeros((3, 0))
        c.translation_D_translation = np.zeros((3, 0)); c.i_R_end = []; c.i_t_end = np.zeros((3, 0))
        return c
    c.i_R_end = [None] * (N + 1)                              # :37-42: bucket i = i_R_end
    c.i_R_end[N] = np.eye(3)
    for i in range(N - 1, -1, -1):
        c.i_R_end[i] = links[i].rotation @ c.i_R_end[i + 1]
    c.i_t_end = np.zeros((3, N + 1))                          # :47-52
    for i in range(N - 1, -1, -1):
        c.i_t_end[:, i] = links[i].rotation @ c.i_t_end[:, i + 1] + links[i].translation
    c.translation_D_translation = np.zeros((3, 3 * N))        # :56-61: d(0_t_N)/d(i_t_[i+1]) = 0_R_i
    start_R_i = np.eye(3)
    for i in range(N):
        c.translation_D_translation[:, 3 * i:3 * i + 3] = start_R_i
        start_R_i = start_R_i @ links[i].rotation
    c.translation_D_rotation = np.zeros((3, 3 * N))           # :66-73: start_R_[i+1] [-[i+1]_t_N]_x ; last block zero
    for i in range(N - 1):
        c.translation_D_rotation[:, 3 * i:3 * i + 3] = c.translation_D_translation[:, 3 * (i + 1):3 * (i + 1) + 3] @ skew3(-c.i_t_end[:, i + 1])
    c.rotation_D_rotation = np.zeros((3, 3 * N))              # :76-81: N_R_[i+1] ; last block identity
    for i in range(N - 1):
        c.rotation_D_rotation[:, 3 * i:3 * i + 3] = c.i_R_end[i + 1].T
    c.rotation_D_rotation[:, 3 * (N - 1):] = np.eye(3)
    return c


def compute_all_poses(c: ChainComputationBuffer) -> List[Pose]:
    """ComputeAllPoses, transform_chains.cc:84-92: start_T_i for i = 0 .. N."""
    start_T_end = Pose(c.i_R_end[0], c.i_t_end[:, 0])
    return [start_T_end * Pose(c.i_R_end[i], c.i_t_end[:, i]).inverse() for i in range(len(c.i_R_end))]


class ActuatorLink:
    """transform_chains.hpp:130-161, transform_chains.cc:94-158."""

    def __init__(self, pose: Pose, mask: Sequence[int]):
        self.parent_T_child = pose
        self.active = [int(bool(v)) for v in mask]
        self.rotation_xyz = np.zeros(3)
        if self.active_rotation_count() > 0:                  # :105-117
            self.rotation_xyz = euler_xyz_from_rotation(pose.rotation)
            R, _ = so3_from_euler_xyz(self.rotation_xyz)
            assert np.all(np.abs(R - pose.rotation) < 1.0e-5), "Euler angle decomposition failed"

    def active_count(self) -> int:                            # :94-97
        return sum(self.active)

    def active_rotation_count(self) -> int:                   # :99-102
        return sum(self.active[:3])

    def compute(self, params, position: int) -> Tuple[Pose, np.ndarray]:
        """Compute, :125-158.  Returns (pose, J_out [3, active_rotation_count])."""
        if self.active_rotation_count() == 0:                 # :128-137
            t = self.parent_T_child.translation.copy()
            idx = position
            for i in range(3):
                if self.active[i + 3]:
                    t[i] = params[idx]; idx += 1
            return Pose(self.parent_T_child.rotation, t), np.zeros((3, 0))
        updated = np.concatenate([self.rotation_xyz, self.parent_T_child.translation])   # :139-146
        idx = position
        for i in range(6):
            if self.active[i]:
                updated[i] = params[idx]; idx += 1
        R, D = so3_from_euler_xyz(updated[:3])                # :148-149
        J = np.stack([D[:, a] for a in range(3) if self.active[a]], axis=1)   # :151-155
        return Pose(R, updated[3:].copy()), J


class ActuatorChain:
    """transform_chains.hpp:165-216, transform_chains.cc:165-244 (without the 1e-9 parameter cache of ShouldUpdate, :247-257)."""

    def __init__(self, links: Sequence[ActuatorLink] = ()):
        self.links: List[ActuatorLink] = list(links)
        self.rotation_D_params = None
        self.translation_D_params = None
        self.buffer: ChainComputationBuffer = None
        self.pose_buffer: List[Pose] = []

    def total_active(self) -> int:                            # :259-262
        return sum(l.active_count() for l in self.links)

    def update(self, params) -> None:                         # :165-244
        params = np.asarray(params, float)
        total = self.total_active()
        assert params.shape[0] == total, f"Wrong number of params passed. Expected = {total}, actual = {params.shape[0]}"
        self.rotation_D_params = np.zeros((3, total))
        self.translation_D_params = np.zeros((3, total))
        self.pose_buffer = []
        position = 0
        for link in self.links:                               # :188-196
            pose, J = link.compute(params, position)
            self.rotation_D_params[:, position:position + link.active_rotation_count()] = J
            self.pose_buffer.append(pose)
            position += link.active_count()
        self.buffer = compute_chain(self.pose_buffer)         # :199
        position = 0
        for i, link in enumerate(self.links):                 # :202-243
            ac, nr = link.active_count(), link.active_rotation_count()
            if ac == 0:
                continue
            rot_D_angles = self.rotation_D_params[:, position:position + nr].copy()
            self.translation_D_params[:, position:position + nr] = self.buffer.translation_D_rotation[:, 3 * i:3 * i + 3] @ rot_D_angles   # :221-222
            self.rotation_D_params[:, position:position + nr] = self.buffer.rotation_D_rotation[:, 3 * i:3 * i + 3] @ rot_D_angles       # :226-229
            out = 0
            for axis in range(3):                             # :231-241
                if link.active[axis + 3]:
                    self.translation_D_params[:, position + nr + out] = self.buffer.translation_D_translation[:, 3 * i + axis]
                    out += 1
            position += ac

    def translation(self) -> np.ndarray:                      # transform_chains.hpp:189
        return self.buffer.i_t_end[:, 0].copy()

    def rotation(self) -> np.ndarray:                         # :192
        return self.buffer.i_R_end[0].copy()
