"""CPU restatement of utils/splats.py:7-35 (to_full_cov, extract_rot_scale, matrix_to_quaternion, build_covariance)
-- TEST INFRASTRUCTURE, NOT THE PRODUCT. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.

Pinning: tests/golden/splats.npz holds the outputs of the reference's own functions run on CPU tensors
(tests/golden/make_golden.py:gen_splats); tests/test_oracle_misc.py checks this file against them. Eigenvector signs
are LAPACK's choice and carry no information, so rotations are compared through R diag(s^2) R^T and |cos| of matching
eigenvectors, eigenvalues directly."""
import numpy as np


def to_full_cov(cov6):
    c = np.asarray(cov6, np.float32)
    return c[:, [0, 1, 2, 1, 3, 4, 2, 4, 5]].reshape(-1, 3, 3)


def matrix_to_quaternion(m):
    m = np.asarray(m, np.float32)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = (m[:, i, j] for i in range(3) for j in range(3))
    qa = np.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1).astype(np.float32)
    q_abs = np.where(qa > 0, np.sqrt(np.maximum(qa, 0)), 0).astype(np.float32)
    cand = np.stack([
        np.stack([q_abs[:, 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
        np.stack([m21 - m12, q_abs[:, 1] ** 2, m10 + m01, m02 + m20], -1),
        np.stack([m02 - m20, m10 + m01, q_abs[:, 2] ** 2, m12 + m21], -1),
        np.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[:, 3] ** 2], -1)], -2).astype(np.float32)
    cand = cand / (2.0 * np.maximum(q_abs[..., None], np.float32(0.1)))
    return cand[np.arange(m.shape[0]), q_abs.argmax(-1)].astype(np.float32)


def extract_rot_scale(cov):
    cov = np.asarray(cov, np.float32)
    S, R = np.linalg.eigh((cov + np.eye(3, dtype=np.float32) * np.float32(1e-8)).astype(np.float32), UPLO="U")
    with np.errstate(invalid="ignore"):
        scaling = np.sqrt(S)
    scaling = np.where(np.isnan(scaling), np.float32(1e-6), scaling).astype(np.float32)
    det = np.linalg.det(R).astype(np.float32)
    q = matrix_to_quaternion(R * det[:, None, None])
    q = q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)
    return q.astype(np.float32), scaling


def quaternion_to_matrix(q):
    q = np.asarray(q, np.float64)
    r, i, j, k = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    two_s = 2.0 / (q * q).sum(-1)
    o = np.stack([1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                  two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                  two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)], -1)
    return o.reshape(-1, 3, 3)


def build_covariance(rotation, scaling):
    R = quaternion_to_matrix(rotation)
    S = np.asarray(scaling, np.float64) ** 2
    return np.einsum("nij,nj,nkj->nik", R, S, R)
