"""The two rotation conversions the RLBench ACT head needs at rollout time: 6-D representation -> rotation matrix ->
quaternion (real part first).  Behavioural counterpart of rotation_6d_to_matrix / matrix_to_quaternion /
standardize_quaternion in /root/reference/src/utils/rotation_conversions.py:102-166, 368-380, 556-578 (PyTorch3D's
formulation: Gram-Schmidt on the two 3-vectors; the best-conditioned of the four quaternion candidates)."""
import torch
import torch.nn.functional as F


def rotation_6d_to_matrix(d6):
    """(..., 6) -> (..., 3, 3): rows b1 = normalise(a1), b2 = normalise(a2 - <b1, a2> b1), b3 = b1 x b2."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def _sqrt_positive_part(x):
    """sqrt(max(0, x)) with a zero subgradient at 0."""
    out = torch.zeros_like(x)
    pos = x > 0
    out[pos] = torch.sqrt(x[pos])
    return out


def standardize_quaternion(q):
    """Flip the sign so that the real part is non-negative."""
    return torch.where(q[..., 0:1] < 0, -q, q)


def matrix_to_quaternion(matrix):
    """(..., 3, 3) -> (..., 4), real part first."""
    if matrix.size(-1) != 3 or matrix.size(-2) != 3:
        raise ValueError(f"Invalid rotation matrix shape {matrix.shape}.")
    batch = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(matrix.reshape(batch + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22,
                                             1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1))
    # candidate r: the quaternion multiplied by its own component r (r in {w, x, y, z})
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
    ], dim=-2)
    floor = torch.tensor(0.1, dtype=q_abs.dtype, device=q_abs.device)  # small denominators are never picked
    cand = cand / (2.0 * q_abs[..., None].max(floor))
    pick = F.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5
    return standardize_quaternion(cand[pick, :].reshape(batch + (4,)))
