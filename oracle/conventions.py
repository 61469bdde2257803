"""numpy restatement of the stage-1 -> stage-2 hand-off conventions of the reference (TEST INFRASTRUCTURE).

These ARE pinned: tests/golden/conventions_rot6d.npz holds outputs of the reference's own
ObjTracker/utils/geometry.py generated in the build container by tests/golden/make_golden_conventions.py.
"""
import numpy as np


def rot6d_to_matrix(rot_6d: np.ndarray) -> np.ndarray:
    """ObjTracker/utils/geometry.py:7-25 -- columns a1,a2 -> Gram-Schmidt -> stack(b1,b2,b3, axis=-1).

    The reference calls torch.cross without dim (geometry.py:24), which is wrong for B == 3; this
    restatement uses the last axis, which equals the reference for every B != 3.
    """
    x = np.asarray(rot_6d, dtype=np.float64).reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = a1 / np.maximum(np.linalg.norm(a1, axis=1, keepdims=True), 1e-12)
    t = a2 - np.sum(b1 * a2, axis=1, keepdims=True) * b1
    b2 = t / np.maximum(np.linalg.norm(t, axis=1, keepdims=True), 1e-12)
    b3 = np.cross(b1, b2, axis=-1)
    return np.stack([b1, b2, b3], axis=-1)


def matrix_to_rot6d(rotmat: np.ndarray) -> np.ndarray:
    """ObjTracker/utils/geometry.py:28-38 -- first two columns."""
    return np.asarray(rotmat).reshape(-1, 3, 3)[:, :, :2]


def saved_pose_from_rot6d(rot_6d: np.ndarray) -> np.ndarray:
    """ObjTracker/run.py:166 -- the R written to obj_infos/*.npz is rot6d_to_matrix(.)^T (object -> camera)."""
    return np.transpose(rot6d_to_matrix(rot_6d), (0, 2, 1))


def intrinsics(height: int, width: int) -> np.ndarray:
    """ObjTracker/run.py:119-123 -- f = 1.2*min(H,W), cx = W//2, cy = H//2, one K for all frames."""
    f = 1.2 * min(height, width)
    return np.array([[f, 0, width // 2], [0, f, height // 2], [0, 0, 1]], dtype=np.float32)


def decode_sam_mask(mask_rgb: np.ndarray):
    """ObjTracker/run.py:81-87 -- object: channel 1 == 255, hand: last channel == 255 (bool maps)."""
    hand = mask_rgb[:, :, -1] == 255
    obj = mask_rgb[:, :, 1] == 255
    return obj, hand


def label_map(obj: np.ndarray, hand: np.ndarray) -> np.ndarray:
    """ObjTracker/run.py:66, utils/maskutils.py:24-28 -- 1 object / 0 background / -1 hand; object wins."""
    lab = np.zeros(obj.shape, dtype=np.int8)
    lab[hand] = -1
    lab[obj] = 1
    return lab


def normalize_vertices(verts: np.ndarray) -> np.ndarray:
    """ObjTracker/run.py:110-112 -- centre at vertex mean, max vertex norm 0.5."""
    v = verts - verts.mean(0)
    return v / np.linalg.norm(v, 2, 1).max() * 0.5


def apply_pose(verts_obj: np.ndarray, R: np.ndarray, T: np.ndarray) -> np.ndarray:
    """ObjTracker/vis.py:52 -- x_cam = x_obj @ R.T + T."""
    return verts_obj @ R.T + T.reshape(1, 3)
