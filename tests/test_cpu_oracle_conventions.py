"""CPU: the conventions restatement (oracle/conventions.py) against the golden vectors generated from the
reference's own ObjTracker/utils/geometry.py (tests/golden/make_golden_conventions.py)."""
import os

import numpy as np
import pytest

from oracle import conventions as C

GOLD = os.path.join(os.path.dirname(__file__), "golden", "conventions_rot6d.npz")


@pytest.mark.parametrize("B", [1, 2, 5, 16])
def test_rot6d_matches_reference_golden(B):
    g = np.load(GOLD)
    R = C.rot6d_to_matrix(g[f"in_B{B}"])
    assert np.abs(R - g[f"R_B{B}"]).max() < 2e-6
    assert np.abs(C.saved_pose_from_rot6d(g[f"in_B{B}"]) - g[f"saved_R_B{B}"]).max() < 2e-6
    assert np.array_equal(C.matrix_to_rot6d(g[f"R_B{B}"]), g[f"rot6d_back_B{B}"])
    # proper rotations
    assert np.abs(np.einsum("bij,bkj->bik", R, R) - np.eye(3)).max() < 1e-6
    assert np.abs(np.linalg.det(R) - 1).max() < 1e-6


def test_reference_constants():
    g = np.load(GOLD)
    assert int(g["REND_SIZE"]) == 256 and abs(float(g["BBOX_EXPANSION_FACTOR"]) - 0.3) < 1e-9


def test_intrinsics_mask_and_pose_conventions():
    K = C.intrinsics(512, 512)
    assert K[0, 0] == np.float32(1.2 * 512) and K[0, 2] == 256 and K[1, 2] == 256      # run.py:119-123
    m = np.zeros((4, 4, 3), np.uint8)
    m[0, 0, 1] = 255; m[1, 1, 2] = 255; m[2, 2, 1] = 255; m[2, 2, 2] = 255
    obj, hand = C.decode_sam_mask(m)                                                        # run.py:81-87
    lab = C.label_map(obj, hand)                                                            # run.py:66
    assert lab[0, 0] == 1 and lab[1, 1] == -1 and lab[2, 2] == 1 and lab[3, 3] == 0
    v = np.random.default_rng(0).normal(size=(50, 3))
    vn = C.normalize_vertices(v)                                                            # run.py:110-112
    assert abs(np.linalg.norm(vn, axis=1).max() - 0.5) < 1e-12 and np.abs(vn.mean(0)).max() < 1e-12
    R = C.rot6d_to_matrix(np.random.default_rng(1).normal(size=(1, 3, 2)))[0]
    T = np.array([[0.1, -0.2, 2.0]])
    assert np.allclose(C.apply_pose(vn, R, T), (R @ vn.T).T + T)                            # vis.py:52


def test_product_dataset_decode_matches_oracle_conventions():
    from dynhor_amd.dataset import decode_sam_mask, label_map
    rng = np.random.default_rng(3)
    m = (rng.integers(0, 2, size=(16, 16, 3)) * 255).astype(np.uint8)
    o1, h1 = C.decode_sam_mask(m)
    o2, h2 = decode_sam_mask(m)
    assert np.array_equal(o1, o2) and np.array_equal(h1, h2)
    assert np.array_equal(C.label_map(o1, h1), label_map(o2, h2))
