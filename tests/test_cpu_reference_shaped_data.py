"""CPU: the stage-1 -> stage-2 hand-off against files laid out by the REFERENCE's own writer:
tests/golden/obj_infos_ref/*.npz were produced by tests/golden/make_golden_obj_infos.py, which imports
ObjTracker/utils/geometry.py and executes run.py:165-179 statement for statement (R = rot6d_to_matrix(.)^T, T of shape [1,3],
file stem = jpg name minus 4 characters).  The product loader (dynhor_amd.dataset.Dataset._load_from_disk) must read them
together with frames globbed as `*.jpg` (run.py:99) and SAM masks decoded by the run.py:81-87 channel rule."""
import os

import numpy as np
import torch

from oracle import conventions as C

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "obj_infos_ref")


def _write_frames(root, stems, H, W, rng):
    from PIL import Image
    for sub in ("rgb", "sam_seg"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    labels = []
    for s in stems:
        Image.fromarray(rng.integers(0, 255, (H, W, 3), dtype=np.uint8)).save(os.path.join(root, "rgb", s + ".jpg"))
        m = np.zeros((H, W, 3), np.uint8)
        obj = rng.random((H, W)) > 0.6
        hand = rng.random((H, W)) > 0.8
        m[..., 1][obj] = 255                      # run.py:84: object = channel 1
        m[..., 2][hand] = 255                     # run.py:85: hand = last channel
        Image.fromarray(m).save(os.path.join(root, "sam_seg", s + ".png"))
        labels.append(C.label_map(obj, hand))
    return labels


def test_loader_reads_obj_infos_written_by_the_reference_statements(tmp_path):
    from dynhor_amd.dataset import Dataset
    inp = np.load(os.path.join(GOLD, "_inputs.npz"))
    H, W = int(inp["height"]), int(inp["width"])
    stems = sorted(f[:-4] for f in os.listdir(GOLD) if f.endswith(".npz") and not f.startswith("_"))
    assert stems == ["0001", "0008", "0015", "0022"]
    raw = np.load(os.path.join(GOLD, stems[0] + ".npz"))
    assert raw["R"].shape == (3, 3) and raw["T"].shape == (1, 3) and raw["K"].shape == (3, 3)          # run.py:172-176
    assert raw["R"].dtype == raw["T"].dtype == raw["K"].dtype == np.float32
    rng = np.random.default_rng(0)
    root = str(tmp_path / "custom_seq")
    labels = _write_frames(root, stems, H, W, rng)
    frames = Dataset._load_from_disk({"dataroot": root, "obj_infos": GOLD})
    assert frames["stems"] == stems and frames["rgb"].shape == (4, H, W, 3)
    # poses: the saved R is rot6d_to_matrix(.)^T (object -> camera) -- oracle/conventions.py is pinned to the reference by
    # tests/golden/conventions_rot6d.npz; T arrives as [3]
    R_expect = C.saved_pose_from_rot6d(inp["rotations_object"]).astype(np.float32)
    assert np.allclose(frames["R"].numpy(), R_expect, atol=1e-6)
    assert frames["T"].shape == (4, 3) and np.allclose(frames["T"].numpy(), inp["translations_object"].reshape(4, 3))
    assert np.array_equal(frames["K"].numpy(), C.intrinsics(H, W))
    for i in range(4):
        assert np.array_equal(frames["label"][i].numpy(), labels[i]), "SAM channel rule + 1/0/-1 label map"
    # camera centre in the object frame = -R^T T, and a vertex moved by vis.py:52 lands where R x + T says
    R, T = frames["R"][0].double(), frames["T"][0].double()
    v = torch.tensor([[0.1, -0.2, 0.3]], dtype=torch.float64)
    assert torch.allclose(torch.from_numpy(C.apply_pose(v.numpy(), R.numpy(), T.numpy())), v @ R.T + T)
    cam = -(R.T @ T)
    assert torch.allclose(R @ cam + T, torch.zeros(3, dtype=torch.float64), atol=1e-5)     # R is fp32-orthogonal
    assert abs(float(torch.det(R)) - 1.0) < 1e-5 and torch.allclose(R @ R.T, torch.eye(3, dtype=torch.float64), atol=1e-5)


def test_missing_pose_file_is_reported(tmp_path):
    import pytest
    from dynhor_amd.dataset import Dataset
    rng = np.random.default_rng(1)
    root = str(tmp_path / "seq")
    _write_frames(root, ["0001", "0002"], 16, 16, rng)           # 0002 has no pose in the golden folder
    with pytest.raises(FileNotFoundError):
        Dataset._load_from_disk({"dataroot": root, "obj_infos": GOLD})
