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


def test_frames_without_a_pose_file_are_skipped_like_vis_py(tmp_path, capsys):
    """ObjTracker/vis.py:44: `if os.path.exists(obj_info_path)` -- a frame without a pose is skipped, not an error."""
    import pytest
    from dynhor_amd.dataset import Dataset
    rng = np.random.default_rng(1)
    root = str(tmp_path / "seq")
    _write_frames(root, ["0001", "0002", "0008"], 16, 16, rng)           # 0002 has no pose in the golden folder
    frames = Dataset._load_from_disk({"dataroot": root, "obj_infos": GOLD})
    assert frames["stems"] == ["0001", "0008"] and frames["skipped"] == ["0002"] and frames["rgb"].shape[0] == 2
    assert "1 of 3 frames have no pose file" in capsys.readouterr().out
    root2 = str(tmp_path / "seq2")
    _write_frames(root2, ["0002", "0003"], 16, 16, rng)
    with pytest.raises(FileNotFoundError):                               # no frame with a pose at all
        Dataset._load_from_disk({"dataroot": root2, "obj_infos": GOLD})


def test_obj_scale_is_honoured(tmp_path):
    """vis.py:48-52: x_cam = R (s x_obj) + T.  tests/golden/obj_infos_scaled/ (written by make_golden_obj_infos.py from the
    reference's statements) holds the SAME cameras as obj_infos_ref/ with T multiplied by s and the key obj_scale = s: a reader
    that honours the key recovers the canonical translations, and frame 0015 (no file there) is skipped."""
    from dynhor_amd.dataset import Dataset
    inp = np.load(os.path.join(GOLD, "_inputs.npz"))
    H, W = int(inp["height"]), int(inp["width"])
    scaled = os.path.join(os.path.dirname(GOLD), "obj_infos_scaled")
    raw = np.load(os.path.join(scaled, "0001.npz"))
    assert "obj_scale" in raw.files and float(raw["obj_scale"]) == 2.0
    rng = np.random.default_rng(2)
    root = str(tmp_path / "custom_seq")
    _write_frames(root, ["0001", "0008", "0015", "0022"], H, W, rng)
    ref = Dataset._load_from_disk({"dataroot": root, "obj_infos": GOLD})
    got = Dataset._load_from_disk({"dataroot": root, "obj_infos": scaled})
    assert got["stems"] == ["0001", "0008", "0022"] and got["skipped"] == ["0015"]
    assert got["obj_scale"].tolist() == [2.0, 0.5, 1.25]
    keep = [0, 1, 3]
    assert np.allclose(got["T"].numpy(), ref["T"].numpy()[keep], rtol=1e-6, atol=1e-7)
    assert np.array_equal(got["R"].numpy(), ref["R"].numpy()[keep])
    # the reference's own formula: a canonical vertex lands, up to the factor s a pinhole camera cannot see, where R x + T/s says
    for k, i in enumerate(keep):
        info = np.load(os.path.join(scaled, got["stems"][k] + ".npz"))
        v = np.array([[0.1, -0.2, 0.3]], np.float64)
        vis = (float(info["obj_scale"]) * v) @ info["R"].astype(np.float64).T + info["T"].astype(np.float64)       # vis.py:52
        ours = v @ got["R"][k].double().numpy().T + got["T"][k].double().numpy()
        assert np.allclose(vis / float(info["obj_scale"]), ours, atol=1e-6)
        K = got["K"].double().numpy()
        pa, pb = (K @ vis.T).T, (K @ ours.T).T
        assert np.allclose(pa[:, :2] / pa[:, 2:], pb[:, :2] / pb[:, 2:], atol=1e-4)                             # same pixel


def test_per_frame_intrinsics_must_agree(tmp_path):
    import pytest
    from dynhor_amd.dataset import Dataset
    rng = np.random.default_rng(3)
    root = str(tmp_path / "seq")
    _write_frames(root, ["0001", "0008"], 16, 16, rng)
    poses = tmp_path / "poses"
    poses.mkdir()
    for s in ("0001", "0008"):
        z = dict(np.load(os.path.join(GOLD, s + ".npz")))
        if s == "0008":
            z["K"] = z["K"].copy(); z["K"][0, 0] += 1.0
        np.savez(poses / (s + ".npz"), **z)
    with pytest.raises(ValueError, match="intrinsics differ"):
        Dataset._load_from_disk({"dataroot": root, "obj_infos": str(poses)})


def test_off_image_and_non_finite_matches_are_dropped_at_load_time():
    """a matcher's kpts0 can round to x = W or be negative, kpts1 / conf can be NaN -- dh_gen_rays would index
    another pixel (or past the last frame)."""
    from dynhor_amd.dataset import Dataset
    F, H, W = 2, 8, 10
    frames = {"rgb": torch.zeros(F, H, W, 3, dtype=torch.uint8), "label": torch.zeros(F, H, W, dtype=torch.int8),
              "normal": torch.zeros(F, H, W, 3, dtype=torch.uint8), "R": torch.eye(3).repeat(F, 1, 1), "T": torch.zeros(F, 3),
              "K": torch.tensor(C.intrinsics(H, W))}
    k0 = torch.tensor([[1.2, 2.7], [9.6, 3.0], [-0.6, 1.0], [4.0, 7.49], [4.0, 7.6], [3.0, 3.0], [2.0, 2.0]])
    k1 = torch.tensor([[1.0, 1.0]] * 5 + [[float("nan"), 1.0]] + [[5.0, 5.0]])
    cf = torch.tensor([0.9, 0.9, 0.9, 0.9, 0.9, 0.9, float("inf")])
    frames["matches"] = [{"i": 0, "j": 1, "kpts0": k0, "kpts1": k1, "conf": cf}]
    ds = Dataset(frames=frames, device="cpu")
    assert ds.corr_dropped == 5 and ds.corr.shape == (2, 6)
    assert ds.corr[:, :2].tolist() == [[1.0, 3.0], [4.0, 7.0]]
    assert float(ds.corr[:, 0].max()) <= W - 1 and float(ds.corr[:, 1].max()) <= H - 1
