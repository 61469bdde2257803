"""Generate tests/golden/obj_infos_ref/*.npz by IMPORTING the reference (runs only in the build container).

The stage-1 -> stage-2 hand-off file: ObjTracker/run.py:165-179 saves, per frame,
    R = rot6d_to_matrix(model.rotations_object).transpose(1, 2)[i]     (object -> camera, [3,3] float32)
    T = model.translations_object[i]                                    ([1,3] float32 -- NOT [3])
    K = camintr                                                         ([3,3] float32, run.py:119-123)
under obj_infos/<image stem>.npz with the stem cut by `[:-4]` from a `*.jpg` path (run.py:99,178).  This script performs exactly
those statements on seeded 6-D rotations (reference utils/geometry.py:rot6d_to_matrix does the conversion) so that the loader is
checked against files laid out by the reference's own writer, not by this repo's write_sequence_to_disk.  Data only.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference/ObjTracker"
sys.path.insert(0, REF)
from utils.geometry import rot6d_to_matrix  # noqa: E402


def main():
    g = torch.Generator().manual_seed(20250906)
    n, height, width = 4, 48, 64                                   # never 3 frames: torch.cross without dim
    rotations_object = torch.randn(n, 3, 2, generator=g)           # model.rotations_object
    translations_object = torch.randn(n, 1, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 2.2])
    image_paths = ["/data/custom_seq/rgb/%04d.jpg" % (7 * i + 1) for i in range(n)]
    focal = 1.2 * min(height, width)
    camintr = np.array([[focal, 0, width // 2], [0, focal, height // 2], [0, 0, 1]]).astype(np.float32)
    # ---- run.py:165-179, statement for statement
    obj_rot = rot6d_to_matrix(rotations_object).transpose(1, 2)
    obj_trans = translations_object
    obj_rot_np = obj_rot.detach().cpu().numpy()
    obj_trans_np = obj_trans.detach().cpu().numpy()
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "obj_infos_ref")
    os.makedirs(dst, exist_ok=True)
    for i in range(len(image_paths)):
        data = {"R": obj_rot_np[i], "T": obj_trans_np[i], "K": camintr}
        path_id = image_paths[i].split("/")[-1][:-4]
        np.savez(os.path.join(dst, "{}.npz".format(path_id)), **data)
    # one more sequence folder whose files carry `obj_scale`, the optional key ObjTracker/vis.py:48-52 reads back
    # (`obj_v_trans = (obj_scale * obj_verts_can) @ R.T + T`): same R / K, T = obj_scale * (the T above), so that a reader
    # that honours the key recovers the SAME canonical cameras and one that ignores it is off by the factor.  Frame 0015 has no
    # file: vis.py:44 skips frames without a pose.
    dst_s = os.path.join(os.path.dirname(os.path.abspath(__file__)), "obj_infos_scaled")
    os.makedirs(dst_s, exist_ok=True)
    obj_scales = [2.0, 0.5, None, 1.25]
    for i in range(len(image_paths)):
        if obj_scales[i] is None:
            continue
        data = {"R": obj_rot_np[i], "T": obj_trans_np[i] * np.float32(obj_scales[i]), "K": camintr,
                "obj_scale": np.float32(obj_scales[i])}
        path_id = image_paths[i].split("/")[-1][:-4]
        np.savez(os.path.join(dst_s, "{}.npz".format(path_id)), **data)
    np.savez(os.path.join(dst, "_inputs.npz"), rotations_object=rotations_object.numpy(), translations_object=translations_object.numpy(),
             height=height, width=width)
    print("wrote", dst, sorted(os.listdir(dst)))


if __name__ == "__main__":
    main()
