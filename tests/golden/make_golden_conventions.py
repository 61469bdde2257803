"""Generate tests/golden/conventions_rot6d.npz by IMPORTING the reference (runs only in the build container).

Only ObjTracker/utils/geometry.py and utils/constants.py import here; batch sizes avoid B == 3
where the reference's dim-less torch.cross misbehaves.  The .npz holds inputs and the reference's
outputs -- data only.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference/ObjTracker"
sys.path.insert(0, REF)
from utils.geometry import matrix_to_rot6d, rot6d_to_matrix  # noqa: E402
from utils import constants  # noqa: E402


def main():
    g = torch.Generator().manual_seed(20250905)
    out = {}
    for B in (1, 2, 5, 16):
        x = torch.randn(B, 3, 2, generator=g)
        R = rot6d_to_matrix(x.clone())
        out[f"in_B{B}"] = x.numpy()
        out[f"R_B{B}"] = R.numpy()
        out[f"saved_R_B{B}"] = R.transpose(1, 2).numpy()          # run.py:166
        out[f"rot6d_back_B{B}"] = matrix_to_rot6d(R).numpy()
    out["REND_SIZE"] = np.array(constants.REND_SIZE)
    out["BBOX_EXPANSION_FACTOR"] = np.array(constants.BBOX_EXPANSION_FACTOR)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conventions_rot6d.npz")
    np.savez(dst, **out)
    print("wrote", dst, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
