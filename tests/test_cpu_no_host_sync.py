"""CPU: the per-iteration path of both model families must not read device values back ("zero .item() /
int(tensor) in train_step_core of either family").  Source-level guard (the GPU tests additionally run the step under
torch.cuda.set_sync_debug_mode("error")): the functions below may not contain a tensor -> host conversion."""
import inspect
import re

import pytest

FORBIDDEN = [r"\.item\(", r"\.tolist\(", r"\.cpu\(", r"\.numpy\(", r"\bint\(\s*csum", r"\bint\(\s*\(?\s*m\.cnt", r"\bfloat\(\s*self\.binary",
             r"torch\.cuda\.synchronize\("]


def _hot_functions():
    from dynhor_amd import hash_fields, renderer
    H, N = hash_fields.HashNeuSRenderer, renderer.NeuSRenderer
    return [H.train_step_core, H.march, H._forward_packed, H._backward_packed, H._net_forward, H._net_backward, H._net_sdf_nograd,
            H.update_grid, hash_fields.OccupancyGrid.update, N.train_step_core, N._forward_core, N._backward_core, N.sample_z,
            N._net_forward, N._net_backward, N._net_sdf_nograd]


@pytest.mark.parametrize("fn", _hot_functions(), ids=lambda f: f.__qualname__)
def test_no_device_to_host_read_on_the_training_path(fn):
    src = inspect.getsource(fn)
    src = re.sub(r'"""[\s\S]*?"""', "", src)
    src = "\n".join(l.split("#")[0] for l in src.splitlines())
    for pat in FORBIDDEN:
        assert not re.search(pat, src), f"{fn.__qualname__} contains {pat}"
