"""One-node launcher of the data-parallel path: `python bench.py --gpus N` and
`python -m dynhor_amd.run --gpus N` start N child processes (one per GPU) through `torch.distributed.run` themselves when
nobody has done it for them (WORLD_SIZE unset).

The parent NEVER touches the GPU: it does not import the HIP library, makes no HIP call and does not ask torch whether a
device is available -- it only starts the children, lets them write to its own stdout / stderr (so rank 0's one JSON line
is the parent's one JSON line) and exits with their exit code.  A process that has initialised the GPU is never re-exec'ed.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launched_by_torchrun() -> bool:
    """True inside a rank started by torch.distributed.run (or any launcher that exports the rendezvous variables)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def rank_command(target, argv, nproc: int, port: int | None = None, module: bool = False):
    """argv of the torch.distributed.run call that starts `nproc` ranks of `target` (a script path, or a module name with
    module=True) on this node, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(nproc)}",
           "--master-addr", "127.0.0.1", "--master-port", str(port if port is not None else free_port())]
    cmd += ["-m", target] if module else [target]
    return cmd + list(argv)


def spawn_ranks(target, argv, nproc: int, module: bool = False, timeout: float | None = None) -> int:
    """Start the ranks as children, relay their output and return their exit code (nonzero if any rank failed)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")                 # silences torchrun's warning; the ranks are GPU-bound
    env.setdefault("PYTHONUNBUFFERED", "1")
    cmd = rank_command(target, argv, nproc, module=module)
    try:
        return subprocess.run(cmd, env=env, timeout=timeout).returncode
    except subprocess.TimeoutExpired:
        print(f"dynhor_amd.launch: {nproc} ranks did not finish within {timeout} s", file=sys.stderr, flush=True)
        return 124
