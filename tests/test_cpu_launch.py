"""CPU: the one-node launcher (dynhor_amd/launch.py) -- `python bench.py --gpus N` / `python -m dynhor_amd.run --gpus N`
start their own ranks when WORLD_SIZE is unset (VERDICT r2 missing #1).  Here: command form, a real world-2 gloo launch of a
tiny target through spawn_ranks, exit-code relay, and bench.py's parent path never touching a GPU (there is none here)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TARGET = '''
import json, os, sys
import torch, torch.distributed as dist
dist.init_process_group("gloo")
t = torch.tensor([float(dist.get_rank() + 1)])
dist.all_reduce(t)
if "--fail-rank-1" in sys.argv and dist.get_rank() == 1:
    sys.exit(7)
if dist.get_rank() == 0:
    print(json.dumps({"sum": t.item(), "world": dist.get_world_size(), "argv": sys.argv[1:]}), flush=True)
dist.destroy_process_group()
'''


def test_rank_command_form():
    from dynhor_amd import launch
    cmd = launch.rank_command("/x/bench.py", ["--gpus", "8", "--steps", "5"], 8, port=29511)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "8", "--steps", "5"]
    cmd = launch.rank_command("dynhor_amd.run", ["--gpus", "2"], 2, port=1, module=True)
    assert cmd[-4:] == ["-m", "dynhor_amd.run", "--gpus", "2"]


def _spawn(tmp_path, extra):
    target = tmp_path / "target.py"
    target.write_text(TARGET)
    code = ("import sys; sys.path.insert(0, %r); from dynhor_amd import launch; "
            "sys.exit(launch.spawn_ranks(%r, %r, 2, timeout=240))" % (ROOT, str(target), ["--tag", "x"] + extra))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)


def test_spawn_ranks_runs_world_2_and_relays_rank0_stdout(tmp_path):
    p = _spawn(tmp_path, [])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out == {"sum": 3.0, "world": 2, "argv": ["--tag", "x"]}


def test_spawn_ranks_relays_a_failing_rank(tmp_path):
    p = _spawn(tmp_path, ["--fail-rank-1"])
    assert p.returncode != 0


def test_bench_parent_starts_ranks_without_touching_a_gpu_and_relays_their_failure():
    """The parent makes no device query at all (torch.cuda.device_count() can initialise the HIP runtime when amdsmi
    discovery fails): it only starts the ranks.  Here there is no GPU, so every rank dies in torch.cuda.set_device -- and the
    parent leaves with their non-zero exit code instead of a JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")]
    src = open(os.path.join(ROOT, "bench.py")).read()
    parent = src[src.index("if args.gpus > 1 and not launch.launched_by_torchrun():"):src.index("world = int(os.environ.get(\"WORLD_SIZE\"")]
    assert "torch.cuda" not in parent.replace("torch.cuda.set_device", "")


def test_bench_rejects_mismatched_world_size():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode == 2 and "WORLD_SIZE=4" in p.stderr


TARGET8 = '''
import json, os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from dynhor_amd.schedules import FramePermutation, frame_slot
from dynhor_amd.dist import allreduce_sum_
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
    sys.exit(9)                                  # before the first collective: the others are torn down by the launcher
perm = FramePermutation(64, seed=777)            # every rank owns a copy of the same seeded permutation
mine = [perm.frame(frame_slot(it, rank, world)) for it in range(20)]          # 20 iterations = 2.5 epochs at W = 8, F = 64
every = [None] * world
dist.all_gather_object(every, mine)
bucket = torch.full((802491,), float(rank + 1))                                # the flat gradient bucket's size
allreduce_sum_(bucket)
if rank == 0:
    disjoint = all(len({every[r][it] for r in range(world)}) == world for it in range(20))
    epoch_cover = sorted(every[r][it] for it in range(8) for r in range(world)) == list(range(64))
    print(json.dumps({"world": world, "disjoint_every_iteration": disjoint, "first_epoch_covers_all_frames": epoch_cover,
                      "bucket_sum": bucket[0].item(), "bucket_uniform": bool((bucket == bucket[0]).all())}), flush=True)
dist.destroy_process_group()
'''


def _spawn8(tmp_path, extra):
    target = tmp_path / "target8.py"
    target.write_text(TARGET8 % ROOT)
    code = ("import sys; sys.path.insert(0, %r); from dynhor_amd import launch; "
            "sys.exit(launch.spawn_ranks(%r, %r, 8, timeout=500))" % (ROOT, str(target), extra))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)


def test_eight_ranks_rendezvous_draw_disjoint_frames_and_reduce_the_bucket(tmp_path):
    """cfg3's process layout without GPUs: 8 self-launched gloo ranks; every iteration the ranks hold 8 distinct
    frames of the shared permutation (F = 64: one epoch = 8 iterations covers every frame once), and the one collective of the
    path -- the SUM all-reduce of the 802,491-float bucket -- gives every rank 1 + 2 + ... + 8."""
    p = _spawn8(tmp_path, [])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out == {"world": 8, "disjoint_every_iteration": True, "first_epoch_covers_all_frames": True, "bucket_sum": 36.0,
                   "bucket_uniform": True}


def test_eight_ranks_relay_the_exit_code_of_a_failing_rank(tmp_path):
    p = _spawn8(tmp_path, ["--fail-rank", "5"])
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")]
