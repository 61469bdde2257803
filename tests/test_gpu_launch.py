"""GPU: the multi-GPU entry points as the driver calls them (VERDICT r2 missing #1 / next #1).
  * RCCL executes the path's one collective at least once: `bench.py --gpus 1 --force-dist --backend nccl`.
  * `python bench.py --gpus 2` with NO torchrun on the command line starts its own two ranks (gloo, both on cuda:0).
  * `python -m dynhor_amd.run --gpus 2` does the same for the Runner CLI."""
import json
import math
import os
import subprocess
import sys

import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def _run(cmd, timeout=420):
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=_clean_env(), cwd=ROOT)
    except subprocess.TimeoutExpired as e:
        pytest.fail(f"{' '.join(cmd[1:4])} did not finish in {timeout} s: " + str(e.stdout)[-1500:] + str(e.stderr)[-1500:])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    return p


def test_single_rank_rccl_allreduce_runs():
    p = _run([sys.executable, "bench.py", "--gpus", "1", "--force-dist", "--backend", "nccl", "--steps", "3", "--warmup", "1",
              "--frames", "8", "--no-cpu-baseline"])
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), "stdout is ONE JSON line (RCCL's version banner must not land in it): " + p.stdout[:400]
    out = json.loads(lines[0])
    comm = out["comm"]
    assert comm["backend"] == "nccl" and comm["nranks"] == 1
    assert comm["rccl_version"] and all(c.isdigit() or c == "." for c in comm["rccl_version"]), comm
    assert math.isfinite(comm["allreduce_only_ms"]) and comm["allreduce_only_ms"] >= 0.0
    assert comm["bucket_bytes"] == 802491 * 4 and comm["bucket_persistent"] is True
    assert out["n_gpus"] == 1 and out["value"] > 0 and math.isfinite(out["final_stats"]["loss"])


def test_bench_starts_its_own_two_ranks():
    """The driver's command form: no torch.distributed.run on the command line."""
    p = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--kernel-steps", "2", "--frames", "8",
              "--backend", "gloo", "--share-gpu", "--no-cpu-baseline"])
    assert "check-sync ok" in p.stderr, "at N > 1 the parameter / frame check runs by default (VERDICT r3 next #6)"
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0's) must reach the parent's stdout"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["value"] > 0
    comm = out["comm"]
    assert comm["nranks"] == 2 and comm["bucket_persistent"] is True and "identical parameters" in comm["check_sync"]
    # first-contact diagnostics: every rank's own step time and device
    rm = comm["rank_ms_per_step"]
    assert len(rm["all"]) == 2 and rm["min"] <= rm["max"] and rm["rank_of_max"] in (0, 1) and rm["max"] <= out["ms_per_step"] * 1.001 + 1e-3
    assert [d["rank"] for d in comm["devices"]] == [0, 1] and all(d["arch"] and "gfx950" in d["arch"] and d["cus"] == 256 for d in comm["devices"])
    assert "secondary" not in out, "the secondary lines belong to the single-GPU default run"


def test_runner_cli_starts_its_own_two_ranks(tmp_path):
    conf = {"seq_name": "launch_test", "exp_name": "dp2",
            "data_info": {"synthetic": {"n_frames": 4, "H": 64, "W": 64, "seed": 5}},
            "train": {"batch_size": 256, "end_iter": 4, "report_freq": 2, "save_freq": 4, "val_freq": 0, "normal_weight": 0.05}}
    cpath = tmp_path / "c.yaml"
    cpath.write_text(yaml.safe_dump(conf))
    p = _run([sys.executable, "-m", "dynhor_amd.run", "--config_path", str(cpath), "--gpus", "2", "--backend", "gloo",
              "--share-gpu", "--exp_root", str(tmp_path / "exps")])
    assert "trained to iteration 4 on 2 rank(s)" in p.stdout
    exp = tmp_path / "exps" / "launch_test" / "dp2"
    assert (exp / "checkpoints" / "ckpt_000004.pth").exists()
    recs = [json.loads(l) for l in (exp / "scalars.jsonl").read_text().splitlines()]
    assert [r["iter"] for r in recs] == [2, 4] and all(math.isfinite(r["Loss/loss"]) for r in recs)
    # the reference's logger (run.py:127, jointopt.py:151-153): one scalar per key per step in <exp>/board
    import glob
    from dynhor_amd import tb_events
    ev = glob.glob(str(exp / "board" / "events.out.tfevents.*"))
    assert len(ev) == 1, "rank 0 alone writes the board"
    sc = tb_events.read_scalars(ev[0])
    assert [(st, v) for st, t, v in sc if t == "Loss/loss"] == [(r["iter"], pytest.approx(r["Loss/loss"], rel=1e-6)) for r in recs]


HASH_OVERLAP_TARGET = '''
import json, os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
torch.cuda.set_device(0)                                   # both ranks share the one GPU of the test box
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
from dynhor_amd.runner import Runner
conf = {"seq_name": "ov", "exp_name": "r%%d" %% rank, "data_info": {"synthetic": {"n_frames": 4, "H": 64, "W": 64, "seed": 5}},
        "train": {"batch_size": 256, "report_freq": 10 ** 9, "save_freq": 10 ** 9, "val_freq": 0, "normal_weight": 0.05},
        "model": {"family": "hash"}}
r = Runner(conf=conf, device="cuda:0", exp_root=sys.argv[1])
assert r.overlap_table_reduce and r.renderer.table_grad_hook is not None
tf = r.store.table_floats
seen = {}
inner = r.renderer.table_grad_hook
def hook(t):
    seen["local_table"] = t.clone()                        # this rank's table gradient before the collective touches it
    seen["calls"] = seen.get("calls", 0) + 1
    return inner(t)
r.renderer.table_grad_hook = hook
import dynhor_amd.dist as dd
orig = dd.allreduce_sum_
def spy(t):
    seen["rest_numel"] = t.numel()
    seen["local_rest"] = t.clone()
    return orig(t)
dd.allreduce_sum_ = spy
ok = True
for it in range(3):
    r.train_iteration()
    torch.cuda.synchronize()
    # the serial path on the same local gradients: ONE all-reduce of the whole bucket
    whole = torch.cat([seen["local_table"], seen["local_rest"]])
    dist.all_reduce(whole)
    got = r.store.grad_flat
    ok = ok and bool(torch.equal(whole, got)) and seen["rest_numel"] == got.numel() - tf
flat = r.store.flat.clone()
ref = flat.clone(); dist.broadcast(ref, src=0)
if rank == 0:
    print(json.dumps({"world": world, "overlapped_equals_serial_bitwise": ok, "hook_calls": seen["calls"], "table_floats": tf,
                      "rest_floats": seen["rest_numel"], "ranks_in_sync": bool(torch.equal(ref, flat))}), flush=True)
else:
    assert torch.equal(ref, flat)
dist.destroy_process_group()
'''


def test_hash_family_table_gradient_allreduce_overlaps_and_equals_the_serial_path(tmp_path):
    """the hash family's 49 MB table gradient is reduced by a collective of its own, issued behind the table scatter
    and joined before Adam; two ranks (gloo, one GPU): every iteration's reduced bucket equals ONE all-reduce of the same local
    gradients bit for bit, and the ranks stay in sync."""
    target = tmp_path / "ov.py"
    target.write_text(HASH_OVERLAP_TARGET % ROOT)
    code = ("import sys; sys.path.insert(0, %r); from dynhor_amd import launch; "
            "sys.exit(launch.spawn_ranks(%r, [%r], 2, timeout=400))" % (ROOT, str(target), str(tmp_path / "exps")))
    p = _run([sys.executable, "-c", code], timeout=500)
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["world"] == 2 and out["overlapped_equals_serial_bitwise"] and out["ranks_in_sync"]
    assert out["hook_calls"] == 3 and out["table_floats"] + out["rest_floats"] == 12206065


def test_bench_reports_two_collectives_for_the_hash_family():
    p = _run([sys.executable, "bench.py", "--gpus", "2", "--family", "hash", "--steps", "3", "--warmup", "1", "--kernel-steps", "2",
              "--frames", "8", "--backend", "gloo", "--share-gpu", "--no-cpu-baseline"])
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    comm = out["comm"]
    assert comm["collectives_per_step"] == 2 and sum(comm["collective_bytes"].values()) == 12206065 * 4
    assert "identical parameters" in comm["check_sync"]
