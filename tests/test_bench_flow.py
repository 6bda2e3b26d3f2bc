"""bench.py's multi-rank control flow on CPU (gloo, world size 2): `python bench.py --gpus 2` must launch 2 ranks by itself,
every rank must enter the collective-bearing extra (the gradient all-reduce), timing is max-over-ranks, rank 0 prints ONE
JSON line with n_gpus = 2 (--gpus was ignored and a rank-0-only all-reduce would have hung)."""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env,
                          cwd=ROOT)


def _json_lines(out):
    lines = []
    for ln in out.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                lines.append(json.loads(ln))
            except ValueError:
                pass
    return lines


def test_gpus_flag_self_launches_two_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                       # ONE line, from rank 0
    d = lines[0]
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["extra"]["dry_run_cpu"] and d["extra"]["grad_sync_ok"]          # both ranks took part in the all-reduce
    # max over ranks: rank 1's stand-in step sleeps 4 ms, rank 0's 2 ms
    assert d["ms_per_step"] >= 3.9, d["ms_per_step"]
    assert d["config"]["parallelism"].startswith("roi-shard x2")


def test_single_rank_dry_run_and_world_mismatch():
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--dry-run-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_lines(r.stdout)[0]
    assert d["n_gpus"] == 1 and d["extra"]["grad_sync_ok"]
    # launched by an external torchrun with another world size than --gpus: refuse instead of mislabelling the line
    r = _run(["--gpus", "4", "--dry-run-cpu"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
