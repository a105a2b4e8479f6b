"""bench.py's multi-GPU entry, on CPU: `python bench.py --gpus 2` without a launcher must start two ranks itself
(re-exec under torch.distributed.run), run the sharded step with its collective inside the timed region, and print ONE
JSON line from rank 0.  The compute engine is the bench's `stub` (gloo, no kernels): what is tested is the launch /
barrier / max-over-ranks / reporting path that the driver's SCALE run depends on."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _json_lines(text):
    out = []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("{"):
            out.append(json.loads(line))
    return out


def test_gpus_flag_spawns_the_ranks_itself():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--engine", "stub", "--steps", "4",
                        "--warmup", "1"], capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout  # rank 0 only
    d = lines[0]
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 4 and d["warmup"] == 1
    assert d["scaling"] == "weak" and d["config"]["global_batch"] == 64
    assert d["collective"]["bytes"] == (640 * 640 + 640 + 640 * 28 + 28) * 4 and d["collective"]["bucket_ok"]
    assert d["value"] is None and d["engine"] == "stub"  # a stub run can never be mistaken for a measurement


def test_single_rank_needs_no_launcher_and_a_contradicting_launcher_is_refused():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--engine", "stub", "--steps", "2", "--warmup", "0"],
                       capture_output=True, text=True, env=_env(), timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_lines(r.stdout)[0]["n_gpus"] == 1
    env = _env()
    env.update(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--engine", "stub"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "contradicts" in (r.stderr + r.stdout)
