"""bench.py started as `python bench.py --gpus N` (no torch.distributed.run around it) launches its
own ranks, relays rank 0's result line and propagates a failure; with too few GPUs it refuses before
anything touches a device.  These run on the CPU: the ranks are stand-in scripts, the launcher is real."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_launcher_relays_rank0_line_and_uses_gloo_rendezvous(tmp_path, capsys, monkeypatch):
    monkeypatch.setenv("SPMV_BENCH_BACKEND", "gloo")
    script = _script(tmp_path, """
        import json, os, sys
        import torch.distributed as dist
        dist.init_process_group("gloo")
        import torch
        t = torch.tensor([dist.get_rank() + 1.0])
        dist.all_reduce(t)
        if dist.get_rank() == 0:
            print("chatter that is not the result")
            print(json.dumps({"metric": "m", "value": float(t), "world": int(os.environ["WORLD_SIZE"]), "argv": sys.argv[1:]}))
        dist.destroy_process_group()
    """)
    argv = ["--gpus", "2", "--steps", "3"]
    rc = bench.launch_ranks(bench.parse_args(argv), argv, script=script)
    out = capsys.readouterr()
    assert rc == 0
    lines = [ln for ln in out.out.splitlines() if ln.strip()]
    assert len(lines) == 1                                   # ONE JSON line on stdout
    rec = json.loads(lines[0])
    assert rec["value"] == 3.0 and rec["world"] == 2 and rec["argv"] == argv
    assert "chatter" in out.err


def test_launcher_propagates_a_failing_rank(tmp_path, capsys, monkeypatch):
    monkeypatch.setenv("SPMV_BENCH_BACKEND", "gloo")
    script = _script(tmp_path, """
        import os, sys
        sys.exit(7 if os.environ["RANK"] == "1" else 0)
    """)
    argv = ["--gpus", "2"]
    rc = bench.launch_ranks(bench.parse_args(argv), argv, script=script)
    assert rc != 0
    assert "exited with status" in capsys.readouterr().err


def test_more_ranks_than_gpus_is_refused_with_both_numbers():
    import torch
    have = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SPMV_BENCH_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 2)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert f"--gpus {have + 2}" in p.stderr and f"has {have}" in p.stderr


def test_world_size_that_differs_from_gpus_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE is 2" in p.stderr and "--gpus is 4" in p.stderr


def test_traffic_records_are_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    h = bench.kernel_source_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "hbm_traffic.json").write_text(json.dumps({
        "good": {"bytes": 123, "kernel_sources_sha256": h, "source": "x.json"},
        "stale": {"bytes": 456, "kernel_sources_sha256": "0" * 64, "source": "y.json"},
        "legacy": 789}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: h)
    assert bench.recorded_traffic("good") == (123, "x.json")
    assert bench.recorded_traffic("stale")[0] is None
    assert bench.recorded_traffic("legacy")[0] is None
    assert bench.recorded_traffic("absent")[0] is None
