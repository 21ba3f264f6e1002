"""bench.py as its own launcher: `python bench.py --gpus N` (no WORLD_SIZE) must start N ranks before any GPU call.

--dry-launch makes every rank print the rendezvous environment it was given and exit, so the launcher's process
handling is testable without a GPU (VERDICT r1 item 1a / ADVICE bench.py:129)."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(argv, env_extra=None, timeout=120):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)


def test_dry_launch_starts_two_fresh_ranks():
    res = _run(["--gpus", "2", "--dry-launch"])
    assert res.returncode == 0, res.stderr
    # rank 0's line is relayed on stdout (exactly one JSON line), rank 1's goes to stderr as diagnostics
    out_lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(out_lines) == 1
    r0 = json.loads(out_lines[0])
    r1 = json.loads([ln for ln in res.stderr.splitlines() if ln.startswith("[rank 1] {")][0][len("[rank 1] "):])
    assert (r0["RANK"], r0["LOCAL_RANK"], r0["WORLD_SIZE"]) == ("0", "0", "2")
    assert (r1["RANK"], r1["LOCAL_RANK"], r1["WORLD_SIZE"]) == ("1", "1", "2")
    assert r0["MASTER_ADDR"] == r1["MASTER_ADDR"] == "127.0.0.1"
    assert r0["MASTER_PORT"] == r1["MASTER_PORT"] and int(r0["MASTER_PORT"]) > 0
    assert r0["pid"] != r1["pid"]
    # children are fresh interpreters that had not imported torch when they reported
    assert r0["torch_imported"] is False and r1["torch_imported"] is False


def test_torchrun_form_is_not_relaunched():
    # with WORLD_SIZE set (the torchrun form) the script IS a rank: it must not spawn anything
    res = _run(["--gpus", "2", "--dry-launch"], {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1", "MASTER_ADDR": "127.0.0.1",
                                                 "MASTER_PORT": "29999"})
    assert res.returncode == 0, res.stderr
    r = json.loads(res.stdout.strip().splitlines()[-1])
    assert r["RANK"] == "1" and r["MASTER_PORT"] == "29999"


def test_launcher_reports_a_failed_rank():
    # without a GPU every real rank exits non-zero ("bench.py needs a GPU"): the launcher must say so, not print a result
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("needs a box without a GPU")
    res = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], timeout=300)
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert "failed" in res.stderr


def test_parent_never_imports_torch():
    # the launcher branch runs before any `import torch` / `import kofft_amd` statement executes
    src = (ROOT / "bench.py").read_text()
    main_src = src[src.index("def main():"):]
    assert main_src.index("launch_ranks(") < main_src.index("run_rank(args)")
    top = src[:src.index("def parse(")]
    assert "import torch" not in top and "import kofft_amd" not in top and "import numpy" not in top


def test_dry_protocol_eight_ranks_one_line_with_every_workload_key():
    """The N = 8 contract without a GPU: the launcher starts eight fresh ranks, they rendezvous on 127.0.0.1 (gloo),
    run the barrier / all-reduce steps of the timing protocol, and exactly ONE JSON line comes back, carrying every key of the real line
    and the workloads an 8-rank run measures (BASELINE configs #3, #4, #5; the 8(f) rows are N = 1 only)."""
    res = _run(["--gpus", "8", "--dry-launch", "--dry-protocol"], timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    out_lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(out_lines) == 1
    line = json.loads(out_lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "workloads"):
        assert key in line, key
    assert line["n_gpus"] == 8 and line["WORLD_SIZE"] == "8" and line["launcher"] == "self"
    assert list(line["workloads"]) == ["stft1024", "rfft2048", "c64_2p20"]
    assert "multi_single_process" in line  # rank 0's single-process RCCL-vs-direct gather A/B of config #4 (N > 1 only)
    # one rank: the SURVEY 8(f) rows ride along
    res1 = _run(["--dry-launch", "--dry-protocol"], timeout=300)
    assert res1.returncode == 0, res1.stderr[-2000:]
    line1 = json.loads([ln for ln in res1.stdout.splitlines() if ln.startswith("{")][-1])
    assert list(line1["workloads"]) == ["stft1024", "rfft2048", "c64_2p20", "istft1024", "magnitudes1024", "fft2d_4096", "bluestein1000"]
