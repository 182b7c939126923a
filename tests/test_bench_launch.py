"""bench.py's N>1 entry: `python bench.py --gpus N` must start its own N ranks
(one process per GPU under torch.distributed.run) instead of demanding a
torchrun environment.  Host logic only — no GPU, no rendezvous."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _dry(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    r = subprocess.run([sys.executable, BENCH, *argv, "--dry-launch"], env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])["launch"]


def test_gpus_2_builds_the_torchrun_command():
    cmd = _dry("--gpus", "2", "--steps", "7", "--warmup", "3", "--master-port", "29712")
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29712"
    at = cmd.index(BENCH)
    tail = cmd[at + 1:]
    assert tail[:6] == ["--gpus", "2", "--steps", "7", "--warmup", "3"]
    assert "--dry-launch" not in tail


def test_gpus_8_and_forced_world_1():
    cmd = _dry("--gpus", "8")
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0          # a free port was picked
    cmd = _dry("--gpus", "1", env={"BENCH_FORCE_DIST": "1"})
    assert cmd[cmd.index("--nproc-per-node") + 1] == "1"


def test_world_size_mismatch_is_a_message_not_an_assert():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "AssertionError" not in r.stderr
    assert "WORLD_SIZE=4" in r.stderr


def test_default_line_options_and_the_t1_cpu_baseline_path(monkeypatch):
    """The driver's default command keeps the CPU linearity check and the `other_configs` sub-records on; the full-batch
    CPU baseline of BASELINE configs[1] (oracle uniform replay + torch-CPU DQN step, bench.CpuPathT1) runs on a host
    without a GPU and reports every field the bench line quotes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(sys, "argv", [BENCH])
    args = bench.parse()
    assert args.cpu_linearity_check and not args.no_other_configs and args.gpus == 1 and args.other_steps == 200
    monkeypatch.setattr(sys, "argv", [BENCH, "--no-cpu-linearity-check", "--no-other-configs"])
    args = bench.parse()
    assert not args.cpu_linearity_check and args.no_other_configs
    cpu = bench.CpuPathT1("dqn_uniform", fill_steps=80)
    assert cpu.B == 256 and not cpu.per and not cpu.iqn
    parts = cpu.learner_step()
    assert len(parts) == 3 and all(p >= 0 for p in parts) and parts[1] > 0
    assert cpu.acting_rate(EA=8, reps=1) > 0
    assert bench.file_date(os.path.join(ROOT, "profiles", "gather_traffic.json")).startswith("2026-10-01")
