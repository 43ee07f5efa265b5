"""CPU-only checks of bench.py's launcher side: `python bench.py --gpus N` without a distributed environment must start
its own N ranks as fresh children (torch.distributed.run, 127.0.0.1) and must not touch a GPU itself."""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_gpus_n_spawns_torchrun_children(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 7, "the launcher must exit with the children's code"
    else:
        raise AssertionError("launcher did not exit")
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_launcher_never_initialises_the_gpu():
    """The parent of the ranks must stay GPU-free (a process that initialised HIP must not be the one that forks ranks):
    run the launcher path with a stub torch.distributed.run module and check torch.cuda was never initialised."""
    code = (
        "import sys, os, subprocess, torch\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "subprocess.call = lambda cmd, env=None: 0\n"
        "sys.argv = ['bench.py', '--gpus', '2']\n"
        "os.environ.pop('WORLD_SIZE', None)\n"
        "try:\n    bench.main()\nexcept SystemExit as e:\n    assert e.code == 0\n"
        "assert not torch.cuda.is_initialized()\nprint('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-500:]
