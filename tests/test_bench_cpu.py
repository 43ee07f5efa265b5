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


def test_pmc_traffic_attached_only_at_the_same_parallelism(tmp_path, monkeypatch):
    """the N > 1 line carried TP = 1 PMC bytes beside TP = N algorithmic bytes.  `traffic` is attached
    only when the summary's recorded parallelism (or, for files older than r05, the one its command line implies) is this
    run's; otherwise a note says why."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    d = tmp_path / "profiles" / "r99"
    d.mkdir(parents=True)
    pmc = {"grouped_gemm_down": {"hbm_bytes_corrected": 123}, "_meta": {
        "kernel_source_digest": "abc", "command": "bench.py --layers 3 --no-graph",
        "parallelism": {"world": 1, "emulate_tp": 1, "expert_parallel": False}}}
    (d / "pmc_traffic.json").write_text(json.dumps(pmc))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "_kernel_source_digest", lambda: "abc")
    monkeypatch.setattr(bench, "_git", lambda *a: None)
    k = {"grouped_gemm_down": {"bound": "hbm"}}
    bench.attach_pmc_traffic(k, {"world": 1, "emulate_tp": 1, "expert_parallel": False})
    assert k["grouped_gemm_down"]["traffic"] == 123
    for par in ({"world": 2, "emulate_tp": 1, "expert_parallel": False}, {"world": 1, "emulate_tp": 4, "expert_parallel": False},
                {"world": 2, "emulate_tp": 1, "expert_parallel": True}):
        k = {"grouped_gemm_down": {"bound": "hbm"}}
        bench.attach_pmc_traffic(k, par)
        assert "traffic" not in k["grouped_gemm_down"] and "not attached" in k["grouped_gemm_down"]["traffic_note"]
    # a pre-r05 summary: no `parallelism`, the command line says TP = 1
    del pmc["_meta"]["parallelism"]
    (d / "pmc_traffic.json").write_text(json.dumps(pmc))
    k = {"grouped_gemm_down": {"bound": "hbm"}}
    bench.attach_pmc_traffic(k, {"world": 2, "emulate_tp": 1, "expert_parallel": False})
    assert "traffic" not in k["grouped_gemm_down"]
    k = {"grouped_gemm_down": {"bound": "hbm"}}
    bench.attach_pmc_traffic(k)
    assert k["grouped_gemm_down"]["traffic"] == 123
    # other kernel sources: never attached
    monkeypatch.setattr(bench, "_kernel_source_digest", lambda: "other")
    k = {"grouped_gemm_down": {"bound": "hbm"}}
    bench.attach_pmc_traffic(k)
    assert "traffic" not in k["grouped_gemm_down"] and "other kernel sources" in k["grouped_gemm_down"]["traffic_note"]


def _collective_record_worker(rank, world, port, q):
    try:
        for p in (ROOT, os.path.join(ROOT, "glm4_moe_nano-vllm_amd")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import bench
        from nanovllm.distributed.parallel_state import get_tp_group
        tp = get_tp_group()
        tp.reset_counters()
        for n in (8, 4096, 4096):          # a "step": three all-reduce sites
            t = torch.full((n,), float(rank + 1), dtype=torch.bfloat16)
            tp.all_reduce(t)
            assert float(t[0]) == sum(range(1, world + 1))
        rec = bench.collective_record(tp, world, torch.device("cpu"))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, rec))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()))


def test_collective_record_proves_the_world_gloo():
    """The `collective` object of an N > 1 bench line is filled from the live process group: backend, world, the number and
    size of the all-reduces one step issued (the TP coordinator's own counters) and, per rank, the world size, the rank
    and the sum of ones it saw.  Two gloo ranks on the CPU."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    ps = [ctx.Process(target=_collective_record_worker, args=(r, world, 29763, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=240) for _ in ps)
    for p in ps:
        p.join(30)
    for r in range(world):
        rec = res[r]
        assert isinstance(rec, dict), rec
        assert rec["backend"] == "gloo" and rec["world"] == 2 and rec["ranks_agree"] is True
        assert rec["allreduce_per_step"] == 3 and rec["bytes_per_allreduce"] == 8192 and rec["oneshot"] is False
        assert rec["allreduce_bytes_per_step"] == 16 + 8192 + 8192
        assert rec["rccl_world_per_rank"] == [[2, 0, 2], [2, 1, 2]]
