"""bench.py's --gpus handling, without a GPU: a WORLD_SIZE that contradicts --gpus ends the process with a non-zero
status before anything is imported or measured, and --gpus N > 1 without a torch.distributed environment starts the N
ranks itself (the command is checked, not run)."""
import os
import subprocess
import sys

from conftest import ROOT


def test_world_size_mismatch_exits_non_zero():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0
    assert "--gpus 4 but WORLD_SIZE=2" in r.stderr
    assert r.stdout.strip() == ""          # no JSON line from a mis-launched run


def test_gpus_without_a_launcher_spawns_one_rank_per_gpu(monkeypatch):
    import bench
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 7                 # the child's status is this process's status
    else:
        raise AssertionError("bench.main() must exit with the launcher's status")
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ


def test_weak_scaling_view_sets_repeat_the_one_rank_pattern():
    """Under weak scaling every rank renders the pattern of views the one-rank run renders (plus a small per-rank
    offset), view k belonging to rank k % world: the per-rank workload does not change with the number of ranks."""
    import math
    import types
    import numpy as np
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    import bench
    gs = types.SimpleNamespace(View=lambda cam, R, t: R)
    one = bench.make_views(gs, None, 8)
    assert all(np.allclose(a, b) for a, b in zip(one, bench.make_views(gs, None, 8, group=8, world=1)))
    for world in (2, 4, 8):
        views = bench.make_views(gs, None, 8 * world, group=8, world=world)
        assert len(views) == 8 * world
        for rank in range(world):
            mine = [views[k] for k in range(8 * world) if k % world == rank]
            for j, R in enumerate(mine):
                yaw = math.degrees(math.atan2(R[0, 2], R[2, 2]))      # R = Ry Rx
                assert abs(yaw - ((j - 3.5) * 1.5 + 0.11 * rank)) < 1e-9
                if rank == 0:
                    assert np.allclose(R, one[j])
        assert max(abs(math.degrees(math.atan2(R[0, 2], R[2, 2]))) for R in views) < 7.0
