"""N > 1 path of bench.py on CPU: world_size 2, gloo, 127.0.0.1 rendezvous.  Ranks run independent scenes (distinct
seeds); the job rate is (units of all ranks) / (max elapsed over ranks); no data-path collective exists."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from gps_slam_amd.dist_util import Group, scene_seed
    g = Group(backend="gloo")
    g.barrier()
    elapsed = 1.0 + rank  # rank 1 is the slow one
    rate = g.aggregate_rate(100, elapsed)
    mx = g.max_over_ranks(elapsed)
    g.barrier()
    q.put((rank, scene_seed(rank), rate, mx))
    g.close()


def test_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] != res[1][1]                     # different scenes
    assert all(abs(r[3] - 2.0) < 1e-12 for r in res)  # max over ranks
    assert all(abs(r[2] - 200 / 2.0) < 1e-9 for r in res)  # 2 ranks x 100 units / slowest


def test_single_process_needs_no_group():
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    from gps_slam_amd.dist_util import Group
    g = Group()
    assert g.world == 1 and g.dist is None
    assert g.aggregate_rate(50, 2.0) == 25.0


def _bench_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import bench
    from gps_slam_amd.dist_util import scene_seed
    r, lr, w, grp, placement, device = bench.setup_ranks(backend="gloo", need_gpu=False)   # bench.py's own N > 1 entry path
    grp.barrier()
    first, prologue = bench.timed_window(5)
    dt = grp.max_over_ranks(0.5 + 0.25 * r)
    grp.barrier()
    q.put((r, w, scene_seed(r), first, prologue, dt, placement, sorted(os.sched_getaffinity(0))))
    grp.close()


def test_bench_multi_gpu_entry_path_with_gloo():
    """`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`: the rank set-up bench.py runs before any GPU work
    (device choice aside) -- environment ranks, NUMA / core pinning per rank, process-group init, barrier, max-over-ranks --
    exercised with world_size 2 on gloo; plus the timed-window arithmetic every rank must agree on."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1] and all(r[1] == 2 for r in res)
    assert res[0][2] != res[1][2]                                  # independent scenes
    assert all(r[3] == 30 and r[4] == 25 for r in res)             # --warmup 5: timed step 0 = frame 30 (a keyframe), 25 prologue frames
    assert all(abs(r[5] - 0.75) < 1e-12 for r in res)              # slowest rank's time on every rank
    assert all(r[6].startswith("affinity:") for r in res)
    if len(os.sched_getaffinity(0)) >= 4:                          # the two ranks were given disjoint core sets
        assert not (set(res[0][7]) & set(res[1][7]))


def test_timed_window_is_whole_keyframe_periods():
    import bench
    for w in (0, 5, 10, 20, 25, 31, 40):
        first, prologue = bench.timed_window(w)
        assert first % bench.PERIOD == 0 and first >= 30 and first - prologue == w and prologue >= 0
