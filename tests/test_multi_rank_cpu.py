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
