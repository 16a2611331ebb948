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


import pytest


@pytest.mark.parametrize("world", [2, 8])
def test_bench_multi_gpu_entry_path_with_gloo(world):
    """`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`: the rank set-up bench.py runs before any GPU work
    (device choice aside) -- environment ranks, NUMA / core pinning per rank, process-group init, barrier, max-over-ranks --
    exercised with world_size 2 and 8 (BASELINE configs[4]: one scene per GPU of an 8-GPU node) on gloo; plus the timed-window
    arithmetic every rank must agree on."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world)) and all(r[1] == world for r in res)
    assert len({r[2] for r in res}) == world                       # independent scenes
    assert all(r[3] == 30 and r[4] == 25 for r in res)             # --warmup 5: timed step 0 = frame 30 (a keyframe), 25 prologue frames
    assert all(abs(r[5] - (0.5 + 0.25 * (world - 1))) < 1e-12 for r in res)   # slowest rank's time on every rank
    assert all(r[6].startswith("affinity:") for r in res)
    if len(os.sched_getaffinity(0)) >= world:                      # the ranks were given disjoint, non-empty core sets
        sets = [set(r[7]) for r in res]
        assert all(sets) and sum(len(x) for x in sets) == len(set().union(*sets))


def _cpus(text):
    from gps_slam_amd.dist_util import _parse_cpulist
    return sorted(_parse_cpulist(text))


@pytest.mark.parametrize("name,node_cpus,gpu_nodes,allowed", [
    # two sockets, SMT siblings numbered after the physical cores (the usual EPYC layout), 4 GPUs per socket
    ("2 nodes contiguous", {0: "0-63,128-191", 1: "64-127,192-255"}, [0, 0, 0, 0, 1, 1, 1, 1], "0-255"),
    ("2 nodes interleaved", {0: "0-63,128-191", 1: "64-127,192-255"}, [0, 1, 0, 1, 0, 1, 0, 1], "0-255"),
    ("2 nodes uneven", {0: "0-63,128-191", 1: "64-127,192-255"}, [0, 0, 0, 0, 0, 0, 1, 1], "0-255"),
    ("1 node", {0: "0-255"}, [0] * 8, "0-255"),
    ("1 node, container cpuset", {0: "0-255"}, [0] * 8, "8-39"),
    ("2 nodes, cpuset inside node 0 only", {0: "0-63,128-191", 1: "64-127,192-255"}, [0, 0, 0, 0, 1, 1, 1, 1], "0-31"),
    ("no numa information", {}, [None] * 8, "0-63"),
    ("numa_node = -1 (a VM)", {0: "0-63"}, [-1] * 8, "0-63"),
    ("8 nodes (NPS4 x 2 sockets)", {n: "%d-%d" % (16 * n, 16 * n + 15) for n in range(8)}, list(range(8)), "0-127"),
])
def test_eight_ranks_get_disjoint_non_empty_core_sets(name, node_cpus, gpu_nodes, allowed):
    """BASELINE configs[4] placement without the hardware: plan_affinity (what pin_to_gpu_numa applies) on cpulist fixtures of
    2-node, 1-node, 8-node and NUMA-less hosts, with contiguous / interleaved / uneven GPU-to-node maps and container cpusets:
    the 8 slices are non-empty, pairwise disjoint, inside the allowed set, and on the rank's own node whenever the plan is `numa`."""
    from gps_slam_amd.dist_util import plan_affinity
    nodes = {n: _cpus(t) for n, t in node_cpus.items()}
    allow = _cpus(allowed)
    # SMT siblings as Linux numbers them on these hosts: cpu c and c + 128 are one physical core (fixtures with < 128 cpus: no SMT)
    core_of = {c: c % 128 for c in allow} if max(allow) >= 128 else None
    plans, how = plan_affinity(8, gpu_nodes, nodes, allow, core_of=core_of)
    assert len(plans) == 8 and all(plans), name
    if core_of:   # no physical core is shared by two ranks
        owners = {}
        for r, p in enumerate(plans):
            for c in p:
                assert owners.setdefault(core_of[c], r) == r, (name, c)
    flat = [c for p in plans for c in p]
    assert len(flat) == len(set(flat)) and set(flat) <= set(allow), name
    if how == "numa":
        for r, p in enumerate(plans):
            assert set(p) <= set(nodes[gpu_nodes[r]]), (name, r)
        # ranks of one node share it evenly (sizes differ by nothing: integer division of the node's allowed cores)
        for n in set(gpu_nodes):
            sizes = {len(plans[r]) for r in range(8) if gpu_nodes[r] == n}
            assert len(sizes) == 1, (name, n, sizes)
    else:
        assert name.startswith(("no numa", "numa_node", "2 nodes, cpuset inside")), name


def test_fewer_cores_than_ranks_still_gives_every_rank_a_core():
    from gps_slam_amd.dist_util import plan_affinity
    plans, _ = plan_affinity(8, [None] * 8, {}, [0, 1, 2])
    assert len(plans) == 8 and all(plans) and all(set(p) <= {0, 1, 2} for p in plans)


class _StubScene:
    """what bench.main() touches of a Scene: run(lo, hi), close(), cli.uploadedBytes, pipe.stats()"""

    class _Cli:
        uploadedBytes = 0

    class _Pipe:
        def __init__(self):
            self.n = dict(frames=0, opt_iters=0, raycasts=0, added=0, pruned=0)

        def stats(self):
            return dict(self.n)

    def __init__(self, overlap, delay):
        self.cli, self.pipe, self.overlap, self.delay, self.log = self._Cli(), self._Pipe(), overlap, delay, []

    def run(self, lo, hi):
        import time
        self.log.append((lo, hi))
        time.sleep(self.delay * (hi - lo))
        self.pipe.n["frames"] += hi - lo
        self.pipe.n["opt_iters"] += 2 * (hi - lo)
        self.cli.uploadedBytes += 6 * 64 * 48 * (hi - lo)

    def close(self):
        pass


def _main_worker(rank, world, port, out_dir):
    import contextlib
    import io
    import json
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), GPS_BENCH_SIDE_DIR=os.path.join(out_dir, "side%d" % rank))
    os.makedirs(os.environ["GPS_BENCH_SIDE_DIR"])
    import bench
    scenes = []

    def factory(overlap):
        scenes.append(_StubScene(overlap, delay=0.002 * (1 + rank)))  # rank 1 is twice as slow per frame
        return scenes[-1]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", "10", "--warmup", "5", "--windows", "3"], scene_factory=factory, backend="gloo",
                   need_gpu=False, extras=False)
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump({"stdout": buf.getvalue(), "log": [s.log for s in scenes], "overlap": [s.overlap for s in scenes]}, f)


def _reject_constant(name):
    raise ValueError("%s is not JSON" % name)


def test_compact_line_of_a_full_record_fits_and_is_strict_json():
    """bench.compact_line on a real full record (round 5's 23 KB line, profiles/r05_bench_line.json): under 4 KB, parses with
    NaN / Infinity rejected, carries the headline, the flat scalars, `roofline` with bound / achieved / peak / unit / frac /
    traffic and `cpu_baseline` with value / unit / cores / kind / sample; emit() refuses a record holding a NaN."""
    import io
    import json
    import contextlib
    import bench
    full = json.load(open(os.path.join(os.path.dirname(__file__), "..", "profiles", "r05_bench_line.json")))
    text = json.dumps(bench.compact_line(full), allow_nan=False, separators=(",", ":"))
    assert len(text.encode()) < 4096
    line = json.loads(text, parse_constant=_reject_constant)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in line, k
    assert abs(line["value"] - full["value"]) < 1e-4 * full["value"] and line["dtype"] == "f32"
    assert all(not isinstance(v, (dict, list)) for v in line["config"].values())
    for k in ("workload", "sequential_fps", "overlap_fps", "whole_run_fps", "cfg1_fps", "cfg3_fps", "cfgR_fps", "keyframe_theta_deg", "gaussians"):
        assert k in line["config"], k
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "avg_launch_us", "frame_frac", "iteration_frac"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample", "threads", "tracked_value"):
        assert k in line["cpu_baseline"], k
    bad = dict(full, value=float("nan"))
    with pytest.raises(AssertionError), contextlib.redirect_stdout(io.StringIO()):
        bench.emit(bad, side_dir="/nonexistent")


@pytest.mark.parametrize("world", [2, 8])
def test_bench_main_end_to_end_on_gloo(tmp_path, world):
    """The WHOLE bench.main() -- argument parsing, rank set-up, per-rank seeds, both schedules, prologue + warm-up, the timed
    windows each bracketed by barriers, max-over-ranks, whole-job aggregation, the rank-0-only JSON line, the final barrier
    before the group is torn down -- with world_size 2 and 8 (the driver's N = 8 launch of BASELINE configs[4], one independent
    scene per rank) on gloo and a stub scene whose frames take 2 ms x (1 + rank): the line must carry n_gpus = world, value =
    world x K frames / the SLOWEST rank's median window, and no cpu_baseline."""
    import json
    port = _free_port()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=_main_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(240)
        assert p.exitcode == 0
    res = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(world)]
    assert all(res[r]["stdout"].strip() == "" for r in range(1, world))     # only rank 0 prints
    lines = [l for l in res[0]["stdout"].splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{")                      # ONE line, the last (and only) thing on stdout
    assert len(lines[0].encode()) < 4096                                     # (round 5's 23 KB line came back from the driver unparsed)
    line = json.loads(lines[0], parse_constant=_reject_constant)
    assert line["n_gpus"] == world and line["steps"] == 10 and line["warmup"] == 5 and line["scaling"] == "weak"
    assert "cpu_baseline" not in line and line["higher_is_better"] is True and line["unit"] == "frames/s"
    assert line["config"]["windows"] == 3 and line["config"]["schedule"] == "overlap" and "workload" in line["config"]
    assert all(not isinstance(v, (dict, list)) for v in line["config"].values())   # flat scalars only
    # the slowest rank sets the time: ~2 ms x world per frame (rank 0 alone needs ~2 ms) -> world x 10 frames / that for the job
    assert 1.95 * world < line["ms_per_step"] < 4.0 * world, line["ms_per_step"]
    assert abs(line["value"] - world * 10 / (line["ms_per_step"] * 1e-3 * 10)) < 1e-4 * line["value"]
    # the full record: ONE side file per job, written by rank 0 only
    assert [os.path.exists(tmp_path / ("side%d" % r) / "bench_full.json") for r in range(world)] == [True] + [False] * (world - 1)
    out = json.loads(open(tmp_path / "side0" / "bench_full.json").read(), parse_constant=_reject_constant)
    assert out["n_gpus"] == world and abs(out["value"] - line["value"]) < 1e-4 * out["value"]
    assert len(out["config"]["windows_ms_per_step"]) == 3
    assert sorted(out["config"]["schedules"]) == ["overlap", "sequential"]
    assert out["config"]["stats"]["frames"] == 10                           # stats of ONE window
    # every rank ran the same frame ranges: prologue + warm-up to frame 30, then three 10-frame windows, per schedule
    for r in res:
        assert r["overlap"] == [False, True]
        assert r["log"] == [[[0, 30], [30, 40], [40, 50], [50, 60]]] * 2


def test_timed_window_is_whole_keyframe_periods():
    import bench
    for w in (0, 5, 10, 20, 25, 31, 40):
        first, prologue = bench.timed_window(w)
        assert first % bench.PERIOD == 0 and first >= 30 and first - prologue == w and prologue >= 0


def test_pin_to_gpu_numa_with_only_a_local_rank():
    """pin_to_gpu_numa(local_rank) with the world left at its default (a caller that only knows LOCAL_RANK): planned as the last
    of local_rank + 1 ranks, no IndexError, a non-empty core set inside what the process may use (round-4 advisor finding)."""
    from gps_slam_amd.dist_util import pin_to_gpu_numa
    before = os.sched_getaffinity(0)
    try:
        for lr in (0, 3):
            os.sched_setaffinity(0, before)
            desc = pin_to_gpu_numa(lr)
            assert desc.startswith("affinity:") and "cpus" in desc, desc
            now = os.sched_getaffinity(0)
            assert now and now <= before
        os.sched_setaffinity(0, before)
        desc = pin_to_gpu_numa(1, 2, device_of_rank=lambda r: 0)   # two ranks sharing GPU 0 (the rehearsal's map)
        assert "cpus" in desc
    finally:
        os.sched_setaffinity(0, before)
        torch.set_num_threads(max(1, min(8, len(before))))


def test_cap_host_threads_respects_the_container_quota(monkeypatch):
    """cap_host_threads: the intra-op pool never exceeds the limit, the affinity mask or HALF the cgroup's CPU quota (the box shows
    256 CPUs to a container that may use 16; a pool sized by the former froze the whole process for the rest of a 100 ms
    accounting period after every parallel CPU op: LABBOOK section 14)."""
    import builtins
    from gps_slam_amd import dist_util
    before = torch.get_num_threads()
    real_open = builtins.open

    def fake(quota_line):
        def _open(path, *a, **k):
            if str(path) == "/sys/fs/cgroup/cpu.max":
                import io
                return io.StringIO(quota_line)
            return real_open(path, *a, **k)
        return _open
    try:
        monkeypatch.setattr(builtins, "open", fake("1600000 100000\n"))
        assert dist_util.cpu_quota() == 16.0
        torch.set_num_threads(before)
        assert dist_util.cap_host_threads(limit=64, cpus=256) == min(8, before) and torch.get_num_threads() == min(8, before)
        monkeypatch.setattr(builtins, "open", fake("300000 100000\n"))
        assert dist_util.cpu_quota() == 3.0 and dist_util.cap_host_threads(limit=8, cpus=256) == 1
        monkeypatch.setattr(builtins, "open", fake("max 100000\n"))
        assert dist_util.cpu_quota() is None
        torch.set_num_threads(before)
        assert dist_util.cap_host_threads(limit=4, cpus=256) == min(4, before)
    finally:
        monkeypatch.undo()
        torch.set_num_threads(before)
