"""Multi-GPU = independent scenes, one process per GPU (the SLAM loop is sequential per scene, so the
path does not shard).  The only communication is the bench contract's barrier and max-over-ranks of the elapsed
time; nccl (= RCCL) on GPUs, gloo in the CPU tests."""
import os

import torch


def env_ranks():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def scene_seed(rank, base=1234):
    """every rank fuses / optimises a different synthetic scene"""
    return base + 7919 * rank


class Group:
    def __init__(self, backend=None, device=None):
        self.rank, self.local_rank, self.world = env_ranks()
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch.distributed as dist
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = torch.device(device)
            dist.init_process_group(backend or "gloo", **kw)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if self.dist is None:
            return float(x)
        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device if self.device else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def aggregate_rate(self, units_per_rank, seconds_this_rank):
        """whole-job throughput: units of ALL ranks / slowest rank's time"""
        return self.world * units_per_rank / self.max_over_ranks(seconds_this_rank)

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


# ----------------------------------------------------------------------------- host placement
def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_node(local_rank):
    """NUMA node of the GPU this rank drives: /sys/class/drm/card*/device/{numa_node} of the PCI device torch reports for
    the index (falls back to the index-th render device).  -1 / None when the platform does not say."""
    try:
        import torch
        bus = None
        if torch.cuda.is_available():
            p = torch.cuda.get_device_properties(local_rank)
            if hasattr(p, "pci_bus_id"):
                bus = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
        if bus and os.path.exists("/sys/bus/pci/devices/%s/numa_node" % bus):
            return int(open("/sys/bus/pci/devices/%s/numa_node" % bus).read())
    except Exception:
        pass
    try:
        cards = sorted(d for d in os.listdir("/sys/class/drm") if d.startswith("renderD"))
        path = "/sys/class/drm/%s/device/numa_node" % cards[local_rank]
        return int(open(path).read())
    except Exception:
        return None


def plan_affinity(world, gpu_nodes, node_cpus, allowed, core_of=None):
    """Pure placement rule -> the cpu list of every local rank (tests/test_multi_rank_cpu.py drives it with fixtures).
    gpu_nodes[r]: NUMA node of rank r's GPU (None or < 0: unknown); node_cpus: {node: cpus of that node}; allowed: cpus this
    process may run on.  Ranks whose GPUs sit on the same node split that node's allowed cores evenly, in rank order -- whatever
    the GPU-to-node mapping is (contiguous, interleaved, uneven); if any rank's node is unknown or has no allowed core, every
    rank takes an equal slice of `allowed` instead.  Slices are non-empty, and disjoint whenever there is at least one core per
    rank to hand out.  core_of: {cpu: physical core key}; the cpus are dealt out core by core (SMT siblings stay together:
    Linux numbers them far apart, "0-63,128-191" being the 64 cores of one socket twice), so two ranks never share a physical core
    when there are enough of them."""
    key = (lambda c: core_of.get(c, c)) if core_of else (lambda c: c)

    def deal(cpus, n):
        """n slices of cpus: whole physical cores each while there are at least n cores, single cpus otherwise"""
        cores = {}
        for c in sorted(cpus, key=lambda c: (key(c), c)):
            cores.setdefault(key(c), []).append(c)
        groups = list(cores.values())
        if len(groups) < n:
            groups = [[c] for g in groups for c in g]
        share = max(1, len(groups) // n)
        out = []
        for k in range(n):
            part = groups[k * share:(k + 1) * share] or groups[-share:]
            out.append(sorted(c for g in part for c in g))
        return out

    plans, usable = [None] * world, True
    by_node = {}
    for r in range(world):
        n = gpu_nodes[r] if r < len(gpu_nodes) else None
        if n is None or n < 0 or not (set(node_cpus.get(n, ())) & set(allowed)):
            usable = False
            break
        by_node.setdefault(n, []).append(r)
    if usable:
        for n, ranks in by_node.items():
            for r, part in zip(ranks, deal(set(node_cpus[n]) & set(allowed), len(ranks))):
                plans[r] = part
        return plans, "numa"
    return deal(allowed, max(1, world)), "even"


def _core_of():
    """{cpu: (package, core)} from /sys/devices/system/cpu/cpu*/topology (empty when the platform does not say)"""
    out = {}
    base = "/sys/devices/system/cpu"
    try:
        for d in os.listdir(base):
            if d.startswith("cpu") and d[3:].isdigit():
                t = os.path.join(base, d, "topology")
                out[int(d[3:])] = (int(open(os.path.join(t, "physical_package_id")).read()), int(open(os.path.join(t, "core_id")).read()))
    except (OSError, ValueError):
        return {}
    return out


def _node_cpus():
    out = {}
    try:
        for d in os.listdir("/sys/devices/system/node"):
            if d.startswith("node") and d[4:].isdigit():
                out[int(d[4:])] = sorted(_parse_cpulist(open("/sys/devices/system/node/%s/cpulist" % d).read()))
    except OSError:
        pass
    return out


def cpu_quota():
    """CPUs this process's container may use per accounting period (cgroup v2 cpu.max / v1 cpu.cfs_quota_us), or None without a
    quota.  os.cpu_count() and sched_getaffinity() know nothing of it: the MI355X boxes show 256 logical CPUs to a container that
    may use 16."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cap_host_threads(limit=8, cpus=None):
    """Bound libtorch's intra-op pool by what this process may actually run on: `limit`, the CPUs of its affinity mask (or `cpus`)
    and HALF the container's CPU quota (the frame thread, the map worker and the runtime's own threads need the other half).
    libtorch sizes the pool by the machine's cores; after every parallel CPU op -- a fill of a megabyte is one -- each of those
    threads spins for a while, and a container over its quota is frozen WHOLE until the accounting period (100 ms) ends: measured
    on the MI355X box (256 CPUs shown, quota 16, pool of 128): a 90 ms hole in the first frames of 6 runs out of 10 that started
    right after a scene had been built with torch::zeros images (tools/probe/early_stall.py; LABBOOK section 14).  The host side
    of this path is bookkeeping: a handful of threads is plenty.  -> the number of threads set."""
    import torch
    try:
        n = len(os.sched_getaffinity(0)) if cpus is None else int(cpus)
    except AttributeError:
        n = os.cpu_count() or 1
    q = cpu_quota()
    if q is not None:
        n = min(n, max(1, int(q // 2)))
    n = max(1, min(int(limit), n, torch.get_num_threads()))
    torch.set_num_threads(n)
    return n


def pin_to_gpu_numa(local_rank, world=1, device_of_rank=None):
    """Pin this process (and the threads it starts later: the tracker's frame thread, the mapping worker) to its share of the
    cores of its GPU's NUMA node (plan_affinity: the ranks of a node split it evenly; without NUMA information every rank takes
    an equal slice of the cores this process may use).  device_of_rank(r) -> GPU index of local rank r (default: r; a rehearsal
    on fewer GPUs than ranks maps several ranks to one device).  Returns a short description for the bench line.  One scene per
    GPU means the only shared host resources are cores and PCIe root complexes -- keeping each rank's threads and pinned
    buffers next to its GPU is all the placement the path needs."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return "affinity: unsupported"
    dev = device_of_rank or (lambda r: r)
    # `world` is the number of ranks on THIS host (LOCAL_WORLD_SIZE); a caller that only knows its local rank (world left at 1) is
    # planned as the last of local_rank + 1 ranks.  One index, r, everywhere below.
    world = max(1, int(world), int(local_rank) + 1)
    r = int(local_rank) % world
    gpu_nodes = [gpu_numa_node(dev(k)) for k in range(world)]
    plans, how = plan_affinity(world, gpu_nodes, _node_cpus(), allowed, core_of=_core_of())
    mine = plans[r]
    where = "numa node %d" % gpu_nodes[r] if how == "numa" else "no numa info"
    try:
        os.sched_setaffinity(0, mine)
    except OSError as e:
        return "affinity: %s" % e
    # libtorch's intra-op pool defaults to one thread per LOGICAL cpu of the machine: more threads than this rank may run on
    # turns the first parallel CPU op into a 50 ms barrier storm -- or, in a container with a CPU quota, into a frozen process
    # (cap_host_threads)
    import torch
    cap_host_threads(8, cpus=len(mine))
    q = cpu_quota()
    # (every rank keeps ~2 threads busy -- the tracking thread polls its mailbox, the map worker enqueues and waits: a quota below
    # ~2.5 CPUs per rank means frozen accounting periods, and the line says so instead of leaving a slow number unexplained)
    quota = "" if q is None else ", container quota %.3g cpus%s" % (q, " (< 2.5 per rank: expect throttling)" if q < 2.5 * world else "")
    return "affinity: %s, %d cpus (%d..%d), %d intra-op threads%s" % (where, len(mine), min(mine), max(mine), torch.get_num_threads(), quota)
