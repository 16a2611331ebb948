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


def pin_to_gpu_numa(local_rank, world=1):
    """Pin this process (and the threads it starts later: the tracker's frame thread, the mapping worker) to the cores of its
    GPU's NUMA node, divided evenly between the ranks that share the node; if the node is unknown, to an equal slice of the
    cores this process may use.  Returns a short description for the bench line.  One scene per GPU means the only shared
    host resources are cores and PCIe root complexes -- keeping each rank's threads and pinned buffers next to its GPU is
    all the placement the path needs."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return "affinity: unsupported"
    node = gpu_numa_node(local_rank)
    cpus = None
    if node is not None and node >= 0:
        try:
            cpus = sorted(_parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read()) & set(allowed))
        except OSError:
            cpus = None
    if cpus:
        # ranks whose GPUs sit on the same node split it; without topology knowledge of the other ranks assume GPUs are
        # spread evenly over the nodes (8 GPUs / 2 sockets -> 4 ranks per node)
        n_nodes = max(1, len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node")]))
        per_node = max(1, (world + n_nodes - 1) // n_nodes)
        k = local_rank % per_node
        share = max(2, len(cpus) // per_node)
        mine = cpus[k * share:(k + 1) * share] or cpus
        where = "numa node %d" % node
    else:
        share = max(2, len(allowed) // max(1, world))
        mine = allowed[local_rank * share:(local_rank + 1) * share] or allowed
        where = "no numa info"
    try:
        os.sched_setaffinity(0, mine)
    except OSError as e:
        return "affinity: %s" % e
    # libtorch's intra-op pool defaults to one thread per LOGICAL cpu of the machine: more threads than this rank may run on
    # turns the first parallel CPU op (randperm / sort of the new-Gaussian subset) into a 50 ms barrier storm.  The host side
    # of this path is bookkeeping: a handful of threads is plenty.
    import torch
    torch.set_num_threads(max(1, min(8, len(mine))))
    return "affinity: %s, %d cores (%d..%d), %d intra-op threads" % (where, len(mine), mine[0], mine[-1], torch.get_num_threads())
