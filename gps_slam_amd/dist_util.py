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
