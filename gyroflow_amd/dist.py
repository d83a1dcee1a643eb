"""Multi-GPU glue.  Frames are independent, so ranks exchange no data: the process group only
carries the start/stop barrier and the max-over-ranks wall time of a run (gloo; one process per GPU)."""
import os

import torch
import torch.distributed as dist


def init_from_env():
    """-> (rank, local_rank, world).  Initialises a gloo group when launched by torch.distributed.run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, local_rank, world


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def sum_over_ranks(value):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0])


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


# ----------------------------------------------------------------------------- host placement
def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_node(device_index, sysfs="/sys"):
    """NUMA node of a GPU from its PCI address (hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<addr>/numa_node), or None."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) != 0:
            return None
        addr = buf.value.decode().lower()
        with open(os.path.join(sysfs, "bus", "pci", "devices", addr, "numa_node")) as f:
            node = int(f.read().strip())
        return node if node >= 0 else None
    except (OSError, ValueError, AttributeError):
        return None


def pin_to_numa_node(node, sysfs="/sys"):
    """Restrict this process to the CPUs of a NUMA node (intersected with its current affinity mask, so a container's CPU
    share is respected).  Returns the CPU set in force, or None when nothing was changed (no such node, empty
    intersection, platform without sched_setaffinity)."""
    if node is None or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        with open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")) as f:
            cpus = _parse_cpulist(f.read())
        allowed = cpus & os.sched_getaffinity(0)
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return allowed
    except (OSError, ValueError):
        return None


def pin_to_gpu(device_index):
    """One worker per GPU, on the CPUs next to it: the host side of a sharded render is bound by PCIe and host DRAM, so the
    pinned ring and the copy threads should live on the GPU's own NUMA node.  No-op where the platform does not say."""
    return pin_to_numa_node(gpu_numa_node(device_index))
