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
        # sched_setaffinity(0, ...) moves the calling thread only: threads that already exist (HIP runtime, OpenMP, torch's
        # pools) keep their mask, so every task of the process is moved; threads created later inherit from their creator
        try:
            tids = [int(t) for t in os.listdir("/proc/self/task")]
        except OSError:
            tids = []
        for tid in tids:
            try:
                os.sched_setaffinity(tid, allowed)
            except OSError:          # a thread that exited meanwhile
                pass
        os.sched_setaffinity(0, allowed)
        return allowed
    except (OSError, ValueError):
        return None


def pin_to_gpu(device_index):
    """One worker per GPU, on the CPUs next to it: the host side of a sharded render is bound by PCIe and host DRAM, so the
    pinned ring and the copy threads should live on the GPU's own NUMA node.  Every thread the process has at the time of
    the call is moved (call it before the pinned ring and the copy pool are created).  No-op where the platform does not say."""
    return pin_to_numa_node(gpu_numa_node(device_index))


def free_port():
    """A TCP port that is free on 127.0.0.1 right now (for the rendezvous of ranks this process is about to start)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(argv, world, extra_env=None, poll_s=0.2):
    """Start `world` fresh processes `argv` (one rank each: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment),
    wait for all of them and return the largest exit code.  When one rank fails the others are stopped (by their own PIDs), so
    nobody is left waiting in a barrier.  The caller must not have touched the GPU: the children own the devices."""
    import subprocess
    import time
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), **(extra_env or {}))
        procs.append(subprocess.Popen(argv, env=env))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            bad = [p.returncode for p in procs if p.poll() not in (None, 0)]
            if bad:
                rc = bad[0]
                break
            time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                if rc:
                    p.terminate()
                try:
                    p.wait(timeout=30 if rc else None)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
    codes = [p.returncode for p in procs]
    return max([abs(c) for c in codes] + [abs(rc)])
