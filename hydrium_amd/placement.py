"""Where the host side of the encoder should run.  On a two-socket box the drop-in API's uploads (pinned staging ->
GPU) run at 55 GB/s from the GPU's own NUMA node and at 45 GB/s from the other socket: an 8192x8192 RGB16 frame through
hyd_send_tile takes 11.6 ms against 13.8 (measured with taskset on both sockets of the 2 x EPYC 9575F host, round 3).
A deployment binds each encoder process to its GPU's node (numactl --cpunodebind / taskset); bench.py does the same
through this helper so that its host-path legs do not depend on where the scheduler happened to start it."""
import os


def gpu_numa_node(device_index: int):
    """NUMA node of the GPU's PCI function from sysfs, or None."""
    import torch

    p = torch.cuda.get_device_properties(device_index)
    bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{getattr(p, 'pci_device_id', 0):02x}.0"
    try:
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
    except (OSError, ValueError):
        return None
    return node if node >= 0 else None


def node_cpus(node: int):
    cpus = set()
    try:
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
    except (OSError, ValueError):
        return set()
    return cpus


def bind_near_gpu(device_index: int):
    """Restrict this process (and the threads it starts from here on) to the CPUs of the GPU's NUMA node.  Returns a
    short description for logs, or None if nothing was changed (single-node box, no sysfs, cpuset too narrow)."""
    node = gpu_numa_node(device_index)
    if node is None:
        return None
    allowed = os.sched_getaffinity(0)
    want = node_cpus(node) & allowed
    if not want or want == allowed:
        return None
    os.sched_setaffinity(0, want)
    return f"NUMA node {node} of GPU {device_index} ({len(want)} CPUs)"
