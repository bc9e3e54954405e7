"""Host-side placement of a rank's launch thread (one process per GPU).

A training step is ~500 kernel launches issued from ONE Python thread per rank (6 - 8 ms of host time per step at N = 1, bench.py
`host`).  With 8 ranks on one node those threads must not migrate across sockets or share cores: each rank is pinned to the CPUs of
the NUMA node its GPU's PCIe function hangs off (sysfs), and inside that node to its own slice when several ranks share it.
Best effort: any failure leaves the affinity untouched and is reported in the returned dict (bench.py prints it in the side file)."""
import os


def _cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_node(device_index):
    """NUMA node of GPU `device_index` from /sys/bus/pci/devices/<domain:bus:device.0>/numa_node (-1 / None: unknown)."""
    import torch
    p = torch.cuda.get_device_properties(device_index)
    bdf = '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    path = '/sys/bus/pci/devices/%s/numa_node' % bdf
    if not os.path.exists(path):
        return None, bdf
    return int(open(path).read().strip()), bdf


def plan(node_cpus, allowed, ranks_on_node, slot):
    """CPUs for rank `slot` of `ranks_on_node` ranks that share a NUMA node: an equal contiguous slice of the node's allowed CPUs
    (at least one CPU; slices wrap when there are more ranks than CPUs)."""
    cpus = sorted(set(node_cpus) & set(allowed)) or sorted(allowed)
    n = max(1, len(cpus) // max(1, ranks_on_node))
    start = (slot * n) % len(cpus)
    return set((cpus + cpus)[start:start + n])


def pin_rank(local_rank, local_world, device_index=None):
    """Pin the calling process (its launch thread and the threads it spawns later) for rank `local_rank` of `local_world` on this
    node.  Returns {'pinned': bool, 'cpus': [...], 'numa_node': n, 'why': text}."""
    out = {'pinned': False, 'cpus': None, 'numa_node': None, 'why': ''}
    try:
        if not hasattr(os, 'sched_setaffinity'):
            out['why'] = 'no sched_setaffinity on this platform'
            return out
        allowed = os.sched_getaffinity(0)
        dev = local_rank if device_index is None else device_index
        node, bdf = gpu_numa_node(dev)
        node_cpus, sharers, slot = allowed, local_world, local_rank
        if node is not None and node >= 0 and os.path.exists('/sys/devices/system/node/node%d/cpulist' % node):
            node_cpus = _cpulist(open('/sys/devices/system/node/node%d/cpulist' % node).read())
            # the ranks that share this node = the local ranks whose GPUs report the same node
            import torch
            same = [r for r in range(local_world) if r < torch.cuda.device_count() and gpu_numa_node(r)[0] == node]
            if local_rank in same:
                sharers, slot = len(same), same.index(local_rank)
            out['numa_node'] = node
        elif local_world <= 1:
            out['why'] = 'GPU %s reports no NUMA node and there is one rank: nothing to separate' % bdf
            return out
        cpus = plan(node_cpus, allowed, sharers, slot)
        os.sched_setaffinity(0, cpus)
        out.update(pinned=True, cpus=sorted(cpus), why='GPU %s, %d rank(s) on its node' % (bdf, sharers))
    except Exception as e:  # noqa: BLE001
        out['why'] = 'not pinned: %r' % (e,)
    return out
