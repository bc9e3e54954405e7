"""Host-side placement of a rank's launch thread (one process per GPU).

A training step is ~500 kernel launches issued from ONE Python thread per rank (6 - 8 ms of host time per step at N = 1, bench.py
`host`).  With 8 ranks on one node those threads must not migrate across sockets or share cores: each rank is pinned to the CPUs of
the NUMA node its GPU's PCIe function hangs off (sysfs), and inside that node to its own slice when several ranks share it.
Best effort: any failure leaves the affinity untouched and is reported in the returned dict (bench.py prints it in the side file).

The affinity is the PROCESS's: it also covers the RCCL proxy thread, autograd's device thread (which runs the step's backward hooks)
and any later worker.  A slice of fewer than MIN_CPUS_PER_RANK CPUs would put all of them on one or two cores — worse than not pinning
— so below that the rank is confined to its whole NUMA node (no slicing), and below that on the node itself nothing is changed.
WGS_NO_PIN=1 disables pinning altogether."""
import os

MIN_CPUS_PER_RANK = 4        # launch thread + autograd device thread + RCCL proxy + one spare


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


def choose(node_cpus, allowed, ranks_on_node, slot, min_cpus=MIN_CPUS_PER_RANK):
    """(cpus or None, why): the rank's own slice when it holds >= min_cpus CPUs; else its whole NUMA node (shared with the node's other
    ranks) when THAT holds >= min_cpus; else None = leave the affinity alone."""
    own = plan(node_cpus, allowed, ranks_on_node, slot)
    if len(own) >= min_cpus:
        return own, 'own slice of %d CPUs' % len(own)
    node = set(node_cpus) & set(allowed)
    if len(node) >= min_cpus and node != set(allowed):
        return node, 'slice of %d CPU(s) < %d: confined to the NUMA node (%d CPUs), shared by its %d rank(s)' % (len(own), min_cpus, len(node), ranks_on_node)
    return None, 'cpuset too small to slice (%d CPUs for %d rank(s) on the node, < %d each): affinity left alone' % (len(node) or len(allowed), ranks_on_node, min_cpus)


def _nodes_of(local_world):
    """NUMA node and PCI address of every local rank's GPU, queried once."""
    import torch
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    return [gpu_numa_node(r) if r < n else (None, None) for r in range(local_world)]


def _plan_rank(local_rank, local_world, allowed, nodes, device_index=None):
    """What pin_rank() would apply for this rank: {'cpus': sorted list or None, 'numa_node', 'why'} — no side effects."""
    out = {'cpus': None, 'numa_node': None, 'why': ''}
    if os.environ.get('WGS_NO_PIN') == '1':
        out['why'] = 'WGS_NO_PIN=1'
        return out
    node, bdf = nodes[local_rank] if device_index is None else gpu_numa_node(device_index)
    node_cpus, sharers, slot = allowed, local_world, local_rank
    if node is not None and node >= 0 and os.path.exists('/sys/devices/system/node/node%d/cpulist' % node):
        node_cpus = _cpulist(open('/sys/devices/system/node/node%d/cpulist' % node).read())
        same = [r for r in range(local_world) if nodes[r][0] == node]      # the local ranks whose GPUs report the same node
        if local_rank in same:
            sharers, slot = len(same), same.index(local_rank)
        out['numa_node'] = node
    elif local_world <= 1:
        out['why'] = 'GPU %s reports no NUMA node and there is one rank: nothing to separate' % bdf
        return out
    cpus, why = choose(node_cpus, allowed, sharers, slot)
    out.update(cpus=sorted(cpus) if cpus else None, why='GPU %s, %d rank(s) on its node: %s' % (bdf, sharers, why))
    return out


def plan_all(local_world):
    """The placement of every local rank (bench.py's side file: the plan can be read even where it is not applied)."""
    try:
        allowed = os.sched_getaffinity(0) if hasattr(os, 'sched_getaffinity') else set(range(os.cpu_count() or 1))
        nodes = _nodes_of(local_world)
        return [_plan_rank(r, local_world, allowed, nodes) for r in range(local_world)]
    except Exception as e:  # noqa: BLE001
        return [{'cpus': None, 'numa_node': None, 'why': 'no plan: %r' % (e,)} for _ in range(local_world)]


def pin_rank(local_rank, local_world, device_index=None):
    """Pin the calling process (its launch thread and the threads it spawns later) for rank `local_rank` of `local_world` on this
    node.  Returns {'pinned': bool, 'cpus': [...], 'numa_node': n, 'why': text}."""
    out = {'pinned': False, 'cpus': None, 'numa_node': None, 'why': ''}
    try:
        if not hasattr(os, 'sched_setaffinity'):
            out['why'] = 'no sched_setaffinity on this platform'
            return out
        allowed = os.sched_getaffinity(0)
        pl = _plan_rank(local_rank, local_world, allowed, _nodes_of(local_world), device_index)
        out.update(numa_node=pl['numa_node'], why=pl['why'])
        if pl['cpus']:
            os.sched_setaffinity(0, set(pl['cpus']))
            out.update(pinned=True, cpus=pl['cpus'])
    except Exception as e:  # noqa: BLE001
        out['why'] = 'not pinned: %r' % (e,)
    return out
