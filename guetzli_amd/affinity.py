"""Host-core binding of one-process-per-GPU jobs (the reference's form of a batch is one
process per image under `xargs -P`, tests/golden_test.sh:24-26 -- it leaves placement to the scheduler;
a rank that feeds a GPU should sit on that GPU's NUMA node).

plan(): rank r -> the CPUs of GPU r's NUMA node (sysfs `local_cpulist` / `numa_node` of the GPU's PCI
function), divided evenly among the ranks whose GPUs share the node, whole physical cores (SMT siblings
stay together).  Where sysfs says nothing (no such device, numa_node -1, a container's empty tree) the
allowed CPUs are cut into contiguous per-rank shares of whole cores, as rounds 1-5 did with logical CPUs.

Pure functions over a sysfs ROOT, so that the CPU suite can run them on a faked tree of a 2-socket /
8-GPU / 256-CPU host (tests/test_affinity.py)."""
import os


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return sorted(set(out))


def format_cpulist(cpus):
    cpus = sorted(set(cpus))
    runs, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        runs.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(runs)


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def normalise_bus_id(bus_id):
    """'0000:05:00.0' from what the runtimes return ('0000:05:00.0', '05:00.0', upper case, ...)."""
    b = bus_id.strip().lower()
    if b.count(":") == 1:
        b = "0000:" + b
    return b


def gpu_locality(bus_id, sysfs="/sys"):
    """(numa_node, [cpus]) of a PCI function, or (None, None) when sysfs does not say."""
    if not bus_id:
        return None, None
    base = os.path.join(sysfs, "bus", "pci", "devices", normalise_bus_id(bus_id))
    node = _read(os.path.join(base, "numa_node"))
    cpus = _read(os.path.join(base, "local_cpulist"))
    try:
        node = int(node) if node is not None else None
    except ValueError:
        node = None
    cpus = parse_cpulist(cpus) if cpus else None
    if node is not None and node < 0:
        # a single-node host reports -1 and lists every CPU: nothing to choose between
        return None, cpus
    if node is not None and not cpus:
        cpus = _read(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist"))
        cpus = parse_cpulist(cpus) if cpus else None
    return node, cpus


def physical_cores(cpus, sysfs="/sys"):
    """The CPUs grouped into physical cores ([[cpu, sibling, ...], ...], ordered by first CPU): SMT
    siblings from topology/thread_siblings_list; a CPU whose topology is unreadable is its own core."""
    cpus = sorted(set(cpus))
    allowed = set(cpus)
    seen, cores = set(), []
    for c in cpus:
        if c in seen:
            continue
        sib = _read(os.path.join(sysfs, "devices", "system", "cpu", f"cpu{c}", "topology", "thread_siblings_list"))
        group = [x for x in (parse_cpulist(sib) if sib else [c]) if x in allowed and x not in seen] or [c]
        if c not in group:
            group.append(c)
        group = sorted(set(group))
        seen.update(group)
        cores.append(group)
    return cores


def _share(cores, index, count):
    """The index-th of `count` contiguous shares of whole cores (the first `len % count` shares take one more)."""
    n = len(cores)
    base, extra = divmod(n, count)
    start = index * base + min(index, extra)
    return cores[start:start + base + (1 if index < extra else 0)]


def plan(local_rank, world, bus_ids, allowed, sysfs="/sys"):
    """The CPU set of rank `local_rank` of `world` ranks on one host.
    bus_ids: PCI bus id of every local rank's GPU, in rank order (None / missing entries allowed);
    allowed: the CPUs this process may use now (os.sched_getaffinity(0)).
    Returns {"cpus": [...], "numa_node": int | None, "how": str, "gpu_pci_bus_id": str | None,
             "ranks_on_node": [...]}; "cpus" is empty when there is nothing to partition (fewer cores than
    ranks: the caller leaves the scheduler alone)."""
    allowed = sorted(set(allowed))
    bus_ids = list(bus_ids or []) + [None] * world
    mine = bus_ids[local_rank]
    out = {"cpus": [], "numa_node": None, "how": "", "gpu_pci_bus_id": normalise_bus_id(mine) if mine else None,
           "ranks_on_node": list(range(world))}
    loc = [gpu_locality(bus_ids[r], sysfs) for r in range(world)]
    node, node_cpus = loc[local_rank]
    if node is not None and node_cpus:
        local = [c for c in node_cpus if c in set(allowed)]
        sharing = [r for r in range(world) if loc[r][0] == node]
        cores = physical_cores(local, sysfs)
        if len(cores) >= len(sharing) >= 1:
            part = _share(cores, sharing.index(local_rank), len(sharing))
            out.update(cpus=sorted(c for g in part for c in g), numa_node=node, ranks_on_node=sharing,
                       how=f"NUMA node {node} of GPU {out['gpu_pci_bus_id']}: share {sharing.index(local_rank) + 1} of "
                           f"{len(sharing)} of its {len(cores)} physical cores (SMT siblings kept together)")
            return out
    # nothing known about the GPU's node: contiguous shares of the allowed CPUs, whole cores
    cores = physical_cores(allowed, sysfs)
    if len(cores) >= world:
        part = _share(cores, local_rank, world)
        out.update(cpus=sorted(c for g in part for c in g),
                   how=f"no NUMA information for the GPU: contiguous share {local_rank + 1} of {world} of the "
                       f"{len(cores)} allowed physical cores")
    else:
        out["how"] = f"{len(cores)} cores for {world} ranks: not binding (the ranks share the cores)"
    return out


def device_bus_ids(world, library=None):
    """PCI bus ids of HIP devices 0 .. world-1: through the C ABI (gz_device_pci_bus_id = hipDeviceGetPCIBusId) when
    the loaded library is given, else through PyTorch's device properties (no context is created on the other
    ranks' devices either way); None where neither says."""
    if library is not None:
        ids = []
        for i in range(world):
            try:
                ids.append(library.device_pci_bus_id(i))
            except Exception:
                ids.append(None)
        if any(ids):
            return ids
    try:
        import torch
        ids = []
        for i in range(world):
            try:
                p = torch.cuda.get_device_properties(i)
                ids.append("%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))
            except Exception:
                ids.append(None)
        return ids
    except Exception:
        return [None] * world
