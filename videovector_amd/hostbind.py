"""CPU placement of the data-parallel ranks of one node.

One process per GPU, each with its own sampler (a walk thread + stage threads), the consumer thread and the collective
library's helper threads.  Left to the scheduler, two ranks can end up on the same cores (the sampler pins its stage
threads to the 8-CPU group its creator happens to run on, sampler.cc stage_cpu_set).  plan() gives every rank of the
node its own contiguous block of physical cores -- on the NUMA node its GPU hangs off when sysfs says which -- and
bind() applies it to the calling process (threads created afterwards inherit it).  Everything degrades to "no binding"
when the topology cannot be read or the share is too small.
"""
import glob
import os


def _read_int(path, default=None):
    try:
        return int(open(path).read().strip())
    except (OSError, ValueError):
        return default


def _parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def read_topology(allowed=None, sys_root="/sys"):
    """-> (cores, node_of_cpu): cores = {(package, core_id): [logical cpus]} over the allowed CPUs."""
    if allowed is None:
        allowed = sorted(os.sched_getaffinity(0))
    cores = {}
    for c in allowed:
        base = "%s/devices/system/cpu/cpu%d/topology/" % (sys_root, c)
        pkg, core = _read_int(base + "physical_package_id"), _read_int(base + "core_id")
        if pkg is None or core is None:
            return None, None
        cores.setdefault((pkg, core), []).append(c)
    node_of = {}
    for d in glob.glob("%s/devices/system/node/node[0-9]*" % sys_root):
        try:
            n = int(os.path.basename(d)[4:])
            for c in _parse_cpulist(open(d + "/cpulist").read()):
                node_of[c] = n
        except (OSError, ValueError):
            pass
    return cores, node_of


def gpu_numa_node(pci_domain, pci_bus, pci_device, sys_root="/sys"):
    v = _read_int("%s/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (sys_root, pci_domain, pci_bus, pci_device), -1)
    return v if v is not None else -1


def plan(cores, node_of, gpu_nodes, local_rank, min_cores=4):
    """cores / node_of from read_topology; gpu_nodes[r] = NUMA node of local rank r's GPU (-1 unknown).
    -> sorted logical CPUs for local_rank, or None (no binding)."""
    if not cores or local_rank < 0 or local_rank >= len(gpu_nodes):
        return None
    keys = sorted(cores)                                   # (package, core_id): neighbours share a CCD / L3 on EPYC hosts
    nodes_known = all(n >= 0 for n in gpu_nodes) and node_of and all(cores[k][0] in node_of for k in keys)
    if nodes_known:
        mine = gpu_nodes[local_rank]
        pool = [k for k in keys if node_of[cores[k][0]] == mine]
        peers = [r for r, n in enumerate(gpu_nodes) if n == mine]
        if not pool:                                       # the GPU's node has no allowed CPU: fall back to an even split
            nodes_known = False
    if not nodes_known:
        pool, peers = keys, list(range(len(gpu_nodes)))
    share = len(pool) // len(peers)
    if share < min_cores:
        return None
    i = peers.index(local_rank)
    mine_cores = pool[i * share:(i + 1) * share]
    return sorted(c for k in mine_cores for c in cores[k])


def bind(local_rank, gpu_pci, sys_root="/sys"):
    """gpu_pci: [(domain, bus, device)] of the GPUs of local ranks 0..n-1.  Applies plan() to this process.
    -> a description string for the log / the benchmark line."""
    try:
        cores, node_of = read_topology(sys_root=sys_root)
        if cores is None:
            return "not bound (CPU topology not readable)"
        gpu_nodes = [gpu_numa_node(*p, sys_root=sys_root) for p in gpu_pci]
        cpus = plan(cores, node_of, gpu_nodes, local_rank)
        if not cpus:
            return "not bound (fewer than 4 cores per rank)"
        os.sched_setaffinity(0, cpus)
        return "rank-local block of %d logical CPUs (%d..%d), GPU NUMA node %s" % (
            len(cpus), cpus[0], cpus[-1], gpu_nodes[local_rank] if gpu_nodes[local_rank] >= 0 else "unknown")
    except Exception as e:                                 # placement is an optimisation: never fail the run over it
        return "not bound (%s)" % e
