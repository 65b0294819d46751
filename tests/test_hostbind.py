"""CPU placement plan of the data-parallel ranks (videovector_amd/hostbind.py): pure host logic."""
from videovector_amd import hostbind


def two_socket(cores_per_socket=64, smt=True):
    cores, node_of = {}, {}
    n = 2 * cores_per_socket
    for pkg in range(2):
        for core in range(cores_per_socket):
            cpu = pkg * cores_per_socket + core
            cpus = [cpu] + ([cpu + n] if smt else [])
            cores[(pkg, core)] = cpus
            for c in cpus:
                node_of[c] = pkg
    return cores, node_of


def test_eight_ranks_get_disjoint_blocks_on_their_gpus_socket():
    cores, node_of = two_socket()
    gpu_nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    seen = set()
    for r in range(8):
        cpus = hostbind.plan(cores, node_of, gpu_nodes, r)
        assert len(cpus) == 32 and not (seen & set(cpus))          # 16 cores x 2 hardware threads
        seen |= set(cpus)
        assert {node_of[c] for c in cpus} == {gpu_nodes[r]}
        prim = [c for c in cpus if c < 128]
        assert prim == list(range(prim[0], prim[0] + 16)) and prim[0] % 16 == 0      # whole 8-core groups
        assert sorted(c - 128 for c in cpus if c >= 128) == prim    # each core with its sibling
    assert len(seen) == 256


def test_unknown_numa_splits_the_machine_evenly_and_small_shares_are_refused():
    cores, node_of = two_socket(smt=False)
    a = hostbind.plan(cores, node_of, [-1, -1], 0)
    b = hostbind.plan(cores, node_of, [-1, -1], 1)
    assert a == list(range(0, 64)) and b == list(range(64, 128))
    small = {k: v for k, v in cores.items() if k[1] < 3 and k[0] == 0}
    assert hostbind.plan(small, node_of, [0, 0], 0) is None
    assert hostbind.plan(cores, node_of, [0, 0], 5) is None
    # all GPUs on one socket: that socket's cores are shared, the other socket stays free
    c = hostbind.plan(cores, node_of, [1, 1, 1, 1], 2)
    assert c == list(range(64 + 32, 64 + 48))


def test_reads_this_machines_topology():
    cores, node_of = hostbind.read_topology()
    if cores is None:
        return
    assert sum(len(v) for v in cores.values()) >= 1
    assert hostbind.gpu_numa_node(0, 0xff, 0x1f) == -1             # no such device: unknown
    assert isinstance(hostbind.bind(0, [(0, 0xff, 0x1f)], sys_root="/nonexistent"), str)
