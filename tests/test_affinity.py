"""guetzli_amd/affinity.py on a faked sysfs tree: rank -> CPUs of its GPU's NUMA node.

The tree is the bench box's kind of host: 2 sockets x 64 cores x 2 threads = 256 logical CPUs, SMT siblings
enumerated in the upper half (cpu i and cpu i + 128), one NUMA node per socket, GPUs 0-3 behind socket 0 and
GPUs 4-7 behind socket 1.  The contiguous split of rounds 1-5 put ranks 2-3 on socket 1 and ranks 4-5 on
socket 0's SMT siblings there."""
import os

import pytest

from guetzli_amd import affinity

BUS = ["0000:05:00.0", "0000:15:00.0", "0000:65:00.0", "0000:75:00.0",
       "0000:85:00.0", "0000:95:00.0", "0000:e5:00.0", "0000:f5:00.0"]


def fake_sysfs(root, numa=True, smt=True, gpu_nodes=(0, 0, 0, 0, 1, 1, 1, 1)):
    node_cpus = {0: "0-63,128-191", 1: "64-127,192-255"} if smt else {0: "0-63", 1: "64-127"}
    ncpu = 256 if smt else 128
    for c in range(ncpu):
        d = os.path.join(root, "devices", "system", "cpu", f"cpu{c}", "topology")
        os.makedirs(d)
        sib = f"{c % 128},{c % 128 + 128}" if smt else str(c)
        open(os.path.join(d, "thread_siblings_list"), "w").write(sib + "\n")
    for n, cpus in node_cpus.items():
        d = os.path.join(root, "devices", "system", "node", f"node{n}")
        os.makedirs(d)
        open(os.path.join(d, "cpulist"), "w").write(cpus + "\n")
    for b, n in zip(BUS, gpu_nodes):
        d = os.path.join(root, "bus", "pci", "devices", b)
        os.makedirs(d)
        open(os.path.join(d, "numa_node"), "w").write(f"{n if numa else -1}\n")
        open(os.path.join(d, "local_cpulist"), "w").write((node_cpus[n] if numa else f"0-{ncpu - 1}") + "\n")
    return ncpu


def test_cpulist_round_trip():
    assert affinity.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert affinity.format_cpulist([11, 10, 8, 3, 2, 1, 0]) == "0-3,8,10-11"
    assert affinity.parse_cpulist("") == []
    assert affinity.normalise_bus_id("05:00.0") == "0000:05:00.0"
    assert affinity.normalise_bus_id("0000:E5:00.0") == "0000:e5:00.0"


def test_two_sockets_eight_gpus(tmp_path):
    root = str(tmp_path)
    ncpu = fake_sysfs(root)
    allowed = list(range(ncpu))
    seen = set()
    for r in range(8):
        p = affinity.plan(r, 8, BUS, allowed, root)
        node = 0 if r < 4 else 1
        k = r % 4
        lo = 64 * node + 16 * k
        want = list(range(lo, lo + 16)) + list(range(lo + 128, lo + 144))   # 16 cores and their SMT siblings
        assert p["cpus"] == want, (r, affinity.format_cpulist(p["cpus"]))
        assert p["numa_node"] == node
        assert p["ranks_on_node"] == ([0, 1, 2, 3] if node == 0 else [4, 5, 6, 7])
        assert p["gpu_pci_bus_id"] == BUS[r]
        assert not seen & set(p["cpus"])          # disjoint
        seen |= set(p["cpus"])
    assert seen == set(allowed)                   # ... and nothing left over


def test_world_sizes_of_the_scaling_bench(tmp_path):
    """N = 1, 2, 4 of SCALE: the ranks' GPUs all hang off socket 0 and share ITS cores only."""
    root = str(tmp_path)
    ncpu = fake_sysfs(root)
    allowed = list(range(ncpu))
    for world in (1, 2, 4):
        cpus = [affinity.plan(r, world, BUS[:world], allowed, root)["cpus"] for r in range(world)]
        for c in cpus:
            assert len(c) == 128 // world
            assert all(x % 128 < 64 for x in c)   # socket 0 and its siblings
        assert len(set().union(*map(set, cpus))) == 128


def test_restricted_affinity_mask(tmp_path):
    """A job that was given 32 CPUs of socket 0 and 32 of socket 1: each rank takes its share of what is allowed
    on its GPU's node."""
    root = str(tmp_path)
    fake_sysfs(root)
    allowed = list(range(0, 16)) + list(range(128, 144)) + list(range(64, 80)) + list(range(192, 208))
    p0 = affinity.plan(0, 8, BUS, allowed, root)
    assert p0["cpus"] == [0, 1, 2, 3, 128, 129, 130, 131]
    p5 = affinity.plan(5, 8, BUS, allowed, root)
    assert p5["cpus"] == [68, 69, 70, 71, 196, 197, 198, 199]


def test_uneven_gpu_placement(tmp_path):
    """Six GPUs on node 0 and two on node 1: the shares follow the node, not the rank count."""
    root = str(tmp_path)
    ncpu = fake_sysfs(root, gpu_nodes=(0, 0, 0, 0, 0, 0, 1, 1))
    p = [affinity.plan(r, 8, BUS, list(range(ncpu)), root) for r in range(8)]
    assert [len(x["cpus"]) for x in p] == [22, 22, 22, 22, 20, 20, 64, 64]   # 64 cores / 6: 11, 11, 11, 11, 10, 10
    assert p[6]["ranks_on_node"] == [6, 7]


@pytest.mark.parametrize("numa", [False])
def test_falls_back_without_numa_information(tmp_path, numa):
    """numa_node = -1 (single-node hosts, many containers), or no sysfs entry at all: contiguous shares of whole
    cores of the allowed CPUs."""
    root = str(tmp_path)
    ncpu = fake_sysfs(root, numa=numa)
    p = affinity.plan(3, 8, BUS, list(range(ncpu)), root)
    assert p["numa_node"] is None and "no NUMA information" in p["how"]
    assert p["cpus"] == list(range(48, 64)) + list(range(176, 192))
    # no tree at all, no bus ids: logical CPUs are their own cores
    q = affinity.plan(1, 2, [None, None], list(range(8)), os.path.join(root, "nothing"))
    assert q["cpus"] == [4, 5, 6, 7]
    # fewer cores than ranks: nothing to partition
    z = affinity.plan(1, 8, [None] * 8, [0, 1, 2], os.path.join(root, "nothing"))
    assert z["cpus"] == [] and "not binding" in z["how"]
