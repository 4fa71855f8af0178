"""sfgs.affinity (host logic; runs on CPU): the CPU-set choice that removes the bimodal small-scene step floor
(profiles/r6_cpu_affinity_small_scenes.txt). Pure functions against a fake sysfs tree, then the real call on this host."""
import os

import pytest

from sfgs import affinity as A


def _fake_sysfs(root, n_domains=4, cores_per_domain=4, smt=True):
    """cpu numbering as on the GPU box: cores 0 .. C-1, their SMT siblings C .. 2C-1; an L3 per `cores_per_domain` cores"""
    C = n_domains * cores_per_domain
    for c in range(2 * C if smt else C):
        core = c % C
        d = core // cores_per_domain
        lo = d * cores_per_domain
        shared = f"{lo}-{lo + cores_per_domain - 1}" + (f",{C + lo}-{C + lo + cores_per_domain - 1}" if smt else "")
        (root / f"cpu{c}" / "cache" / "index3").mkdir(parents=True)
        (root / f"cpu{c}" / "topology").mkdir(parents=True)
        (root / f"cpu{c}" / "cache" / "index3" / "shared_cpu_list").write_text(shared + "\n")
        (root / f"cpu{c}" / "topology" / "thread_siblings_list").write_text((f"{core},{core + C}" if smt else f"{core}") + "\n")
    return C


def test_cpulist_format():
    assert A.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert A.parse_cpulist("") == [] and A.parse_cpulist("5") == [5]


def test_domains_put_physical_cores_before_their_siblings(tmp_path):
    C = _fake_sysfs(tmp_path)
    doms = A.l3_domains(range(2 * C), sysfs=str(tmp_path))
    assert len(doms) == 4
    assert doms[0] == [0, 1, 2, 3, 16, 17, 18, 19] and doms[3] == [12, 13, 14, 15, 28, 29, 30, 31]
    # a restricted CPU set (container, taskset): only what is allowed, domains that lost every CPU disappear
    doms = A.l3_domains([2, 3, 18, 12], sysfs=str(tmp_path))
    assert doms == [[2, 3, 18], [12]]
    # no topology files at all: one domain of everything
    assert A.l3_domains([0, 1, 2], sysfs=str(tmp_path / "missing")) == [[0, 1, 2]]


def test_ranks_get_a_domain_each(tmp_path):
    C = _fake_sysfs(tmp_path)
    doms = A.l3_domains(range(2 * C), sysfs=str(tmp_path))
    picks = [A.choose_cores(doms, r, cores=2) for r in range(6)]
    assert picks[:4] == [[0, 1], [4, 5], [8, 9], [12, 13]]       # physical cores of four different L3 domains
    assert picks[4] == picks[0] and picks[5] == picks[1]          # more ranks than domains: they wrap
    assert A.choose_cores(doms, 0, cores=6) == [0, 1, 2, 3, 16, 17]   # more than the domain's cores: its SMT siblings, never a
    assert A.choose_cores(doms, 0, cores=99) == sorted(doms[0])       # second domain
    assert A.choose_cores([], 0, 4) == []


@pytest.mark.skipif(not hasattr(os, "sched_setaffinity"), reason="no sched_setaffinity on this platform")
def test_pin_and_unpin_on_this_host(monkeypatch):
    import torch
    before = sorted(os.sched_getaffinity(0))
    threads = torch.get_num_threads()
    try:
        monkeypatch.setenv("SFGS_PIN", "0")
        assert A.pin() is None and sorted(os.sched_getaffinity(0)) == before
        monkeypatch.delenv("SFGS_PIN")
        assert A.pin(min_cpus=len(before) + 1) is None           # a small CPU set is somebody's choice already: left alone
        assert sorted(os.sched_getaffinity(0)) == before
        got = A.pin(local_rank=0, cores=2, min_cpus=1)
        assert got is not None and 1 <= len(got) <= 2 and set(got) <= set(before)
        assert sorted(os.sched_getaffinity(0)) == got and torch.get_num_threads() == len(got)
        A.unpin()
        assert sorted(os.sched_getaffinity(0)) == before and torch.get_num_threads() == threads
    finally:
        os.sched_setaffinity(0, before)
        torch.set_num_threads(threads)
        A._ORIGINAL = A._ORIGINAL_THREADS = None


@pytest.mark.skipif(not hasattr(os, "sched_setaffinity"), reason="no sched_setaffinity on this platform")
def test_auto_pins_small_scenes_and_releases_grown_ones():
    """auto(): on_frame(N) -- called by diff_gauss with every forward's Gaussian count -- pins below `below`, releases above
    `above`, does nothing in between (hysteresis) and nothing at all when the mechanism is off."""
    import threading
    before = sorted(os.sched_getaffinity(0))
    try:
        A.on_frame(10)                                   # off: nothing happens
        assert A.state() == {"policy": None, "pinned_to": None}
        A.auto(local_rank=0, cores=1, below=1000, above=2000, min_cpus=1)
        assert "1 cores" in A.state()["policy"]
        A.on_frame(5000)
        assert A.state()["pinned_to"] is None and sorted(os.sched_getaffinity(0)) == before
        # a second thread that already exists must follow the switch too (autograd's device thread, the HIP runtime's)
        seen, go, done = {}, threading.Event(), threading.Event()

        def other():
            go.wait(10)
            seen["cpus"] = sorted(os.sched_getaffinity(0))
            done.set()
        t = threading.Thread(target=other)
        t.start()
        A.on_frame(500)
        got = A.state()["pinned_to"]
        assert got is not None and len(got) == 1 and sorted(os.sched_getaffinity(0)) == got
        go.set(); done.wait(10); t.join()
        assert seen["cpus"] == got
        A.on_frame(1500)                                 # between the thresholds: stays as it is
        assert A.state()["pinned_to"] == got
        A.on_frame(2500)
        assert A.state()["pinned_to"] is None and sorted(os.sched_getaffinity(0)) == before
        A.auto(cores=0)
        A.on_frame(10)
        assert A.state() == {"policy": None, "pinned_to": None}
    finally:
        A.auto(cores=0)
        A.unpin()
        os.sched_setaffinity(0, before)
        A._ORIGINAL = A._ORIGINAL_THREADS = A._PINNED = None
