"""Keep a training process on a few neighbouring CPU cores (optional; host logic, no kernel of ours).

Below ~500 k Gaussians a step is bound by the HOST: torch's dispatch and autograd engine, this library's one C call per
direction with its dozen launches, the plan's one wait. That work bounces between three threads -- the caller, autograd's
device thread, the HIP runtime's -- and on a 2 x 64-core host the scheduler spreads them over the machine: every hand-over
then wakes a core in another L3 domain. Measured on the GPU box (tools/diag_numa.py, bench.py --n N; profiles/
r6_cpu_affinity_small_scenes.txt): the same step takes 0.154 ms at N = 1 000 and 0.178 ms at N = 100 000 with the process
confined to <= 8 cores of ONE L3 domain (a CCD), and 0.23 - 0.34 ms, bimodal from process to process, with 16 cores or more --
whether those are on the GPU's NUMA node or not makes no difference. From 500 k Gaussians up the GPU is the bottleneck and
the CPU set stops helping (0.32 against 0.33 - 0.36 ms).

From 500 k Gaussians up the GPU is the bottleneck -- and there the confinement COSTS: the headline step (2 M) reads 0.928 ms
unpinned and 0.974 on 4 cores, 0.936 on 8 or 16 (medians of 8 fresh processes each, same box): the runtime's helper threads and
the two launching threads want a core each and nothing waits on their hand-overs any more. Hence `auto()`: the process is
pinned while the scene is small and released when it has grown (every rasterizer forward reports its N; the set is switched on
crossing `below` / `above`, for ALL threads of the process).

The reference starts one training process per scene and GPU (scripts/run_jax.py:52-87); nothing in it sets an affinity.
`auto()` is what tools/launch_scenes.py calls per rank and what a user adds as the first line of train.py's process:

    import sfgs.affinity; sfgs.affinity.auto(local_rank=int(os.environ.get("LOCAL_RANK", 0)))

`pin()` / `unpin()` are the two switches themselves. They change the CPU set of every thread of the CALLING process
(sched_setaffinity per task); threads and worker processes started afterwards inherit the current set (and torch's intra-op
pool, if torch is already imported, is resized with it), so give it more cores (cores=8) when the process also decodes images
in workers. Never applied implicitly: importing sfgs / diff_gauss does not touch the affinity."""
import os
import sys

__all__ = ["auto", "on_frame", "state", "pin", "unpin", "parse_cpulist", "l3_domains", "choose_cores", "gpu_local_cpus"]

_ORIGINAL = None
_ORIGINAL_THREADS = None
_AUTO = None       # auto(): dict(local_rank, cores, below, above, device_index); None = off
_PINNED = None     # the CPU list the process is confined to right now


def parse_cpulist(s):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out = []
    for part in str(s).strip().split(","):
        part = part.strip()
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def l3_domains(cpus, sysfs="/sys/devices/system/cpu"):
    """The allowed CPUs grouped by the last-level cache they share, each group as (physical cores first, then their SMT
    siblings), groups in order of their lowest CPU. Without the sysfs entries: one group of everything."""
    cpus = sorted(cpus)
    allowed = set(cpus)
    groups, seen = [], set()
    for c in cpus:
        if c in seen:
            continue
        shared = _read(f"{sysfs}/cpu{c}/cache/index3/shared_cpu_list")
        members = [x for x in (parse_cpulist(shared) if shared else cpus) if x in allowed and x not in seen]
        if not members:
            members = [c]
        seen.update(members)
        first, rest = [], []
        for x in members:   # a core's first hardware thread before anybody's second
            sib = _read(f"{sysfs}/cpu{x}/topology/thread_siblings_list")
            sibs = [y for y in (parse_cpulist(sib) if sib else [x]) if y in allowed]
            (first if not sibs or x == min(sibs) else rest).append(x)
        groups.append(first + rest)
    return groups


def choose_cores(domains, local_rank=0, cores=4):
    """`cores` CPUs of ONE L3 domain for this rank: rank r takes domain r (modulo their number), so the ranks of a node do not
    share a domain while there are enough of them; within the domain physical cores come first. Domains too small for the
    request give what they have."""
    domains = [d for d in domains if d]
    if not domains:
        return []
    d = domains[int(local_rank) % len(domains)]
    return sorted(d[:max(1, int(cores))])


def gpu_local_cpus(device_index=0):
    """CPUs of the NUMA node the GPU hangs on (sysfs local_cpulist of its PCI function), or None when that cannot be read.
    (Locality made no measurable difference to the step floor; it is used to spread the ranks of a node sensibly.)"""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{getattr(p, 'pci_device_id', 0):02x}.0"
    except Exception:
        return None
    s = _read(f"/sys/bus/pci/devices/{bdf}/local_cpulist")
    return parse_cpulist(s) if s else None


def _set_all_threads(cpus):
    """sched_setaffinity acts on ONE task: every thread of this process, the caller included (threads created later inherit
    their creator's set)."""
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = []
    for t in tids:
        try:
            os.sched_setaffinity(t, cpus)
        except OSError:        # a thread that ended meanwhile
            pass
    os.sched_setaffinity(0, cpus)


def pin(local_rank=0, cores=4, device_index=None, min_cpus=16):
    """Confine this process to `cores` CPUs of one L3 domain (see the module text). Returns the chosen CPU list, or None when
    nothing was changed: SFGS_PIN=0 in the environment, a platform without sched_setaffinity, or fewer than `min_cpus`
    allowed CPUs (somebody -- a container, taskset, a job scheduler -- already chose)."""
    global _ORIGINAL, _ORIGINAL_THREADS, _PINNED
    if os.environ.get("SFGS_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    if _ORIGINAL is None:
        _ORIGINAL = allowed
    if len(allowed) < min_cpus:
        return None
    pool = allowed
    if device_index is not None:
        local = gpu_local_cpus(device_index)
        if local:
            near = [c for c in allowed if c in set(local)]
            if len(near) >= cores:
                pool = near
    chosen = choose_cores(l3_domains(pool), local_rank, cores)
    if not chosen:
        return None
    _set_all_threads(chosen)
    _PINNED = chosen
    torch = sys.modules.get("torch")
    if torch is not None:   # its intra-op pool was sized for the CPUs it saw at import
        if _ORIGINAL_THREADS is None:
            _ORIGINAL_THREADS = torch.get_num_threads()
        torch.set_num_threads(len(chosen))
    return chosen


def unpin():
    """Back to the CPU set (and torch intra-op thread count) found by the first pin()."""
    global _ORIGINAL, _ORIGINAL_THREADS, _PINNED
    if _ORIGINAL is not None and hasattr(os, "sched_setaffinity"):
        _set_all_threads(_ORIGINAL)
        _ORIGINAL = None
    _PINNED = None
    torch = sys.modules.get("torch")
    if torch is not None and _ORIGINAL_THREADS is not None:
        torch.set_num_threads(_ORIGINAL_THREADS)
    _ORIGINAL_THREADS = None


def auto(local_rank=0, cores=4, below=500_000, above=800_000, device_index=None, min_cpus=16):
    """Pinned while the scene is small, released once it has grown: from now on every forward of the rasterizer reports its
    Gaussian count (diff_gauss -> on_frame) and the process is confined (pin) when a frame has fewer than `below` Gaussians,
    released (unpin) when one has more than `above`. cores <= 0 switches the mechanism off again.
    The defaults are those of a bare rasterizer loop (bench.py: host-bound below ~500 k). A TRAINING iteration issues ~100
    torch launches around the rasterizer and stays host-sensitive much longer -- the reference's real classes with the hooks:
    1.41 -> 1.17 ms at 100 k, 1.44 -> 1.16 at 500 k, 1.45 -> 1.22 at 1 M, 1.85 -> 1.68 at 2 M, even at 4 M -- so
    tools/launch_scenes.py passes below = 2.5 M, above = 3.5 M."""
    global _AUTO
    if cores <= 0 or os.environ.get("SFGS_PIN", "1") == "0":
        _AUTO = None
        return
    _AUTO = dict(local_rank=int(local_rank), cores=int(cores), below=int(below), above=int(above), device_index=device_index,
                 min_cpus=int(min_cpus))


def on_frame(n):
    """Called by diff_gauss.GaussianRasterizer's forward with the frame's Gaussian count (two integer compares per frame while
    nothing changes)."""
    a = _AUTO
    if a is None:
        return
    if _PINNED is None:
        if n < a["below"]:
            pin(a["local_rank"], a["cores"], a["device_index"], a["min_cpus"])
    elif n > a["above"]:
        unpin()


def state():
    """{'policy': ..., 'pinned_to': [...] | None} for logs (bench.py prints it in its JSON line)."""
    a = _AUTO
    return {"policy": None if a is None else f"{a['cores']} cores of one L3 domain while N < {a['below']}, released above {a['above']}",
            "pinned_to": _PINNED}

