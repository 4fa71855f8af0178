#!/usr/bin/env python
"""Is the small-scene step floor (bimodal with the PROCESS: profiles/r6_small_scene_binning_direct_vs_auto.txt) a matter of
which CPUs the process runs on? Prints the GPU's PCI address, its NUMA node and local CPU list, the host's NUMA layout and
this process's affinity, then times bench.py --n N (host-bound sizes) pinned to the GPU-local CPUs, to the other node's, to
single cores, and unpinned. Design tool; runs on the GPU box."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpulist(s):
    out = []
    for part in s.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def gpu_pci():
    import torch
    p = torch.cuda.get_device_properties(0)
    dom, bus, dev = getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", 0)
    return None if bus is None else f"{dom:04x}:{bus:02x}:{dev:02x}.0"


def bench(n, cpus, steps=200):
    env = dict(os.environ)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--n", str(n), "--steps", str(steps), "--warmup", "20", "--cpu-sample", "0"]
    pre = (lambda: os.sched_setaffinity(0, cpus)) if cpus else None
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, preexec_fn=pre, timeout=300)
    for line in r.stdout.splitlines()[::-1]:
        if line.startswith("{"):
            return json.loads(line)["ms_per_step"]
    return None


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [1000, 100000, 2000000]
    bdf = gpu_pci()
    print("gpu pci", bdf)
    local = None
    if bdf and os.path.isdir(f"/sys/bus/pci/devices/{bdf}"):
        node = open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip()
        local = cpulist(open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read())
        print("gpu numa_node", node, "local cpus", len(local), local[:4], "...", local[-4:] if local else None)
    nodes = {}
    for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        nodes[os.path.basename(d)] = cpulist(open(d + "/cpulist").read())
    print("host nodes", {k: (len(v), v[0], v[-1]) for k, v in nodes.items() if v})
    aff = sorted(os.sched_getaffinity(0))
    print("affinity of this process", len(aff), aff[:4], "...", aff[-4:])
    allowed = set(aff)
    sets = {"unpinned": None}
    loc = [c for c in (local or aff) if c in allowed]
    ncore = len(loc) // 2 if len(loc) >= 4 else len(loc)          # (Linux numbers the SMT siblings of a node's cores behind them)
    c0 = loc[ncore // 2 // 8 * 8] if ncore >= 16 else loc[0]      # first core of a CCD (8 cores share an L3 on this host)
    sib = {c: loc[loc.index(c) + ncore] for c in loc[:ncore]} if len(loc) == 2 * ncore else {}
    sets["1 core"] = [c0]
    if sib:
        sets["1 core + its SMT sibling"] = [c0, sib[c0]]
    for k in (2, 4, 8, 16, 32):
        if k <= ncore:
            sets[f"{k} neighbouring cores"] = [c for c in range(c0, c0 + k) if c in allowed]
    if sib:
        sets["4 cores + siblings"] = [c for c in range(c0, c0 + 4)] + [sib[c] for c in range(c0, c0 + 4) if c in sib]
    sets["all gpu-local cpus"] = loc
    if local:
        rem = [c for c in aff if c not in set(local)]
        if rem:
            sets["2 remote cores"] = rem[8:10]
    for n in sizes:
        for rep in range(2):
            for name, cpus in sets.items():
                ms = bench(n, cpus)
                print(f"N={n} rep={rep} {name:22s} ({'all' if cpus is None else len(cpus)} cpus): {ms if ms is None else round(ms, 4)} ms/step", flush=True)


if __name__ == "__main__":
    main()
