"""Rank -> CPU affinity: pin a rank's host threads to the CPUs local to ITS GPU (same NUMA node / PCIe root).

Counterpart of the reference's utils/distributed.py:19-62 (`set_affinity`, which asks NVML for the GPU's CPU mask before
torch.cuda.set_device).  On ROCm the same information is in sysfs: /sys/bus/pci/devices/<domain:bus:dev.fn>/
local_cpulist (and numa_node).  It matters on this path: every frame ends with the calling thread polling a pinned host
word the GPU writes (gcr_forward), and the frame loop is host-enqueue-bound at small scenes -- a thread that migrates to the
other socket pays a cross-socket hop per poll and per doorbell.  With eight ranks on one node each rank keeps to its GPU's
node; the pinned read-back word is allocated by the first frame, i.e. after the binding, on that node (first touch).

No GPU / no sysfs / an empty list -> nothing is changed and the reason is returned (never an error: affinity is an
optimisation).  GCR_NO_AFFINITY=1 disables it.
"""
import os


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    cpus = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            cpus.extend(range(int(lo), int(hi) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def gpu_pci_address(device_index):
    """'0000:c1:00.0' of torch device `device_index`, or None."""
    import torch
    try:
        pr = torch.cuda.get_device_properties(device_index)
        dom, bus, dev = int(getattr(pr, "pci_domain_id", 0)), int(pr.pci_bus_id), int(pr.pci_device_id)
    except Exception:
        return None
    return "%04x:%02x:%02x.0" % (dom, bus, dev)


def kfd_gpu_pci_addresses(topology="/sys/class/kfd/kfd/topology/nodes"):
    """PCI addresses of the GPU agents in KFD node order -- HIP's device order -- read from sysfs WITHOUT touching the
    HIP runtime, so that a rank can bind itself before the runtime (and its helper threads, and the first pinned
    allocations) exist.  HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES lists of plain indices are honoured; anything else
    (UUIDs) -> [] and the caller falls back to the runtime's answer."""
    try:
        nodes = sorted((int(n) for n in os.listdir(topology) if n.isdigit()))
    except OSError:
        return []
    gpus = []
    for n in nodes:
        try:
            props = dict(line.split(None, 1) for line in open(os.path.join(topology, str(n), "properties")) if " " in line)
            if int(props.get("simd_count", "0")) <= 0:
                continue  # a CPU agent
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
        except (OSError, ValueError, KeyError):
            return []
        gpus.append("%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            try:
                gpus = [gpus[int(i)] for i in v.split(",") if i.strip() != ""]
            except (ValueError, IndexError):
                return []
    return gpus


def gpu_local_cpus(device_index, sysfs="/sys/bus/pci/devices", pci=None):
    """(cpus, numa_node, why): the CPUs local to the GPU according to sysfs; cpus == [] with a reason otherwise."""
    addr = pci if pci is not None else gpu_pci_address(device_index)
    if addr is None:
        return [], None, "no PCI address for device %s" % device_index
    base = os.path.join(sysfs, addr)
    try:
        cpus = parse_cpulist(open(os.path.join(base, "local_cpulist")).read())
    except OSError as e:
        return [], None, "sysfs: %s" % e
    node = None
    try:
        node = int(open(os.path.join(base, "numa_node")).read())
    except (OSError, ValueError):
        pass
    return cpus, node, "ok" if cpus else "empty local_cpulist for %s" % addr


def bind_rank_early(device_index, sysfs="/sys/bus/pci/devices", topology="/sys/class/kfd/kfd/topology/nodes"):
    """bind_rank_to_gpu() from sysfs alone, to be called BEFORE the process initialises HIP, so that the runtime's
    helper threads and the first pinned allocations come up on the GPU's NUMA node (a binding made after
    torch.cuda.set_device leaves them where they were).  Where the KFD topology is not visible (the gpurun boxes of
    this project: their containers do not mount it) the caller falls back to the late binding."""
    if os.environ.get("GCR_NO_AFFINITY") == "1":
        return {"bound": False, "cpus": 0, "numa_node": None, "why": "GCR_NO_AFFINITY=1"}
    gpus = kfd_gpu_pci_addresses(topology)
    if device_index >= len(gpus):
        return {"bound": False, "cpus": 0, "numa_node": None, "why": "no KFD topology entry for device %d" % device_index}
    r = bind_rank_to_gpu(device_index, sysfs, pci=gpus[device_index])
    r["when"] = "before HIP initialisation"
    return r


def bind_rank_to_gpu(device_index, sysfs="/sys/bus/pci/devices", pci=None):
    """os.sched_setaffinity(this process, CPUs local to the GPU).  Returns a dict for logs:
    {"bound": bool, "cpus": n, "numa_node": k, "why": str}."""
    if os.environ.get("GCR_NO_AFFINITY") == "1":
        return {"bound": False, "cpus": 0, "numa_node": None, "why": "GCR_NO_AFFINITY=1"}
    cpus, node, why = gpu_local_cpus(device_index, sysfs, pci)
    if not cpus:
        return {"bound": False, "cpus": 0, "numa_node": node, "why": why}
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))  # never widen a mask the launcher / cgroup set
    if not allowed:
        return {"bound": False, "cpus": 0, "numa_node": node, "why": "GPU-local CPUs are outside this process's mask"}
    try:
        os.sched_setaffinity(0, allowed)
    except OSError as e:
        return {"bound": False, "cpus": 0, "numa_node": node, "why": "sched_setaffinity: %s" % e}
    return {"bound": True, "cpus": len(allowed), "numa_node": node, "why": "ok", "cpulist": format_cpulist(allowed),
            "pci": pci if pci is not None else gpu_pci_address(device_index)}


def format_cpulist(cpus):
    """[0, 1, 2, 3, 8, 10, 11] -> '0-3,8,10-11' (inverse of parse_cpulist)."""
    cpus = sorted(set(cpus))
    parts, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        parts.append("%d-%d" % (cpus[i], cpus[j]) if j > i else "%d" % cpus[i])
        i = j + 1
    return ",".join(parts)


def summarize_rccl_log(text):
    """What an RCCL log written with NCCL_DEBUG=INFO (NCCL_DEBUG_FILE) says about the communicator this rank built:
    channel count, ring / tree graphs, transports (xGMI peer-to-peer, shared memory, network) and the algorithm /
    protocol lines RCCL prints when NCCL_DEBUG_SUBSYS includes TUNING.  Best effort -- the log format is RCCL's, not
    an interface: fields that are not found are simply absent."""
    import re
    out = {}
    m = re.findall(r"(\d+) coll channels", text)
    if m:
        out["coll_channels"] = int(m[-1])
    out["ring_lines"] = len(re.findall(r"NCCL INFO (?:Channel \d+/\d+ :|Ring \d+ :)", text))
    out["tree_lines"] = len(re.findall(r"NCCL INFO Trees? ", text))
    tr = {}
    for name, pat in (("p2p_xgmi_or_ipc", r"via P2P/"), ("shm", r"via SHM"), ("net", r"via NET/")):
        n = len(re.findall(pat, text))
        if n:
            tr[name] = n
    if tr:
        out["transports"] = tr
    algo = sorted(set(re.findall(r"\b(?:Algo|algorithm)[ =:]+(\w+)", text)))
    if algo:
        out["algorithms"] = algo
    proto = sorted(set(re.findall(r"\b(?:Proto|protocol)[ =:]+(\w+)", text)))
    if proto:
        out["protocols"] = proto
    ver = re.search(r"(?:RCCL|NCCL) version ([\w.+-]+)", text)
    if ver:
        out["version"] = ver.group(1)
    return out
