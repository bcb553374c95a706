"""Time sharding of an export across the GPUs of one node (one process per GPU, no data-path collective).

The reference renders one frame after another in a single process (export/exporter.py:538-549).  Frames only
depend on the past through the analysis halo (previous block / previous magnitudes, rebuilt inside the STFT
kernel) and three scalar recurrences (carried state, ``HipExportEngine.carried_state``), so contiguous frame
ranges are independent units of work: rank r renders ``frame_range(total, world, r)`` and the host writes the
segments to the sink in rank order.  Whole tracks (BASELINE.json configs[4]: 64 tracks on 8 GPUs) are dealt
round-robin with ``track_assignment``.
"""
from __future__ import annotations

import os


def frame_range(total_frames: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced [begin, end) of ``rank``; ranges of all ranks tile [0, total_frames) in rank order."""
    total_frames, world, rank = int(total_frames), int(world), int(rank)
    if world < 1 or not (0 <= rank < world) or total_frames < 0:
        raise ValueError(f"bad shard request: total={total_frames} world={world} rank={rank}")
    base, extra = divmod(total_frames, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def track_assignment(n_tracks: int, world: int, rank: int) -> list[int]:
    """Indices of the tracks ``rank`` exports when whole tracks are dealt round-robin."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad shard request: world={world} rank={rank}")
    return list(range(int(rank), int(n_tracks), int(world)))


def max_over_ranks(seconds: float, device=None) -> float:
    """Slowest rank's time (the job's time).  Uses the default process group if one is initialised."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_digests_in_order(digest: str, n_bytes: int, dst: int = 0):
    """The only exchange of a time-sharded export between ranks that share a process group: every rank writes its segment
    to its own file (or its own FFmpeg) and hands ``dst`` a (rank, sha256 hex digest, byte count) record -- never the
    frames themselves -- so ``dst`` can verify the segments it joins in rank order = frame order.  Returns the records
    on ``dst`` and None elsewhere.  (``aurora_amd.export.sharded`` needs no process group at all: its children are
    independent processes and the supervisor reads the segment files.)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [(0, str(digest), int(n_bytes))]
    rec = (dist.get_rank(), str(digest), int(n_bytes))
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(rec, out, dst=dst)
    return out


# ---- one process per GPU, started by a parent that never touches a GPU ---------------------------------------------

def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def launch_ranks(world: int, argv, script: str, env=None, python: str | None = None, poll_s: float = 0.2) -> int:
    """Starts ``world`` fresh child processes ``python script *argv`` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set (what ``torch.distributed.run`` would set for one node) and waits for them.  Rank 0 inherits this
    process's stdout (its one JSON line is the job's); the other ranks' stdout goes to stderr.  A rank that fails takes
    the others down with it (they would wait in a barrier for good otherwise).  Returns the worst exit status.

    The caller must not have touched the GPU: children are started with ``subprocess`` (fork + exec of a new
    interpreter), never by re-executing the running program."""
    import subprocess
    import sys
    import time
    world = int(world)
    if world < 1:
        raise ValueError(f"world {world}")
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base.setdefault("MASTER_PORT", str(free_port()))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base["WORLD_SIZE"] = base["LOCAL_WORLD_SIZE"] = str(world)
    procs = []
    try:
        for r in range(world):
            e = dict(base)
            e["RANK"] = e["LOCAL_RANK"] = str(r)
            procs.append(subprocess.Popen([python or sys.executable, script, *argv], env=e,
                                          stdout=None if r == 0 else sys.stderr))
        status = [None] * world
        while any(s is None for s in status):
            for r, p in enumerate(procs):
                if status[r] is None:
                    status[r] = p.poll()
            if any(s not in (None, 0) for s in status):
                break
            time.sleep(poll_s)
    finally:
        for p in procs:                       # a failed rank (or an interrupt here) ends the job
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    codes = [p.returncode for p in procs]
    bad = [c for c in codes if c != 0]
    if not bad:
        return 0
    pos = [c for c in bad if c > 0]
    return max(pos) if pos else 128 + max(-c for c in bad)      # a rank killed by signal s reads as 128 + s


# ---- placement of a rank on the host ---------------------------------------------------------------------------------
# Every exported frame crosses the GPU's PCIe link into a pinned host ring (24.9 MB per 4K frame, 57 GB/s per GPU).  On a
# two-socket node the ring and the threads that drain it belong on the socket the GPU hangs off: a ring on the other
# socket sends 8 x 57 GB/s of DMA writes over the inter-socket links.  One process per GPU makes this a matter of CPU
# affinity: pages are allocated on the node of the thread that allocates (and HIP's own default picks the node nearest
# the current device), so the rank binds itself before it allocates anything.

def parse_cpulist(text: str) -> set:
    """"0-15,128-143" -> {0..15, 128..143} (the format of /sys/.../local_cpulist)."""
    cpus = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def device_numa(bus_id: str, sysfs: str = "/sys/bus/pci/devices"):
    """(numa_node, cpus local to it) of the PCI device ``dddd:bb:dd.f``; (None, set()) where the host does not say
    (single-node machines and most virtual machines report -1)."""
    base = os.path.join(sysfs, bus_id.strip().lower())
    try:
        with open(os.path.join(base, "numa_node")) as fh:
            node = int(fh.read().strip())
        with open(os.path.join(base, "local_cpulist")) as fh:
            cpus = parse_cpulist(fh.read())
    except (OSError, ValueError):
        return None, set()
    return (node if node >= 0 else None), cpus


def pci_bus_id_from_sysfs(device: int, kfd: str = "/sys/class/kfd/kfd/topology/nodes", env=None):
    """PCI address ``dddd:bb:dd.f`` of HIP device ``device`` WITHOUT touching the HIP runtime: the GPU nodes of the KFD
    topology in node order (``simd_count`` > 0; ``location_id`` = bus << 8 | device << 3 | function, ``domain``), filtered
    the way the runtime filters them -- ``ROCR_VISIBLE_DEVICES`` first, then ``HIP_VISIBLE_DEVICES`` /
    ``CUDA_VISIBLE_DEVICES`` on what is left (integer lists only; UUID forms return None and the caller asks HIP).
    None when the host does not show the topology."""
    env = os.environ if env is None else env
    try:
        nodes = sorted((n for n in os.listdir(kfd) if n.isdigit()), key=int)
    except OSError:
        return None
    gpus = []
    for n in nodes:
        props = {}
        try:
            with open(os.path.join(kfd, n, "properties")) as fh:
                for line in fh:
                    k, _, v = line.strip().partition(" ")
                    if v.strip().lstrip("-").isdigit():
                        props[k] = int(v)
        except OSError:
            continue
        if props.get("simd_count", 0) > 0:
            gpus.append(props)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = env.get(var)
        if val is None or (var == "CUDA_VISIBLE_DEVICES" and env.get("HIP_VISIBLE_DEVICES") is not None):
            continue
        toks = [t.strip() for t in val.split(",") if t.strip()]
        if not all(t.isdigit() for t in toks):
            return None
        picked = []
        for t in toks:
            if int(t) >= len(gpus):
                break                         # the runtime stops at the first index it cannot resolve
            picked.append(gpus[int(t)])
        gpus = picked
    if not (0 <= int(device) < len(gpus)) or "location_id" not in gpus[int(device)]:
        return None
    loc, dom = gpus[int(device)]["location_id"], gpus[int(device)].get("domain", 0)
    return "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)


def _bind_other_threads(cpus) -> int:
    """Applies the affinity mask to the process's other threads; returns how many took it (a thread may exit meanwhile)."""
    n = 0
    import threading
    me = threading.get_native_id()
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except (OSError, ValueError):
        return 0
    for tid in tids:
        if tid == me:
            continue
        try:
            os.sched_setaffinity(tid, cpus)
            n += 1
        except OSError:
            pass
    return n


def bind_near_device(lib, device: int, sysfs: str = "/sys/bus/pci/devices",
                     kfd: str | None = "/sys/class/kfd/kfd/topology/nodes"):
    """Restricts EVERY thread of the calling process to the CPUs next to HIP device ``device`` (those it is allowed to
    use anyway).  Asking for the device's PCI address initialises the HIP runtime, and callers have usually set the device
    or started a process group by then, so helper threads (HIP, RCCL, torch) already exist: ``sched_setaffinity(0, ..)``
    would bind the calling thread only, so the mask is applied to every task under ``/proc/self/task``; threads started
    later inherit it from their creator.  What was allocated before the call stays where it is -- call this before the
    pinned frame ring is allocated (the workers and ``bench.py`` do).  ``lib`` = None: only the KFD topology in sysfs is
    asked (``pci_bus_id_from_sysfs``: no HIP call at all), which lets a rank bind itself before the runtime starts and
    allocates anything; with a library the sysfs route is still tried first.  Returns a dict describing what was done;
    never raises: a host that does not expose the topology is left alone."""
    import ctypes as C
    info = {"device": int(device), "bus_id": None, "numa_node": None, "bound_cpus": 0}
    bus = pci_bus_id_from_sysfs(device, kfd) if kfd else None
    if bus is None:
        if lib is None:
            return info
        buf = C.create_string_buffer(64)
        try:
            if lib.aur_device_pci_bus_id(int(device), buf, 64) != 0:
                return info
        except Exception:
            return info
        bus = buf.value.decode("ascii", "ignore")
    else:
        info["via"] = "sysfs"
    info["bus_id"] = bus
    node, cpus = device_numa(info["bus_id"], sysfs)
    info["numa_node"] = node
    if node is None or not cpus or not hasattr(os, "sched_setaffinity"):
        return info
    try:
        allowed = os.sched_getaffinity(0)
        want = allowed & cpus
        if want and want != allowed:
            os.sched_setaffinity(0, want)
            info["bound_threads"] = 1 + _bind_other_threads(want)
        info["bound_cpus"] = len(want)
    except OSError:
        pass
    return info
