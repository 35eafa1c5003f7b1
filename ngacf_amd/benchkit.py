"""Measurement helpers shared by bench.py's one-GPU and node-partitioned legs (no model code here).

  * launch_command / self_launch: `python bench.py --gpus N` without a torchrun environment starts
    `python -m torch.distributed.run ... bench.py --gpus N ...` as a CHILD process and relays its one JSON line.  The
    parent never imports torch and never touches the GPU (re-exec'ing a process that has initialised HIP takes the box
    down on this pool).
  * timed_steps: K steps of R passes each between two barriers + synchronize() (the bench contract), with one event per
    step so the median step ("window") is reported next to the whole-region figure.
  * roofs: the applicable ceilings of a row-gather kernel by table size (MI355X_MICROARCH.md, "Indexed rows").
"""
import json
import math
import os
import socket
import subprocess
import sys
import time

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable (float4 copy)
# MI355X_MICROARCH.md, "Indexed rows: gather into LDS" (chip-wide, measured): rows served from ...
ROOF_L2_GATHER_GBS = 17800.0     # ... an XCD's 4 MiB L2 (16.8-18.8 TB/s)
ROOF_MALL_GATHER_GBS = 8600.0    # ... the 256 MiB Infinity Cache, random rows of a 38 MB table (151 MB: 7.4-7.9 TB/s)
ROOF_HBM_GATHER_GBS = 5800.0     # ... HBM, random whole rows of a table far beyond the Infinity Cache, each fetched once
L2_PEAK_GBS = 34500.0            # MI355X_MICROARCH.md, "L2 (per XCD)": ~34.5 TB/s aggregate, the cache's own measured peak
L2_BYTES_PER_XCD = 4 << 20
MALL_BYTES = 256 << 20


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(n, argv, port, python=None, script=None):
    """The child command of `bench.py --gpus n <argv>`: one rank per GPU on this node, rendezvous on 127.0.0.1."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(sys.argv[0])] + list(argv)


def self_launch(n, argv, timeout=None):
    """Run the distributed bench as a child process; relay its JSON line on stdout, everything else on stderr.
    Returns the child's exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC (RCCL needs it on this driver)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = launch_command(n, argv, free_port())
    print("[bench] launching: %s" % " ".join(cmd), file=sys.stderr, flush=True)
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    try:
        out, _ = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        p.kill()
        out, _ = p.communicate()
        print("[bench] child timed out", file=sys.stderr, flush=True)
        return 124
    line = None
    for ln in out.splitlines():
        s = ln.strip()
        if s.startswith("{") and s.endswith("}"):
            try:
                json.loads(s)
                line = s
            except ValueError:
                pass
        elif s:
            print(s, file=sys.stderr)
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    elif p.returncode == 0:
        print("[bench] child printed no JSON line", file=sys.stderr, flush=True)
        return 1
    return p.returncode


def passes_per_step(t_pass_s, steps, target_s=0.6, cap=20000):
    """R such that the timed region (steps x R passes) lasts about target_s: a 3 ms region moves by several percent
    with one scheduler hiccup."""
    if t_pass_s <= 0:
        return 1
    return int(max(1, min(cap, math.ceil(target_s / (max(steps, 1) * t_pass_s)))))


def pass_time(run, sync, groups=3, n=5):
    """Seconds per call of run() for sizing the timed region: the FASTEST of `groups` groups of n calls -- the first calls on a
    fresh box (pages still faulting in, a graph's first uploads) can run an order of magnitude slow and would shrink the region."""
    best = float("inf")
    for _ in range(groups):
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            run()
        sync()
        best = min(best, (time.perf_counter() - t0) / n)
    return best


def timed_steps(run, steps, passes, sync, torch):
    """EXACTLY `steps` steps of `passes` calls of run() between sync()s.  Returns (seconds of the whole region, [ms per
    step] from one event per step on the current stream -- no host synchronisation inside the region)."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    sync()
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(steps):
        for _ in range(passes):
            run()
        ev[k + 1].record()
    sync()
    t = time.perf_counter() - t0
    return t, [ev[k].elapsed_time(ev[k + 1]) for k in range(steps)]


def median(xs):
    s = sorted(xs)
    n = len(s)
    return 0.0 if n == 0 else (s[n // 2] if n % 2 else 0.5 * (s[n // 2 - 1] + s[n // 2]))


def gather_regime(table_bytes, half_bytes=None):
    """Which memory level serves a random-row gather of a table of this size, and the ceiling that applies.
    half_bytes: the larger of the two halves a traversal half re-reads (user rows gather item rows only)."""
    hb = table_bytes if half_bytes is None else half_bytes
    if hb <= L2_BYTES_PER_XCD:
        return {"regime": "L2-resident", "roof_GBps": ROOF_L2_GATHER_GBS,
                "note": "the %.1f MB half a traversal half re-reads fits the 4 MiB L2 of every XCD" % (hb / 1e6)}
    if table_bytes <= MALL_BYTES // 2:
        return {"regime": "cache-resident (L2 + Infinity Cache)", "roof_GBps": ROOF_MALL_GATHER_GBS,
                "note": "%.1f MB table: beyond an XCD's 4 MiB L2, inside the 256 MiB Infinity Cache -- the HBM fraction of "
                        "such a launch is not an HBM statement (it may exceed 1); the ceiling is the Infinity-Cache row-gather "
                        "rate, partly lifted by L2 hits" % (table_bytes / 1e6)}
    return {"regime": "HBM", "roof_GBps": HBM_PEAK_GBS,
            "note": "%.2f GB table: far beyond the 256 MiB Infinity Cache; rows come from HBM (guide: random whole rows "
                    "fetched once read at 5.5-5.8 TB/s, a float4 copy at 6.3 TB/s, spec 8 TB/s)" % (table_bytes / 1e9)}


def copy_rate_GBps(torch, dev, gib=1, reps=10):
    """The box's own streaming rate: device-to-device copy of `gib` GiB (read + write counted)."""
    a = torch.empty(gib << 30, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return round(reps * 2 * (gib << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)


def usable_cpus():
    """CPUs this process may really use: the smallest of os.cpu_count(), the scheduler affinity and the cgroup CPU quota (a GPU
    box hands a one-GPU job a share of its host: 16 of 256 CPUs on the MI355X pool)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(math.ceil(int(txt[0]) / float(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(math.ceil(q / float(per)))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_model_name():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


_ROUNDS = ("r05", "r04", "r03", "r02", "r01")


def traffic_from_profile(root, workload, dtype, stages, d):
    """(bytes per launch, source) from the committed PMC passes of this command (tools/traffic.py: separate --pmc
    FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE x2 on gfx950), newest round first; (None, None) when there is none.
    The counters are NOT collected by this run -- the line says where they come from."""
    if stages != 3 or d != 64:
        return None, None
    for rnd in _ROUNDS:
        rel = os.path.join("profiles", "%s_traffic_%s.json" % (rnd, workload))
        p = os.path.join(root, rel)
        if os.path.exists(p):
            by = json.load(open(p)).get("by_dtype", {}).get(dtype)
            if by:
                return by.get("traffic_bytes"), "from_profile: %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, " \
                                                "FETCH_SIZE x2 per the gfx950 correction; not collected in this run)" % rel
    return None, None


def l2_hit_from_profile(root, workload, dtype):
    """(L2 hit rate of the forward traversal kernel, source): TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) of the prop_fwd_kernel
    launches of this bench command (tools/traffic.py's third pass), or -- older rounds -- of tools/reqcount.py's kbench passes;
    newest round first; (None, None) without one."""
    for rnd in _ROUNDS:
        rel = os.path.join("profiles", "%s_traffic_%s.json" % (rnd, workload))
        p = os.path.join(root, rel)
        if os.path.exists(p):
            by = json.load(open(p)).get("by_dtype", {}).get(dtype) or {}
            if by.get("l2_hit_rate") is not None:
                return by["l2_hit_rate"], "from_profile: %s (rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum pass of this command)" % rel
        rel = os.path.join("profiles", "%s_reqcount_%s_%s.json" % (rnd, workload, dtype))
        p = os.path.join(root, rel)
        if os.path.exists(p):
            ks = json.load(open(p)).get("kernels", {})
            rates = [v["per_directed_edge"]["L2_hit_rate"] for k, v in sorted(ks.items())
                     if k.startswith("prop_fwd_kernel") and ", true, false, false" in k and "L2_hit_rate" in v.get("per_directed_edge", {})]
            if rates:
                return round(sum(rates) / len(rates), 4), "from_profile: %s (rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum over tools/kbench.py)" % rel
    return None, None


def bwd_hit_from_profile(root, workload, dtype):
    """(L2 hit rate of the dense edge backward, source): tools/traffic.py trainstep -- the kernels of the replayed train step under
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum, per workload and dtype.  (Only files of the round that wrote the current edge
    backward count: the split-sign kernel gathers other tables than rounds 1-3's.)"""
    for rnd in _ROUNDS[:2]:                                  # (r04 wrote the split-sign backward; r05 kept its gathers)
        rel = os.path.join("profiles", "%s_traffic_trainstep%s.json" % (rnd, "" if workload == "gowalla" else "_" + workload))
        p = os.path.join(root, rel)
        if os.path.exists(p):
            k = (json.load(open(p)).get("by_dtype", {}).get(dtype) or {}).get("kernels", {}).get("prop_bwd_kernel dense")
            if k and k.get("l2_hit_rate") is not None:
                return k["l2_hit_rate"], "from_profile: %s" % rel
    return None, None


# ---------------------------------------------------------------------------------------------------------------------------
# The MEASURED roof of the traversal's access pattern (tools/gather_roof.py; include/ngacf.h NGACF_ROOF_ADD / NGACF_ROOF_XOR): the
# traversal kernel's own stripped instantiation over the SAME plan -- same work order, lane layout, index loads, row gathers, loop
# structure and stores, the arithmetic reduced to one xor per gathered dword ("xor": the memory system's ceiling for this pattern)
# or one add per gathered element ("add").  bench.py grades both storage types against it (one yardstick).
# ---------------------------------------------------------------------------------------------------------------------------
def _timed_us(torch, ops, fn, tag, n, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    with ops.KernelTimer() as kt:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    ms = kt.durations_ms(tag)
    if len(ms) != n:
        raise RuntimeError("gather roof: timed %d launches, expected %d" % (len(ms), n))
    return 1e3 * sum(ms) / n


def gather_roof_fwd_us(torch, ops, plan, U, I, dtype, heads, n=100):
    """{"xor": us, "add": us}: mean duration of the stripped forward traversal (d = 64) over `plan`."""
    dev, N = plan.device, U + I
    g = torch.Generator(device=dev).manual_seed(11)
    H = (torch.randn(N, 64, device=dev, generator=g) * 0.1).to(dtype)
    S = torch.randn(N, heads, device=dev, generator=g) * 0.1
    avec = torch.randn(2, 64, device=dev, generator=g) * 0.2
    out = torch.empty(N, 64, dtype=dtype, device=dev)
    T = ops.KernelTimer
    return {k: round(_timed_us(torch, ops, lambda: ops.prop_fwd(plan, H, S, S, avec, heads, out=out, roof=r), T.TAG_PROP_FWD, n), 2)
            for k, r in (("xor", 2), ("add", 1))}


def gather_roof_bwd_us(torch, ops, plan, U, I, dtype, heads, n=100):
    """The same for the dense edge backward (one Gn row + one (S, t) pair per head per edge)."""
    dev, N = plan.device, U + I
    g = torch.Generator(device=dev).manual_seed(12)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.1
    H, out, gout, Gn = (rnd(N, 64).to(dtype) for _ in range(4))
    agg = rnd(2, N, 64).to(dtype)
    S, R = rnd(N, heads), rnd(2, N, heads).abs() + 0.1
    avec = rnd(2, 64)
    ST, P1 = rnd(N, heads, 2), [rnd(N, heads) for _ in range(2)]
    k = [0]

    def run(r):
        k[0] += 1
        return ops.prop_bwd(plan, H, S, S, avec, out, agg, R, gout, heads, node=(Gn, ST, P1[k[0] & 1]), roof=r)
    T = ops.KernelTimer
    return {key: round(_timed_us(torch, ops, lambda: run(r), T.TAG_PROP_BWD, n), 2) for key, r in (("xor", 2), ("add", 1))}
