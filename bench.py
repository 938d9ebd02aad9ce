#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on MI355X: protein-ligand graphs/s for one full SINGA training step
(zero_grad -> forward -> CrossEntropy -> backward -> [grad all-reduce] -> clip(inf) -> Adam), plus the achieved
HBM GB/s of the fused alpha-scale + rotate-back + scatter kernel ("scatter-TP", k10) and of gather+rotate (k4)
against the roofline, plus the CPU oracle timed on this box's host cores.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3_b128_l4] [--scaling weak|strong] [--sweep]

Default workload = BASELINE.json configs[2]: 128 CrossDocked-shaped ragged synthetic graphs,
l_max = 4, full step, one GPU.  N > 1: one rank per GPU over RCCL.  Launched by the driver under torch.distributed.run
the ranks are taken from the environment; launched bare (`python bench.py --gpus 8`) it starts its own ranks as child
processes before touching the GPU and relays rank 0's JSON line.  `--scaling weak` (default): every rank trains on its
own 128 graphs; `--scaling strong` = BASELINE.json configs[3] (config 4): the SAME 128-graph batch is split over the
ranks by edge count (dp.shard_ranges_by_cost) and the gradients are combined token-weighted.
Prints ONE JSON line on rank 0.  Exit code 3 (after the line) if the HIP path's loss or total gradient norm on the
CPU-oracle sample is further than 1e-4 relative from the oracle's (north_star's tolerance).
`--sweep`: N = 1, 2, 4, 8 (as many as there are GPUs) back to back, every N a fresh child process, one combined line.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
ORACLE_TOL = 1e-4     # north_star: losses / gradients within 1e-4 relative of the reference's CPU path
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32 MFMA peak (MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3_b128_l4")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="N > 1 only.  Default strong = BASELINE.json configs[3] (the SAME batch split over the ranks); the "
                         "weak figure (every rank its own full batch) is then measured too and reported under 'weak'")
    ap.add_argument("--proxy-workload", default="cfg4_shard_r0of8",
                    help="N = 1, default workload only: also time the shard one rank of the 8-GPU strong split owns")
    ap.add_argument("--proxy-steps", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prefetch-priority", choices=("high", "normal"), default="normal",
                    help="priority of the stream the next batch is prepared on (TrainStep.prefetch_priority)")
    ap.add_argument("--allreduce-overlap", action="store_true",
                    help="several ranks: two-phase backward in the main timed region - the transformer's buckets travel while the "
                         "embedding's backward pass computes.  Default: one all-reduce after the backward pass; the overlapped order is "
                         "then measured AND checked against it (same parameters after the same steps) in a short second region and "
                         "reported under 'allreduce_overlap'")
    ap.add_argument("--no-overlap-check", action="store_true", help="N > 1: skip the 'allreduce_overlap' comparison region")
    ap.add_argument("--sweep", action="store_true", help="run N = 1, 2, 4, 8 (<= visible GPUs) as fresh child processes, one combined line")
    ap.add_argument("--other-workloads", default="cfg2_b32_l2,cfg5_l6",
                    help="N = 1, default workload only: further BASELINE configs timed with the same engine after the main region "
                         "(value, median step, k10 roofline fraction, lap_pe_ms, peak HBM) and reported under 'other_workloads'; '' = none")
    ap.add_argument("--other-steps", type=int, default=6)
    ap.add_argument("--ingraph-steps", type=int, default=3,
                    help="replays of a second, event-instrumented capture of the step (external event-record nodes around the k4 / "
                         "k10 / S2 dispatches) from which roofline.avg_launch_us is taken; 0 = eager dispatch timing only")
    ap.add_argument("--eager", action="store_true", help="run the step eagerly instead of replaying HIP graphs")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="build each batch's graph structure inside its own step instead of on a second stream during the previous one")
    ap.add_argument("--distinct-batches", type=int, default=3,
                    help="different resident batches cycled through the timed steps (ragged sizes -> padded / bucketed replay; "
                         "1 = the same batch every step)")
    ap.add_argument("--growth", type=float, default=1.04, help="width of a size class of the bucketed replay")
    ap.add_argument("--lap-pe-resident", action="store_true",
                    help="keep the Laplacian encodings made at generation time instead of recomputing them inside every step")
    ap.add_argument("--roofline-steps", type=int, default=2, help="instrumented eager steps after the timed region")
    ap.add_argument("--cpu-graphs", type=int, default=8, help="graphs in the bounded CPU-oracle sample")
    ap.add_argument("--cpu-graphs-1t", type=int, default=2, help="graphs of the one-thread CPU-oracle figure (0 = skip)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--ingraph-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-state", type=str, default=None, help=argparse.SUPPRESS)
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ algorithmic bytes
def kernel_bytes(tag, E, N, L, C=16, CH=112, heads=7):
    """Algorithmic HBM bytes of one launch (SURVEY.md §8d formulas; the 36-byte edge frame of the survey's formula is
    replaced by the WSZ*4-byte reduced Wigner rows this design streams, as DESIGN.md states).  E = edges of the launch,
    N = node rows written / read by it (k4: N_src + N_dst, here both = the pass's node count)."""
    from singa_amd import so3
    lay = so3.layout(L, 2)
    K, KR, WSZ, R = lay.K, lay.KR, lay.WSZ, lay.rad_rows * 2 * C
    if tag == "k10_fwd":      # reads msg + alpha + Wigner rows + row_ptr, writes the node rows
        return E * (KR * CH * 4 + heads * 4 + WSZ * 4) + N * K * CH * 4 + (N + 1) * 4
    if tag == "k10_bwd":      # reads node-row gradients, msg, alpha, Wigner rows; writes d msg and d alpha
        return N * K * CH * 4 + (N + 1) * 4 + E * (KR * CH * 4 + heads * 4 + WSZ * 4) + E * (KR * CH * 4 + heads * 4)
    if tag == "k4_fwd":       # reads both endpoint rows once per node, radial weights, Wigner rows, indices; writes [E, KR, 2C]
        return 2 * N * K * C * 4 + E * (KR * 2 * C * 4 + R * 4 + WSZ * 4 + 8)
    if tag == "k4_bwd_rad":   # reads the same as the forward plus d out; writes d rad
        return 2 * N * K * C * 4 + E * (KR * 2 * C * 4 + WSZ * 4 + 8) + E * R * 4
    if tag in ("k4_bwd_dst", "k4_bwd_src"):   # reads one half of d out and of rad, Wigner rows, the CSR; writes N node rows
        return E * (KR * C * 4 + R // 2 * 4 + WSZ * 4 + (4 if tag == "k4_bwd_src" else 0)) + N * K * C * 4 + (N + 1) * 4
    # k8, separable S2 activation.  Attention grid: E edge rows of KR coefficients x 128 channels (+ the 128 gate
    # scalars); feed-forward grid: E = node rows of K coefficients x 512 channels (+ 512 gate scalars)
    if tag == "s2_edge_fwd":
        return E * ((KR + 1) * 128 * 4 + KR * 128 * 4)
    if tag == "s2_edge_bwd":  # reads the input, the gate and d out; writes d input and d gate
        return E * ((KR + 1) * 128 * 4 + KR * 128 * 4 + (KR + 1) * 128 * 4)
    if tag == "s2_node_fwd":
        return E * ((K + 1) * 512 * 4 + K * 512 * 4)
    if tag == "s2_node_bwd":  # the fused form (singa_s2act_ffn_bwd): reads the input, the gate and the SMALL gradient [K, 16]; writes d input, d gate
        return E * ((K + 1) * 512 * 4 + K * 16 * 4 + (K + 1) * 512 * 4)
    raise KeyError(tag)


def lib_source_sha():
    h = hashlib.sha1()
    for f in ("singa_amd/csrc/singa_hip.hip", "singa_amd/csrc/so3_index.h"):
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def committed_profile(workload):
    """The newest profiles/<tag>/ directory that was made from THIS workload (its bench_*.json names it): PMC traffic
    per launch and MFMA counters of the same command, produced by tools/prof.sh.  rocprofv3 --pmc cannot run inside this
    process, so these numbers are read back; `lib_sha` tells whether the profiled library source is the current one."""
    import glob
    best = None
    for d in sorted(glob.glob(os.path.join(ROOT, "profiles", "*"))):
        meta = os.path.join(d, "meta.json")
        if os.path.isfile(meta):
            try:
                m = json.load(open(meta))
            except Exception:
                continue
            if m.get("workload") == workload:
                best = (d, m)
    return best


def pmc_traffic(prof, kernel_substr, grid):
    """HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes: FETCH_SIZE x 2 (gfx950 correction for this access
    shape, confirmed by the calibration copy of the same pass) + WRITE_SIZE, KiB -> bytes."""
    import csv
    if prof is None:
        return None
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = os.path.join(prof[0], f"pmc_{c}.csv")
        if not os.path.exists(f):
            return None
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["kernel"] and int(r["grid"]) == grid:
                vals[c] = float(r["avg_value"])
    if len(vals) != 2:
        return None
    return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)


def baseline_metric():
    """The metric string of BASELINE.json, verbatim.  (It says "GAN step (G+D fwd+bwd)"; the reference defines no
    discriminator - SURVEY.md F2 - so the step measured is the whole training step the reference runs: the generator's
    forward + CrossEntropy + backward + clip + Adam, train.py:113-133; `config.step` spells it out.)"""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "protein-ligand graphs/sec GAN step (G+D fwd+bwd); scatter-TP kernel HBM GB/s"


def host_cores():
    """CPU threads this process may really use: min(affinity, cgroup quota), capped at the one-GPU box share."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


# ------------------------------------------------------------------------------------------------ CPU baseline (oracle)
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline_worker(workload, n_graphs, threads, state=None, n_graphs_1t=2):
    """Runs in a child process that never touches the GPU: times the CPU oracle (kind 'port': oracle/singa_oracle.py,
    pinned to the reference by tests/golden) on a bounded sample of the same workload - the whole step of train.py:113-133
    (zero_grad, forward, CrossEntropy, backward, gradient norm, Adam) on the first `n_graphs` synthetic graphs with all
    host threads, then on the first `n_graphs_1t` graphs with ONE thread - and prints one JSON object."""
    torch.set_num_threads(threads)
    from oracle import singa_oracle as O
    from singa_amd import graph as G
    from singa_amd.config import load_config
    from singa_amd.model.GAN import SINGA
    L, kw, ids, _ = G.resolve_workload(workload)
    cfg = load_config(lmax=L)
    torch.manual_seed(cfg.train.seed)
    model = SINGA(cfg, device="cpu")          # parameter container only; the product forward is never called here
    if state:                                 # the GPU run's initial parameters (drawn with the device generator there)
        model.load_state_dict(torch.load(state, map_location="cpu"))
    graphs = [G.synthetic_graph(i, **G.graph_sizes(i, **kw)) for i in ids[:n_graphs]]
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    leaves = [v for v in sd.values() if v.requires_grad]
    adam = torch.optim.Adam(leaves, lr=cfg.train.optimizer.lr, betas=(cfg.train.optimizer.beta1, cfg.train.optimizer.beta2))

    def step(inputs, update):
        b, rots, lap_p, lap_l = inputs
        adam.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        loss = O.train_step_loss(sd, b, rots, L, lap_p, lap_l)
        loss.backward()
        gn = torch.sqrt(sum((v.grad.double() ** 2).sum() for v in leaves if v.grad is not None))
        if update:
            adam.step()
        return time.perf_counter() - t0, float(loss.detach()), float(gn)

    full = O.batch_from_graphs(graphs)
    lap_file = os.path.join(os.path.dirname(state), "sample_lap.pt") if state else None
    if lap_file and os.path.exists(lap_file):
        # the SAME encoding tensors the HIP path was given (an eigenvector basis of a repeated eigenvalue is not unique, and
        # LAPACK's choice depends on the process's BLAS thread count)
        lp = torch.load(lap_file)
        full = (full[0], full[1], lp["lap_p"], lp["lap_l"])
    # first pass warms the table caches and reports loss / gradient norm at the INITIAL parameters (no update); the second,
    # complete with its Adam update, is the timed one
    t_warm, loss0, gn0 = step(full, update=False)
    print(f"[cpu-baseline] warm pass: {t_warm:.2f} s", file=sys.stderr, flush=True)
    t_full, _, _ = step(full, update=True)
    print(f"[cpu-baseline] timed pass, {threads} threads: {t_full:.2f} s", file=sys.stderr, flush=True)
    out = {"loss": loss0, "grad_norm": gn0, "value": round(n_graphs / t_full, 4), "unit": "graphs/s", "cores": threads,
           "kind": "port", "cpu": cpu_model(),
           "sample": f"first {n_graphs} graphs of {workload}: oracle zero_grad+forward+CE+backward+grad-norm+Adam, "
                     f"{threads} torch threads, {t_full:.1f} s"}
    if n_graphs_1t > 0:
        torch.set_num_threads(1)
        small = O.batch_from_graphs(graphs[:n_graphs_1t])             # (timing only: its own encodings)
        t_1, _, _ = step(small, update=True)
        print(f"[cpu-baseline] timed pass, 1 thread: {t_1:.2f} s", file=sys.stderr, flush=True)
        out["one_thread"] = {"value": round(n_graphs_1t / t_1, 4), "unit": "graphs/s", "cores": 1,
                             "sample": f"first {n_graphs_1t} graphs, same step, torch.set_num_threads(1), {t_1:.1f} s"}
    print(json.dumps(out), flush=True)


def cpu_baseline(workload, n_graphs, limit_s=400, state=None, n_graphs_1t=2):
    threads = host_cores()
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads),
               MKL_NUM_THREADS=str(threads))
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--workload", workload, "--cpu-graphs",
           str(n_graphs), "--cpu-graphs-1t", str(n_graphs_1t), "--cpu-threads", str(threads)] + (["--cpu-state", state] if state else [])
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, timeout=limit_s, check=True)
        return json.loads(r.stdout.decode().strip().splitlines()[-1])
    except Exception as e:  # a baseline that cannot be timed must not take the GPU measurement down with it
        return {"value": None, "unit": "graphs/s", "cores": threads, "kind": "port", "cpu": cpu_model(),
                "sample": f"not measured: {type(e).__name__}"}


# ------------------------------------------------------------------------------------------------ launching N ranks
def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a CHILD process (this
    process has not touched the GPU and never will) and pass its output through."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def rocprof_in_graph_us(prof, kernel_substr, grid):
    """Average duration of a kernel (by name + grid) in the committed rocprofv3 kernel trace of this command
    (profiles/<tag>/singa_kernels_by_grid.csv: the replayed steps of the timed region dominate its launches)."""
    import csv
    if prof is None:
        return None
    f = os.path.join(prof[0], "singa_kernels_by_grid.csv")
    if not os.path.exists(f):
        return None
    for r in csv.DictReader(open(f)):
        if kernel_substr in r["kernel"] and int(r["grid_x"]) == grid:
            return float(r["avg_us"])
    return None


def copy_ceiling_gbs(dev, n=64 * 1024 * 1024, reps=10, wide=True):
    """Measured device-copy bandwidth (read + write bytes / time) of the library's own copy kernels on 256 MB operands.
    wide: singa_calib_copy16 (16 bytes per lane) over a small sweep of launch shapes (workgroups per CU x loads in flight
    per lane) - the best is the practical HBM ceiling of this box next to the 8 TB/s spec peak; returns (GB/s, shape).
    Otherwise singa_calib_copy (one dword per lane, the segment kernels' access shape)."""
    import ctypes
    from singa_amd import _lib
    a = torch.randn(n, device=dev)
    b = torch.empty_like(a)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib = _lib.lib()
    pa, pb = ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr())

    def rate(fn):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return round(2.0 * 4 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)

    if not wide:
        return rate(lambda: lib.singa_calib_copy(pa, pb, n, st))
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    best = (0.0, None)
    for per_cu in (4, 8, 16, 32):
        for unroll in (1, 2, 4):
            r = rate(lambda: lib.singa_calib_copy16(pa, pb, n, cus * per_cu, unroll, st))
            if r > best[0]:
                best = (r, f"{per_cu} workgroups per CU x {unroll} loads in flight per lane")
    return best


HOST_SUBMIT_MS = [None]     # median host time of engine.step() in the last timed_region
HOST_CPU_MS = [None]        # CPU time (thread_time) of the timed loop per step
HOST_PREFETCH_MS = [None]   # ... and of engine.prefetch() (the next batch's preparation, issued on the second stream)


def timed_region(engine, batches, steps, multi, dev, prefetch=True):
    """EXACTLY `steps` steps bracketed by barrier + synchronize on both sides -> (elapsed seconds = MAX over ranks, per-step
    milliseconds from events on the compute stream, MAX over ranks per step, last loss, this rank's own per-step
    milliseconds)."""
    nb = len(batches)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    nxt = batches[0]
    loss = None
    host, host_pf = [], []
    # diagnostic (SINGA_BENCH_REPLAY_ONLY=1, NOT a valid bench line): the SAME staged batch every step, no preparation of a
    # next one - what the replay alone costs when the host has nothing else to do between two submissions
    replay_only = os.environ.get("SINGA_BENCH_REPLAY_ONLY") == "1"
    if replay_only and prefetch:
        nxt = engine.prefetch(nxt)
        torch.cuda.synchronize()
        marks[0].record()
    cpu0 = time.thread_time()
    for i in range(steps):
        th = time.perf_counter()
        loss = engine.step(nxt)
        host.append(1e3 * (time.perf_counter() - th))
        marks[i + 1].record()
        if replay_only:
            continue
        nxt = batches[(i + 1) % nb]
        if prefetch:
            th = time.perf_counter()
            nxt = engine.prefetch(nxt)          # (bucket mode: returns the padded batch the next step replays)
            host_pf.append(1e3 * (time.perf_counter() - th))
    cpu1 = time.thread_time()
    torch.cuda.synchronize()
    HOST_PREFETCH_MS[:] = [median(host_pf) if host_pf else 0.0]
    # CPU time this thread actually burned per step (not waiting): close to the step time = the loop is HOST-bound
    HOST_CPU_MS[:] = [1e3 * (cpu1 - cpu0) / max(1, steps)]
    # host time of the step() calls themselves (no synchronisation inside: for a replayed step this is what the runtime needs to
    # SUBMIT the captured graph's ~1,500 kernel nodes) - when it approaches the step time the GPU is waiting for the host
    HOST_SUBMIT_MS[:] = [median(host)]
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    own = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    per = torch.tensor(own, device=dev, dtype=torch.float64)
    if multi:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(per, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    return elapsed, per.tolist(), loss, own


def median(xs):
    xs = sorted(xs)
    n = len(xs)
    return 0.5 * (xs[(n - 1) // 2] + xs[n // 2]) if n else None


def sweep(args):
    """N = 1, 2, 4, 8 (<= visible GPUs) back to back, every N a fresh child process started before this process touches the
    GPU (it never does).  N = 1 runs with SINGA_RCCL_SELFTEST=1, i.e. through the same process-group code path as N > 1 (a
    one-rank RCCL communicator), so that the points of the curve are measured by one code path."""
    n_dev = torch.cuda.device_count()                 # (counting devices does not initialise the GPU)
    skip_next, clean = False, []
    for a in sys.argv[1:]:
        if skip_next:
            skip_next = False
            continue
        if a == "--sweep":
            continue
        if a == "--gpus":
            skip_next = True
            continue
        if a.startswith("--gpus="):
            continue
        clean.append(a)
    points, rc = [], 0
    for n in (1, 2, 4, 8):
        if n > max(n_dev, 1):
            break
        env = dict(os.environ)
        if n == 1:
            env["SINGA_RCCL_SELFTEST"] = "1"
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(n)] + clean + (["--no-cpu-baseline"] if n > 1 else [])
        print(f"[bench sweep] N = {n}: {' '.join(cmd)}", file=sys.stderr, flush=True)
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
        rc = rc or r.returncode
        line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
        points.append(json.loads(line[-1]) if line else {"n_gpus": n, "error": f"no result line (rc {r.returncode})"})
    base = next((p for p in points if p.get("n_gpus") == 1 and "value" in p), None)
    out = {"metric": baseline_metric(), "sweep": [{k: p.get(k) for k in ("n_gpus", "value", "unit", "ms_per_step", "ms_per_step_median",
                                                                          "scaling", "rccl_ranks", "per_rank", "allreduce_exposed_ms",
                                                                          "allreduce_overlap", "weak", "error") if k in p}
                                                   for p in points],
           "speedup_vs_n1": {str(p["n_gpus"]): round(p["value"] / base["value"], 3) for p in points if base and "value" in p},
           "note": "every N is a fresh child process; N = 1 on a one-rank RCCL communicator (the N > 1 code path)", "lines": points}
    print(json.dumps(out), flush=True)
    return rc


class Run:
    """What every workload of one bench process shares: ranks, device, flags, logging."""

    def __init__(self, args, world, rank, local, dev, multi, selftest, scaling):
        self.args, self.world, self.rank, self.local, self.dev = args, world, rank, local, dev
        self.multi, self.selftest, self.scaling = multi, selftest, scaling

    def log(self, msg):
        if self.rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def gather_ranks(R, values):
    """[world][len(values)] floats: every rank's `values` (all_gather on the device)."""
    t = torch.tensor([float(v) for v in values], device=R.dev, dtype=torch.float64)
    if not R.multi:
        return [t.tolist()]
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]


def roofline_rows(recs, L, tags_k=("k10_fwd", "k10_bwd", "k4_fwd", "k4_bwd_rad", "k4_bwd_dst", "k4_bwd_src"),
                  tags_s2=("s2_edge_fwd", "s2_edge_bwd", "s2_node_fwd", "s2_node_bwd")):
    """Per-kernel roofline rows from (tag, ms, E, N) dispatch records: the bonded-edge union pass (the launches with the
    most edges) for k4 / k10, the largest launch of each kind for the S2 activation."""
    per = {}
    recs = [r for r in recs if r[1] >= 0]
    if not recs:
        return per, 0, 0
    k10 = [r for r in recs if r[0] == "k10_fwd"]
    big = max((r[2] for r in (k10 or recs)), default=0)
    n_union = max((r[3] for r in recs if r[2] == big and r[0] == "k10_fwd"), default=0)
    for tag in tags_k:
        sel = [r for r in recs if r[0] == tag and r[2] == big]
        if not sel:
            continue
        us = 1e3 * sum(r[1] for r in sel) / len(sel)
        by = kernel_bytes(tag, big, n_union, L)
        ach = by / (us * 1e-6) / 1e9
        per[tag] = {"achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "avg_launch_us": round(us, 2),
                    "bytes_per_launch": by, "launches": len(sel)}
    for tag in tags_s2:
        rows = max((r[2] for r in recs if r[0] == tag), default=0)
        sel = [r for r in recs if r[0] == tag and r[2] == rows]
        if not sel:
            continue
        us = 1e3 * sum(r[1] for r in sel) / len(sel)
        by = kernel_bytes(tag, rows, 0, L)
        ach = by / (us * 1e-6) / 1e9
        per[tag] = {"achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "avg_launch_us": round(us, 2),
                    "bytes_per_launch": by, "launches": len(sel), "rows": rows}
    return per, big, n_union


def run_workload(R, workload, steps, warmup, main=True):
    """One workload through the step engine: generation, warm-up / capture, the timed region, and the measurements around
    it.  main: the workload `value` is quoted on (strong proxy, eager roofline pass, N > 1 extras); otherwise the lighter
    set reported under `other_workloads`.  Returns a dict of results; the model / engine are dropped on return."""
    import copy
    import gc
    from singa_amd import dp, graph as G, ops
    from singa_amd.config import load_config
    from singa_amd.engine import TrainStep
    from singa_amd.model import EF_layers
    from singa_amd.model.GAN import SINGA
    from singa_amd.optim import Adam

    args, world, rank, dev, multi, selftest, scaling, log = R.args, R.world, R.rank, R.dev, R.multi, R.selftest, R.scaling, R.log
    L, kw, base_ids, n_parent = G.resolve_workload(workload)
    n_graphs = len(base_ids)
    cfg = load_config(lmax=L)
    torch.cuda.reset_peak_memory_stats(dev)
    torch.manual_seed(cfg.train.seed)                     # same-seed init on every rank (no broadcast)
    model = SINGA(cfg, device=dev)
    model.train()
    reducer = dp.GradAllReducer(model, always=selftest, phases=bool(args.allreduce_overlap))
    reducer.check_same_init()
    use_graph = not args.eager
    opt = Adam(model.parameters(), lr=cfg.train.optimizer.lr, betas=(cfg.train.optimizer.beta1, cfg.train.optimizer.beta2))
    D = max(1, args.distinct_batches)
    bucket_mode = (not args.eager) and D > 1
    lap_in_step = bucket_mode and not args.lap_pe_resident

    def make_batches(ids, stride, count):
        """`count` different batches of the graphs `ids` (batch k: ids + k * stride), resident in HBM.  Their Laplacian
        positional encodings are recomputed by every step that takes them (`lap_pe_in_step`: the reference runs dgl.lap_pe
        inside forward, GAN.py:71,77) - with the library's eigensolver, in the step's preparation phase; such batches are
        generated without encodings (numpy's dense eigensolver is 90 % of the generation time)."""
        out = [G.synthetic_batch(len(ids), ids=[i + k * stride for i in ids], with_lap=not lap_in_step, **kw).to(dev)
               for k in range(count)]
        if lap_in_step:
            for b in out:
                b.extras["lap_pe_in_step"] = True
        return out

    # ---- this rank's graphs (resident in HBM before the timed region starts)
    if scaling == "strong" and world > 1:
        costs = [G.graph_cost(G.graph_sizes(i, **kw)) for i in base_ids]
        lo, hi = dp.shard_ranges_by_cost(costs, world)[rank]
        ids = base_ids[lo:hi]
        reducer.set_shard_weight(len(ids), n_graphs)
        graphs_per_step = n_graphs
        stride = n_parent
    else:
        ids = [i + rank * n_parent for i in base_ids]
        graphs_per_step = n_graphs * world
        stride = n_parent * world
    t_gen = time.perf_counter()
    # D different batches of this rank's graphs, resident in HBM, cycled through the steps.  Their atom / edge counts differ
    # (config 3 is ragged), so the replayed step pads each batch to its size class (TrainStep bucket mode).  D = 1: the
    # same batch every step (two resident copies, so that batch i+1 can be prepared while step i computes).
    batches = make_batches(ids, stride, D)
    batch = batches[0]
    if D == 1 and not args.no_prefetch:
        batches.append(copy.deepcopy(batch))
    bucket = use_graph and D > 1
    engine = TrainStep(model, opt, reducer if multi else None, use_graph=use_graph,
                       max_grad_norm=float(cfg.train.max_grad_norm), bucket=bucket, growth=args.growth, max_cached=6)
    engine.prefetch_priority = -1 if args.prefetch_priority == "high" else 0
    # the run's initial parameters, for the CPU oracle (and for the HIP path's own loss / gradient norm on the CPU sample,
    # computed at the very end of the run)
    state_file = None
    if main and rank == 0 and world == 1 and not args.no_cpu_baseline:
        import tempfile
        state_file = os.path.join(tempfile.mkdtemp(prefix="singa_bench_"), "init_state.pt")
        torch.save({k: v.cpu() for k, v in model.state_dict().items()}, state_file)
    n_nodes = batch[G.PA]["x"].shape[0] + batch[G.LA]["x"].shape[0]
    n_edges = sum(int(batch[et]["edge_index"].shape[1]) for et in (G.E_PP, G.E_LL, G.E_LP, G.E_PL))
    gen_s = time.perf_counter() - t_gen
    log(f"workload {workload}: {len(ids)} graphs on this GPU ({n_nodes} atoms, {n_edges} edges), L={L}, generated in "
        f"{gen_s:.1f} s; {'HIP-graph replay' if use_graph else 'eager'} step")

    def warm(bs, n):
        for i in range(max(n, len(bs) if bucket else 0)):      # (bucket mode: every batch once, so that all captures exist)
            t_w = time.perf_counter()
            engine.step(bs[i % len(bs)])
            torch.cuda.synchronize()
            log(f"warmup step {i}: {time.perf_counter() - t_w:.3f} s")

    warm(batches, warmup)
    captures_before = engine.captures
    if multi:
        engine.comm_events = []
    # ---- timed region: exactly K steps, barrier + synchronize on both sides, MAX over ranks.  Every step handles its
    # batch as newly arrived: the graph structure (edge sorting, kNN graphs, dense maps) is rebuilt K times inside the
    # region - by default on a second stream while the previous step computes (TrainStep.prefetch), the way a loader
    # thread would; with --no-prefetch at the start of the step itself.
    elapsed, per_step, loss, own_ms = timed_region(engine, batches, steps, multi, dev, not args.no_prefetch)
    final_loss = float(loss.detach())
    host_submit_ms, host_prefetch_ms, host_cpu_ms = HOST_SUBMIT_MS[0], HOST_PREFETCH_MS[0], HOST_CPU_MS[0]
    captures_timed = engine.captures - captures_before
    exposed = None
    if engine.comm_events:
        exposed = median([a.elapsed_time(b) for a, b in engine.comm_events])
    engine.comm_events = None
    log(f"timed {steps} steps in {elapsed:.3f} s, median step {median(per_step):.3f} ms ({engine.captures} graph capture(s) so far)")
    # how much of a step is per-batch graph preparation (edge sorting, kNN graphs, dense maps) - measured
    # separately, it is already inside every timed step
    torch.cuda.synchronize()
    t_p = time.perf_counter()
    for _ in range(3):
        EF_layers._edge_cache.clear()
        batch.extras.pop("prepared", None)
        model.prepare(batch)
    torch.cuda.synchronize()
    prepare_ms = (time.perf_counter() - t_p) / 3 * 1e3
    # the Laplacian positional encoding of one batch (reference: dgl.lap_pe inside forward, GAN:71,77), timed
    # on its own (it is inside every timed step already)
    torch.cuda.synchronize()
    t_p = time.perf_counter()
    for _ in range(3):
        for nt, et in ((G.PA, G.E_PP), (G.LA, G.E_LL)):
            G.laplacian_pe_batched(batch[et]["edge_index"], batch[nt]["batch"], batch.num_graphs, cfg.model.encoder.lap_dim)
    torch.cuda.synchronize()
    lap_pe_ms = (time.perf_counter() - t_p) / 3 * 1e3

    res = {"workload": workload, "L": L, "kw": kw, "ids": ids, "base_ids": base_ids, "n_graphs": n_graphs, "n_nodes": n_nodes,
           "n_edges": n_edges, "graphs_per_step": graphs_per_step, "elapsed": elapsed, "per_step": per_step, "own_ms": own_ms,
           "final_loss": final_loss, "captures": engine.captures, "captures_timed": captures_timed, "prepare_ms": prepare_ms,
           "lap_pe_ms": lap_pe_ms, "use_graph": use_graph, "bucket": bucket, "D": D, "gen_s": gen_s, "state_file": state_file,
           "two_phase": bool(getattr(engine, "two_phase", False)), "exposed_ms": exposed, "lap_in_step": lap_in_step,
           "weak": None, "proxy": None, "roof": None, "overlap": None, "per_rank": None, "rccl_ranks": None,
           "host_submit_ms": host_submit_ms, "host_prefetch_ms": host_prefetch_ms,
           "host_cpu_ms": host_cpu_ms}

    # ---- N > 1: who ran where (the first thing to look at when a scaling point looks wrong)
    if multi:
        props = torch.cuda.get_device_properties(dev)
        rows = gather_ranks(R, [rank, torch.cuda.current_device(), len(ids), n_nodes, n_edges, median(own_ms), min(own_ms), max(own_ms),
                                exposed if exposed is not None else -1.0])
        res["per_rank"] = [{"rank": int(r[0]), "device": int(r[1]), "graphs": int(r[2]), "atoms": int(r[3]), "edges": int(r[4]),
                            "step_ms_median": round(r[5], 3), "step_ms_min": round(r[6], 3), "step_ms_max": round(r[7], 3),
                            "allreduce_exposed_ms": (round(r[8], 3) if r[8] >= 0 else None)} for r in rows]
        res["rccl_ranks"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                             "device_ordinals": [int(r[1]) for r in rows], "device_name": props.name,
                             "distinct_devices": len({int(r[1]) for r in rows})}
    res["reducer"] = {"payload_bytes": reducer.payload_bytes,
                      "phase0_bytes": (sum(f.numel() * 4 for f, k in zip(reducer.flat, reducer.bucket_phase) if k == 0)
                                       if reducer.buckets is not None else 0)}

    # ---- N > 1: the overlapped (two-phase) order next to the default one.  Same initial state, same batches, dropout off,
    # the same number of steps in both orders: step time, exposed all-reduce time, and the parameters afterwards must agree.
    if main and world > 1 and not args.no_overlap_check and use_graph:
        res["overlap"] = overlap_check(R, model, opt, batches, ids, n_graphs, scaling, cfg, bucket, log)

    # ---- N > 1, strong split: the weak figure too (every rank its own full batch), same engine, its own captures
    if main and world > 1 and scaling == "strong" and args.scaling is None:
        reducer.clear_shard_weight()
        w_ids = [i + rank * n_parent for i in base_ids]
        w_batches = make_batches(w_ids, n_parent * world, 2)
        warm(w_batches, 2)
        w_steps = max(4, steps // 2)
        w_el, w_per, _, _ = timed_region(engine, w_batches, w_steps, multi, dev, not args.no_prefetch)
        wv = n_graphs * world * w_steps / w_el
        res["weak"] = {"value": round(wv, 3), "unit": "graphs/s", "global_batch": n_graphs * world, "steps": w_steps,
                       "ms_per_step": round(1e3 * w_el / w_steps, 3), "ms_per_step_median": round(median(w_per), 3),
                       "scaling": "weak", "value_per_gpu": round(wv / world, 3),
                       "value_per_gpu_note": "every rank trains on a full 128-graph batch through the process-group code path: "
                                             "compare with the N = 1 line's value"}
        del w_batches

    # ---- N = 1, default workload: the shard ONE rank of the 8-GPU strong split (config 4) would own, timed on this GPU
    if main and world == 1 and not multi and use_graph and args.proxy_steps > 0 and args.proxy_workload in G.WORKLOADS \
            and G.WORKLOADS[args.proxy_workload].get("parent") == workload:
        _, _, p_ids, _ = G.resolve_workload(args.proxy_workload)
        p_batches = make_batches(p_ids, n_parent, D)
        warm(p_batches, 3)
        p_el, p_per, _, _ = timed_region(engine, p_batches, args.proxy_steps, False, dev, not args.no_prefetch)
        pw = G.WORKLOADS[args.proxy_workload]["shard"][1]
        res["proxy"] = {"workload": args.proxy_workload, "graphs": len(p_ids), "ranks_of_split": pw, "steps": args.proxy_steps,
                        "ms_per_step": round(1e3 * p_el / args.proxy_steps, 3), "ms_per_step_median": round(median(p_per), 3),
                        "host_submit_ms_per_step": round(HOST_SUBMIT_MS[0], 3),
                        "host_prefetch_ms_per_step": round(HOST_PREFETCH_MS[0], 3),
                        "host_cpu_ms_per_step": round(HOST_CPU_MS[0], 3),
                        "implied_speedup_at_8": round(median(per_step) / median(p_per), 2),
                        "note": "1-GPU step time of rank 0's cost-balanced shard of the same batch (no all-reduce: add the RCCL "
                                "ring of grad_allreduce_bytes, ~1.2 ms over xGMI)"}
        log(f"strong proxy: {len(p_ids)} graphs, median step {res['proxy']['ms_per_step_median']} ms")
        del p_batches

    # ---- instrumented eager pass: the same step run eagerly with start/stop events attached to every tagged dispatch
    if main and args.roofline_steps > 0:
        if use_graph:
            engine.release()              # give the graph pool back before the eager instrumented steps
        engine.eager_step(batch)          # one un-instrumented eager step first (allocator warm-up)
        ops.profile_start()
        for _ in range(args.roofline_steps):
            engine.eager_step(batch)
        torch.cuda.synchronize()
        recs = ops.profile_collect()
        if os.environ.get("SINGA_CALIB") == "1" and rank == 0:
            # known-byte launches for calibrating FETCH_SIZE / WRITE_SIZE under `rocprofv3 --pmc`
            import ctypes
            from singa_amd import _lib
            n = 64 * 1024 * 1024
            a = torch.randn(n, device=dev)
            b = torch.empty_like(a)
            for _ in range(3):
                _lib.lib().singa_calib_copy(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), n,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            del a, b
        per, big, n_union = roofline_rows(recs, L)
        if per:
            log("bonded-edge launches, eager dispatches (us): " + "; ".join(f"{k} {v['avg_launch_us']}" for k, v in per.items()))
        res["eager_rows"] = (per, big, n_union)

    res["peak_hbm_gb"] = round(torch.cuda.max_memory_allocated(dev) / 1e9, 2)
    res["peak_hbm_reserved_gb"] = round(torch.cuda.max_memory_reserved(dev) / 1e9, 2)
    free, total = torch.cuda.mem_get_info(dev)
    res["hbm_total_gb"] = round(total / 1e9, 1)

    # ---- the HIP path's loss and total gradient norm on the CPU-baseline sample (the first graphs of the workload, the run's
    # INITIAL parameters, dropout off): compared with what the oracle computes while it is being timed.  Both sides get the
    # SAME input tensors: the sample's Laplacian encodings are computed once, here, and handed to the oracle's process (an
    # eigenvector basis of a repeated eigenvalue is not unique - round 3's "8-graph discrepancy" was two processes with
    # different BLAS thread counts computing different, equally valid encodings)
    if state_file is not None:
        if use_graph:
            engine.release()
        model.load_state_dict(torch.load(state_file, map_location=dev))
        sample_cpu = G.synthetic_batch(args.cpu_graphs, ids=base_ids[:args.cpu_graphs], **kw)
        torch.save({"lap_p": sample_cpu[G.PA]["lap_pe"].clone(), "lap_l": sample_cpu[G.LA]["lap_pe"].clone()},
                   os.path.join(os.path.dirname(state_file), "sample_lap.pt"))
        sample = sample_cpu.to(dev)
        model.eval()
        EF_layers._edge_cache.clear()
        model.prepare(sample)
        model.zero_grad(set_to_none=True)
        ls = torch.nn.functional.cross_entropy(model(sample), sample["ligand_data"]["smiIndices_tgt"].reshape(-1))
        ls.backward()
        gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None))
        res["hip_sample"] = (float(ls.detach()), float(gn))
        del sample, ls

    # ---- drop everything this workload holds on the device
    engine.release()
    EF_layers._edge_cache.clear()
    EF_layers._edge_pinned.clear()
    del engine, opt, model, reducer, batches, batch
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()

    # ---- roofline, measured INSIDE the replayed step: a second capture of the same step with a device-timestamp kernel in
    # front of and behind every k4 / k10 / S2 / GEMM dispatch (the library's tagging in graph mode; external event-record nodes
    # are refused under capture by this ROCm build), replayed a few times.  In a CHILD process (one GPU, after this workload's
    # memory has been handed back): an instrumented capture is not the product path, and whatever goes wrong in it must not
    # take the measured line down with it.
    if use_graph and args.ingraph_steps > 0 and world == 1 and not multi and rank == 0:
        recs, err = ingraph_child(args, workload)
        if recs is not None:
            per_graph, big_g, n_union_g = roofline_rows(recs, L)
            res["ingraph"] = (per_graph, big_g, n_union_g)
            log("in-graph launches (us): " + "; ".join(f"{k} {v['avg_launch_us']}" for k, v in per_graph.items()))
        else:
            res["ingraph_error"] = err
            log(f"in-graph dispatch timing failed: {err}")
    return res


def ingraph_worker(args):
    """Child process of `ingraph_child`: the step of one workload captured with the library's tagging in graph mode, replayed
    `--ingraph-steps` times; prints {"records": [[tag, ms, edges, nodes], ...]} (one line) on stdout."""
    out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import __graft_entry__
    __graft_entry__.build()
    from singa_amd import graph as G, ops
    from singa_amd.config import load_config
    from singa_amd.engine import TrainStep
    from singa_amd.model.GAN import SINGA
    from singa_amd.optim import Adam
    L, kw, base_ids, _ = G.resolve_workload(args.workload)
    cfg = load_config(lmax=L)
    torch.manual_seed(cfg.train.seed)
    model = SINGA(cfg, device=dev)
    model.train()
    opt = Adam(model.parameters(), lr=cfg.train.optimizer.lr, betas=(cfg.train.optimizer.beta1, cfg.train.optimizer.beta2))
    engine = TrainStep(model, opt, None, use_graph=True, max_grad_norm=float(cfg.train.max_grad_norm), bucket=True,
                       growth=args.growth, max_cached=2)
    batch = G.synthetic_batch(len(base_ids), ids=base_ids, with_lap=False, **kw).to(dev)
    batch.extras["lap_pe_in_step"] = True
    engine.pre_capture_hook = lambda: ops.profile_start(in_graph=True)
    engine.post_capture_hook = ops.profile_pause
    engine.step(batch)                                 # capture (tagged) + first replay
    engine.pre_capture_hook = engine.post_capture_hook = None
    acc = []
    for _ in range(max(1, args.ingraph_steps)):
        engine.step(batch)
        torch.cuda.synchronize()
        acc += ops.profile_read()
    print(json.dumps({"records": [[t, float(ms), int(e), int(n)] for t, ms, e, n in acc]}), file=out, flush=True)
    engine.release()
    ops.profile_end()
    return 0


def ingraph_child(args, workload, limit_s=240):
    """-> (records, None) or (None, reason)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--ingraph-worker", "--workload", workload, "--ingraph-steps",
           str(args.ingraph_steps), "--growth", str(args.growth)]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SINGA_RCCL_SELFTEST"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, timeout=limit_s)
        lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return None, f"child exited with code {r.returncode}"
        return [tuple(x) for x in json.loads(lines[-1])["records"]], None
    except Exception as e:                                 # noqa: BLE001
        return None, f"{type(e).__name__}: {e}"


def overlap_check(R, model, opt, batches, ids, n_graphs, scaling, cfg, bucket, log, k_steps=6):
    """N > 1.  The default order (one all-reduce after the backward pass) and the overlapped one (two-phase backward: the
    transformer's buckets travel while the embedding's backward pass computes) from the same state on the same batches with
    dropout off: median step time and exposed all-reduce time of both, and the largest relative parameter difference
    afterwards (it has to be rounding: the two orders add the same numbers).  Leaves the model where it found it."""
    from singa_amd import dp
    from singa_amd.engine import TrainStep
    was_training = model.training
    model.eval()
    snap = opt.snapshot()
    out, finals = {}, {}
    try:
        for name, phases in (("single_phase", False), ("overlapped", True)):
            opt.restore(snap)
            red = dp.GradAllReducer(model, always=R.selftest, phases=phases)
            if scaling == "strong" and R.world > 1:
                red.set_shard_weight(len(ids), n_graphs)
            eng = TrainStep(model, opt, red, use_graph=True, max_grad_norm=float(cfg.train.max_grad_norm), bucket=bucket,
                            growth=R.args.growth, max_cached=6)
            for b in batches:                                   # captures (side-effect free) + one replay each
                eng.step(b)
            torch.cuda.synchronize()
            opt.restore(snap)                                   # both orders start from the same state
            eng.comm_events = []
            el, per, _, _ = timed_region(eng, batches, k_steps, True, R.dev, not R.args.no_prefetch)
            ex = median([a.elapsed_time(b) for a, b in eng.comm_events]) if eng.comm_events else None
            finals[name] = [p.detach().clone() for p in model.parameters()]
            out[name] = {"ms_per_step_median": round(median(per), 3), "allreduce_exposed_ms": round(ex, 3) if ex is not None else None,
                         "two_phase_engine": bool(eng.two_phase)}
            eng.release()
            del eng, red
        worst = 0.0
        for a, b in zip(finals["single_phase"], finals["overlapped"]):
            worst = max(worst, float((a - b).double().norm() / (a.double().norm() + 1e-30)))
        chk = torch.stack([p.double().sum() for p in finals["overlapped"]]).sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        w = torch.tensor([worst], device=R.dev, dtype=torch.float64)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        out.update(steps=k_steps, params_max_rel_diff=float(w), params_agree=bool(float(w) < 1e-5),
                   ranks_hold_identical_parameters=bool(float(hi - lo) == 0.0),
                   note="same initial state, same batches, dropout off; the headline region runs "
                        + ("the overlapped order (--allreduce-overlap)" if R.args.allreduce_overlap else "single_phase"))
        log(f"all-reduce overlap check: {out}")
    finally:
        opt.restore(snap)
        model.train(was_training)
    return out


def main():
    args = parse()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args.workload, args.cpu_graphs, args.cpu_threads or host_cores(), args.cpu_state,
                                   args.cpu_graphs_1t)
    if args.ingraph_worker:
        return ingraph_worker(args)
    if args.sweep:
        return sweep(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    # stdout carries exactly ONE line, the JSON result: native libraries write to file descriptor 1 as well (RCCL prints a
    # version banner there when its first communicator comes up), so everything else is sent to stderr from here on
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    # SINGA_RCCL_SELFTEST=1 with one rank: take every multi-rank branch below on a ONE-rank RCCL communicator (process
    # group, barriers, bucketed all-reduce between the replayed graphs, capture next to the RCCL watchdog) - what a one-GPU
    # box can exercise of the N-GPU path
    selftest = world == 1 and os.environ.get("SINGA_RCCL_SELFTEST") == "1"
    if selftest:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    multi = world > 1 or selftest
    if multi:
        # RCCL ("nccl") is the product path.  SINGA_DIST_BACKEND=gloo exists only to rehearse the multi-rank control flow
        # on a one-GPU box (several ranks sharing a device, which RCCL refuses).
        backend = os.environ.get("SINGA_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            local = local % max(n_dev, 1)
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # N > 1 defaults to BASELINE.json configs[3]: the SAME batch split over the ranks (strong scaling)
    scaling = args.scaling or ("strong" if world > 1 else "weak")

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if multi:
        dist.barrier()
    from singa_amd import graph as G

    R = Run(args, world, rank, local, dev, multi, selftest, scaling)
    t_all = time.perf_counter()
    res = run_workload(R, args.workload, args.steps, args.warmup, main=True)

    # ---- further BASELINE configs with the same engine (N = 1, default workload only): configs[1] and the full per-GPU share of
    # configs[4], so that the driver's record carries them - value, median step, k10 roofline fraction inside the replayed step,
    # Laplacian-encoding time and peak HBM
    others = {}
    if world == 1 and not multi and not args.eager and args.workload == "cfg3_b128_l4" and args.other_workloads:
        for wl in [w for w in args.other_workloads.split(",") if w]:
            if wl not in G.WORKLOADS or wl == args.workload:
                continue
            try:
                o = run_workload(R, wl, args.other_steps, 2, main=False)
            except Exception as e:                        # noqa: BLE001 - reported, must not take the main line down
                others[wl] = {"error": f"{type(e).__name__}: {e}"}
                R.log(f"other workload {wl} failed: {others[wl]['error']}")
                torch.cuda.empty_cache()
                continue
            med_o = median(o["per_step"])
            k10 = (o.get("ingraph") or ({},))[0].get("k10_fwd")
            others[wl] = {"value": round(o["graphs_per_step"] * args.other_steps / o["elapsed"], 3), "unit": "graphs/s",
                          "graphs": o["n_graphs"], "lmax": o["L"], "atoms": o["n_nodes"], "edges": o["n_edges"],
                          "steps": args.other_steps, "ms_per_step": round(1e3 * o["elapsed"] / args.other_steps, 3),
                          "ms_per_step_median": round(med_o, 3), "value_at_median": round(o["graphs_per_step"] / (med_o * 1e-3), 3),
                          "k10_frac": k10["frac"] if k10 else None, "k10_avg_launch_us_in_graph": k10["avg_launch_us"] if k10 else None,
                          "k10_bytes_per_launch": k10["bytes_per_launch"] if k10 else None,
                          "lap_pe_ms": round(o["lap_pe_ms"], 2), "prepare_ms_of_step": round(o["prepare_ms"], 2),
                          "peak_hbm_gb": o["peak_hbm_gb"], "peak_hbm_reserved_gb": o["peak_hbm_reserved_gb"],
                          "final_loss": round(o["final_loss"], 5), "captures_in_timed_region": o["captures_timed"]}
            R.log(f"other workload {wl}: {others[wl]}")

    rc = 0
    if rank == 0:
        L, kw, ids, base_ids = res["L"], res["kw"], res["ids"], res["base_ids"]
        per_step, elapsed, graphs_per_step = res["per_step"], res["elapsed"], res["graphs_per_step"]
        n_nodes, n_edges, D, bucket, use_graph = res["n_nodes"], res["n_edges"], res["D"], res["bucket"], res["use_graph"]
        sizes = [G.graph_sizes(i, **kw) for i in ids]
        mean = lambda k: round(sum(s[k] for s in sizes) / len(sizes), 1)
        med = median(per_step)
        # ---- the roofline block: k10 forward on the bonded edges.  frac / achieved / avg_launch_us come from the dispatches of
        # the REPLAYED step of this run (event-record nodes inside the captured graph); the eager-dispatch timing and the
        # committed rocprofv3 trace are cross-checks
        roof = None
        per_g, big_g, nu_g = res.get("ingraph") or ({}, 0, 0)
        per_e, big_e, nu_e = res.get("eager_rows") or ({}, 0, 0)
        src_rows, how = (per_g, "in_graph") if "k10_fwd" in per_g else ((per_e, "eager") if "k10_fwd" in per_e else ({}, None))
        if how:
            big, n_union = (big_g, nu_g) if how == "in_graph" else (big_e, nu_e)
            k = src_rows["k10_fwd"]
            prof = committed_profile(args.workload)
            # PMC traffic: the counter passes run the step eagerly on the UNPADDED batch, so their k10 launch is the eager rows'
            # (edges, nodes), not the padded one of the replayed step: looked up by either grid, reported with its own bytes
            tr, tr_launch = None, None
            for e_, n_ in ((big, n_union), (big_e, nu_e)):
                if n_ and tr is None:
                    tr = pmc_traffic(prof, f"rotate_back_scatter_kernel<{L}, 2, false", n_ * 128)
                    if tr is not None:
                        alg = kernel_bytes("k10_fwd", e_, n_, L)
                        tr_launch = {"edges": e_, "dst_nodes": n_, "algorithmic_bytes": alg, "traffic_over_algorithmic": round(tr / alg, 4)}
            in_prof = rocprof_in_graph_us(prof, f"rotate_back_scatter_kernel<{L}, 2, false", n_union * 128)
            sha = lib_source_sha()
            same = prof is not None and prof[1].get("lib_sha") == sha
            src = (f"{os.path.relpath(prof[0], ROOT)} (library source sha {prof[1].get('lib_sha')}; "
                   f"{'same as' if same else 'OLDER than'} this build {sha})") if prof else None
            roof = {"bound": "hbm", "kernel": "rotate_back_scatter_kernel (k10 fwd, 'scatter-TP', on the bonded edges: "
                                              "protein-protein U ligand-ligand pass)",
                    "achieved": k["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k["frac"],
                    "frac_uses": ("avg_launch_us = the kernel's dispatches INSIDE a replayed step, measured by a child process of this run: a capture of "
                                  "the same step with a device-timestamp kernel (100 MHz wall clock) in front of and behind each "
                                  "tagged launch, minus a calibration pair with nothing in between (so it still includes ~one "
                                  f"dependent-launch gap); {args.ingraph_steps} replays" if how == "in_graph" else
                                  "avg_launch_us (events attached to each EAGER dispatch, measured in this process)"),
                    "traffic": tr, "traffic_launch": tr_launch,
                    "traffic_source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command: {src}" if tr else None,
                    "bytes_per_launch": k["bytes_per_launch"], "avg_launch_us": k["avg_launch_us"], "launches": k["launches"],
                    "avg_launch_us_eager": per_e.get("k10_fwd", {}).get("avg_launch_us"),
                    "avg_launch_us_rocprof": in_prof,
                    "rocprof_source": f"rocprofv3 --kernel-trace of this command (replayed steps): {src}" if in_prof else None,
                    "hbm_copy_ceiling": None,
                    "hbm_copy_ceiling_note": "GB/s, read + write bytes of the library's 16-byte-per-lane copy kernel on 256 MB operands, best "
                                             "of a sweep of launch shapes, measured in this process (hbm_copy_dword: one dword per lane, the "
                                             "segment kernels' access shape)",
                    "hbm_copy_dword": copy_ceiling_gbs(dev, wide=False),
                    "edges": big, "dst_nodes": n_union,
                    "other_kernels": {t: v for t, v in src_rows.items() if t != "k10_fwd"},
                    "other_kernels_eager": {t: {"avg_launch_us": v["avg_launch_us"], "frac": v["frac"]} for t, v in per_e.items()}
                    if how == "in_graph" else None}
            roof["hbm_copy_ceiling"], roof["hbm_copy_ceiling_shape"] = copy_ceiling_gbs(dev, wide=True)
            roof["frac_of_copy_ceiling"] = round(k["achieved"] / roof["hbm_copy_ceiling"], 4) if roof["hbm_copy_ceiling"] else None
            if prof and prof[1].get("mfma"):
                roof["mfma"] = dict(prof[1]["mfma"], source=os.path.relpath(prof[0], ROOT), peak_tflops=MFMA_F32_PEAK_TFLOPS)
        if res.get("ingraph_error"):
            roof = roof or {}
            roof["ingraph_error"] = res["ingraph_error"]
        out = {"metric": baseline_metric(),
               "value": round(graphs_per_step * args.steps / elapsed, 3), "unit": "graphs/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
               "ms_per_step_median": round(med, 3), "value_at_median": round(graphs_per_step / (med * 1e-3), 3),
               "host_submit_ms_per_step": round(res["host_submit_ms"], 3) if res.get("host_submit_ms") is not None else None,
               "host_prefetch_ms_per_step": round(res["host_prefetch_ms"], 3) if res.get("host_prefetch_ms") is not None else None,
               "host_cpu_ms_per_step": round(res["host_cpu_ms"], 3) if res.get("host_cpu_ms") is not None else None,
               "higher_is_better": True, "scaling": scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic",
               "config": {"workload": args.workload + (" split over the ranks (BASELINE.json configs[3])"
                                                       if world > 1 and scaling == "strong" else ""),
                          "global_batch": graphs_per_step, "graphs_per_gpu": len(ids), "lmax": L,
                          "mmax": 2, "ragged": "ragged" in kw,
                          "atoms_on_this_gpu": n_nodes, "edges_on_this_gpu": n_edges,
                          "mean_protein_atoms": mean("n_protein"), "mean_ligand_atoms": mean("n_ligand"),
                          "mean_edges_per_graph": round(n_edges / len(ids), 1),
                          "parallelism": f"dp{world}" + (" (RCCL self-test: the collectives run on a 1-rank communicator)" if selftest else ""),
                          "step": "prepare+zero_grad+fwd+CE+bwd+allreduce+clip+Adam of the generator (the reference has no discriminator)",
                          "launch": "hipGraph replay" if use_graph else "eager",
                          "value_is": "graphs of all ranks x steps / wall time of the timed region (barrier + synchronize on both "
                                      "sides, MAX over ranks); ms_per_step_median = median of the per-step event times",
                          "batches": (f"{D} different resident batches per GPU cycled through the steps, each padded to its "
                                      f"size class (x{args.growth} per class) and its graph structure rebuilt every step"
                                      if bucket else "one resident batch per GPU, its graph structure rebuilt every step"),
                          "captures_in_timed_region": res["captures_timed"],
                          "prepare": "in step" if args.no_prefetch else "prefetched on a second stream during the previous step",
                          "prepare_ms_of_step": round(res["prepare_ms"], 2), "lap_pe_ms": round(res["lap_pe_ms"], 2),
                          "lap_pe": ("recomputed inside every step (preparation phase, prefetch stream) by the library's batched "
                                     "eigensolver singa_lap_eig; lap_pe_ms = the same computation timed on its own"
                                     if res["lap_in_step"] else
                                     "carried by the batches from generation time; lap_pe_ms = graph.laplacian_pe_batched timed on its own"),
                          "graph_captures": res["captures"], "generation_s": round(res["gen_s"], 1),
                          "grad_allreduce_bytes": res["reducer"]["payload_bytes"],
                          "grad_allreduce": (("two-phase backward: the transformer's buckets (" + str(res["reducer"]["phase0_bytes"])
                                              + " bytes) are reduced while the embedding's backward pass computes, the embedding's after it")
                                             if res["two_phase"] else
                                             "one bucketed all-reduce after the backward pass" if multi else "none (one rank)")},
               "final_loss": round(res["final_loss"], 5), "roofline": roof,
               "peak_hbm_gb": res["peak_hbm_gb"], "peak_hbm_reserved_gb": res["peak_hbm_reserved_gb"], "hbm_total_gb": res["hbm_total_gb"]}
        if multi:
            out["rccl_ranks"] = res["rccl_ranks"]
            out["per_rank"] = res["per_rank"]
            out["allreduce_exposed_ms"] = round(res["exposed_ms"], 3) if res["exposed_ms"] is not None else None
            out["allreduce_exposed_ms_is"] = ("median over the steps of the time between the end of the backward pass and the arrival of "
                                              "the last bucket on the compute stream (rank 0; per_rank has every rank's)")
        if res["overlap"] is not None:
            out["allreduce_overlap"] = res["overlap"]
        if res["weak"] is not None:
            out["weak"] = res["weak"]
        if res["proxy"] is not None:
            out["strong_proxy"] = res["proxy"]
            out["strong_proxy_ms"] = res["proxy"]["ms_per_step_median"]
        if others:
            out["other_workloads"] = others
        if res.get("hip_sample") is not None:
            R.log("timing the CPU oracle on the bounded sample ...")
            state_file = res["state_file"]
            cb = cpu_baseline(args.workload, args.cpu_graphs, state=state_file, n_graphs_1t=args.cpu_graphs_1t)
            import shutil
            shutil.rmtree(os.path.dirname(state_file), ignore_errors=True)
            lo, gno = cb.pop("loss", None), cb.pop("grad_norm", None)
            out["cpu_baseline"] = cb
            hip_sample = res["hip_sample"]
            if lo is not None:
                dl, dg = abs(hip_sample[0] - lo) / abs(lo), abs(hip_sample[1] - gno) / abs(gno)
                ok = dl <= ORACLE_TOL and dg <= ORACLE_TOL
                out["oracle_check"] = {"what": f"CrossEntropy and total gradient 2-norm of the first {args.cpu_graphs} graphs at the "
                                               "initial parameters, dropout off, identical input tensors: HIP path vs CPU oracle",
                                       "loss_hip": round(hip_sample[0], 6), "loss_oracle": round(lo, 6), "rel_diff": dl,
                                       "grad_norm_hip": round(hip_sample[1], 6), "grad_norm_oracle": round(gno, 6),
                                       "grad_norm_rel_diff": dg, "tolerance": ORACLE_TOL, "pass": ok,
                                       "gate": "bench.py exits with code 3 (after this line) when pass is false"}
                if not ok:
                    rc = 3
        out["bench_wall_s"] = round(time.perf_counter() - t_all, 1)
        print(json.dumps(out), file=json_out, flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main() or 0)
