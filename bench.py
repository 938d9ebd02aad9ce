#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on MI355X: protein-ligand graphs/s for one full SINGA training step
(zero_grad -> forward -> CrossEntropy -> backward -> [grad all-reduce] -> clip(inf) -> Adam), plus the achieved
HBM GB/s of the fused alpha-scale + rotate-back + scatter kernel ("scatter-TP", k10) against the roofline, plus
the CPU oracle timed on this box's host cores.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2_b32_l2]

N > 1 is launched by the driver with torch.distributed.run (one rank per GPU, RCCL): every rank trains on its own
shard of `n_graphs` synthetic graphs (weak scaling), gradients are averaged with one bucketed all-reduce per step.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2_b32_l2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="run the step eagerly instead of replaying HIP graphs")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="build each batch's graph structure inside its own step instead of on a second stream during the previous one")
    ap.add_argument("--roofline-steps", type=int, default=3, help="instrumented eager steps after the timed region")
    ap.add_argument("--cpu-graphs", type=int, default=12, help="graphs in the bounded CPU-oracle sample")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=0, help=argparse.SUPPRESS)
    return ap.parse_args()


def k10_algorithmic_bytes(E, N, L, CH=112, heads=7):
    """SURVEY.md §8d formula for rotate_back_scatter forward, with the 36-byte frame replaced by the streamed
    reduced Wigner rows (WSZ floats per edge), as DESIGN.md states."""
    from singa_amd import so3
    lay = so3.layout(L, 2)
    return E * (lay.KR * CH * 4 + heads * 4 + lay.WSZ * 4) + N * lay.K * CH * 4 + (N + 1) * 4


def pmc_traffic(L, n_dst):
    """HBM bytes per launch of the scatter-TP forward kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/<round>/pmc_{FETCH,WRITE}_SIZE.csv, produced by tools/prof.sh): FETCH_SIZE x 2 (gfx950 correction
    for this access shape, confirmed by the calibration copy in the same pass) + WRITE_SIZE, KiB -> bytes.  None if no
    profile matches the launch geometry (grid = dst nodes x 128)."""
    import csv
    import glob
    best = None
    for d in sorted(glob.glob(os.path.join(ROOT, "profiles", "*"))):
        vals = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            f = os.path.join(d, f"pmc_{c}.csv")
            if not os.path.exists(f):
                continue
            for r in csv.DictReader(open(f)):
                if f"rotate_back_scatter_kernel<{L}, 2, false" in r["kernel"] and int(r["grid"]) == n_dst * 128:
                    vals[c] = float(r["avg_value"])
        if len(vals) == 2:
            best = (int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), os.path.relpath(d, ROOT))
    return best


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


def cpu_baseline_worker(workload, n_graphs, threads):
    """Runs in a child process that never touches the GPU: times the CPU oracle (kind 'port':
    oracle/singa_oracle.py, pinned to the reference by tests/golden) on a bounded sample of the same workload -
    forward + CrossEntropy + backward of `n_graphs` synthetic graphs - and prints one JSON object."""
    torch.set_num_threads(threads)
    from oracle import singa_oracle as O
    from singa_amd import graph as G
    from singa_amd.config import load_config
    from singa_amd.model.GAN import SINGA
    wl = dict(G.WORKLOADS[workload])
    L = wl["lmax"]
    kw = {k: v for k, v in wl.items() if k not in ("n_graphs", "lmax")}
    cfg = load_config(lmax=L)
    torch.manual_seed(cfg.train.seed)
    model = SINGA(cfg, device="cpu")          # parameter container only; the product forward is never called here
    graphs = [G.synthetic_graph(10_000 + i, **kw) for i in range(n_graphs)]
    og = []
    for g in graphs:
        og.append({"x_p": g[G.PA]["x"], "pos_p": g[G.PA]["pos"], "z_p": g["atomicnum"][G.PA],
                   "x_l": g[G.LA]["x"], "pos_l": g[G.LA]["pos"], "z_l": g["atomicnum"][G.LA],
                   "ei_pp": g[G.E_PP]["edge_index"], "ei_ll": g[G.E_LL]["edge_index"],
                   "ei_lp": g[G.E_LP]["edge_index"], "ei_pl": g[G.E_PL]["edge_index"],
                   "tok_in": g["ligand_data"]["smiIndices_input"], "tok_tgt": g["ligand_data"]["smiIndices_tgt"],
                   "props": torch.tensor([g["ligand_data"][k] for k in ("vina_score", "qed", "sas")],
                                         dtype=torch.float64)})
    b = O.collate(og)
    rand = {k: torch.cat([g.extras["rot_rand"][k] for g in graphs], 0) for k in ("pp", "ll", "lp")}
    vec = {"pp": b["pos_p"][b["ei_pp"][0]] - b["pos_p"][b["ei_pp"][1]],
           "ll": b["pos_l"][b["ei_ll"][0]] - b["pos_l"][b["ei_ll"][1]],
           "lp": b["pos_l"][b["ei_lp"][0]] - b["pos_p"][b["ei_lp"][1]]}
    rots = {k: O.edge_rot_mat(vec[k], rand[k]) for k in vec}
    lap_p = torch.cat([g[G.PA]["lap_pe"] for g in graphs], 0)
    lap_l = torch.cat([g[G.LA]["lap_pe"] for g in graphs], 0)
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    times = []
    for it in range(2):   # first pass warms the table caches; the second is reported
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        loss = O.train_step_loss(sd, b, rots, L, lap_p, lap_l)
        loss.backward()
        times.append(time.perf_counter() - t0)
        print(f"[cpu-baseline] pass {it}: {times[-1]:.2f} s", file=sys.stderr, flush=True)
    print(json.dumps({"value": round(n_graphs / times[-1], 4), "unit": "graphs/s", "cores": threads, "kind": "port",
                      "sample": f"{n_graphs} graphs of {workload}: oracle forward+CE+backward (no Adam), "
                                f"{threads} torch threads, {times[-1]:.1f} s"}), flush=True)


def cpu_baseline(workload, n_graphs, limit_s=300):
    import subprocess
    threads = host_cores()
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads),
               MKL_NUM_THREADS=str(threads))
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--workload", workload, "--cpu-graphs",
           str(n_graphs), "--cpu-threads", str(threads)]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, timeout=limit_s, check=True)
        return json.loads(r.stdout.decode().strip().splitlines()[-1])
    except Exception as e:  # a baseline that cannot be timed must not take the GPU measurement down with it
        return {"value": None, "unit": "graphs/s", "cores": threads, "kind": "port",
                "sample": f"not measured: {type(e).__name__}"}


def main():
    args = parse()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args.workload, args.cpu_graphs, args.cpu_threads or host_cores())
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    if world > 1:
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

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    from singa_amd import dp, graph as G, ops
    from singa_amd.config import load_config
    from singa_amd.model import EF_layers
    from singa_amd.model.GAN import SINGA

    wl = dict(G.WORKLOADS[args.workload])
    n_graphs, L = wl["n_graphs"], wl["lmax"]
    kw = {k: v for k, v in wl.items() if k not in ("n_graphs", "lmax")}
    cfg = load_config(lmax=L)
    torch.manual_seed(cfg.train.seed)                     # same-seed init on every rank (no broadcast)
    model = SINGA(cfg, device=dev)
    model.train()
    reducer = dp.GradAllReducer(model)
    reducer.check_same_init()
    use_graph = not args.eager
    from singa_amd.optim import Adam
    opt = Adam(model.parameters(), lr=cfg.train.optimizer.lr,
               betas=(cfg.train.optimizer.beta1, cfg.train.optimizer.beta2))
    batch = G.synthetic_batch(n_graphs, first_id=rank * n_graphs, **kw).to(dev)   # this rank's shard, resident in HBM
    # a second resident copy: steps alternate between the two, so that batch i+1 can be prepared while step i computes
    batches = [batch, batch if args.no_prefetch else G.synthetic_batch(n_graphs, first_id=rank * n_graphs, **kw).to(dev)]
    from singa_amd.engine import TrainStep
    engine = TrainStep(model, opt, reducer if world > 1 else None, use_graph=use_graph,
                       max_grad_norm=float(cfg.train.max_grad_norm))

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    log(f"workload {args.workload}: {n_graphs} graphs/GPU, L={L}; model + batch ready; "
        f"{'HIP-graph replay' if use_graph else 'eager'} step")
    for i in range(args.warmup):
        t_w = time.perf_counter()
        loss = engine.step(batch)
        torch.cuda.synchronize()
        log(f"warmup step {i}: {time.perf_counter() - t_w:.3f} s")
    # ---- timed region: exactly K steps, barrier + synchronize on both sides, MAX over ranks.  Every step handles its
    # batch as newly arrived: the graph structure (edge sorting, kNN graphs, dense maps) is rebuilt K times inside the
    # region - by default on a second stream while the previous step computes (TrainStep.prefetch), the way a loader
    # thread would; with --no-prefetch at the start of the step itself.
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = engine.step(batches[i % 2])
        if not args.no_prefetch:
            engine.prefetch(batches[(i + 1) % 2])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss.detach())
    log(f"timed {args.steps} steps in {elapsed:.3f} s ({engine.captures} graph capture(s) so far)")
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

    # ---- instrumented pass for the roofline: the same step run eagerly with start/stop events attached to every
    # scatter-TP forward dispatch (graph replays cannot carry per-dispatch events); not part of `value`.
    if use_graph and args.roofline_steps > 0:
        engine.release()              # give the graph pool back before the eager instrumented steps
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

    roof = None
    if recs:
        big = max(r[1] for r in recs)                     # dispatches with the most edges = the protein-protein passes
        sel = [r for r in recs if r[1] == big]
        ms = sum(r[0] for r in sel) / len(sel)
        log("k10 forward launches on the bonded edges (us): " + " ".join(f"{r[0] * 1e3:.1f}" for r in sel))
        E, N = sel[0][1], sel[0][2]
        by = k10_algorithmic_bytes(E, N, L)
        ach = by / (ms * 1e-3) / 1e9
        tr = pmc_traffic(L, N)
        roof = {"bound": "hbm", "kernel": "rotate_back_scatter_kernel (k10 fwd on the bonded edges: protein-protein U "
                                          "ligand-ligand pass)",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": tr[0] if tr else None,
                "traffic_source": (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, {tr[1]}" if tr else None),
                "bytes_per_launch": by, "avg_launch_us": round(ms * 1e3, 2), "launches": len(sel),
                "edges": E, "dst_nodes": N, "timing": f"start/stop events attached to each dispatch, {args.roofline_steps} instrumented eager steps "
                          "right after the timed region (same process, same batch)"}

    if rank == 0:
        total_graphs = n_graphs * world * args.steps
        out = {"metric": baseline_metric(),
               "value": round(total_graphs / elapsed, 3), "unit": "graphs/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": args.workload, "graphs_per_gpu": n_graphs, "lmax": L, "mmax": 2,
                          "nodes_per_graph": kw["n_protein"] + kw["n_ligand"],
                          "edges_per_graph": kw["e_pp"] + kw["e_ll"] + 2 * kw["e_x"],
                          "parallelism": f"dp{world}",
                          "step": "prepare+zero_grad+fwd+CE+bwd+allreduce+clip+Adam of the generator (the reference has no discriminator)",
                          "launch": "hipGraph replay" if use_graph else "eager",
                          "prepare": "in step" if args.no_prefetch else "prefetched on a second stream during the previous step", "prepare_ms_of_step": round(prepare_ms, 2),
                          "grad_allreduce_bytes": reducer.payload_bytes},
               "final_loss": round(final_loss, 5), "roofline": roof}
        if not args.no_cpu_baseline and world == 1:
            log("timing the CPU oracle on the bounded sample ...")
            out["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_graphs)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
