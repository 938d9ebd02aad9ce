"""Lab: host time per phase of a replayed data-parallel step on ONE rank (RCCL communicator of one rank, as
SINGA_RCCL_SELFTEST=1 bench.py), next to the same step without a reducer.  Where does the step wait?
    [AUX_PRIORITY=normal] [MODES="plain;reducer;reducer, one phase"] python tools/lab/dp_timeline.py [workload]
Finding (profiles/r03z/dp_timeline.txt): with the prefetch stream at normal priority, the step is 9 ms longer in SOME runs -
whichever engine's prefetch stream HIP happened to map onto the compute stream's hardware queue - with or without a reducer;
at high priority never."""
import os, sys, time, socket, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
dist.init_process_group(os.environ.get("SINGA_DIST_BACKEND", "nccl"), device_id=torch.device("cuda", 0))
from singa_amd import dp, graph as G
from singa_amd.config import load_config
from singa_amd.engine import TrainStep
from singa_amd.model.GAN import SINGA
from singa_amd.optim import Adam

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3_b128_l4"
L, kw, ids, n_parent = G.resolve_workload(wl)
acc = collections.defaultdict(float)


def timed(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        acc[label] += time.perf_counter() - t
        return r
    setattr(obj, name, g)


for mode in os.environ.get("MODES", "plain;reducer;reducer, one phase").split(";"):
    torch.manual_seed(0)
    model = SINGA(load_config(lmax=L), device="cuda").train()
    red = None if mode == "plain" else dp.GradAllReducer(model, always=True, phases=True if mode == "reducer" else False)
    eng = TrainStep(model, Adam(model.parameters(), lr=1e-4), red, use_graph=True, bucket=True, growth=1.04, max_cached=6)
    eng.prefetch_priority = -1 if os.environ.get("AUX_PRIORITY", "normal") == "high" else 0
    batches = [G.synthetic_batch(len(ids), ids=[i + k * n_parent for i in ids], **kw).to("cuda") for k in range(3)]
    for b in batches:
        b.extras["lap_pe_in_step"] = True
    for i in range(5):
        eng.step(batches[i % 3])
    torch.cuda.synchronize()
    for name in ("_stage", "_load", "_replay", "prefetch", "_join_prefetch", "_activate"):
        timed(eng, name)
    if red is not None:
        for name in ("any_rank", "max_ints", "launch", "wait"):
            timed(red, name, "reducer." + name)
    acc.clear()
    steps = 15
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    marks[0].record()
    nxt = batches[0]
    t0 = time.perf_counter()
    for i in range(steps):
        t = time.perf_counter()
        eng.step(nxt)
        acc["step (host)"] += time.perf_counter() - t
        marks[i + 1].record()
        nxt = eng.prefetch(batches[(i + 1) % 3])
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    print(f"== {mode}: median step {per[steps // 2]:.2f} ms (GPU events); host loop {t_host / steps * 1e3:.2f} ms per step before the final sync")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f"     {k:24s} {v / steps * 1e3:8.3f} ms per step (host)")
    eng.release()
    del eng, model, red, batches
    torch.cuda.empty_cache()
dist.destroy_process_group()
