"""-m gpu: the data-parallel training step end to end on ONE GPU box: two ranks (two processes sharing the device, gloo
collectives - RCCL refuses two ranks on one device) run TrainStep with the bucketed all-reduce between the replayed
forward+backward graph and the optimizer graph, each on its own shard of graphs; the result must equal one process
training on the union batch (a shard of G graphs is a reference batch of size G; equal token counts ->
the average of the rank gradients is the global-batch gradient).  Covers what the 8-GPU run does apart from the
transport: same-seed init check, shard weighting, collective capture decision, replay + eager all-reduce + replay."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(n_protein=40, n_ligand=12, e_pp=200, e_ll=24, e_x=30)
STEPS = 3


def _train(rank, world, ids, use_graph):
    sys.path.insert(0, ROOT)
    from singa_amd import dp, graph as G
    from singa_amd.config import load_config
    from singa_amd.engine import TrainStep
    from singa_amd.model.GAN import SINGA
    from singa_amd.optim import Adam
    torch.manual_seed(11)
    model = SINGA(load_config(lmax=2), device="cuda").eval()       # dropout off: the ranks must see the same numbers
    reducer = dp.GradAllReducer(model, phases=True) if world > 1 else None      # the two-phase (overlapped) order
    if reducer:
        reducer.check_same_init()
        reducer.set_shard_weight(len(ids), len(ids) * world)
    eng = TrainStep(model, Adam(model.parameters(), lr=1e-4), reducer, use_graph=use_graph)
    batch = G.synthetic_batch(len(ids), ids=ids, **KW).to("cuda")
    losses = [float(eng.step(batch).detach()) for _ in range(STEPS)]
    probe = {n: p.detach().cpu().clone() for n, p in model.named_parameters()
             if n in ("model.projection.weight", "embedding.blocks.0.ga.alpha_dot", "embedding.blocks.2.ffn.so3_linear_2.weight")}
    return losses, probe, (reducer.payload_bytes if reducer else 0)


def _worker(rank, world, port, out, use_graph):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ids = [100, 101] if rank == 0 else [102, 103]
    res = _train(rank, world, ids, use_graph)
    torch.save(res, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_ranks_equal_one_process_on_the_union_batch(tmp_path, use_graph):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r")
    mp.spawn(_worker, args=(2, port, out, use_graph), nprocs=2, join=True)
    (l0, p0, bytes0), (l1, p1, _) = torch.load(out + ".0"), torch.load(out + ".1")
    ref_l, ref_p, _ = _train(0, 1, [100, 101, 102, 103], False)
    assert bytes0 > 60e6                                              # L = 2 model: 64 MB of gradients per step
    for n in ref_p:                                                   # both ranks hold the same parameters ...
        assert torch.equal(p0[n], p1[n]), n
        d = (p0[n] - ref_p[n]).norm() / (ref_p[n].norm() + 1e-12)     # ... and they are the union-batch parameters
        # two shards vs the union batch = different fp32 summation orders in every reduction over atoms / edges; through the
        # depth of the network that is ~1e-4 relative in individual gradient tensors (the same size as own GEMM vs BLAS
        # library on one batch), and Adam's g / sqrt(v) turns sign changes of near-zero entries into full-size updates
        assert float(d) < 2e-4, (n, float(d))
    # the union batch's loss is the mean of the shard losses (equal token counts)
    for a, b, r in zip(l0, l1, ref_l):
        assert abs(0.5 * (a + b) - r) < 2e-4 * abs(r), (l0, l1, ref_l)


# ---- BASELINE.json configs[3] in miniature: ONE ragged config-3 batch (the bench's generator, L = 4) split over two ranks by
# edge cost (unequal shards), token-weighted combination, bucketed HIP-graph replay with sizes agreed over the ranks - the
# path `bench.py --gpus N` takes by default
CFG3_IDS = [0, 1, 2, 3, 4]
CFG3_STRIDE = 128


def _cfg3_kw():
    from singa_amd import graph as G
    return G.resolve_workload("cfg3_b128_l4")[1]


def _train_cfg3(rank, world, use_graph):
    sys.path.insert(0, ROOT)
    from singa_amd import dp, graph as G
    from singa_amd.config import load_config
    from singa_amd.engine import TrainStep
    from singa_amd.model.GAN import SINGA
    from singa_amd.optim import Adam
    kw = _cfg3_kw()
    torch.manual_seed(11)
    model = SINGA(load_config(lmax=4), device="cuda").eval()
    reducer = dp.GradAllReducer(model) if world > 1 else None
    ids = CFG3_IDS
    if reducer:
        reducer.check_same_init()
        lo, hi = dp.shard_ranges_by_cost([G.graph_cost(G.graph_sizes(i, **kw)) for i in CFG3_IDS], world)[rank]
        ids = CFG3_IDS[lo:hi]
        reducer.set_shard_weight(len(ids), len(CFG3_IDS))
    eng = TrainStep(model, Adam(model.parameters(), lr=1e-4), reducer, use_graph=use_graph, bucket=use_graph, growth=1.04)
    batches = [G.synthetic_batch(len(ids), ids=[i + k * CFG3_STRIDE for i in ids], **kw).to("cuda") for k in range(2)]
    losses = [float(eng.step(batches[k % 2]).detach()) for k in range(4)]
    probe = {n: p.detach().cpu().clone() for n, p in model.named_parameters()
             if n in ("model.projection.weight", "embedding.blocks.0.ga.alpha_dot", "embedding.blocks.2.ffn.so3_linear_2.weight",
                      "model.encoder.layers.0.pos_ffn.conv1.weight")}
    return losses, probe, len(ids), eng.captures


def _worker_cfg3(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.save(_train_cfg3(rank, world, True), f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_strong_split_of_a_config3_batch_equals_the_union(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "s")
    mp.spawn(_worker_cfg3, args=(2, port, out), nprocs=2, join=True)
    (l0, p0, n0, c0), (l1, p1, n1, c1) = torch.load(out + ".0"), torch.load(out + ".1")
    assert n0 + n1 == len(CFG3_IDS) and n0 != n1                     # unequal shards
    assert c0 == c1                                                   # the ranks captured in lockstep
    ref_l, ref_p, _, _ = _train_cfg3(0, 1, False)                     # one process, eager, on the whole batch
    for n in ref_p:
        assert torch.equal(p0[n], p1[n]), n
        d = (p0[n] - ref_p[n]).norm() / (ref_p[n].norm() + 1e-12)
        assert float(d) < 2e-4, (n, float(d))
    # the whole batch's loss is the token-weighted mean of the shard losses
    for a, b, r in zip(l0, l1, ref_l):
        assert abs((n0 * a + n1 * b) / (n0 + n1) - r) < 2e-4 * abs(r), (l0, l1, ref_l)


def test_rccl_selftest_one_rank_bench():
    """The N-rank code path of bench.py on a ONE-rank RCCL communicator (SINGA_RCCL_SELFTEST=1): process group on `nccl`,
    same-init check, collective capture decision over the side gloo group, barriers, the bucketed all-reduce between the
    replayed forward+backward graph and the optimizer graph, max-over-ranks timing.  (More ranks need more GPUs: RCCL
    refuses two ranks on one device.)  The run must produce a normal bench line that moved the gradient payload."""
    import json
    import subprocess
    env = dict(os.environ, SINGA_RCCL_SELFTEST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg2_b32_l2", "--steps", "4", "--warmup", "2",
                        "--no-cpu-baseline", "--roofline-steps", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["final_loss"] == line["final_loss"]
    assert "RCCL self-test" in line["config"]["parallelism"]
    assert line["config"]["grad_allreduce_bytes"] > 60e6            # the L = 2 model's 64 MB of gradients went through RCCL


def test_two_rank_bench_line_explains_itself():
    """`bench.py --gpus 2` on ONE GPU (SINGA_DIST_BACKEND=gloo: two ranks sharing the device - the multi-rank control flow
    with gloo in RCCL's place): the N > 1 line names who ran where (`rccl_ranks`, `per_rank`), what the all-reduce cost the
    step (`allreduce_exposed_ms`), carries the weak figure per GPU, and its `allreduce_overlap` block shows the two-phase
    order to end with the same parameters as the default one on every rank."""
    import json
    import subprocess
    env = dict(os.environ, SINGA_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "SINGA_RCCL_SELFTEST"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cfg2_b32_l2", "--steps", "4",
                        "--warmup", "1", "--roofline-steps", "0", "--ingraph-steps", "0"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["rccl_ranks"]["world_size"] == 2 and len(line["rccl_ranks"]["device_ordinals"]) == 2
    pr = line["per_rank"]
    assert [p["rank"] for p in pr] == [0, 1] and sum(p["graphs"] for p in pr) == 32
    assert all(p["step_ms_min"] <= p["step_ms_median"] <= p["step_ms_max"] for p in pr)
    assert line["allreduce_exposed_ms"] is not None and line["allreduce_exposed_ms"] >= 0
    ov = line["allreduce_overlap"]
    assert ov["overlapped"]["two_phase_engine"] and not ov["single_phase"]["two_phase_engine"]
    assert ov["params_agree"] and ov["ranks_hold_identical_parameters"], ov
    assert line["weak"]["global_batch"] == 64 and line["weak"]["value_per_gpu"] > 0
    assert "after the backward pass" in line["config"]["grad_allreduce"]          # the default order of the headline region


def _train_one_rank(reducer_on, use_graph):
    sys.path.insert(0, ROOT)
    from singa_amd import dp, graph as G
    from singa_amd.config import load_config
    from singa_amd.engine import TrainStep
    from singa_amd.model.GAN import SINGA
    from singa_amd.optim import Adam
    torch.manual_seed(11)
    model = SINGA(load_config(lmax=2), device="cuda").eval()
    reducer = dp.GradAllReducer(model, always=True, phases=True) if reducer_on else None      # one-rank group: every collective still runs
    eng = TrainStep(model, Adam(model.parameters(), lr=1e-4), reducer, use_graph=use_graph)
    batch = G.synthetic_batch(3, ids=[100, 101, 102], **KW).to("cuda")
    losses = [float(eng.step(batch).detach()) for _ in range(4)]
    return losses, {n: p.detach().cpu().clone() for n, p in model.named_parameters()}, eng


def test_two_phase_backward_equals_one_backward_call():
    """engine.TrainStep with a reducer that has phases (SINGA.backward_phases): loss -> transformer parameters + embedding
    outputs, all-reduce of the transformer's buckets launched, then embedding outputs -> embedding parameters (eager, and
    as two HIP graphs sharing one pool).  Same parameters after four steps as the plain single-backward step."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        for use_graph in (False, True):
            ref_l, ref_p, _ = _train_one_rank(False, use_graph)                  # one backward() call, no reducer
            l, p, eng = _train_one_rank(True, use_graph)
            assert eng.two_phase and eng.reducer.bucket_phase == sorted(eng.reducer.bucket_phase) and set(eng.reducer.bucket_phase) == {0, 1}
            assert (eng.g_b is not None) == use_graph
            for a, b in zip(l, ref_l):
                assert abs(a - b) < 1e-6 * abs(b), (use_graph, l, ref_l)
            worst = max(float((p[n] - ref_p[n]).norm() / (ref_p[n].norm() + 1e-12)) for n in ref_p)
            assert worst < 1e-6, (use_graph, worst)
    finally:
        dist.destroy_process_group()
