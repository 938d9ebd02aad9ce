"""Lab: distribution of the per-parameter relative gradient errors (HIP path vs CPU oracle) on two-graph batches of the
bench generators - what the tolerances of tests/test_workloads_gpu.py are set from."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import singa_oracle as O
from singa_amd import graph as G
from singa_amd.config import load_config
from singa_amd.model.GAN import SINGA

for workload in ("cfg2_b32_l2", "cfg3_b128_l4", "cfg5_l6"):
    L, kw, _, _ = G.resolve_workload(workload)
    graphs = [G.synthetic_graph(i, **G.graph_sizes(i, **kw)) for i in (3, 4)]
    torch.manual_seed(7)
    model = SINGA(load_config(lmax=L), device="cuda").eval()
    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    b, rots, lap_p, lap_l = O.batch_from_graphs(graphs)
    O.train_step_loss(sd, b, rots, L, lap_p, lap_l).backward()
    batch = G.collate(graphs).to("cuda")
    logits = model(batch)
    torch.nn.functional.cross_entropy(logits, batch["ligand_data"]["smiIndices_tgt"].reshape(-1)).backward()
    total = float(torch.sqrt(sum((v.grad.double() ** 2).sum() for v in sd.values() if v.grad is not None)))
    errs = []
    for name, p in model.named_parameters():
        go = sd[name].grad
        if go is None or float(go.norm()) == 0.0:
            continue
        diff = p.grad.detach().cpu().double() - go.double()
        errs.append((float(diff.norm() / go.double().norm()), float(diff.norm()) / total, name))
    errs.sort(reverse=True)
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
    print(f"{workload}: total grad norm rel err {abs(gn - total) / total:.2e}; worst per-parameter relative errors:")
    for e, a, n in errs[:8]:
        print(f"    {e:.2e} (abs / total norm {a:.1e})  {n}")
    real = [(e, a, n) for e, a, n in errs if a > 1e-6]
    print("    worst among parameters whose error is above 1e-6 of the total gradient norm:")
    for e, a, n in real[:6]:
        print(f"    {e:.2e} (abs / total norm {a:.1e})  {n}")
    import statistics
    print(f"    median {statistics.median(e for e, _, _ in errs):.2e}; parameters above 1e-3: {sum(e > 1e-3 for e, _, _ in errs)}, above 2e-4: "
          f"{sum(e > 2e-4 for e, _, _ in errs)} of {len(errs)}")
