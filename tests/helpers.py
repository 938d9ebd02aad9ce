"""Shared test helpers (CPU side): golden loading + synthetic weights."""
import os

import numpy as np
import torch

from oracle import weights as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["3wi2_4tpp", "4agq_5a7b", "5cp5_4nue"]


def state_from_spec(tag):
    z = np.load(os.path.join(GOLDEN, f"param_spec_{tag}.npz"))
    spec = [(str(n), tuple(int(v) for v in str(s).split(",")) if str(s) else (), float(m), float(sd))
            for n, s, m, sd in zip(z["names"], z["shapes"], z["mean"], z["std"])]
    return W.synth_state(spec)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rel_err(a, b):
    a, b = torch.as_tensor(a, dtype=torch.float64), torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-30))


def product_batch(names, z=None, device="cuda", with_lap=False):
    """Golden graphs as a singa_amd HeteroGraph batch; `z` (a golden npz) pins rot-mats / kNN lists / lap-PE."""
    import os as _os

    from singa_amd import graph as G
    b = G.collate([G.load_npz(_os.path.join(GOLDEN, f"graph_{n}.npz"), with_lap=with_lap) for n in names])
    if z is not None:
        b.extras["edge_rot_mat"] = {k: torch.tensor(z[f"rot_{k}"]) for k in ("pp", "ll", "lp")}
        if "knn_p" in z.files:
            b.extras["knn"] = {G.PA: torch.tensor(z["knn_p"]), G.LA: torch.tensor(z["knn_l"])}
            b.nodes[G.PA]["lap_pe"], b.nodes[G.LA]["lap_pe"] = torch.tensor(z["lap_p"]), torch.tensor(z["lap_l"])
    return b.to(device)


BEAM_CASES = ["b1_k20", "b2_k4", "b2_k6_eos", "b2_k5_flat"]
SMI_VOC = None


def smi_voc():
    global SMI_VOC
    if SMI_VOC is None:
        import yaml
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        SMI_VOC = list(yaml.safe_load(open(os.path.join(root, "config", "train.yml")))["model"]["decoder"]["smiVoc"])
    return SMI_VOC


def apply_beam_gains(weight, z):
    """The two documented rescalings of the vocabulary projection a beam golden was made with (oracle/make_golden_beam.py)."""
    with torch.no_grad():
        weight.mul_(float(z["proj_gain"]))
        weight[smi_voc().index("$")] *= float(z["eos_gain"])


def grad_sample_errors(named_grads, z, tol):
    """Compare the element-wise gradient samples of a singa_L*_B3 golden (up to 512 evenly spaced elements of every
    parameter's gradient, oracle/make_golden.py) with `named_grads` = {name: grad or None}.  Returns the list of
    parameters whose samples differ by more than `tol` relative (L2 over the parameter's samples)."""
    bad, off = [], 0
    flat = z["grad_samples"]
    for n, ref in zip(z["grad_names"], z["grad_norms"]):
        if ref < 0:
            continue
        gr = named_grads[str(n)]
        idx = W.sample_index(gr.numel())
        want = torch.as_tensor(flat[off:off + len(idx)], dtype=torch.float64)
        off += len(idx)
        got = gr.detach().reshape(-1).cpu()[torch.as_tensor(idx)].double()
        err = float((got - want).norm() / (want.norm() + 1e-12))
        if err > tol and float((got - want).abs().max()) > 1e-7:
            bad.append((str(n), err))
    assert off == len(flat), (off, len(flat))
    return bad
