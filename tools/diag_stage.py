#!/usr/bin/env python3
"""Diagnostic: per-stage, per-pose errors of one forward vs the fp64 oracle on a synthetic batch (which stage / which pose deviates)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffphore_amd import config, synthetic, weights  # noqa: E402
from diffphore_amd.hetero import Batch  # noqa: E402
from oracle import sampler as osampler  # noqa: E402
from oracle.model import OracleScoreModel  # noqa: E402
from oracle.tables import load_tables  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def main():
    from diffphore_amd.score_model import TensorProductScoreModel
    torch.set_num_threads(32)
    seed, n, nl, nt, npn = (int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (812, 10, 110, 10, 40)))
    gs = synthetic.synthetic_pairs(seed, n, nl, nt, npn)
    sd = weights.init_state_dict()
    om = OracleScoreModel(sd, config.ModelConfig(), dtype=torch.float64, tables=load_tables())
    batch = Batch.from_data_list(gs)
    osampler.set_time_phore(batch, 0.45, 0.45, 0.45, batch.num_graphs)
    om(batch)
    d = om.debug
    os.environ["DIFFPHORE_CONV_ALGO"] = "edge"
    m = TensorProductScoreModel(device="cuda").load_state_dict(sd)
    m(batch)
    torch.cuda.synchronize()
    t = m._engine.ws.t
    print("status", t["status"].cpu().tolist())
    lb, pb = batch["ligand"].batch, batch["phore"].batch
    print("x_emb", rel(t["x_emb"], d["e_x"]), "x_sh", rel(t["x_sh"], d["x_sh"]), "x_nsh", rel(t["x_nsh"], d["x_nsh"]))
    dx = (t["x_sh"].double().cpu() - d["x_sh"].double()).abs().max(1).values
    e = int(dx.argmax())
    print("worst cross edge", e, "hip", t["x_sh"][e].cpu().numpy(), "oracle", d["x_sh"][e].numpy(), "dir", d["cross_dir"][e].numpy() if d["cross_dir"].shape[0] == dx.shape[0] else None)
    x_ptr = m._engine.pb.t["x_ptr"].cpu() if hasattr(m._engine.pb, "t") else None
    print("x_ptr", None if x_ptr is None else x_ptr.tolist())
    order = torch.argsort(batch["phore", "phore"].edge_index[0], stable=True)
    print("pp_emb", rel(t["pp_emb"], d["e_pp"][order]))
    rp, re_, nbr = t["ll_rowptr"].cpu(), t["ll_rowend"].cpu(), t["ll_nbr"].cpu()
    got = sorted((i, int(nbr[k])) for i in range(rp.numel()) for k in range(int(rp[i]), int(re_[i])))
    want = sorted((int(a), int(b)) for a, b in d["lig_ei"].T)
    print("lig edges: got", len(got), "want", len(want), "equal", got == want)
    if got != want:
        sg, sw = set(got), set(want)
        print("  only HIP:", sorted(sg - sw)[:20], "only oracle:", sorted(sw - sg)[:20])
        deg = np.bincount(d["lig_ei"][1].numpy())
        print("  oracle max in-degree", deg.max())
    for key, buf, w in (("h_l4", "h_l0", 100), ("h_l3", "h_l1", 100), ("h_p3", "h_p1", 100), ("h_p2", "h_p0", 80)):
        a, b = t[buf][:, :w].double().cpu(), d[key].double()
        idx = lb if key.startswith("h_l") else pb
        per = [((a[idx == p] - b[idx == p]).abs().max() / b.abs().max()).item() for p in range(n)]
        print(key, " ".join(f"{x:.1e}" for x in per))


if __name__ == "__main__":
    main()

