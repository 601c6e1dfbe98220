#!/usr/bin/env python3
"""Diagnostic: forward of >96-atom ligands through every conv configuration vs the fp64 oracle (which layer / which switch deviates)."""
import os
import sys

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
    sd = weights.init_state_dict()
    om = OracleScoreModel(sd, config.ModelConfig(), dtype=torch.float64, tables=load_tables())
    a, b = synthetic.synthetic_pairs(811, 10, 100, 14, 24), synthetic.synthetic_pairs(812, 10, 110, 10, 40)
    for name, gs in (("10x100/24", a), ("10x110/40", b), ("mixed", a + b), ("mixed-rev", b + a), ("20x100", a + synthetic.synthetic_pairs(815, 10, 100, 14, 24))):
        batch = Batch.from_data_list(gs)
        osampler.set_time_phore(batch, 0.45, 0.45, 0.45, batch.num_graphs)
        tr_o, rot_o, tor_o = om(batch)
        for env in (dict(DIFFPHORE_CONV_ALGO="edge"), dict(DIFFPHORE_CONV_ALGO="node"), dict(DIFFPHORE_CONV_ALGO="node", DIFFPHORE_CONV_FUSE="0"),
                    dict(DIFFPHORE_CONV_ALGO="node", DIFFPHORE_TWO_STREAMS="0"), dict(DIFFPHORE_CONV_ALGO="node", DIFFPHORE_TWO_STREAMS="0", DIFFPHORE_CONV_FUSE="0")):
            for k in ("DIFFPHORE_CONV_ALGO", "DIFFPHORE_CONV_FUSE", "DIFFPHORE_TWO_STREAMS"):
                os.environ.pop(k, None)
            os.environ.update(env)
            m = TensorProductScoreModel(device="cuda").load_state_dict(sd)
            tr, rot, tor = m(batch)
            torch.cuda.synchronize()
            t = m._engine.ws.t
            e = dict(tr=rel(tr, tr_o), h_l4=rel(t["h_l0"][:, :100], om.debug["h_l4"]), h_l3=rel(t["h_l1"][:, :100], om.debug["h_l3"]),
                     h_p3=rel(t["h_p1"][:, :100], om.debug["h_p3"]), h_p2=rel(t["h_p0"][:, :80], om.debug["h_p2"]))
            print(name, env, {k: f"{v:.1e}" for k, v in e.items()}, flush=True)


if __name__ == "__main__":
    main()
