#!/usr/bin/env python3
"""Wall time of the whole drop-in call `sampling_phore(data_list, ...)`: host packing + H2D, the device-resident loop, D2H write-back
(the figure DESIGN.md quotes next to the HBM-resident bench value)."""
import argparse
import json
import time
import types

import torch

from diffphore_amd import config, synthetic, weights
from diffphore_amd.diffusion import get_t_schedule, t_to_sigma_fn
from diffphore_amd.sampling import randomize_position, sampling_phore
from diffphore_amd.score_model import TensorProductScoreModel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--poses", type=int, default=512)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--max-poses", type=int, default=8192, help="device batch size (several batches are pipelined)")
    a = ap.parse_args()
    c = synthetic.CONFIGS[a.config]
    gs = synthetic.synthetic_pairs(20250321 + a.config, a.poses, c["n_atoms"], c["n_tor"], c["n_phore"])
    cfg = config.ModelConfig()
    args = types.SimpleNamespace(**{k: getattr(cfg, k) for k in ("tr_sigma_min", "tr_sigma_max", "rot_sigma_min", "rot_sigma_max",
                                                                 "tor_sigma_min", "tor_sigma_max")}, no_torsion=False)
    m = TensorProductScoreModel(device="cuda", precision=a.precision).load_state_dict(weights.init_state_dict())
    sched = get_t_schedule(a.steps)
    torch.manual_seed(0)
    out = {}
    for rep in range(2):   # first repetition warms up the library / allocator
        import copy
        dl = copy.deepcopy(gs)
        torch.cuda.synchronize()
        t0 = time.time()
        randomize_position(dl, False, False, args.tr_sigma_max)
        t1 = time.time()
        tm = {}
        sampling_phore(dl, m, a.steps, sched, sched, sched, "cuda", t_to_sigma_fn(args), args, batch_size=20, timing=tm, max_poses=a.max_poses)
        torch.cuda.synchronize()
        t2 = time.time()
        out = dict(poses=a.poses, steps=a.steps, config=a.config, precision=a.precision, randomize_s=t1 - t0, sampling_s=t2 - t1,
                   pose_steps_per_s=a.poses * a.steps / (t2 - t1), poses_per_s=a.poses / (t2 - t1), **tm)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
