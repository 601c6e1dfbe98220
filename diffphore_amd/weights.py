"""Reference-layout `state_dict` handling: random initialisation (no checkpoint ships) and key validation.

Key names / shapes mirror `TensorProductScoreModel.state_dict()` of the released configuration
(src/models/score_model_phore.py:248-291, 524-641; SURVEY.md §8 "Parameter inventory"), so a real
`best_ema_inference_epoch_model.pt` can be loaded on a machine that has it.  e3nn-internal entries (`*.tp.weight`
empty buffers, `*.tp.output_mask`, compiled `_w3j*` buffers) are accepted and ignored.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict

import torch

from . import config as C


def _linear(sd, prefix, fan_in, fan_out, g, bias=True, scale=1.0):
    bound = 1.0 / math.sqrt(fan_in)
    sd[prefix + ".weight"] = (torch.rand(fan_out, fan_in, generator=g) * 2 - 1) * bound * scale
    if bias:
        sd[prefix + ".bias"] = (torch.rand(fan_out, generator=g) * 2 - 1) * bound * scale


def _embedding(sd, prefix, num, dim, g, scale=1.0):
    bound = math.sqrt(6.0 / (num + dim))          # xavier_uniform (score_model_phore.py:57)
    sd[prefix + ".weight"] = (torch.rand(num, dim, generator=g) * 2 - 1) * bound * scale


def _bn(sd, prefix, irreps, g):
    ns, nf = C.bn_sizes(irreps)
    sd[prefix + ".running_mean"] = torch.randn(ns, generator=g) * 0.1
    sd[prefix + ".running_var"] = torch.rand(nf, generator=g) + 0.5
    sd[prefix + ".weight"] = torch.rand(nf, generator=g) * 0.4 + 0.8
    sd[prefix + ".bias"] = torch.randn(ns, generator=g) * 0.1


def _conv(sd, prefix, n_edge, hidden, numel, out_irreps, g, w_gain):
    _linear(sd, prefix + ".fc.0", n_edge, hidden, g)
    # e3nn tensor-product weights are meant to be O(1); a default-initialised Linear would make every message
    # ~1e-2 of the residual stream and hide kernel errors, so the synthetic last layer carries a gain.
    _linear(sd, prefix + ".fc.3", hidden, numel, g, scale=w_gain)
    _bn(sd, prefix + ".batch_norm", out_irreps, g)


def init_state_dict(seed: int = 20250321, cfg: C.ModelConfig = None, w_gain: float = 3.0,
                    head_gain: float = 0.3, emb_gain: float = 0.25) -> "OrderedDict[str, torch.Tensor]":
    """Random weights in the reference layout (SURVEY.md §8d "Weights"); deterministic in `seed`."""
    cfg = cfg or C.ModelConfig()
    cfg.check_supported()
    g = torch.Generator().manual_seed(seed)
    ns, sd_dim, dd = cfg.ns, cfg.sigma_embed_dim, cfg.distance_embed_dim
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    e = "encoder."
    # Synthetic-weight conditioning (no trained checkpoint ships): the radial MLP makes the TP weights a function of the
    # node scalars, so an un-normalised random net grows ~quadratically per layer.  Small node embeddings (emb_gain) keep
    # the 4-layer stack and 20-step trajectories bounded; the TP weights get w_gain so messages still matter.
    for i in range(len(cfg.clash_cutoff)):
        _embedding(sd, f"{e}boarder_embedding.atom_embedding_list.{i}", 2, ns, g, scale=emb_gain)
    _linear(sd, e + "boarder_embedding.linear", 1, ns, g, scale=0.05 * emb_gain)     # multiplies a distance in Angstrom
    for i, d in enumerate(C.LIG_FEATURE_DIMS):
        _embedding(sd, f"{e}lig_node_embedding.atom_embedding_list.{i}", d, ns, g, scale=emb_gain)
    _linear(sd, e + "lig_node_embedding.linear", sd_dim, ns, g, scale=emb_gain)
    _linear(sd, e + "lig_edge_embedding.0", cfg.in_lig_edge_features + sd_dim + dd, ns, g)
    _linear(sd, e + "lig_edge_embedding.3", ns, ns, g)
    for i, d in enumerate(C.PHORE_FEATURE_DIMS):
        _embedding(sd, f"{e}phore_node_embedding.atom_embedding_list.{i}", d, ns, g, scale=2 * emb_gain)
    _linear(sd, e + "phore_node_embedding.linear", 2 + sd_dim, ns, g, scale=emb_gain)
    _linear(sd, e + "phore_edge_embedding.0", sd_dim + dd, ns, g)
    _linear(sd, e + "phore_edge_embedding.3", ns, ns, g)
    _linear(sd, e + "cross_edge_embedding.0", sd_dim + cfg.cross_distance_embed_dim + 33, ns, g)
    _linear(sd, e + "cross_edge_embedding.3", ns, ns, g)
    sd[e + "lig_distance_expansion.offset"] = torch.linspace(0.0, cfg.lig_max_radius, dd)
    sd[e + "phore_distance_expansion.offset"] = torch.linspace(0.0, cfg.phore_max_radius, dd)
    sd[e + "cross_distance_expansion.offset"] = torch.linspace(0.0, cfg.cross_max_distance, cfg.cross_distance_embed_dim)
    cd = cfg.cross_distance_embed_dim
    _linear(sd, e + "cross_distance_transition.0", cd, cd // 2, g)
    # The reference computes exp(100 * softplus(.) * softplus(.)) un-stabilised (score_model_phore.py:811,839):
    # keep both softplus arguments near 0 so that 100*w stays ~[40,56] (exp finite in fp32) while still varying per edge.
    _linear(sd, e + "cross_distance_transition.3", cd // 2, 1, g, scale=0.3)
    _linear(sd, e + "phore_direction_transition.0", 1, C.NUM_PHORETYPE, g)
    _linear(sd, e + "phore_direction_transition.3", C.NUM_PHORETYPE, 1, g)
    _linear(sd, e + "phoretype_match_transition.0", 3 * C.NUM_PHORETYPE, C.NUM_PHORETYPE, g)
    _linear(sd, e + "phoretype_match_transition.3", C.NUM_PHORETYPE, 1, g, scale=0.3)
    # Put the zero crossing of the direction MLP at the typical weight w0 = 100*softplus(0)^2 so both signs of the
    # cross edge vector occur in synthetic runs (a random init would give one constant sign).
    w0 = torch.tensor([[100.0 * math.log(2.0) ** 2]])
    hid = torch.nn.functional.leaky_relu(torch.nn.functional.linear(
        w0, sd[e + "phore_direction_transition.0.weight"], sd[e + "phore_direction_transition.0.bias"]))
    sd[e + "phore_direction_transition.3.bias"] = -(hid @ sd[e + "phore_direction_transition.3.weight"].T).reshape(1)
    for fam in C.CONV_FAMILIES:
        for l in range(cfg.num_conv_layers):
            _, out_ir = C.layer_irreps(l)
            _conv(sd, f"{e}{fam}.{l}", 3 * ns, 3 * ns, C.layer_numel(l), out_ir, g, w_gain)
    sd["center_distance_expansion.offset"] = torch.linspace(0.0, cfg.center_max_distance, dd)
    _linear(sd, "center_edge_embedding.0", dd + sd_dim, ns, g)
    _linear(sd, "center_edge_embedding.3", ns, ns, g)
    _conv(sd, "final_conv", 2 * ns, 2 * ns, C.tp_paths(C.SEQ[3], C.SH, C.FINAL_OUT)[1], C.FINAL_OUT, g, w_gain)
    _linear(sd, "tr_final_layer.0", 1 + sd_dim, ns, g)
    # untrained score heads would throw the ligand tens of Angstrom per step (score ~ |MLP| / sigma): damp the synthetic heads
    _linear(sd, "tr_final_layer.3", ns, 1, g, scale=head_gain)
    _linear(sd, "rot_final_layer.0", 1 + sd_dim, ns, g)
    _linear(sd, "rot_final_layer.3", ns, 1, g, scale=head_gain)
    _linear(sd, "final_edge_embedding.0", dd, ns, g)
    _linear(sd, "final_edge_embedding.3", ns, ns, g)
    _conv(sd, "tor_bond_conv", 3 * ns, 3 * ns, C.tp_paths(C.SEQ[3], C.TOR_SH, C.TOR_OUT)[1], C.TOR_OUT, g, w_gain)
    _linear(sd, "tor_final_layer.0", 2 * ns, ns, g, bias=False)
    _linear(sd, "tor_final_layer.3", ns, 1, g, bias=False, scale=3 * head_gain)
    return sd


_IGNORED_SUFFIXES = (".tp.weight", ".tp.output_mask", "final_tp_tor.output_mask", "final_tp_tor.weight",
                     "num_batches_tracked")


def clean_state_dict(state_dict: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    """Strip DataParallel 'module.' prefixes (src/utils/utils.py:346-354) and e3nn-internal buffers; check shapes."""
    want = init_state_dict(0)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in state_dict.items():
        if k.startswith("module."):
            k = k[len("module."):]
        if k.endswith(_IGNORED_SUFFIXES) or "._w3j" in k or ".tp._" in k or "final_tp_tor." in k:
            continue
        out[k] = v.detach().to(torch.float32).cpu()
    missing = [k for k in want if k not in out]
    unexpected = [k for k in out if k not in want]
    if missing or unexpected:
        raise KeyError(f"state_dict mismatch: missing={missing[:8]}{'...' if len(missing) > 8 else ''} "
                       f"unexpected={unexpected[:8]}{'...' if len(unexpected) > 8 else ''}")
    for k in want:
        if tuple(out[k].shape) != tuple(want[k].shape):
            raise ValueError(f"shape mismatch for {k}: got {tuple(out[k].shape)}, want {tuple(want[k].shape)}")
    return OrderedDict((k, out[k]) for k in want)
