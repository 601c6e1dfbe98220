"""Oracle restatement of the DiffPhore score model forward (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Follows, function by function, /root/reference/src/models/score_model_phore.py for the released configuration
(SURVEY.md §8, Appendix A).  Pure torch-CPU; `dtype=torch.float64` gives the high-precision answer the fp32 HIP
path is compared against, `torch.float32` is the reference-faithful precision used for the CPU baseline timing.

third-party pieces restated here (absent from /root/reference, see oracle/e3.py for e3nn):
  torch_cluster==1.6.0  radius_graph / radius  (call sites :720, :427): CUDA-kernel semantics - for every centre y,
        scan candidate x of the same graph in index order, keep d^2 < r^2 (strict), stop after max_num_neighbors
        (32 for `radius`; 32+1 for `radius_graph`, which then drops the self loop).
  torch_scatter==2.0.9  scatter(reduce='mean'|'sum')  (:141,:839): sum / max(count,1).
  torch_geometric       to_dense_batch (:908-910): only used for a per-graph masked min; restated directly.

Deviation note: `boarder_analyze` uses torch.cdist (:912), which for >25 points switches to a matmul-based distance;
the oracle uses the direct sqrt(sum (a-b)^2).  (Score-model parity is unpinned anyway.)
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import e3
from .tables import so3_score_norm, torus_score_norm

NS = 20
SEQ = ["20x0e", "20x0e + 10x1o", "20x0e + 10x1o + 10x1e", "20x0e + 10x1o + 10x1e + 20x0o"]


def sinusoidal_embedding(timesteps: torch.Tensor, embedding_dim: int = 20, max_positions: int = 10000) -> torch.Tensor:
    """src/utils/diffusion_utils.py:82-93 (always computed in float32, like the reference)."""
    half = embedding_dim // 2
    emb = math.log(max_positions) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float32) * -emb)
    emb = timesteps.float()[:, None] * emb[None, :]
    return torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)


def timestep_embedding(t: torch.Tensor, scale: float = 10000.0, dim: int = 20) -> torch.Tensor:
    """get_timestep_embedding('sinusoidal') : src/utils/diffusion_utils.py:109-133."""
    return sinusoidal_embedding(scale * t.float(), dim)


def t_to_sigma_f32(t: torch.Tensor, cfg):
    """`self.t_to_sigma(*complex_t)` inside the model (:333): float32 tensors in, float32 pow."""
    t = t.float()
    return (cfg.tr_sigma_min ** (1 - t) * cfg.tr_sigma_max ** t,
            cfg.rot_sigma_min ** (1 - t) * cfg.rot_sigma_max ** t,
            cfg.tor_sigma_min ** (1 - t) * cfg.tor_sigma_max ** t)


def gaussian_smearing(d: torch.Tensor, stop: float, n: int = 20) -> torch.Tensor:
    """GaussianSmearing.forward (:1007-1015): offsets linspace(0,stop,n), coeff = -0.5/step^2 (python float of f32)."""
    offset = torch.linspace(0.0, stop, n)
    coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
    return torch.exp(coeff * (d.reshape(-1, 1) - offset.to(d.dtype).reshape(1, -1)) ** 2)


def radius_neighbors(x: torch.Tensor, batch_x: torch.Tensor, y: torch.Tensor, batch_y: torch.Tensor, r: float,
                     max_num_neighbors: int):
    """torch_cluster.radius (CUDA semantics): returns (row=y index, col=x index), y-major, x ascending, capped.

    Dense restatement: d^2 = ((x - y)^2).sum(-1) in the input precision, strict `<`, first `max_num_neighbors` hits per y
    in x-index order."""
    if y.shape[0] == 0 or x.shape[0] == 0:
        return torch.zeros(2, 0, dtype=torch.long)
    d2 = ((y[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    hit = (d2 < r * r) & (batch_y[:, None] == batch_x[None, :])
    hit = hit & (torch.cumsum(hit.to(torch.int64), dim=1) <= max_num_neighbors)
    rows, cols = torch.nonzero(hit, as_tuple=True)
    return torch.stack([rows, cols])


def radius_graph(pos: torch.Tensor, r: float, batch: torch.Tensor, max_num_neighbors: int = 32) -> torch.Tensor:
    """torch_cluster.radius_graph(loop=False, flow='source_to_target'): edge_index[0]=neighbour, [1]=centre."""
    ei = radius_neighbors(pos, batch, pos, batch, r, max_num_neighbors + 1)
    row, col = ei[1], ei[0]
    keep = row != col
    return torch.stack([row[keep], col[keep]])


def _graph_min(val: torch.Tensor, graph: torch.Tensor, B: int) -> torch.Tensor:
    """per-graph minimum of `val` (1e9 where a graph has no entry)"""
    out = torch.full((B,), 1e9, dtype=torch.float64)
    if val.numel():
        out = out.scatter_reduce(0, graph.long().reshape(-1), val.double().reshape(-1), reduce="amin", include_self=True)
    return out


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    out = src.new_zeros(dim_size, src.shape[1]).index_add_(0, index, src)
    cnt = torch.bincount(index, minlength=dim_size).clamp_min(1).to(src.dtype)
    return out / cnt[:, None]


def scatter_sum(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    return src.new_zeros((dim_size,) + tuple(src.shape[1:])).index_add_(0, index, src)


def angle_vectors(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """:1018-1035"""
    an = a.norm(dim=-1, keepdim=True)
    bn = b.norm(dim=-1, keepdim=True)
    return 2 * torch.atan2((a * bn - an * b).norm(dim=-1), (a * bn + an * b).norm(dim=-1))


class OracleScoreModel:
    """TensorProductScoreModel.forward restated (:294-310).  `state_dict` uses the reference's key layout."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg, dtype=torch.float64, tables=None):
        self.cfg = cfg
        self.dtype = dtype
        self.sd = {k: v.to(dtype) if v.is_floating_point() else v for k, v in state_dict.items()}
        self.tables = tables  # (so3_table, torus_table) float64 numpy; None -> load from diffphore_amd/data
        self.sh_irreps = e3.SH_IRREPS
        self.seq = [e3.parse_irreps(s) for s in SEQ]
        self.final_out = e3.parse_irreps("2x1o + 2x1e")
        self.tor_out = e3.parse_irreps("20x0o + 20x0e")
        self.tor_sh_irreps, _ = e3.full_tp_irreps(self.sh_irreps, e3.parse_irreps("1x2e"))
        self.debug: Dict[str, torch.Tensor] = {}
        # decision margins (tests only): with track_margins the forward records, per graph, how close each DISCONTINUOUS decision of the
        # reference algorithm came to flipping (relative distance from its threshold): radius cuts, d_min <= k, sign of the direction
        # MLP, |d2| < |d1|.  A trajectory that passes one within fp32 resolution has no well-defined fp32 continuation.
        self.track_margins = False
        self.margins: Dict[str, torch.Tensor] = {}

    # ---- small modules -------------------------------------------------------------------------------------------
    def lin(self, name, x):
        return F.linear(x, self.sd[name + ".weight"], self.sd.get(name + ".bias"))

    def mlp(self, name, x, act=F.relu):
        """nn.Sequential(Linear, act, Dropout, Linear) - parameters at indices 0 and 3; dropout is identity in eval."""
        return self.lin(name + ".3", act(self.lin(name + ".0", x)))

    def atom_encoder(self, name, x_cat: torch.Tensor, x_scalar: Optional[torch.Tensor]):
        """AtomEncoder.forward (:64-73)."""
        out = 0
        for i in range(x_cat.shape[1]):
            out = out + self.sd[f"{name}.atom_embedding_list.{i}.weight"][x_cat[:, i].long()]
        if x_scalar is not None:
            out = out + self.lin(name + ".linear", x_scalar)
        return out

    def conv(self, name, irreps_in, irreps_sh, irreps_out, node_attr, edge_index, edge_attr, edge_sh, out_nodes):
        """TensorProductConvLayer.forward (:134-149), residual=False, reduce='mean', e3nn BatchNorm eval."""
        edge_src, edge_dst = edge_index[0], edge_index[1]
        w = self.mlp(name + ".fc", edge_attr)
        tp = e3.fctp_apply(node_attr[edge_dst], edge_sh, w, irreps_in, irreps_sh, irreps_out)
        out = scatter_mean(tp, edge_src, out_nodes)
        bn = name + ".batch_norm"
        return e3.batchnorm_eval(out, irreps_out, self.sd[bn + ".running_mean"], self.sd[bn + ".running_var"],
                                 self.sd[bn + ".weight"], self.sd[bn + ".bias"])

    # ---- graph builders ------------------------------------------------------------------------------------------
    def build_lig_conv_graph(self, data):
        """:715-739"""
        L = data["ligand"]
        dt = self.dtype
        L.node_sigma_emb = timestep_embedding(L.node_t["tr"], self.cfg.embedding_scale, self.cfg.sigma_embed_dim)
        pos = L.pos.to(dt)
        radius_edges = radius_graph(L.pos.float(), self.cfg.lig_max_radius, L.batch)
        if self.track_margins:      # distance of every same-graph atom pair from the 5 A cut, relative (decision: d^2 < r^2)
            B = int(L.batch.max()) + 1
            d2 = ((pos[:, None, :] - pos[None, :, :]) ** 2).sum(-1)
            same = (L.batch[:, None] == L.batch[None, :]) & ~torch.eye(pos.shape[0], dtype=torch.bool)
            rel = (d2 / self.cfg.lig_max_radius ** 2 - 1).abs()
            self.margins = {"lig_radius": _graph_min(rel[same], L.batch[:, None].expand_as(same)[same], B)}
        bond_ei = data["ligand", "ligand"].edge_index.long()
        edge_index = torch.cat([bond_ei, radius_edges], 1)
        edge_attr = torch.cat([data["ligand", "ligand"].edge_attr.to(dt),
                               torch.zeros(radius_edges.shape[1], self.cfg.in_lig_edge_features, dtype=dt)], 0)
        edge_attr = torch.cat([edge_attr, L.node_sigma_emb[edge_index[0]].to(dt)], 1)
        src, dst = edge_index
        edge_vec = pos[dst] - pos[src]
        edge_attr = torch.cat([edge_attr, gaussian_smearing(edge_vec.norm(dim=-1), self.cfg.lig_max_radius)], 1)
        edge_sh = e3.spherical_harmonics_lmax2(edge_vec)
        return edge_index, edge_attr, edge_sh

    def build_phore_conv_graph(self, data):
        """:742-756"""
        P = data["phore"]
        dt = self.dtype
        P.node_sigma_emb = timestep_embedding(P.node_t["tr"], self.cfg.embedding_scale, self.cfg.sigma_embed_dim)
        edge_index = data["phore", "phore"].edge_index.long()
        src, dst = edge_index
        pos = P.pos.to(dt)
        edge_vec = pos[dst] - pos[src]
        edge_attr = torch.cat([P.node_sigma_emb[src].to(dt),
                               gaussian_smearing(edge_vec.norm(dim=-1), self.cfg.phore_max_radius)], 1)
        return edge_index, edge_attr, e3.spherical_harmonics_lmax2(edge_vec)

    def build_cross_conv_graph(self, data):
        """_build_phoretype_cross_conv_graph (:759-895) for new=True, phoretype_match, angle_match, atom_weight='phore',
        multiple=False, use_phore_match_feat=True, use_att=False."""
        L, P = data["ligand"], data["phore"]
        dt = self.dtype
        # dense bipartite edges per graph, sorted by (ligand node, phore node)   (:770-781, :1038-1097)
        srcs, dsts = [], []
        for b in range(int(data.num_graphs)):
            li = torch.nonzero(L.batch == b).flatten()
            pi = torch.nonzero(P.batch == b).flatten()
            srcs.append(li.repeat_interleave(pi.numel()))
            dsts.append(pi.repeat(li.numel()))
        src, dst = torch.cat(srcs), torch.cat(dsts)
        lpos, ppos = L.pos.to(dt), P.pos.to(dt)
        edge_vec = ppos[dst] - lpos[src]
        edge_length_emb = gaussian_smearing(edge_vec.norm(dim=-1), self.cfg.cross_max_distance,
                                            self.cfg.cross_distance_embed_dim)
        edge_attr = torch.cat([L.node_sigma_emb[src].to(dt), edge_length_emb], 1)
        phoretype = P.phoretype.to(dt)
        phorefp = L.phorefp.to(dt)
        is_ex = phoretype[:, -1] == 1
        agreement = phoretype[dst] * phorefp[src]
        agreement = torch.where(is_ex[dst][:, None], torch.zeros_like(agreement), agreement)       # (:790-793)
        phoretype_attr = torch.cat([agreement, phoretype[dst], phorefp[src]], -1)                  # (:796)
        distance = F.softplus(self.mlp("encoder.cross_distance_transition", edge_length_emb))      # (:802)
        feat_match = F.softplus(self.mlp("encoder.phoretype_match_transition", phoretype_attr))    # (:807)
        total_weight = feat_match * (distance * 1.0) * self.cfg.scaler                             # (:803-811)
        d_out = F.leaky_relu(self.mlp("encoder.phore_direction_transition", total_weight, act=F.leaky_relu))
        direction = torch.pow(torch.tensor(-1.0, dtype=dt), (d_out < 0).to(dt))                    # (:814)
        edge_vec = edge_vec * direction
        ew = total_weight.exp()
        atom_weight = ew / scatter_sum(ew, src, L.pos.shape[0])[src]                               # (:839)
        edge_vec = edge_vec * atom_weight                                                          # multiple=False (:845-848)
        edge_attr = torch.cat([edge_attr, phoretype_attr], -1)                                     # (:862)
        # normal-vector rule (:874-889)
        lnorm = L.norm.to(dt)[src].reshape(-1, 11, 3)
        lig_norm = (agreement.unsqueeze(-1) * lnorm).sum(1)
        pn = P.norm.to(dt)[dst]
        rotate_norm = torch.clip(torch.linalg.cross(lig_norm, pn, dim=-1), 1e-12) * agreement.sum(-1, keepdim=True)
        rotate_norm = F.normalize(rotate_norm)
        curr_angle = angle_vectors(lig_norm, pn).unsqueeze(-1)
        a1 = (agreement * L.norm_angle1.to(dt)[src]).sum(-1, keepdim=True)
        a2 = (agreement * L.norm_angle2.to(dt)[src]).sum(-1, keepdim=True)
        d1, d2 = curr_angle - a1, curr_angle - a2
        norm_real = torch.where(d2.abs() < d1.abs(), d2, d1)       # sort(...).indices[:,0]; first wins ties
        rotate_norm = rotate_norm * norm_real
        edge_sh = e3.spherical_harmonics_lmax2(edge_vec)
        edge_norm_sh = e3.spherical_harmonics_lmax2(rotate_norm)
        self.debug.update(cross_dir=direction, cross_aw=atom_weight, cross_tw=total_weight)
        if self.track_margins:
            B = int(L.batch.max()) + 1
            gsrc = L.batch[src]
            # sign of the direction MLP output (:813-816): |z| relative to the magnitude of the terms it is the (cancelling) sum of,
            # z = b3 + sum_o w3_o leaky(w0_o tw + b0_o) - the scale on which fp32 rounding of the sum acts
            pre = "encoder.phore_direction_transition"
            hdir = F.leaky_relu(self.lin(pre + ".0", total_weight))
            w3, b3 = self.sd[pre + ".3.weight"], self.sd[pre + ".3.bias"]
            zdir = F.linear(hdir, w3, b3).reshape(-1)
            scale = (hdir * w3.reshape(1, -1)).abs().sum(-1) + b3.abs().reshape(-1)
            self.margins["cross_sign"] = _graph_min(zdir.abs() / scale.clamp_min(1e-30), gsrc, B)
            # |d2| < |d1| choice of the normal-vector rule (:883-887); only where it changes the result (d1 != d2, agreement present)
            live = ((d1 - d2).abs().reshape(-1) > 0) & (agreement.sum(-1) > 0)
            gap = (d2.abs() - d1.abs()).abs().reshape(-1)
            self.margins["norm_choice"] = _graph_min(gap[live], gsrc[live], B)
        return torch.stack([src, dst]), edge_attr, edge_sh, edge_norm_sh

    def boarder_analyze(self, data):
        """:898-935 (by_radius=False)."""
        L, P = data["ligand"], data["phore"]
        dt = self.dtype
        lpos, ppos = L.pos.to(dt), P.pos.to(dt)
        is_ex = P.phoretype[:, -1] == 1
        # reference: min over the padded dense row of (cdist + (1-mask)*1e9); a masked-out column always exists -> 1e9 cap
        d = (lpos[:, None, :] - ppos[None, :, :]).norm(dim=-1)
        m = (L.batch[:, None] == P.batch[None, :]) & is_ex[None, :]
        dis_min = torch.where(m, d, torch.full_like(d, 1e9)).min(dim=1).values if ppos.shape[0] else torch.full((lpos.shape[0],), 1e9, dtype=dt)
        cut = torch.tensor(list(self.cfg.clash_cutoff), dtype=dt)
        clashed = dis_min[:, None] <= cut[None, :]
        if self.track_margins:      # d_min <= k thresholds (:912-935)
            B = int(L.batch.max()) + 1
            rel = ((dis_min[:, None] - cut[None, :]).abs() / cut[None, :]).min(dim=1).values
            self.margins["boarder"] = _graph_min(rel, L.batch, B)
        return clashed, dis_min[:, None]

    # ---- encoder -------------------------------------------------------------------------------------------------
    def encoder(self, data):
        """LigPhoreEncoder.forward (:644-712)."""
        L, P = data["ligand"], data["phore"]
        dt = self.dtype
        lig_ei, lig_ea, lig_sh = self.build_lig_conv_graph(data)
        h_l = self.atom_encoder("encoder.lig_node_embedding", L.x, L.node_sigma_emb.to(dt))
        clashed, dmin = self.boarder_analyze(data)
        h_l = h_l + self.atom_encoder("encoder.boarder_embedding", clashed.long(), dmin)
        e_ll = self.mlp("encoder.lig_edge_embedding", lig_ea)
        ph_ei, ph_ea, ph_sh = self.build_phore_conv_graph(data)
        h_p = self.atom_encoder("encoder.phore_node_embedding", P.x[:, :3],
                                torch.cat([P.x[:, 3:5].to(dt), P.node_sigma_emb.to(dt)], 1))
        e_pp = self.mlp("encoder.phore_edge_embedding", ph_ea)
        x_ei, x_ea, x_sh, x_nsh = self.build_cross_conv_graph(data)
        e_x = self.mlp("encoder.cross_edge_embedding", x_ea)
        x_l, x_p = x_ei
        x_ei_flip = torch.flip(x_ei, dims=[0])
        self.debug.update(h_l0=h_l, h_p0=h_p, e_ll=e_ll, e_pp=e_pp, e_x=e_x, lig_ei=lig_ei, lig_sh=lig_sh, ph_sh=ph_sh,
                          x_sh=x_sh, x_nsh=x_nsh)
        n_l, n_p = h_l.shape[0], h_p.shape[0]
        for l in range(self.cfg.num_conv_layers):
            ir_in, ir_out = self.seq[min(l, 3)], self.seq[min(l + 1, 3)]
            last = l == self.cfg.num_conv_layers - 1
            a_ll = torch.cat([e_ll, h_l[lig_ei[0], :NS], h_l[lig_ei[1], :NS]], -1)
            u_ll = self.conv(f"encoder.lig_conv_layers.{l}", ir_in, self.sh_irreps, ir_out, h_l, lig_ei, a_ll, lig_sh, n_l)
            a_x = torch.cat([e_x, h_l[x_l, :NS], h_p[x_p, :NS]], -1)
            u_pl = self.conv(f"encoder.phore_to_lig_conv_layers.{l}", ir_in, self.sh_irreps, ir_out, h_p, x_ei, a_x, x_sh, n_l)
            u_pln = self.conv(f"encoder.phore_to_lig_norm_conv_layers.{l}", ir_in, self.sh_irreps, ir_out, h_p, x_ei, a_x,
                              x_nsh, n_l)
            if not last:
                a_pp = torch.cat([e_pp, h_p[ph_ei[0], :NS], h_p[ph_ei[1], :NS]], -1)
                v_pp = self.conv(f"encoder.phore_conv_layers.{l}", ir_in, self.sh_irreps, ir_out, h_p, ph_ei, a_pp, ph_sh, n_p)
                v_lp = self.conv(f"encoder.lig_to_phore_conv_layers.{l}", ir_in, self.sh_irreps, ir_out, h_l, x_ei_flip, a_x,
                                 x_sh, n_p)
                v_lpn = self.conv(f"encoder.lig_to_phore_norm_conv_layers.{l}", ir_in, self.sh_irreps, ir_out, h_l, x_ei_flip,
                                  a_x, x_nsh, n_p)
            h_l = F.pad(h_l, (0, u_ll.shape[-1] - h_l.shape[-1])) + u_ll + u_pl + u_pln
            if not last:
                h_p = F.pad(h_p, (0, v_pp.shape[-1] - h_p.shape[-1])) + v_pp + v_lp + v_lpn
            self.debug[f"h_l{l + 1}"] = h_l
            self.debug[f"h_p{l + 1}"] = h_p
        return h_l, h_p

    # ---- heads ---------------------------------------------------------------------------------------------------
    def forward(self, data):
        """:294-310 + get_trtheta_score (:313-378)."""
        cfg, dt = self.cfg, self.dtype
        L = data["ligand"]
        h_l, _ = self.encoder(data)
        tr_sigma, rot_sigma, tor_sigma = t_to_sigma_f32(data.complex_t["tr"], cfg)[0], \
            t_to_sigma_f32(data.complex_t["rot"], cfg)[1], t_to_sigma_f32(data.complex_t["tor"], cfg)[2]
        B = int(data.num_graphs)
        pos = L.pos.to(dt)
        # centre graph (:381-406)
        n = pos.shape[0]
        c_ei = torch.stack([L.batch, torch.arange(n)])
        center = scatter_sum(pos, L.batch, B) / torch.bincount(L.batch, minlength=B).to(dt)[:, None]
        c_vec = pos[c_ei[1]] - center[c_ei[0]]
        c_attr = torch.cat([gaussian_smearing(c_vec.norm(dim=-1), cfg.center_max_distance),
                            L.node_sigma_emb[c_ei[1]].to(dt)], 1)
        c_sh = e3.spherical_harmonics_lmax2(c_vec)
        c_attr = self.mlp("center_edge_embedding", c_attr)
        c_attr = torch.cat([c_attr, h_l[c_ei[1], :NS]], -1)
        g = self.conv("final_conv", self.seq[3], self.sh_irreps, self.final_out, h_l, c_ei, c_attr, c_sh, B)
        self.debug["global_pred"] = g
        tr_pred = g[:, :3] + g[:, 6:9]
        rot_pred = g[:, 3:6] + g[:, 9:]
        data.graph_sigma_emb = timestep_embedding(data.complex_t["tr"], cfg.embedding_scale, cfg.sigma_embed_dim)
        gse = data.graph_sigma_emb.to(dt)
        tr_norm = torch.linalg.vector_norm(tr_pred, dim=1).unsqueeze(1)
        tr_pred = tr_pred / tr_norm * self.lin("tr_final_layer.3", F.relu(self.lin("tr_final_layer.0", torch.cat([tr_norm, gse], 1))))
        rot_norm = torch.linalg.vector_norm(rot_pred, dim=1).unsqueeze(1)
        rot_pred = rot_pred / rot_norm * self.lin("rot_final_layer.3", F.relu(self.lin("rot_final_layer.0", torch.cat([rot_norm, gse], 1))))
        if cfg.scale_by_sigma:
            tr_pred = tr_pred / tr_sigma.to(dt).unsqueeze(1)
            rot_pred = rot_pred * torch.as_tensor(so3_score_norm(rot_sigma.numpy(), self._tab(0))).to(dt).unsqueeze(1)
        edge_mask = L.edge_mask.bool()
        if cfg.no_torsion or int(edge_mask.sum()) == 0:
            return tr_pred, rot_pred, torch.empty(0, dtype=dt)
        # torsion head (:361-377, :409-437)
        ll_ei = data["ligand", "ligand"].edge_index.long()
        bonds = ll_ei[:, edge_mask]
        bond_pos = (pos[bonds[0]] + pos[bonds[1]]) / 2
        bond_batch = L.batch[bonds[0]]
        t_ei = radius_neighbors(L.pos.float(), L.batch, bond_pos.float(), bond_batch, cfg.lig_max_radius, 32)
        if self.track_margins:
            B = int(L.batch.max()) + 1
            d2 = ((bond_pos[:, None, :] - pos[None, :, :]) ** 2).sum(-1)
            same = bond_batch[:, None] == L.batch[None, :]
            rel = (d2 / cfg.lig_max_radius ** 2 - 1).abs()
            self.margins["tor_radius"] = _graph_min(rel[same], bond_batch[:, None].expand_as(same)[same], B)
        t_vec = pos[t_ei[1]] - bond_pos[t_ei[0]]
        t_attr = self.mlp("final_edge_embedding", gaussian_smearing(t_vec.norm(dim=-1), cfg.lig_max_radius))
        t_sh = e3.spherical_harmonics_lmax2(t_vec)
        bond_vec = pos[bonds[1]] - pos[bonds[0]]
        bond_attr = h_l[bonds[0]] + h_l[bonds[1]]
        bond_sh = e3.spherical_harmonics_2e(bond_vec)
        t_sh45 = e3.full_tp_apply(t_sh, bond_sh[t_ei[0]], self.sh_irreps, e3.parse_irreps("1x2e"))
        t_attr = torch.cat([t_attr, h_l[t_ei[1], :NS], bond_attr[t_ei[0], :NS]], -1)
        q = self.conv("tor_bond_conv", self.seq[3], self.tor_sh_irreps, self.tor_out, h_l, t_ei, t_attr, t_sh45,
                      int(edge_mask.sum()))
        self.debug["tor_q"] = q
        tor_pred = self.lin("tor_final_layer.3", torch.tanh(self.lin("tor_final_layer.0", q))).squeeze(1)
        edge_sigma = tor_sigma[L.batch][ll_ei[0]][edge_mask]
        if cfg.scale_by_sigma:
            tor_pred = tor_pred * torch.sqrt(torch.as_tensor(torus_score_norm(edge_sigma.numpy(), self._tab(1))).float()).to(dt)
        return tr_pred, rot_pred, tor_pred

    __call__ = forward

    def _tab(self, i):
        return None if self.tables is None else self.tables[i]
