"""Host side of the HIP hot path: parameter upload, batch packing (struct of arrays + CSR), workspace, launches.

PyTorch is used for device memory and streams only; every arithmetic step of the path runs in libdiffphore_hip.so.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib
from . import config as cfgmod
from . import packing
from .weights import clean_state_dict

VOIDP, c_i32, c_f32 = C.c_void_p, C.c_int32, C.c_float
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
DEFAULT_PRECISION = "f32"


# ----------------------------------------------------------------------------------------------------------------------
# ctypes mirrors of include/diffphore_amd.h
# ----------------------------------------------------------------------------------------------------------------------
# the cross pairs with agreement != 0 (DpBatch.xs_* by atom, xt_* by pharmacophore node: include/diffphore_amd.h)
_XS_ARRAYS = ("xs_src", "xs_nbr", "xs_feat", "xs_rowptr", "xt_src", "xt_nbr", "xt_feat", "xt_rowptr")


class DpBatch(C.Structure):
    _fields_ = ([("B", c_i32)] + [(n, c_i32) for n in ("n_lig", "n_ph", "n_bond", "n_tor", "n_pp", "n_x", "ll_cap", "n_llp")]
                + [(n, c_i32) for n in ("max_nl", "max_np", "max_t", "no_tail_nl", "no_tail_np")]
                + [(n, VOIDP) for n in ("lig_ptr", "ph_ptr", "bond_ptr", "tor_ptr", "pp_ptr", "x_ptr", "ll_ptr", "mask_ptr",
                                        "pose_ptr", "llp_ptr", "lig_x", "bond_src", "bond_dst", "bond_type", "tor_u", "tor_v",
                                        "mask_rotate", "phorefp", "norm_angle1", "norm_angle2", "ph_pos", "ph_norm", "ph_x",
                                        "ph_scalar", "ph_type", "pp_src", "pp_dst", "pp_rowptr", "pp_rowend", "pos", "norm",
                                        "t")]
                + [("n_xs", c_i32)] + [(n, VOIDP) for n in _XS_ARRAYS] + [("ph_shared", c_i32)])


class DpGraphView(C.Structure):
    _fields_ = ([(n, c_i32) for n in ("n_atoms", "n_phore", "n_bond_edges", "n_tor", "n_pp")]
                + [(n, VOIDP) for n in ("lig_x", "bond_index", "bond_attr", "edge_mask", "mask_rotate", "phorefp", "norm_angle1", "norm_angle2",
                                        "ph_x", "ph_pos", "ph_norm", "ph_type", "pp_index")])


class DpPoseRef(C.Structure):
    _fields_ = [("view", VOIDP), ("pos", VOIDP), ("norm", VOIDP)]


_PACK_ARRAYS = ("lig_ptr", "ph_ptr", "bond_ptr", "tor_ptr", "pp_ptr", "x_ptr", "ll_ptr", "mask_ptr", "pose_ptr", "llp_ptr",
                "lig_x", "bond_src", "bond_dst", "bond_type", "tor_u", "tor_v", "mask_rotate", "phorefp", "norm_angle1", "norm_angle2",
                "ph_pos", "ph_norm", "ph_x", "ph_scalar", "ph_type", "pp_src", "pp_dst", "pp_rowptr", "pp_rowend", "pos", "norm") + _XS_ARRAYS


class DpPackLayout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in _PACK_ARRAYS] + [("bytes", C.c_int64)]


class DpMlp(C.Structure):
    _fields_ = [(n, VOIDP) for n in ("w0", "b0", "w3", "b3")]


class DpConvW(C.Structure):
    _fields_ = [(n, VOIDP) for n in ("w1p", "w2p", "bn_scale", "bn_shift", "w1n", "w2n", "w2x", "bn_scale_x", "w1h", "w2h")]


class DpWeights(C.Structure):
    _fields_ = ([(n, VOIDP) for n in ("lig_emb", "lig_lin_w", "lig_lin_b", "boarder_emb", "boarder_lin_w", "boarder_lin_b",
                                      "ph_emb", "ph_lin_w", "ph_lin_b")]
                + [(n, DpMlp) for n in ("lig_edge", "ph_edge", "x_edge", "cdt", "pdt", "pmt", "center_edge", "final_edge",
                                        "tr_final", "rot_final")]
                + [("tor_w0", VOIDP), ("tor_w3", VOIDP)]
                + [(n, DpConvW * 4) for n in ("lig_conv", "ph_conv", "l2p_conv", "p2l_conv", "l2p_norm_conv", "p2l_norm_conv")]
                + [("final_conv", DpConvW), ("tor_conv", DpConvW), ("so3_table", VOIDP), ("torus_table", VOIDP)]
                + [(n, VOIDP) for n in ("lig_gauss_mu", "ph_gauss_mu", "x_gauss_mu", "center_gauss_mu", "emb_freq")]
                + [(n, c_f32) for n in ("lig_gauss_coeff", "ph_gauss_coeff", "x_gauss_coeff", "center_gauss_coeff", "scaler",
                                        "tr_smin", "tr_smax", "rot_smin", "rot_smax", "tor_smin", "tor_smax", "emb_scale")]
                + [("precision", c_i32)])


class DpWork(C.Structure):
    _fields_ = ([(n, VOIDP) for n in ("lig_x_emb", "ph_x_emb", "pair_pre", "pair_fm", "pp_pre", "pp_sh", "tparams",
                                      "ll_rowptr", "ll_rowend", "ll_nbr", "ll_src", "ll_emb", "ll_sh", "ll_pair_rad", "ll_pair_bond", "pp_emb", "x_emb", "x_sh",
                                      "x_nsh", "x_ew", "x_emb_bf", "ll_emb_bf")]
                + [("h_l", VOIDP * 2), ("h_p", VOIDP * 2)]
                + [(n, VOIDP) for n in ("c_emb", "c_sh", "gpred", "t_rowptr", "t_rowend", "t_nbr", "t_src", "t_emb", "t_sh",
                                        "t_hb", "t_q", "status", "prof_start", "prof_stop")]
                + [("prof_layer", c_i32), ("prof_conv", c_i32), ("prof_events", VOIDP), ("conv_algo", c_i32)]
                + [(n, VOIDP) for n in ("aux_stream", "ev_fork", "ev_join")] + [("aux_mode", c_i32), ("ll_node_max_atoms", c_i32), ("edge_bf16x6", c_i32), ("conv_part", VOIDP), ("z_noise", VOIDP), ("path_split", c_i32), ("conv_fuse", c_i32), ("node_f16", c_i32),
                                                                                 ("aux2_stream", VOIDP), ("ev_join2", VOIDP), ("ev_x2", VOIDP), ("pp_tmp", VOIDP), ("pp_msg", VOIDP), ("xs_msg", VOIDP), ("xs_place", c_i32), ("xn_sparse", c_i32),
                                                                                 ("t_uniform", c_i32), ("ph_share", c_i32), ("t_row", VOIDP), ("tp_table", VOIDP), ("tp_row", VOIDP)])


class DpStepParams(C.Structure):
    _fields_ = [(n, c_f32) for n in ("tr_c1", "tr_c2", "rot_c1", "rot_c2", "tor_c1", "tor_c2")] + [("no_torsion", c_i32)]


class DpSegment(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("offset", C.c_int64), ("numel", C.c_int64)]


class DpModelScalars(C.Structure):
    _fields_ = [(n, c_f32) for n in ("lig_gauss_coeff", "ph_gauss_coeff", "x_gauss_coeff", "center_gauss_coeff", "scaler", "tr_smin",
                                     "tr_smax", "rot_smin", "rot_smax", "tor_smin", "tor_smax", "emb_scale")]


class DpSchedule(C.Structure):
    _fields_ = [("steps", c_i32), ("sp", C.POINTER(DpStepParams)), ("t", VOIDP), ("z_tr", VOIDP), ("z_rot", VOIDP), ("z_tor", VOIDP),
                ("noise_mode", c_i32), ("uid_base", c_i32), ("pose_uid", VOIDP), ("seed", C.c_uint64)]


def _declare():
    L = _lib.lib()
    if getattr(L, "_dp_engine_declared", False):
        return L
    L.dp_prepare.argtypes = [C.POINTER(DpBatch), C.POINTER(DpWeights), C.POINTER(DpWork), VOIDP]
    L.dp_prepare.restype = C.c_int
    L.dp_forward.argtypes = [C.POINTER(DpBatch), C.POINTER(DpWeights), C.POINTER(DpWork), VOIDP, VOIDP, VOIDP, VOIDP]
    L.dp_forward.restype = C.c_int
    L.dp_forward_launch_of.argtypes = [C.POINTER(DpBatch), C.POINTER(DpWeights), C.POINTER(DpWork), c_i32, c_i32] + [C.POINTER(c_i32)] * 3
    L.dp_forward_launch_of.restype = C.c_int
    L.dp_noise_normals.argtypes = [C.c_uint64, c_i32, C.POINTER(DpBatch), VOIDP, c_i32, VOIDP, VOIDP, VOIDP, VOIDP]
    L.dp_noise_normals.restype = C.c_int
    L.dp_randomize_draws.argtypes = [C.c_uint64, C.POINTER(DpBatch), VOIDP, c_i32, c_f32, VOIDP, VOIDP, VOIDP, VOIDP]
    L.dp_randomize_draws.restype = C.c_int
    L.dp_conformer_update.argtypes = [C.POINTER(DpBatch), C.POINTER(DpStepParams)] + [VOIDP] * 8
    L.dp_conformer_update.restype = C.c_int
    L.dp_sample_step.argtypes = [C.POINTER(DpBatch), C.POINTER(DpWeights), C.POINTER(DpWork), C.POINTER(DpStepParams)] + [VOIDP] * 7
    L.dp_sample_step.restype = C.c_int
    L.dp_work_bytes.argtypes = [C.POINTER(DpBatch)]
    L.dp_work_bytes.restype = C.c_size_t
    L.dp_work_bind.argtypes = [C.POINTER(DpBatch), VOIDP, C.c_size_t, C.POINTER(DpWork)] + [C.POINTER(VOIDP)] * 3
    L.dp_work_bind.restype = C.c_int
    L.dp_work_init.argtypes = [C.POINTER(DpBatch), C.POINTER(DpWork), VOIDP]
    L.dp_work_init.restype = C.c_int
    L.dp_weights_segments.argtypes = [c_i32, C.POINTER(DpSegment), c_i32]
    L.dp_weights_segments.restype = C.c_int
    L.dp_weights_blob_floats.argtypes = [c_i32]
    L.dp_weights_blob_floats.restype = C.c_int64
    L.dp_weights_bind.argtypes = [VOIDP, c_i32, C.POINTER(DpModelScalars), C.POINTER(DpWeights)]
    L.dp_weights_bind.restype = C.c_int
    L.dp_sample.argtypes = [C.POINTER(DpBatch), C.POINTER(DpWeights), C.POINTER(DpWork), C.POINTER(DpSchedule), VOIDP, VOIDP, VOIDP, VOIDP]
    L.dp_sample.restype = C.c_int
    L.dp_sample_graph_create.argtypes = [C.POINTER(DpBatch), C.POINTER(DpWeights), C.POINTER(DpWork), C.POINTER(DpSchedule), VOIDP, VOIDP,
                                         VOIDP, C.POINTER(VOIDP)]
    L.dp_sample_graph_create.restype = C.c_int
    L.dp_sample_graph_launch.argtypes = [VOIDP, VOIDP]
    L.dp_sample_graph_launch.restype = C.c_int
    L.dp_sample_graph_destroy.argtypes = [VOIDP]
    L.dp_sample_graph_destroy.restype = C.c_int
    L.dp_resolve_conv_algo.argtypes = [c_i32, c_i32, c_i32]
    L.dp_resolve_conv_algo.restype = C.c_int
    L.dp_aux_create.argtypes = [C.POINTER(VOIDP)] * 3
    L.dp_aux_create.restype = C.c_int
    L.dp_aux_destroy.argtypes = [VOIDP] * 3
    L.dp_aux_destroy.restype = C.c_int
    L.dp_pack_query.argtypes = [c_i32, VOIDP, C.POINTER(DpBatch), C.POINTER(DpPackLayout)]
    L.dp_pack_query.restype = C.c_int
    L.dp_pack_fill.argtypes = [c_i32, VOIDP, C.POINTER(DpPackLayout), VOIDP, C.POINTER(c_i32)]
    L.dp_pack_fill.restype = C.c_int
    L.dp_pack_bind.argtypes = [C.POINTER(DpPackLayout), VOIDP, c_i32, C.POINTER(DpBatch)]
    L.dp_pack_bind.restype = C.c_int
    L._dp_engine_declared = True
    return L


def load_tables():
    """SO(3) / torus score-norm tables generated by the reference's own code (tests/golden/make_golden.py)."""
    return (np.load(os.path.join(_DATA, "so3_exp_score_norms.npy")), np.load(os.path.join(_DATA, "torus_score_norm.npy")))


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ----------------------------------------------------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------------------------------------------------
class DeviceModel:
    """Uploads a reference-layout state_dict (re-ordered for the kernels) and owns the DpWeights struct."""

    def __init__(self, state_dict, cfg: cfgmod.ModelConfig, device, tables=None, precision: str = None):
        cfg.check_supported()
        precision = precision or os.environ.get("DIFFPHORE_PRECISION", DEFAULT_PRECISION)
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}")
        self.precision = precision
        prec = _lib.PRECISIONS[precision]
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.DiffPhoreLibraryError("diffphore_amd runs on an AMD GPU (device 'cuda' under PyTorch-ROCm); no CPU path exists")
        _declare()
        sd = clean_state_dict(state_dict)
        self.host: Dict[str, torch.Tensor] = {}
        H = self.host
        e = "encoder."
        H["lig_emb"] = torch.cat([sd[f"{e}lig_node_embedding.atom_embedding_list.{i}.weight"] for i in range(16)], 0)
        H["lig_lin_w"], H["lig_lin_b"] = sd[e + "lig_node_embedding.linear.weight"], sd[e + "lig_node_embedding.linear.bias"]
        H["boarder_emb"] = torch.stack([sd[f"{e}boarder_embedding.atom_embedding_list.{i}.weight"] for i in range(5)], 0)
        H["boarder_lin_w"], H["boarder_lin_b"] = sd[e + "boarder_embedding.linear.weight"].reshape(-1), sd[e + "boarder_embedding.linear.bias"]
        H["ph_emb"] = torch.cat([sd[f"{e}phore_node_embedding.atom_embedding_list.{i}.weight"] for i in range(3)], 0)
        H["ph_lin_w"], H["ph_lin_b"] = sd[e + "phore_node_embedding.linear.weight"], sd[e + "phore_node_embedding.linear.bias"]
        mlps = dict(lig_edge=e + "lig_edge_embedding", ph_edge=e + "phore_edge_embedding", x_edge=e + "cross_edge_embedding",
                    cdt=e + "cross_distance_transition", pdt=e + "phore_direction_transition", pmt=e + "phoretype_match_transition",
                    center_edge="center_edge_embedding", final_edge="final_edge_embedding", tr_final="tr_final_layer",
                    rot_final="rot_final_layer")
        for name, pre in mlps.items():
            for part, key in (("w0", ".0.weight"), ("b0", ".0.bias"), ("w3", ".3.weight"), ("b3", ".3.bias")):
                H[f"{name}.{part}"] = sd[pre + key]
        H["tor_w0"], H["tor_w3"] = sd["tor_final_layer.0.weight"], sd["tor_final_layer.3.weight"]
        fams = dict(lig_conv="lig_conv_layers", ph_conv="phore_conv_layers", l2p_conv="lig_to_phore_conv_layers",
                    p2l_conv="phore_to_lig_conv_layers", l2p_norm_conv="lig_to_phore_norm_conv_layers",
                    p2l_norm_conv="phore_to_lig_norm_conv_layers")
        for name, fam in fams.items():
            for l in range(4):
                node = name != "ph_conv" and prec == _lib.PREC_F32
                node_bf = name != "ph_conv" and prec == _lib.PREC_BF16
                # edge-side in an fp32 model (phore-phore always, ligand-ligand for large ligands): two-limb fp16 second layer as well (w2x, bnx)
                # ... and the two normals convs: the pairs whose harmonics are not e0 run edge-side as an explicit list (DP_EDGES_LIST)
                x6 = (((name == "ph_conv" and l < 3) or name in ("lig_conv", "l2p_norm_conv", "p2l_norm_conv")) and prec == _lib.PREC_F32) or \
                    (name in ("l2p_norm_conv", "p2l_norm_conv") and prec == _lib.PREC_BF16)
                # node-side in an fp32 model: the fp32-grade two-limb fp16 fragments as well (DP_PREC_F16X3: what those convs run on by default)
                for part, t in packing.pack_conv(sd, f"{e}{fam}.{l}", l, prec, node_side=node, x6=x6, node_bf16=node_bf, node_f16=node).items():
                    H[f"{name}.{l}.{part}"] = t
        for part, t in packing.pack_conv(sd, "final_conv", _lib.CONV_FINAL, prec).items():
            H[f"final_conv.{part}"] = t
        for part, t in packing.pack_conv(sd, "tor_bond_conv", _lib.CONV_TOR, prec, x6=prec == _lib.PREC_F32).items():
            H[f"tor_conv.{part}"] = t
        so3, torus = tables if tables is not None else load_tables()
        H["so3_table"] = torch.as_tensor(np.asarray(so3), dtype=torch.float64).float()
        H["torus_table"] = torch.as_tensor(np.asarray(torus), dtype=torch.float64).float()
        H["lig_gauss_mu"] = sd[e + "lig_distance_expansion.offset"]
        H["ph_gauss_mu"] = sd[e + "phore_distance_expansion.offset"]
        H["x_gauss_mu"] = sd[e + "cross_distance_expansion.offset"]
        H["center_gauss_mu"] = sd["center_distance_expansion.offset"]
        half = cfg.sigma_embed_dim // 2            # diffusion_utils.py:85-87 (float32, torch CPU)
        H["emb_freq"] = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
        self._upload()

    # one flat blob so multi-GPU start-up is a single RCCL broadcast; its layout is defined by the C library
    # (dp_weights_segments), so that a host without this module can build and bind the same blob
    def segments(self):
        L = _declare()
        prec = _lib.PRECISIONS[self.precision]
        n = L.dp_weights_segments(prec, None, 0)
        segs = (DpSegment * n)()
        assert L.dp_weights_segments(prec, segs, n) == n
        return [(sg.name.decode(), int(sg.offset), int(sg.numel)) for sg in segs]

    def _upload(self, blob: Optional[torch.Tensor] = None):
        L = _declare()
        segs = self.segments()
        missing = [n for n, _, _ in segs if n not in self.host]
        extra = [n for n in self.host if n not in {x[0] for x in segs}]
        if missing or extra:
            raise _lib.DiffPhoreLibraryError(f"parameter blob layout mismatch: missing {missing[:5]}, unexpected {extra[:5]}")
        total = int(L.dp_weights_blob_floats(_lib.PRECISIONS[self.precision]))
        if blob is None:
            flat = torch.zeros(total, dtype=torch.float32)
            for n, off, numel in segs:
                t = self.host[n].reshape(-1).float()
                if t.numel() != numel:
                    raise _lib.DiffPhoreLibraryError(f"parameter segment {n}: {t.numel()} floats, the library expects {numel}")
                flat[off:off + numel] = t
            blob = flat.to(self.device)
        assert blob.numel() == total
        self.blob = blob
        self.dev = {n: blob[off:off + numel] for n, off, numel in segs}
        self._fill_struct()

    def blob_numel(self) -> int:
        return int(self.blob.numel())

    def _fill_struct(self):
        L, cfg = _declare(), self.cfg

        def coeff(mu):                             # GaussianSmearing.__init__ (:1009-1010)
            return -0.5 / (mu[1] - mu[0]).item() ** 2
        sc = DpModelScalars()
        sc.lig_gauss_coeff, sc.ph_gauss_coeff = coeff(self.host["lig_gauss_mu"]), coeff(self.host["ph_gauss_mu"])
        sc.x_gauss_coeff, sc.center_gauss_coeff = coeff(self.host["x_gauss_mu"]), coeff(self.host["center_gauss_mu"])
        sc.scaler = cfg.scaler
        sc.tr_smin, sc.tr_smax = cfg.tr_sigma_min, cfg.tr_sigma_max
        sc.rot_smin, sc.rot_smax = cfg.rot_sigma_min, cfg.rot_sigma_max
        sc.tor_smin, sc.tor_smax = cfg.tor_sigma_min, cfg.tor_sigma_max
        sc.emb_scale = cfg.embedding_scale
        w = DpWeights()
        _lib.check(L.dp_weights_bind(self.blob.data_ptr(), _lib.PRECISIONS[self.precision], C.byref(sc), C.byref(w)), "dp_weights_bind")
        self.scalars = sc
        self.struct = w


# ----------------------------------------------------------------------------------------------------------------------
# batch packing
# ----------------------------------------------------------------------------------------------------------------------
def _i32(x):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.int32)


# Host <-> device copies go through a side stream.  A synchronous copy on the launch stream waits for every kernel already queued there;
# the sampler keeps the NEXT device batch's uploads and the PREVIOUS batch's write-back off that queue so that the GPU always has a loop
# to run (sampling._run_parts).  Ordering is by the host: an upload has landed when h2d returns (pageable source: the call blocks on the
# side stream only), a download is issued after the producing loop's event has been waited for.
_COPY_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def _copy_stream(device) -> "torch.cuda.Stream":
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if idx not in _COPY_STREAMS:
        _COPY_STREAMS[idx] = torch.cuda.Stream(device=idx)
    return _COPY_STREAMS[idx]


def h2d(t: Optional[torch.Tensor], device) -> Optional[torch.Tensor]:
    """CPU tensor -> `device` without touching the launch stream's queue (device tensors pass through)."""
    if t is None or t.device.type != "cpu":
        return t if t is None else t.to(device)
    # The destination comes from the COPY stream's pool of the caching allocator: a block the launch stream has just freed may still be
    # read by a kernel that is queued but has not run (a temporary of the batch being enqueued), and this copy executes at once.
    # record_stream hands the block's lifetime to the launch stream, where its consumers are queued.
    with torch.cuda.stream(_copy_stream(device)):
        dst = torch.empty(t.shape, dtype=t.dtype, device=device)
        dst.copy_(t)
    dst.record_stream(torch.cuda.current_stream(dst.device))
    return dst


def d2h(t: torch.Tensor) -> torch.Tensor:
    """device tensor -> CPU on the side stream; the caller has waited for the kernels that produced it"""
    if t.device.type == "cpu":
        return t
    with torch.cuda.stream(_copy_stream(t.device)):
        return t.cpu()


class _PinnedStage:
    """One page-locked staging buffer per device for the packed batch image: dp_pack_fill writes straight into it and the upload is one
    DMA (a pageable source is staged by the runtime first: ~3x slower for the 5 MB of a 512-pose batch).  Leased to ONE packed batch at a
    time (a batch packed while the lease is taken uses ordinary memory); released when its upload has been enqueued, reused after that
    copy's event."""
    _by_dev: Dict[int, "_PinnedStage"] = {}

    def __init__(self):
        self.buf, self.busy, self.event = None, False, None

    @classmethod
    def lease(cls, device, nbytes):
        if device.type != "cuda" or not torch.cuda.is_available():
            return None
        idx = device.index if device.index is not None else torch.cuda.current_device()
        st = cls._by_dev.setdefault(idx, cls())
        if st.busy:
            return None
        if st.buf is None or st.buf.numel() < nbytes:
            try:
                st.buf = torch.empty(max(int(nbytes * 1.25), 1 << 22), dtype=torch.uint8, pin_memory=True)
            except RuntimeError:
                return None
            st.event = None
        if st.event is not None:
            st.event.synchronize()
        st.busy = True
        return st


class _GraphView:
    """HOST view of one complex's static arrays for the native collate (DpGraphView, csrc/dp_pack.hip): typed, contiguous CPU arrays (the
    complex's own tensors wherever they already are) + the C struct pointing at them.  Cached on the graph object and shared by its samples
    (`HeteroData.sample_copy` copies the attribute); `key` = identity + in-place version of every source object, re-checked once per device
    batch, so a replaced or in-place edited static tensor rebuilds the view."""
    __slots__ = ("struct", "addr", "keep", "key", "n", "src", "_tver", "_other")

    # The view holds a ctypes struct of raw pointers: it is a cache, never state.  A pickled / deep-copied graph simply carries no view (None)
    # and rebuilds it at its next packing - torch.save, multiprocessing hand-off and copy.deepcopy of a packed graph keep working.
    def __reduce__(self):
        return (type(None), ())

    def __deepcopy__(self, memo):
        return None

    @staticmethod
    def sources(g):
        st = getattr(g, "_stores", None)
        if st is not None:                            # diffphore_amd.hetero.HeteroData: plain dict access (this runs once per complex and batch)
            L, P, LL, PP = st["ligand"], st["phore"], st[("ligand", "ligand")], st[("phore", "phore")]
            return (L["x"], LL["edge_index"], LL["edge_attr"], L["edge_mask"], L["mask_rotate"], L["phorefp"], L["norm_angle1"], L["norm_angle2"],
                    P["x"], P["pos"], P["norm"], P["phoretype"], PP["edge_index"])
        L, P = g["ligand"], g["phore"]
        LL, PP = g["ligand", "ligand"], g["phore", "phore"]
        return (L.x, LL.edge_index, LL.edge_attr, L.edge_mask, L.mask_rotate, L.phorefp, L.norm_angle1, L.norm_angle2, P.x, P.pos, P.norm,
                P.phoretype, PP.edge_index)

    @staticmethod
    def _ver(v):
        if hasattr(v, "_version"):
            return v._version
        if isinstance(v, np.ndarray):
            return hash(v.tobytes())
        if isinstance(v, (list, tuple)):
            return hash(tuple(_GraphView._ver(x) for x in v))
        return 0

    @staticmethod
    def key_of(src):
        # identity of every source object + the in-place version counter of the tensors among them; sources without a version counter (the
        # reference keeps mask_rotate as an ndarray or a list of them) are hashed by content, as score_model._cheap_key does
        ver = _GraphView._ver
        return tuple([id(v) for v in src]) + tuple([ver(v) for v in src])

    def valid_for(self, src):
        """key_of(src) == self.key without building the key (the re-check that runs once per complex and device batch: 13 sources x 512
        complexes; `self.src` keeps the cached objects alive, so identity is what the ids in the key compare)"""
        s0 = self.src
        for i, v0 in self._tver:                       # tensors: same object, same in-place version
            v = src[i]
            if v is not s0[i] or v._version != v0:
                return False
        for i, h0 in self._other:                      # sources without a version counter: same object, same content
            v = src[i]
            if v is not s0[i] or self._ver(v) != h0:
                return False
        return True

    def __init__(self, g):
        src = self.sources(g)
        self.key = self.key_of(src)
        self.src = src                                  # keeps the source objects alive: their ids (part of the key) cannot be recycled
        vers = self.key[len(src):]
        self._tver = tuple((i, vers[i]) for i, v in enumerate(src) if hasattr(v, "_version"))
        self._other = tuple((i, vers[i]) for i, v in enumerate(src) if not hasattr(v, "_version"))
        x, ei, ea, em, mr, fp, na1, na2, px, ppos, pnorm, ptype, pei = src

        def T(v, dt, shape=None):
            if torch.is_tensor(v) and v.dtype == dt and v.device.type == "cpu" and v.is_contiguous() and not v.requires_grad:
                t = v                                   # the complex's own tensor, no copy (the usual case)
            else:
                t = (v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v))).detach().to("cpu", dt).contiguous()
            return t.reshape(shape) if shape is not None else t
        n = int(x.shape[0])
        em_t = T(em, torch.bool)
        nt = int(em_t.sum())
        if nt > 0:
            m = mr if isinstance(mr, np.ndarray) else (mr[0] if isinstance(mr, (list, tuple)) else mr)
            mr_t = T(m, torch.bool)
            if tuple(mr_t.shape) != (nt, n):
                raise ValueError(f"mask_rotate has shape {tuple(mr_t.shape)}, expected ({nt}, {n})")
        else:
            mr_t = None
        self.keep = [T(x, torch.int64), T(ei, torch.int64), T(ea, torch.float32, (-1, 4)), em_t, mr_t, T(fp, torch.float32), T(na1, torch.float32),
                     T(na2, torch.float32), T(px, torch.float32), T(ppos, torch.float32), T(pnorm, torch.float32), T(ptype, torch.float32),
                     T(pei, torch.int64)]
        k = self.keep
        if k[0].shape[1] != 16 or k[8].shape[1] != 5 or k[11].shape[1] != 11 or k[1].shape[0] != 2 or k[12].shape[0] != 2:
            raise ValueError("complex has unexpected field shapes (ligand x [N,16], phore x [N,5], phoretype [N,11], edge_index [2,E])")
        v = DpGraphView()
        v.n_atoms, v.n_phore, v.n_bond_edges, v.n_tor, v.n_pp = n, int(k[8].shape[0]), int(k[1].shape[1]), nt, int(k[12].shape[1])
        if k[2].shape[0] != v.n_bond_edges or k[3].numel() != v.n_bond_edges:
            raise ValueError("ligand bond edge_attr / edge_mask do not match edge_index")
        for name, t in zip(("lig_x", "bond_index", "bond_attr", "edge_mask", "mask_rotate", "phorefp", "norm_angle1", "norm_angle2", "ph_x", "ph_pos",
                            "ph_norm", "ph_type", "pp_index"), k):
            setattr(v, name, t.data_ptr() if t is not None else None)
        self.struct = v
        self.addr = C.addressof(v)
        self.n = n

    @classmethod
    def of(cls, g, checked, by_key):
        """the (cached) view of `g`'s complex; `checked`: views already validated for this device batch, `by_key`: the batch's views by
        source key (samples of one complex that were copied before any of them was packed find each other here)"""
        gv = getattr(g, "_dp_view", None)
        src = None
        if gv is not None:
            src = cls.sources(g)
            if id(gv) in checked:
                # validated for a sibling sample in this batch (same objects => same in-place versions): this sample must still hold the SAME objects -
                # a sample whose static tensor was replaced since its last pack must not ride on its siblings' view
                s0 = gv.src
                if all(v is s0[i] for i, v in enumerate(src)):
                    return gv
            elif gv.valid_for(src):
                by_key[gv.key] = gv
                checked.add(id(gv))
                return gv
        key = cls.key_of(src if src is not None else cls.sources(g))
        if gv is None or gv.key != key:
            gv = by_key.get(key)
            if gv is None:
                gv = cls(g)
            try:
                object.__setattr__(g, "_dp_view", gv)
            except (AttributeError, TypeError):
                pass
        by_key[key] = gv
        checked.add(id(gv))
        return gv


class PackedBatch:
    """Struct-of-arrays image of a list of complexes on the device (static topology + diffusion state)."""

    def __init__(self, graphs: Sequence, device, upload: bool = True, host_pack: Optional[str] = None):
        """`upload=False` stops after the host-side packing (no device interaction at all) so that a caller can pack the next batch
        while the GPU works on the current one; `upload()` then moves everything across in one go.  `host_pack`: "native" (default;
        $DIFFPHORE_HOST_PACK) = the C collate dp_pack_query / dp_pack_fill over per-complex views, one host block, one upload; "numpy" =
        the same packing written out in numpy (the readable form; tests hold the two equal byte for byte)."""
        self.device = torch.device(device)
        assert len(graphs) > 0
        host_pack = host_pack or os.environ.get("DIFFPHORE_HOST_PACK", "native")
        if host_pack not in ("native", "numpy"):
            raise ValueError("host_pack must be 'native' or 'numpy'")
        self.static = None
        self._block = None
        self._stage = None
        if host_pack == "native":
            self._pack_native(graphs)
        else:
            self._pack_numpy(graphs)
        if upload:
            self.upload()

    def __del__(self):
        st = getattr(self, "_stage", None)
        if st is not None:                             # packed but never uploaded: give the staging buffer back
            st.busy = False

    def _pack_native(self, graphs):
        L = _declare()
        B = len(graphs)
        refs = (DpPoseRef * B)()
        checked, by_key, keep = set(), {}, []
        f32 = torch.float32
        view_of = _GraphView.of
        flat = []                                      # (view address, pos address, norm address) per pose
        for g in graphs:
            gv = view_of(g, checked, by_key)
            st = getattr(g, "_stores", None)
            lig = st["ligand"] if st is not None else g["ligand"]
            pos, norm = lig["pos"], lig["norm"]
            if pos.dtype is not f32 or not pos.is_contiguous() or pos.device.type != "cpu":
                pos = pos.detach().to("cpu", f32).contiguous()
            if norm.dtype is not f32 or not norm.is_contiguous() or norm.device.type != "cpu":
                norm = norm.detach().to("cpu", f32).contiguous()
            if pos.numel() != 3 * gv.n or norm.numel() != 33 * gv.n:
                raise ValueError("ligand pos / norm do not match the ligand's atom count")
            keep.append((gv, pos, norm))
            flat.append(gv.addr); flat.append(pos.data_ptr()); flat.append(norm.data_ptr())
        # DpPoseRef is three pointers: fill the array in one go
        flat_np = np.asarray(flat, dtype=np.uint64)
        C.memmove(refs, flat_np.ctypes.data, 24 * B)
        tot, lay = DpBatch(), DpPackLayout()
        rc = L.dp_pack_query(B, C.addressof(refs), C.byref(tot), C.byref(lay))
        if rc == -3:
            raise ValueError("diffphore_amd kernel limits: <=256 ligand atoms, <=256 pharmacophore nodes, <=128 torsions per pose "
                             "(or the batch is too large for int32 indexing; split it)")
        _lib.check(rc, "dp_pack_query")
        self._stage = _PinnedStage.lease(self.device, int(lay.bytes))
        block = self._stage.buf[:int(lay.bytes)] if self._stage is not None else torch.empty(int(lay.bytes), dtype=torch.uint8)
        flags = c_i32()
        rc = L.dp_pack_fill(B, C.addressof(refs), C.byref(lay), block.data_ptr(), C.byref(flags))
        if rc == -1:
            raise ValueError("malformed complex in the batch: ligand bond edge_attr must be one-hot over 4 bond types, data['phore'].phoretype "
                             "one-hot over 11 types (pdbbind_phore.py:1173), edge indices inside the complex, edge_mask / mask_rotate consistent")
        _lib.check(rc, "dp_pack_fill")
        del keep
        self._block, self._layout, self._flags, self._totals = block, lay, int(flags.value), tot
        bn = block.numpy()

        def host(name, count, dt=np.int32):
            off = int(getattr(lay, name))
            return bn[off:off + count * np.dtype(dt).itemsize].view(dt)
        self.B = B
        for name in ("lig_ptr", "ph_ptr", "bond_ptr", "tor_ptr", "pp_ptr", "x_ptr", "ll_ptr", "mask_ptr", "llp_ptr"):
            setattr(self, name, host(name, B + 1).astype(np.int64))
        self.nl, self.np_, self.nt = np.diff(self.lig_ptr), np.diff(self.ph_ptr), np.diff(self.tor_ptr)
        self.ll_pairs_ok = not (self._flags & 1)
        self._n_ph = int(tot.n_ph)

    def _pack_numpy(self, graphs):
        B = len(graphs)
        nl = np.asarray([g["ligand"].pos.shape[0] for g in graphs], dtype=np.int64)
        npn = np.asarray([g["phore"].pos.shape[0] for g in graphs], dtype=np.int64)
        nb = np.asarray([g["ligand", "ligand"].edge_index.shape[1] for g in graphs], dtype=np.int64)
        # the samples of one complex share their static tensors (HeteroData.sample_copy): whatever is derived from a static tensor is
        # computed once per tensor OBJECT and reused for its other samples (a screening batch: 40 samples per complex)
        memo: Dict[tuple, object] = {}

        def once(v, fn):
            k = (id(v), fn)
            if k not in memo:
                memo[k] = (v, fn(v))                   # (the tensor is kept alive with its entry: ids are not recycled meanwhile)
            return memo[k][1]
        n_true = lambda m: int(torch.as_tensor(m).sum())      # noqa: E731
        nt = np.asarray([once(g["ligand"].edge_mask, n_true) for g in graphs], dtype=np.int64)
        npp = np.asarray([g["phore", "phore"].edge_index.shape[1] for g in graphs], dtype=np.int64)
        if nl.max() > 256 or npn.max() > 256 or nt.max() > 128:
            raise ValueError("diffphore_amd kernel limits: <=256 ligand atoms, <=256 pharmacophore nodes, <=128 torsions per pose")
        if nl.min() < 1 or npn.min() < 1:
            raise ValueError("every pose needs at least one ligand atom and one pharmacophore node")
        cs = lambda a: np.concatenate([[0], np.cumsum(a)])      # noqa: E731
        self.lig_ptr, self.ph_ptr, self.bond_ptr = cs(nl), cs(npn), cs(nb)
        self.tor_ptr, self.pp_ptr, self.x_ptr = cs(nt), cs(npp), cs(nl * npn)
        self.ll_ptr = cs(nb + nl * np.minimum(np.maximum(nl - 1, 0), 33))
        self.mask_ptr = cs(nt * nl)
        self.llp_ptr = cs(nl * nl)
        self.B, self.nl, self.np_, self.nt = B, nl, npn, nt
        if max(self.x_ptr[-1], self.ll_ptr[-1]) * 20 >= 2 ** 31:
            raise ValueError("batch too large for int32 indexing; split it")
        L = [g["ligand"] for g in graphs]
        P = [g["phore"] for g in graphs]
        # Host packing is numpy, not torch: on a many-core host every small torch CPU op (cat / reshape) pays the intra-op thread
        # pool (measured 1 ms per torch.cat on a 256-thread box: 170 ms to pack 512 poses, 15 ms with numpy).
        def to_np(v):
            if not torch.is_tensor(v):
                return np.asarray(v)
            try:
                return v.numpy()                       # CPU tensor without grad: zero-copy
            except (RuntimeError, TypeError):
                return v.detach().cpu().numpy()

        def A(v):
            return once(v, to_np)

        def cat(vals, dtype, axis=0):
            return np.ascontiguousarray(np.concatenate([A(v) for v in vals], axis=axis).astype(dtype, copy=False))
        H: Dict[str, np.ndarray] = {}
        for n, a in (("lig_ptr", self.lig_ptr), ("ph_ptr", self.ph_ptr), ("bond_ptr", self.bond_ptr), ("tor_ptr", self.tor_ptr),
                     ("pp_ptr", self.pp_ptr), ("x_ptr", self.x_ptr), ("ll_ptr", self.ll_ptr), ("mask_ptr", self.mask_ptr),
                     ("pose_ptr", np.arange(B + 1)), ("llp_ptr", self.llp_ptr)):
            H[n] = np.ascontiguousarray(a, dtype=np.int32)
        H["lig_x"] = cat([l.x for l in L], np.int32)
        ei_all = cat([g["ligand", "ligand"].edge_index for g in graphs], np.int64, axis=1) + np.repeat(self.lig_ptr[:-1], nb)[None, :]
        H["bond_src"], H["bond_dst"] = ei_all[0].astype(np.int32), ei_all[1].astype(np.int32)
        # A ligand graph that lists the same ordered bond twice (a multigraph the reference would simply process) does not fit the
        # N x N pair tables of the node-side ligand-ligand conv: such a batch carries no tables (DpBatch.llp_ptr = NULL) and dp_forward
        # keeps its ligand-ligand convs on the edge-side CSR kernel.  Global atom ids are unique per pose, so one key covers the batch.
        key = ei_all[0] * (int(self.lig_ptr[-1]) + 1) + ei_all[1]
        self.ll_pairs_ok = bool(np.unique(key).size == key.size)
        ea = cat([g["ligand", "ligand"].edge_attr for g in graphs], np.float32).reshape(-1, 4)
        if ea.size and not np.all((ea.sum(1) == 1) & (ea.max(1) == 1)):
            raise ValueError("ligand bond edge_attr must be one-hot over 4 bond types")
        H["bond_type"] = ea.argmax(1).astype(np.int32) if ea.size else np.zeros(0, dtype=np.int32)
        em = cat([l.edge_mask for l in L], bool)
        H["tor_u"], H["tor_v"] = ei_all[0][em].astype(np.int32), ei_all[1][em].astype(np.int32)
        masks = []
        for l, t, n in zip(L, nt, nl):
            mr = l.mask_rotate
            mr = np.asarray(mr if isinstance(mr, np.ndarray) else mr[0]) if t > 0 else np.zeros((0, n), dtype=bool)
            assert mr.shape == (t, n), (mr.shape, t, n)
            masks.append(mr.astype(np.uint8).reshape(-1))
        H["mask_rotate"] = np.concatenate(masks) if masks else np.zeros(0, dtype=np.uint8)
        if H["mask_rotate"].size == 0:
            H["mask_rotate"] = np.zeros(1, dtype=np.uint8)
        H["phorefp"] = cat([l.phorefp for l in L], np.float32)
        H["norm_angle1"] = cat([l.norm_angle1 for l in L], np.float32)
        H["norm_angle2"] = cat([l.norm_angle2 for l in L], np.float32)
        H["ph_pos"] = cat([p.pos for p in P], np.float32)
        H["ph_norm"] = cat([p.norm for p in P], np.float32)
        px = cat([p.x for p in P], np.float32)
        H["ph_x"] = np.ascontiguousarray(px[:, :3].astype(np.int64).astype(np.int32))
        H["ph_scalar"] = np.ascontiguousarray(px[:, 3:5])
        pt = cat([p.phoretype for p in P], np.float32)
        if not np.all((pt.sum(1) == 1) & (pt.max(1) == 1)):
            raise ValueError("data['phore'].phoretype must be one-hot (pdbbind_phore.py:1173)")
        H["ph_type"] = pt.argmax(1).astype(np.int32)
        pe = cat([g["phore", "phore"].edge_index for g in graphs], np.int64, axis=1) + np.repeat(self.ph_ptr[:-1], npp)[None, :]
        # global node ids grow with the pose index, so ONE stable sort by source keeps every pose's edges inside
        # [pp_ptr[p], pp_ptr[p+1]) and orders them by source there (ties keep the input order)
        order = np.argsort(pe[0], kind="stable")
        ps, pd = pe[0][order], pe[1][order]
        H["pp_src"], H["pp_dst"] = ps.astype(np.int32), pd.astype(np.int32)
        n_ph = int(self.ph_ptr[-1])
        deg = np.bincount(ps, minlength=n_ph)
        rowptr = np.cumsum(deg) - deg                 # CSR over the globally sorted edge list == per-pose offsets + in-pose prefix
        H["pp_rowptr"], H["pp_rowend"] = rowptr.astype(np.int32), (rowptr + deg).astype(np.int32)
        # cross pairs with agreement != 0 (node j is not an exclusion sphere and phorefp[i, type(j)] != 0), by atom and by node: DpBatch.xs_* / xt_*
        fp_all, ty_all = H["phorefp"].reshape(-1, 11), H["ph_type"]
        xs, xt = [[], [], []], [[], [], []]
        for p in range(B):
            l0, p0, x0 = int(self.lig_ptr[p]), int(self.ph_ptr[p]), int(self.x_ptr[p])
            n_l, n_p = int(nl[p]), int(npn[p])
            ty = ty_all[p0:p0 + n_p]
            agree = (fp_all[l0:l0 + n_l][:, ty] != 0) & (ty != 10)[None, :]            # [atoms, nodes]
            ai, aj = np.nonzero(agree)                                                # row-major: by (atom, node)
            tj, ti = np.nonzero(agree.T)                                              # by (node, atom)
            for dst, (a_, j_) in ((xs, (ai, aj)), (xt, (ti, tj))):
                dst[2].append(x0 + a_ * n_p + j_)
            xs[0].append(l0 + ai); xs[1].append(p0 + aj)
            xt[0].append(p0 + tj); xt[1].append(l0 + ti)
        i32cat = lambda parts: np.concatenate(parts).astype(np.int32) if parts else np.zeros(0, dtype=np.int32)      # noqa: E731
        for pre, lists, n_nodes in (("xs", xs, int(self.lig_ptr[-1])), ("xt", xt, n_ph)):
            src = i32cat(lists[0])
            H[pre + "_src"], H[pre + "_nbr"], H[pre + "_feat"] = src, i32cat(lists[1]), i32cat(lists[2])
            cnt = np.bincount(src, minlength=n_nodes)
            H[pre + "_rowptr"] = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        self._n_xs = int(H["xs_src"].size)
        # DpBatch.ph_shared: every pose carries pose 0's pharmacophore (byte comparison, as dp_pack_query does it)
        def same_phore(i):
            if P[i] is P[0] and graphs[i]["phore", "phore"] is graphs[0]["phore", "phore"]:
                return True
            if npn[i] != npn[0] or npp[i] != npp[0]:
                return False
            pairs = [(P[i].x, P[0].x, np.float32), (P[i].pos, P[0].pos, np.float32), (P[i].norm, P[0].norm, np.float32),
                     (P[i].phoretype, P[0].phoretype, np.float32),
                     (graphs[i]["phore", "phore"].edge_index, graphs[0]["phore", "phore"].edge_index, np.int64)]
            return all(a is b_ or np.ascontiguousarray(A(a), dtype=dt).tobytes() == np.ascontiguousarray(A(b_), dtype=dt).tobytes() for a, b_, dt in pairs)
        self._ph_shared = int(B > 1 and all(same_phore(i) for i in range(1, B)))
        for k_ in ("xs_src", "xs_nbr", "xs_feat", "xt_src", "xt_nbr", "xt_feat"):
            if H[k_].size == 0:
                H[k_] = np.zeros(1, dtype=np.int32)
        self._host = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in H.items()}
        self._host_pos = torch.from_numpy(cat([l.pos for l in L], np.float32))
        self._host_norm = torch.from_numpy(np.concatenate([A(l.norm).astype(np.float32, copy=False).reshape(A(l.pos).shape[0], 33) for l in L]))
        self._n_ph = n_ph

    def _upload_native(self):
        L = _declare()
        lay, tot = self._layout, self._totals
        if self._stage is not None:
            # page-locked source: an asynchronous DMA on the copy stream; the staging buffer is free again behind its event
            with torch.cuda.stream(_copy_stream(self.device)):
                dev = torch.empty(self._block.shape, dtype=torch.uint8, device=self.device)
                dev.copy_(self._block, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            torch.cuda.current_stream(self.device).wait_event(ev)      # the launch stream reads the batch behind the copy
            dev.record_stream(torch.cuda.current_stream(self.device))
            self._stage.event, self._stage.busy = ev, False
            self._stage = None
        else:
            dev = h2d(self._block, self.device)                # ONE copy for the whole batch
        self._dev_block = dev
        base = dev.data_ptr()
        assert base % 256 == 0

        def view(name, count, dt, shape=None):
            off = int(getattr(lay, name))
            v = dev[off:off + count * dt.itemsize].view(dt)
            return v.view(shape) if shape is not None else v
        n_lig, n_ph, B = int(tot.n_lig), int(tot.n_ph), self.B
        i32, f32, u8 = torch.int32, torch.float32, torch.uint8
        sizes = dict(lig_ptr=B + 1, ph_ptr=B + 1, bond_ptr=B + 1, tor_ptr=B + 1, pp_ptr=B + 1, x_ptr=B + 1, ll_ptr=B + 1, mask_ptr=B + 1, pose_ptr=B + 1,
                     llp_ptr=B + 1, lig_x=n_lig * 16, bond_src=int(tot.n_bond), bond_dst=int(tot.n_bond), bond_type=int(tot.n_bond), tor_u=int(tot.n_tor),
                     tor_v=int(tot.n_tor), mask_rotate=max(int(self.mask_ptr[-1]), 1), phorefp=n_lig * 11, norm_angle1=n_lig * 11, norm_angle2=n_lig * 11,
                     ph_pos=n_ph * 3, ph_norm=n_ph * 3, ph_x=n_ph * 3, ph_scalar=n_ph * 2, ph_type=n_ph, pp_src=int(tot.n_pp), pp_dst=int(tot.n_pp),
                     pp_rowptr=n_ph, pp_rowend=n_ph)
        n_xs = int(tot.n_xs)
        sizes.update(xs_src=n_xs, xs_nbr=n_xs, xs_feat=n_xs, xs_rowptr=n_lig + 1, xt_src=n_xs, xt_nbr=n_xs, xt_feat=n_xs, xt_rowptr=n_ph + 1)
        fl = {"phorefp", "norm_angle1", "norm_angle2", "ph_pos", "ph_norm", "ph_scalar"}
        self.static = {k: view(k, n, u8 if k == "mask_rotate" else (f32 if k in fl else i32)) for k, n in sizes.items()}
        for k, cols in (("lig_x", 16), ("phorefp", 11), ("norm_angle1", 11), ("norm_angle2", 11), ("ph_pos", 3), ("ph_norm", 3), ("ph_x", 3), ("ph_scalar", 2)):
            self.static[k] = self.static[k].view(-1, cols)
        self.pos = view("pos", n_lig * 3, f32, (n_lig, 3))
        self.norm = view("norm", n_lig * 33, f32, (n_lig, 33))
        self.t = torch.zeros(B, 3, dtype=f32, device=self.device)
        s = DpBatch()
        C.memmove(C.byref(s), C.byref(tot), C.sizeof(DpBatch))
        _lib.check(L.dp_pack_bind(C.byref(lay), base, self._flags, C.byref(s)), "dp_pack_bind")
        s.t = self.t.data_ptr()
        self.struct = s
        self._block = None
        return self

    def upload(self):
        if self.static is not None:
            return self
        if self._block is not None:
            return self._upload_native()
        B, nl, npn, nt, n_ph = self.B, self.nl, self.np_, self.nt, self._n_ph
        self.static = {k: h2d(v, self.device) for k, v in self._host.items()}
        self.pos = h2d(self._host_pos, self.device)
        self.norm = h2d(self._host_norm, self.device)
        self._host = self._host_pos = self._host_norm = None
        self.t = torch.zeros(B, 3, dtype=torch.float32, device=self.device)
        s = DpBatch()
        s.B = B
        s.n_lig, s.n_ph, s.n_bond, s.n_tor = int(self.lig_ptr[-1]), n_ph, int(self.bond_ptr[-1]), int(self.tor_ptr[-1])
        s.n_pp, s.n_x, s.ll_cap = int(self.pp_ptr[-1]), int(self.x_ptr[-1]), int(self.ll_ptr[-1])
        s.n_llp = int(self.llp_ptr[-1]) if self.ll_pairs_ok else 0
        s.n_xs = self._n_xs
        s.ph_shared = self._ph_shared
        s.max_nl, s.max_np, s.max_t = int(nl.max()), int(npn.max()), int(nt.max())
        # hints for the node-side convs: no pose ends on a pass of <= 16 edge slots -> the lean kernel instantiation (DpConvArgs.no_tail)
        tail = lambda v: bool((((v - 1) % 32) < 16).any())                     # noqa: E731  (v mod 32 in 1..16)
        s.no_tail_nl, s.no_tail_np = (0 if tail(np.asarray(nl)) else 1), (0 if tail(np.asarray(npn)) else 1)
        for k, v in self.static.items():
            setattr(s, k, v.data_ptr())
        if not self.ll_pairs_ok:
            s.llp_ptr = None
        s.pos, s.norm, s.t = self.pos.data_ptr(), self.norm.data_ptr(), self.t.data_ptr()
        self.struct = s
        return self

    def set_state(self, pos: torch.Tensor, norm: torch.Tensor):
        self.pos.copy_(pos.to(self.device, torch.float32).reshape(self.pos.shape))
        self.norm.copy_(norm.to(self.device, torch.float32).reshape(self.norm.shape))

    def set_time(self, t_tr, t_rot, t_tor):
        """Scalars (one t for the whole batch, sampling.py:218) or per-pose tensors [B]."""
        tt = torch.stack([torch.as_tensor(x, dtype=torch.float32).expand(self.B) if not torch.is_tensor(x) or x.dim() == 0
                          else x.float() for x in (t_tr, t_rot, t_tor)], 1)
        # one time for every pose (known without looking at device data): DpWork.t_uniform of the forwards that follow
        # (a per-pose HOST tensor is looked at: set_time_phore hands forward(data) B copies of one value; a device tensor is not - no sync here)
        def uniform(x):
            if not torch.is_tensor(x) or x.dim() == 0:
                return True
            return x.device.type == "cpu" and x.numel() > 0 and bool((x == x.reshape(-1)[0]).all())
        self.t_uniform = all(uniform(x) for x in (t_tr, t_rot, t_tor))
        self.t.copy_(tt.to(self.device))

    def set_time_row(self, row: torch.Tensor):
        """The sampling loop's set_time_phore (sampling.py:218): one device row (t_tr, t_rot, t_tor) for every pose, no host sync.  (Write the times
        through set_time / set_time_row only: they keep `t_uniform`, which Engine.forward / sample_step pass on as DpWork.t_uniform.)"""
        self.t.copy_(row.reshape(1, 3).expand(self.B, 3))
        self.t_uniform = True


class Workspace:
    """One device block sized and carved by the C library (dp_work_bytes / dp_work_bind); `t[...]` are typed views of its arrays."""
    _SHAPES = dict(lig_x_emb=("f", "n_lig", 20), ph_x_emb=("f", "n_ph", 20), pair_pre=("f", "n_lig", 220), pair_fm=("f", "n_lig", 11),
                   pp_pre=("f", "n_pp", 20), pp_sh=("f", "n_pp", 9), tparams=("f", "B", 32), ll_rowptr=("i", "n_lig", 1),
                   ll_rowend=("i", "n_lig", 1), ll_nbr=("i", "ll_cap", 1), ll_src=("i", "ll_cap", 1), ll_emb=("f", "ll_cap", 20),
                   ll_sh=("f", "ll_cap", 9), ll_pair_rad=("i", "n_llp", 1), ll_pair_bond=("i", "n_llp", 1), pp_emb=("f", "n_pp", 20), x_emb=("f", "n_x", 20), x_sh=("f", "n_x", 9), x_nsh=("f", "n_x", 9),
                   x_ew=("f", "n_x", 1), c_emb=("f", "n_lig", 20), c_sh=("f", "n_lig", 9), gpred=("f", "B", 12), t_rowptr=("i", "nT", 1),
                   t_rowend=("i", "nT", 1), t_nbr=("i", "nT", 32), t_src=("i", "nT", 32), t_emb=("f", "nT", 640), t_sh=("f", "nT", 224),
                   t_hb=("f", "nT", 20), t_q=("f", "nT", 40), status=("i", "B", 1))

    def __init__(self, pb: PackedBatch):
        L = _declare()
        dev, s = pb.device, pb.struct
        nbytes = int(L.dp_work_bytes(C.byref(s)))
        if nbytes <= 0:
            raise _lib.DiffPhoreLibraryError("dp_work_bytes rejected the batch")
        self.block = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        base = (self.block.data_ptr() + 255) // 256 * 256
        w = DpWork()
        ptrs = [VOIDP(), VOIDP(), VOIDP()]
        _lib.check(L.dp_work_bind(C.byref(s), base, nbytes, C.byref(w), *[C.byref(p) for p in ptrs]), "dp_work_bind")
        _lib.check(L.dp_work_init(C.byref(s), C.byref(w), _stream()), "dp_work_init")
        self.struct = w
        lead = base - self.block.data_ptr()
        f32 = self.block[lead:lead + nbytes].view(torch.float32)
        i32 = self.block[lead:lead + nbytes].view(torch.int32)
        dims = dict(n_lig=s.n_lig, n_ph=s.n_ph, n_pp=max(s.n_pp, 1), n_x=s.n_x, ll_cap=max(s.ll_cap, 1), B=s.B, nT=max(s.n_tor, 1),
                    n_llp=max(s.n_llp, 1))

        def view(ptr, kind, rows, cols):
            off = (int(ptr) - base) // 4
            v = (f32 if kind == "f" else i32)[off:off + rows * cols]
            return v if cols == 1 else v.view(rows, cols)
        T = {k: view(getattr(w, k), kind, dims[rk], cols) for k, (kind, rk, cols) in self._SHAPES.items()}
        T["h_l0"], T["h_l1"] = view(w.h_l[0], "f", s.n_lig, 100), view(w.h_l[1], "f", s.n_lig, 100)
        T["h_p0"], T["h_p1"] = view(w.h_p[0], "f", s.n_ph, 100), view(w.h_p[1], "f", s.n_ph, 100)
        self.t = T
        nT = dims["nT"]
        self.tr, self.rot = view(ptrs[0].value, "f", s.B, 3), view(ptrs[1].value, "f", s.B, 3)
        self.tor = view(ptrs[2].value, "f", nT, 1)
        self.n_tor = s.n_tor


class _HipHandles:
    """A second stream + two events for dp_forward's optional fork/join, created and destroyed by the C library itself (dp_aux_create /
    dp_aux_destroy): they come from the same HIP runtime the kernels are launched with.  Cached per device: engines are created per
    device batch, the handles live as long as the process."""
    _cache: Dict[int, "_HipHandles"] = {}

    def __init__(self):
        L = _declare()
        self.L = L
        self.stream, self.ev_fork, self.ev_join = VOIDP(), VOIDP(), VOIDP()
        _lib.check(L.dp_aux_create(C.byref(self.stream), C.byref(self.ev_fork), C.byref(self.ev_join)), "dp_aux_create")
        # the third stream (DpWork.aux2_stream / ev_join2: the phore-phore conv of a small batch beside its ligand->phore convs)
        self.stream2, self._ev_spare, self.ev_join2 = VOIDP(), VOIDP(), VOIDP()
        _lib.check(L.dp_aux_create(C.byref(self.stream2), C.byref(self._ev_spare), C.byref(self.ev_join2)), "dp_aux_create")

    @classmethod
    def for_device(cls, index: int) -> "_HipHandles":
        h = cls._cache.get(index)
        if h is None:
            h = cls._cache[index] = cls()
        return h

    def __del__(self):
        try:
            self.L.dp_aux_destroy(self.stream, self.ev_fork, self.ev_join)
            self.L.dp_aux_destroy(self.stream2, self._ev_spare, self.ev_join2)
        except Exception:      # noqa: BLE001 - interpreter shutdown
            pass


CONV_ALGOS = {"auto": 0, "edge": 1, "node": 2}
CONV_FUSE = {"1": 0, "0": 1, "pairs": 2, "nosplit": 3}        # DIFFPHORE_CONV_FUSE -> DpWork.conv_fuse
TWO_STREAM_MAX_POSES = 256        # below this a single conv launch leaves CUs idle


class Engine:
    """(model, batch, workspace) bound together; thin wrappers over the C ABI entry points.

    `conv_algo`: formulation of the fused layer convs - "edge" (per-edge radial GEMM), "node" (node-side contraction, dense cross
    edge sets) or None / "auto" ($DIFFPHORE_CONV_ALGO, else the library's choice by precision and batch)."""

    def __init__(self, model: DeviceModel, pb: PackedBatch, conv_algo: Optional[str] = None, exact: bool = False):
        self.model, self.pb, self.ws = model, pb, Workspace(pb)
        self.exact = bool(exact)                     # fp32 MFMA everywhere whatever the environment says: what sampling._run_parts replays a flagged batch on
        self.L = _declare()
        conv_algo = conv_algo or os.environ.get("DIFFPHORE_CONV_ALGO", "auto")
        if conv_algo not in CONV_ALGOS:
            raise ValueError(f"conv_algo must be one of {sorted(CONV_ALGOS)}")
        self.ws.struct.conv_algo = CONV_ALGOS[conv_algo]
        self.L.dp_resolve_conv_algo.argtypes = [c_i32, c_i32, c_i32]
        self.L.dp_resolve_conv_algo.restype = C.c_int
        self.conv_algo = {1: "edge", 2: "node"}[self.L.dp_resolve_conv_algo(CONV_ALGOS[conv_algo], model.struct.precision, pb.B)]
        fuse = os.environ.get("DIFFPHORE_CONV_FUSE", "1")
        if fuse not in CONV_FUSE:
            raise ValueError(f"DIFFPHORE_CONV_FUSE must be one of {sorted(CONV_FUSE)} (got {fuse!r})")
        self.ws.struct.conv_fuse = CONV_FUSE[fuse]
        x6 = "0" if exact else os.environ.get("DIFFPHORE_EDGE_X6", "1")
        if x6 not in ("0", "1"):
            raise ValueError(f"DIFFPHORE_EDGE_X6 must be 0 or 1 (got {x6!r})")
        self.ws.struct.edge_bf16x6 = int(x6)         # 1 (default): edge-side convs of an fp32 model as a two-limb fp16 split (until round 4: bf16x6); 0: exact fp32 MFMA
        f16 = "0" if exact else os.environ.get("DIFFPHORE_NODE_F16", "1")
        if f16 not in ("0", "1"):
            raise ValueError(f"DIFFPHORE_NODE_F16 must be 0 or 1 (got {f16!r})")
        self.ws.struct.node_f16 = int(f16)           # 1 (default): node-side convs of an fp32 model as two fp16 limbs per operand (DP_PREC_F16X3); 0: fp32 MFMA
        self.node_side_convs = ("f16x3" if f16 == "1" else "f32") if model.precision == "f32" else model.precision
        self.edge_side_convs = ("f16x3" if x6 == "1" else "f32") if model.precision == "f32" else model.precision
        xn = os.environ.get("DIFFPHORE_XN_SPARSE", "auto")
        if xn not in ("auto", "1", "0"):
            raise ValueError(f"DIFFPHORE_XN_SPARSE must be auto, 1 or 0 (got {xn!r})")
        # the normals convs as their l = 0 part + the listed pairs with agreement != 0 (DpWork.xn_sparse): library's choice / whenever possible / never
        self.ws.struct.xn_sparse = {"auto": 0, "1": 1, "0": 2}[xn]
        if os.environ.get("DIFFPHORE_PP_LIST", "1") == "0":
            self.ws.struct.pp_msg = None             # A/B switch: the phore-phore convs on the CSR kernel whatever the fill of its chunks (dp_pp_as_list)
        if os.environ.get("DIFFPHORE_PH_SHARE", "1") == "0":
            self.ws.struct.ph_share = 2              # A/B switch: the layer-0 phore-phore conv per pose even when the batch is on one pharmacophore (DpBatch.ph_shared)
        self.ws.struct.ll_node_max_atoms = int(os.environ.get("DIFFPHORE_LL_NODE_MAX_ATOMS", "0"))
        ps = os.environ.get("DIFFPHORE_PATH_SPLIT", "auto")
        if ps not in ("auto", "1", "0"):
            raise ValueError(f"DIFFPHORE_PATH_SPLIT must be auto, 1 or 0 (got {ps!r})")
        self.ws.struct.path_split = {"auto": 0, "1": 1, "0": 2}[ps]
        self._ps_B, self._ps_nl = int(pb.B), int(max(pb.nl))
        self._handles = None
        self._graph = None
        # second stream (DpWork.aux_stream): 'auto' runs the phore-side convs and the torsion head beside the ligand-side convs when one
        # conv launch cannot keep the chip busy - small batches, and ragged ones (virtual screening: 15-45 atoms, the launch tails of the
        # big poses leave CUs idle: 640 poses 34.8k vs 33.8k pose-steps/s) - and only the ligand radius graph otherwise (two matrix-bound
        # convs side by side lose on a large homogeneous batch: 512 x 30 atoms 90.1k vs 94.9k)
        two = os.environ.get("DIFFPHORE_TWO_STREAMS", "auto")
        nl_, np_ = np.asarray(pb.nl, dtype=np.float64), np.asarray(pb.np_, dtype=np.float64)
        ragged = bool((nl_ * np_).max() > 1.3 * (nl_ * np_).mean())
        if two in ("1", "graph"):
            self.enable_two_streams(1 if two == "graph" else 0)
        elif two == "auto":
            try:
                # (round 5: with the f16x3 node-side kernels - bound by instruction issue, not by the matrix pipe - the two sides of a layer overlap
                #  at every size: 256 x 80-atom ligands 36.4k -> 40.6k pose-steps/s, 512 x 30 atoms 138.7k either way)
                f16 = self.node_side_convs == "f16x3" and self.conv_algo == "node"
                self.enable_two_streams(0 if (pb.B < TWO_STREAM_MAX_POSES or ragged or f16) else 1)
            except _lib.DiffPhoreLibraryError:          # an optimisation only: a single stream computes the same result
                self._handles = None
        _lib.check(self.L.dp_prepare(C.byref(pb.struct), C.byref(model.struct), C.byref(self.ws.struct), _stream()), "dp_prepare")

    def enable_two_streams(self, aux_mode: int = 0):
        """dp_forward then runs the phore-side convs of a layer and the torsion head on a second stream (bit-identical results;
        pays when the batch is too small to fill the chip with one conv at a time)."""
        if self._handles is None:
            dev = self.pb.device
            self._handles = _HipHandles.for_device(dev.index if dev.index is not None else torch.cuda.current_device())
            w = self.ws.struct
            w.aux_stream, w.ev_fork, w.ev_join = self._handles.stream, self._handles.ev_fork, self._handles.ev_join
            if os.environ.get("DIFFPHORE_THIRD_STREAM", "1") != "0":
                w.aux2_stream, w.ev_join2, w.ev_x2 = self._handles.stream2, self._handles.ev_join2, self._handles._ev_spare
            place = os.environ.get("DIFFPHORE_XS_PLACE", "auto")
            if place not in ("auto", "crossed", "third", "front", "front-reduce"):
                raise ValueError(f"DIFFPHORE_XS_PLACE must be auto, crossed, third or front (got {place!r})")
            # where the list launches of a layer's normals convs run (DpWork.xs_place; profiles/r6_experiments.md: front > third stream > crossed)
            w.xs_place = {"auto": 0, "crossed": 1, "third": 2, "front": 3, "front-reduce": 4}[place]
        self.ws.struct.aux_mode = int(aux_mode)

    def __del__(self):
        if getattr(self, "_graph", None):
            try:
                self.L.dp_sample_graph_destroy(self._graph)
            except Exception:      # noqa: BLE001
                pass

    def limb_kernels(self) -> bool:
        """whether this fp32 model's convs run on the two-limb fp16 kernels (DP_PREC_F16X3): fp32-grade INSIDE fp16's exponent range
        (dp_conv_node_f16_impl.h: |feature|, |T| 2^5 < 65504) - a value beyond it becomes non-finite and surfaces as status bit 0"""
        return self.model.precision == "f32" and (self.edge_side_convs == "f16x3" or (self.node_side_convs == "f16x3" and self.conv_algo == "node"))

    def path_split_on(self) -> bool:
        """whether the ligand-side node-side launches of this batch take the fp32 kernels' path split (dp_graph.hip: two workgroups per node group
        while B x groups <= 256; such launches run the exact-fp32 kernel even with DpWork.node_f16 = 1).  With the f16x3 kernel available the
        library never chooses it by itself (that kernel has its own small-batch form, the conv split): only DIFFPHORE_PATH_SPLIT=1 does."""
        ps = int(self.ws.struct.path_split)
        if ps:
            return ps == 1
        if self.node_side_convs == "f16x3":
            return False
        return self._ps_B * ((self._ps_nl + 15) // 16) <= 256

    def launch_of(self, layer: int, conv: int):
        """(first, count, node_side) of the dp_forward launch that carries conv `conv` of interaction layer `layer` (conv numbering of
        DpWork.prof_conv: 0 lig_conv, 1 phore_to_lig, 2 phore_to_lig_norm, 3 phore_conv, 4 lig_to_phore, 5 lig_to_phore_norm); node_side: 0 = edge-side
        kernel, 1 = node-side kernel, 2 = a normals conv as its s0-only node-side launch + its listed pairs (DpWork.xn_sparse)."""
        f, n, ns = c_i32(), c_i32(), c_i32()
        _lib.check(self.L.dp_forward_launch_of(C.byref(self.pb.struct), C.byref(self.model.struct), C.byref(self.ws.struct), layer, conv,
                                               C.byref(f), C.byref(n), C.byref(ns)), "dp_forward_launch_of")
        return f.value, n.value, int(ns.value)

    def forward(self):
        ws = self.ws
        ws.struct.t_uniform = 1 if getattr(self.pb, "t_uniform", False) else 0
        _lib.check(self.L.dp_forward(C.byref(self.pb.struct), C.byref(self.model.struct), C.byref(ws.struct), ws.tr.data_ptr(),
                                     ws.rot.data_ptr(), ws.tor.data_ptr(), _stream()), "dp_forward")
        return ws.tr, ws.rot, ws.tor[:ws.n_tor]

    def forward_in_range(self):
        """forward() for callers without a status word to look at: if this engine runs the two-limb fp16 kernels (fp32-grade only inside fp16's
        exponent range, `limb_kernels`) and the scores come out non-finite, they are recomputed on an exact-fp32 engine over the same packed
        batch.  -> (the engine that produced the scores - keep using it -, tr, rot, tor).  One host sync; DIFFPHORE_EXACT_REPLAY=0 = plain
        forward()."""
        tr, rot, tor = self.forward()
        if not self.limb_kernels() or os.environ.get("DIFFPHORE_EXACT_REPLAY", "1") == "0":
            return self, tr, rot, tor
        finite = torch.isfinite(tr).all() & torch.isfinite(rot).all()
        if tor.numel():
            finite = finite & torch.isfinite(tor).all()
        if bool(finite):
            return self, tr, rot, tor
        eng = Engine(self.model, self.pb, conv_algo=self.conv_algo, exact=True)
        eng.ws.t["status"].copy_(self.ws.t["status"])        # per-pose flags of earlier steps stay with the batch
        return (eng,) + tuple(eng.forward())

    def update(self, sp: DpStepParams, z_tr=None, z_rot=None, z_tor=None):
        ws = self.ws
        _lib.check(self.L.dp_conformer_update(C.byref(self.pb.struct), C.byref(sp), ws.tr.data_ptr(), ws.rot.data_ptr(),
                                              ws.tor.data_ptr(), _lib.ptr(z_tr), _lib.ptr(z_rot), _lib.ptr(z_tor),
                                              ws.t["status"].data_ptr(), _stream()), "dp_conformer_update")

    def sample_step(self, sp: DpStepParams, z_tr=None, z_rot=None, z_tor=None):
        ws = self.ws
        ws.struct.t_uniform = 1 if getattr(self.pb, "t_uniform", False) else 0
        _lib.check(self.L.dp_sample_step(C.byref(self.pb.struct), C.byref(self.model.struct), C.byref(ws.struct), C.byref(sp),
                                         ws.tr.data_ptr(), ws.rot.data_ptr(), ws.tor.data_ptr(), _lib.ptr(z_tr), _lib.ptr(z_rot),
                                         _lib.ptr(z_tor), _stream()), "dp_sample_step")

    def sample(self, sps: Sequence[DpStepParams], times: torch.Tensor, z_tr=None, z_rot=None, z_tor=None, graph: bool = False,
               seed: Optional[int] = None, pose_uid: Optional[torch.Tensor] = None, uid_base: int = 0):
        """The whole reverse-diffusion loop in ONE library call (dp_sample): `sps` per-step scalars, `times` [S,3] device tensor,
        `z_*` device noise tables [S,B,3] / [S,B,3] / [S,n_tor] or None.  `seed` (int) switches to on-device noise (DP_NOISE_SEED):
        the tables are ignored and every step's normals are a function of (seed, pose uid, step, component), `pose_uid` an int32
        device tensor [B] (default uid_base + position in the batch).  `graph=True` captures the loop into a hipGraph (kept for replays
        with the same tables) and launches it."""
        ws, S = self.ws, len(sps)
        arr = (DpStepParams * S)(*sps)
        assert times.is_contiguous() and tuple(times.shape) == (S, 3) and times.dtype == torch.float32 and times.is_cuda
        sch = DpSchedule()
        sch.steps, sch.sp, sch.t = S, arr, times.data_ptr()
        sch.z_tr, sch.z_rot, sch.z_tor = _lib.ptr(z_tr), _lib.ptr(z_rot), _lib.ptr(z_tor)
        if seed is not None:
            if pose_uid is not None:
                assert pose_uid.is_cuda and pose_uid.dtype == torch.int32 and pose_uid.numel() == self.pb.B and pose_uid.is_contiguous()
            sch.noise_mode, sch.seed, sch.uid_base, sch.pose_uid = 1, int(seed) & 0xFFFFFFFFFFFFFFFF, int(uid_base), _lib.ptr(pose_uid)
        self._keep = (arr, times, z_tr, z_rot, z_tor, pose_uid)            # the graph bakes these pointers in
        args = (C.byref(self.pb.struct), C.byref(self.model.struct), C.byref(ws.struct), C.byref(sch), ws.tr.data_ptr(), ws.rot.data_ptr(),
                ws.tor.data_ptr())
        if not graph:
            _lib.check(self.L.dp_sample(*args, _stream()), "dp_sample")
            return
        if self._graph:
            self.L.dp_sample_graph_destroy(self._graph)
            self._graph = None
        torch.cuda.current_stream().synchronize()               # capture happens on the library's own stream: order it after pending work
        g = VOIDP()
        _lib.check(self.L.dp_sample_graph_create(*args, C.byref(g)), "dp_sample_graph_create")
        self._graph = g
        self.replay()

    def replay(self):
        """Launch the captured loop again (same tables; the caller resets pos / norm / status first)."""
        if not self._graph:
            raise RuntimeError("no captured loop: call sample(..., graph=True) first")
        _lib.check(self.L.dp_sample_graph_launch(self._graph, _stream()), "dp_sample_graph_launch")
