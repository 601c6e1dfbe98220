"""ctypes binding of libdiffphore_hip.so (C ABI: include/diffphore_amd.h).  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DIFFPHORE_LIB: an alternative build of the library (A/B and ablation builds, tools/); the product default is the in-tree build
LIB_PATH = os.environ.get("DIFFPHORE_LIB") or os.path.join(_HERE, "csrc", "libdiffphore_hip.so")

c_i32 = C.c_int32
c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
VOIDP = C.c_void_p

DP_ABI_VERSION = 12

CONV_L0, CONV_L1, CONV_L2, CONV_L3, CONV_FINAL, CONV_TOR = range(6)
PREC_F32, PREC_BF16X3, PREC_BF16X6 = 0, 2, 3
PREC_BF16X9 = 4
PREC_BF16 = 5          # plain bf16 operands, fp32 accumulate: node-side dp_conv_node_bf16_impl.h, edge-side the one-product split-bf16 kernel
PREC_F16X3 = 6         # node-side kernel only, fp32-grade: two fp16 limbs per operand on v_mfma_f32_16x16x32_f16 (dp_conv_node_f16_impl.h); not a
                       # model precision - what the node-side convs of an "f32" model run on unless DIFFPHORE_NODE_F16=0
PRECISIONS = {"f32": PREC_F32, "bf16x3": PREC_BF16X3, "bf16x6": PREC_BF16X6, "bf16x9": PREC_BF16X9, "bf16": PREC_BF16}
EDGES_CSR, EDGES_P2L, EDGES_L2P, EDGES_CENTER, EDGES_LLPAIR, EDGES_LIST = range(6)
ALGO_AUTO, ALGO_EDGE, ALGO_NODE = range(3)


class DpConvArgs(C.Structure):
    _fields_ = [
        ("kind", c_i32), ("mode", c_i32), ("B", c_i32), ("slices", c_i32), ("precision", c_i32), ("max_scatter", c_i32),
        ("scat_ptr", VOIDP), ("lig_ptr", VOIDP), ("ph_ptr", VOIDP), ("x_ptr", VOIDP),
        ("rowptr", VOIDP), ("rowend", VOIDP), ("nbr", VOIDP), ("edge_scat", VOIDP),
        ("gather_tab", VOIDP), ("gather_stride", c_i32),
        ("attr1_tab", VOIDP), ("attr1_stride", c_i32),
        ("attr2_tab", VOIDP), ("attr2_stride", c_i32), ("attr_swap", c_i32),
        ("edge_emb", VOIDP), ("edge_sh", VOIDP), ("sh_stride", c_i32),
        ("w1p", VOIDP), ("w2p", VOIDP), ("bn_scale", VOIDP), ("bn_shift", VOIDP),
        ("out", VOIDP), ("out_stride", c_i32), ("accumulate", c_i32),
        ("resid", VOIDP), ("resid_stride", c_i32), ("resid_width", c_i32), ("phase_split", c_i32),
        ("w1n", VOIDP), ("w2n", VOIDP), ("algo", c_i32), ("path_split", c_i32),
        ("pair_a", VOIDP), ("pair_b", VOIDP), ("pair_ptr", VOIDP), ("edge_emb_bf", VOIDP), ("emb_flag_row", c_i32), ("no_tail", c_i32),
        ("s0_only", c_i32), ("feat_idx", VOIDP), ("msg_out", VOIDP), ("n_list", c_i32), ("sh0_zero", c_i32),
        ("list_msg", VOIDP), ("list_rowptr", VOIDP), ("list_scale", VOIDP),
    ]


class DpBuildInfo(C.Structure):
    _fields_ = [("abi_version", c_i32), ("nabl", c_i32), ("ablate", c_i32), ("prof", c_i32), ("source_hash", C.c_char * 20)]


class DiffPhoreLibraryError(RuntimeError):
    pass


_lib = None


def lib():
    """Load (once) and return the HIP library; raises loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DiffPhoreLibraryError(
                f"{LIB_PATH} not found: build it with `python diffphore_amd/csrc/build.py` "
                "(there is no CPU / PyTorch fallback for the DiffPhore hot path)")
        L = C.CDLL(LIB_PATH)
        L.dp_abi_version.restype = C.c_int
        if L.dp_abi_version() != DP_ABI_VERSION:
            raise DiffPhoreLibraryError(f"ABI mismatch: library {L.dp_abi_version()} != binding {DP_ABI_VERSION}")
        _declare(L)
        info = DpBuildInfo()
        if L.dp_build_info(C.byref(info)) != 0:
            raise DiffPhoreLibraryError("dp_build_info failed")
        if (info.nabl or info.ablate) and os.environ.get("DIFFPHORE_ALLOW_ABLATION") != "1":
            # -DDP_NABL / -DDP_ABLATE builds replace parts of the kernels by stand-ins to TIME the rest: their numbers are wrong by design
            raise DiffPhoreLibraryError(f"{LIB_PATH} is an ablation build (DP_NABL={info.nabl}, DP_ABLATE={info.ablate}): it computes wrong results "
                                        "by construction; set DIFFPHORE_ALLOW_ABLATION=1 to load it for timing")
        _lib = L
    return _lib


def build_info() -> dict:
    """What the loaded library was built from: ABI version, ablation / profiling switches, hash of the kernel sources."""
    info = DpBuildInfo()
    check(lib().dp_build_info(C.byref(info)), "dp_build_info")
    return dict(abi_version=info.abi_version, nabl=info.nabl, ablate=info.ablate, prof=info.prof, source_hash=info.source_hash.decode())


def _declare(L):
    L.dp_tp_conv.argtypes = [C.POINTER(DpConvArgs), VOIDP]
    L.dp_tp_conv.restype = C.c_int
    L.dp_conv_w1p_floats.argtypes = [c_i32]
    L.dp_conv_w1p_floats.restype = C.c_int
    L.dp_conv_w2p_floats.argtypes = [c_i32, c_i32]
    L.dp_conv_w2p_floats.restype = C.c_int
    L.dp_conv_dims.argtypes = [c_i32] + [c_i32p] * 5
    L.dp_conv_dims.restype = C.c_int
    L.dp_pack_conv_weights.argtypes = [c_i32, c_i32, VOIDP, VOIDP, VOIDP, VOIDP, VOIDP, VOIDP]
    L.dp_pack_conv_weights.restype = C.c_int
    L.dp_list_reduce.argtypes = [VOIDP, VOIDP, c_i32, VOIDP, c_i32, VOIDP, c_i32, VOIDP, VOIDP, c_i32, VOIDP]
    L.dp_list_reduce.restype = C.c_int
    L.dp_list_finish.argtypes = [VOIDP, VOIDP, VOIDP, c_i32, c_i32, VOIDP, VOIDP, VOIDP, c_i32, c_i32, VOIDP, c_i32, c_i32, c_i32, VOIDP]
    L.dp_list_finish.restype = C.c_int
    L.dp_tp_conv_node.argtypes = [C.POINTER(DpConvArgs), VOIDP]
    L.dp_tp_conv_node_fused.argtypes = [C.POINTER(DpConvArgs), c_i32, VOIDP]
    L.dp_tp_conv_node_fused.restype = C.c_int
    L.dp_tp_conv_node_fused_split.argtypes = [C.POINTER(DpConvArgs), c_i32, c_i32, VOIDP, C.c_int64, VOIDP]
    L.dp_tp_conv_node_fused_split.restype = C.c_int
    L.dp_tp_conv_node.restype = C.c_int
    L.dp_conv_w1n_floats.argtypes = [c_i32]
    L.dp_conv_w1n_floats.restype = C.c_int
    L.dp_conv_w2n_floats.argtypes = [c_i32]
    L.dp_conv_w2n_floats.restype = C.c_int
    L.dp_pack_conv_weights_node.argtypes = [c_i32] + [VOIDP] * 7
    L.dp_pack_conv_weights_node.restype = C.c_int
    L.dp_conv_node_max_nodes.restype = C.c_int
    L.dp_conv_w1b_floats.argtypes = [c_i32]
    L.dp_conv_w1b_floats.restype = C.c_int
    L.dp_conv_w2b_floats.argtypes = [c_i32]
    L.dp_conv_w2b_floats.restype = C.c_int
    L.dp_pack_conv_weights_node_bf16.argtypes = [c_i32] + [VOIDP] * 7
    L.dp_pack_conv_weights_node_bf16.restype = C.c_int
    L.dp_edge_emb_bf16.argtypes = [VOIDP, c_i32, VOIDP, VOIDP]
    L.dp_edge_emb_bf16.restype = C.c_int
    L.dp_conv_w1h_floats.argtypes = [c_i32]
    L.dp_conv_w1h_floats.restype = C.c_int
    L.dp_conv_w2h_floats.argtypes = [c_i32]
    L.dp_conv_w2h_floats.restype = C.c_int
    L.dp_pack_conv_weights_node_f16.argtypes = [c_i32] + [VOIDP] * 7
    L.dp_pack_conv_weights_node_f16.restype = C.c_int
    L.dp_edge_emb_f16.argtypes = [VOIDP, c_i32, VOIDP, VOIDP]
    L.dp_edge_emb_f16.restype = C.c_int
    L.dp_build_info.argtypes = [C.POINTER(DpBuildInfo)]
    L.dp_build_info.restype = C.c_int


def check(rc: int, what: str):
    if rc != 0:
        raise DiffPhoreLibraryError(f"{what} failed with code {rc}" + (" (bad argument)" if rc == -1 else ""))


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor passed to the C ABI must be contiguous"
    return t.data_ptr()


def conv_dims(kind: int):
    v = [c_i32() for _ in range(5)]
    check(lib().dp_conv_dims(kind, *[C.byref(x) for x in v]), "dp_conv_dims")
    return dict(zip(("d_in", "d_out", "numel", "n_attr", "hidden"), (x.value for x in v)))
