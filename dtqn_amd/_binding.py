"""ctypes binding of libdtqn_hip.so, generated from include/dtqn_hip.h at import time.

The header is the single source of truth for the C ABI: the struct layouts below are parsed out of
it (members are restricted to int32_t / uint32_t / float / pointers there), so the Python side can
never drift from the library.  This module only knows how to LOAD a library and describe its
structs; dtqn_amd.engine decides which library is the product one (the hipcc build for gfx950,
and nothing else -- there is no CPU fallback).
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "dtqn_hip.h")

_SCALARS = {
    "int32_t": ctypes.c_int32, "uint32_t": ctypes.c_uint32, "float": ctypes.c_float, "int": ctypes.c_int,
}


def _strip_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def parse_structs(header_path: str = HEADER) -> Dict[str, List[Tuple[str, object]]]:
    src = _strip_comments(open(header_path).read())
    out: Dict[str, List[Tuple[str, object]]] = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        name, body = m.group(3), m.group(2)
        fields: List[Tuple[str, object]] = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            if "*" in decl:
                for nm in decl.split("*")[-1].split(","):
                    fields.append((nm.strip(), ctypes.c_void_p))
                continue
            toks = decl.split(" ", 1)
            ctype = _SCALARS[toks[0]]
            for nm in toks[1].split(","):
                fields.append((nm.strip(), ctype))
        out[name] = fields
    return out


def parse_defines(header_path: str = HEADER) -> Dict[str, int]:
    src = _strip_comments(open(header_path).read())
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(DTQN_\w+)\s+(-?\d+)\s*$", src, flags=re.M)}


def parse_functions(header_path: str = HEADER) -> List[str]:
    src = _strip_comments(open(header_path).read())
    src = re.sub(r"typedef\s+struct.*?\}\s*\w+\s*;", "", src, flags=re.S)
    return re.findall(r"\b(dtqn_\w+)\s*\(", src)


_STRUCTS = parse_structs()
DEFINES = parse_defines()
FUNCTIONS = sorted(set(parse_functions()))


def _make(name: str):
    return type(name, (ctypes.Structure,), {"_fields_": _STRUCTS[name]})


DtqnNet = _make("DtqnNet")
DtqnWJob = _make("DtqnWJob")
DtqnReplay = _make("DtqnReplay")
DtqnReplayRecord = _make("DtqnReplayRecord")
DtqnTd = _make("DtqnTd")


def load_library(path: str) -> ctypes.CDLL:
    """dlopen `path` and declare the prototypes of every entry point the header lists.
    Raises if the file or any declared symbol is missing."""
    lib = ctypes.CDLL(path)
    missing = [f for f in FUNCTIONS if not hasattr(lib, f)]
    if missing:
        raise OSError(f"{path} does not export {missing} (declared in include/dtqn_hip.h)")
    vp, i32, u32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32
    P = ctypes.POINTER
    protos = {
        "dtqn_net_init": [P(DtqnNet)],
        "dtqn_net_wjobs": [P(DtqnNet), P(DtqnWJob)],
        "dtqn_net_fill_frozen": [P(DtqnNet), vp],
        "dtqn_lds_bytes_forward": [P(DtqnNet), i32],
        "dtqn_lds_bytes_backward": [P(DtqnNet)],
        "dtqn_replay_apply": [P(DtqnReplay), vp, vp, i32, vp],
        "dtqn_replay_sample": [P(DtqnReplay), i32, i32, i32, i32, u32, vp, vp, vp, vp],
        "dtqn_replay_sample_at": [P(DtqnReplay), i32, i32, i32, i32, u32, i32, vp, vp, vp, vp],
        "dtqn_replay_push": [P(DtqnReplay), vp, vp, i32, vp],
        "dtqn_td_prefers_tiled": [P(DtqnNet), i32],
        "dtqn_net_tiled_twin": [P(DtqnNet), P(DtqnNet)],
        "dtqn_replay_gather_bag": [P(DtqnReplay), vp, vp, vp, i32, i32, u32, vp, vp, vp, vp],
        "dtqn_actor_forward": [P(DtqnNet), vp, vp, vp, i32, vp, vp, vp, i32, u32, u32, vp],
        "dtqn_actor_forward_batch": [P(DtqnNet), vp, vp, vp, i32, i32, vp, vp, vp, i32, u32, u32, vp],
        "dtqn_forward_tiled_strided": [P(DtqnNet), vp, vp, vp, i32, i32, i32, vp, vp, vp],
        "dtqn_forward_bag": [P(DtqnNet), vp, vp, vp, vp, vp, i32, i32, vp, vp, i32, u32, u32, vp],
        "dtqn_forward": [P(DtqnNet), vp, vp, vp, i32, i32, vp, vp],
        "dtqn_forward_workspace_floats": [P(DtqnNet), i32],
        "dtqn_forward_tiled": [P(DtqnNet), vp, vp, vp, i32, i32, vp, vp, vp],
        "dtqn_td_row_split": [P(DtqnNet), i32],
        "dtqn_td_latency_mode": [P(DtqnNet), i32],
        "dtqn_td_wgrad_is_direct": [P(DtqnNet), i32],
        "dtqn_td_wgrad_splits": [P(DtqnNet), i32],
        "dtqn_td_wpack_floats": [P(DtqnNet)],
        "dtqn_td_wpack": [P(DtqnNet), P(DtqnTd), vp],
        "dtqn_td_wgrad_is_fused": [P(DtqnNet), P(DtqnTd)],
        "dtqn_td_norm_partials": [P(DtqnNet)],
        "dtqn_td_xch_floats": [P(DtqnNet), i32],
        "dtqn_td_xch_flags": [P(DtqnNet), i32],
        "dtqn_td_forward": [P(DtqnNet), P(DtqnReplay), P(DtqnTd), vp],
        "dtqn_td_forward_part": [P(DtqnNet), P(DtqnReplay), P(DtqnTd), i32, i32, i32, i32, vp],
        "dtqn_td_fwd_slices4_ok": [P(DtqnNet)],
        "dtqn_td_backward": [P(DtqnNet), P(DtqnReplay), P(DtqnTd), vp],
        "dtqn_td_backward_ahead": [P(DtqnNet), P(DtqnReplay), P(DtqnTd), P(DtqnTd), i32, vp],
        "dtqn_td_wgrad": [P(DtqnNet), P(DtqnTd), vp],
        "dtqn_td_reduce": [P(DtqnNet), P(DtqnTd), vp],
        "dtqn_td_gradnorm": [P(DtqnNet), P(DtqnTd), vp],
        "dtqn_td_clip_adam": [P(DtqnNet), P(DtqnTd), vp],
        "dtqn_img_prep_floats": [P(DtqnNet)],
        "dtqn_img_act_floats": [P(DtqnNet), i32],
        "dtqn_img_gact_floats": [P(DtqnNet), i32],
        "dtqn_img_wpart_floats": [P(DtqnNet)],
        "dtqn_img_prep": [P(DtqnNet), vp, vp, vp],
        "dtqn_img_encode": [P(DtqnNet), vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp],
        "dtqn_img_backward": [P(DtqnNet), vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp],
        "dtqn_img_td_lists": [P(DtqnNet), P(DtqnReplay), P(DtqnTd), vp, vp, vp, vp, vp, vp, vp],
        "dtqn_forward_tiled_pre": [P(DtqnNet), vp, vp, vp, i32, i32, vp, vp, i32, u32, u32, vp],
        "dtqn_xch_publish": [vp, i32, vp],
        "dtqn_td_xreduce": [P(DtqnNet), P(DtqnTd), vp, vp, i32, i32, vp, vp, vp, vp],
        "dtqn_td_update": [P(DtqnNet), P(DtqnReplay), P(DtqnTd), vp],
        "dtqn_td_update_pipelined": [P(DtqnNet), P(DtqnReplay), P(DtqnTd), P(DtqnTd), i32, i32, vp],
        "dtqn_td_gradients_pipelined": [P(DtqnNet), P(DtqnReplay), P(DtqnTd), P(DtqnTd), i32, i32, vp],
        "dtqn_target_sync": [P(DtqnNet), vp, vp, vp],
        "dtqn_debug_set_profile_buffer": [vp],
        "dtqn_debug_last_packed_blocks": [],
        "dtqn_abi_version": [],
        "dtqn_build_info": [],
    }
    for name, argtypes in protos.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_char_p if name == "dtqn_build_info" else (
            ctypes.c_longlong if name in ("dtqn_img_act_floats", "dtqn_img_gact_floats", "dtqn_img_wpart_floats") else ctypes.c_int)
    assert set(protos) == set(FUNCTIONS), sorted(set(protos) ^ set(FUNCTIONS))
    if lib.dtqn_abi_version() != DEFINES["DTQN_ABI_VERSION"]:
        raise OSError(f"{path}: ABI version {lib.dtqn_abi_version()} != header {DEFINES['DTQN_ABI_VERSION']}")
    return lib


GATES = {"res": DEFINES["DTQN_GATE_RES"], "gru": DEFINES["DTQN_GATE_GRU"]}
POS = {"learned": DEFINES["DTQN_POS_LEARNED"], "sin": DEFINES["DTQN_POS_SIN"], "none": DEFINES["DTQN_POS_NONE"]}


def make_net(lib, *, obs_dim, num_actions, embed_per_obs_dim=8, action_dim=0, inner_embed_size=64, num_heads=8,
             num_layers=2, history_len=50, gate="res", identity=False, pos="learned", discrete=False,
             vocab_sizes=0, dropout=0.0, bag_size=0, image=None) -> DtqnNet:
    """Build and initialise a DtqnNet from the reference's DTQN constructor arguments
    (dtqn/networks/dtqn.py:41-59)."""
    net = DtqnNet()
    net.obs_dim, net.num_actions, net.embed_per_obs, net.action_dim = obs_dim, num_actions, embed_per_obs_dim, action_dim
    net.d_model, net.num_heads, net.num_layers, net.ctx_len = inner_embed_size, num_heads, num_layers, history_len
    if gate not in GATES:
        raise ValueError("Gate must be one of `gru`, `res`")          # dtqn.py:114
    net.gate, net.identity, net.pos = GATES[gate], int(bool(identity)), POS[str(pos)]
    net.discrete, net.vocab = int(bool(discrete)), int(vocab_sizes or 0)
    net.dropout = float(dropout)
    net.bag_size = int(bag_size)
    if image is not None:                    # obs_dim as the reference's (C, H, W) tuple (dtqn.py:71-77)
        net.img_c, net.img_h, net.img_w = (int(v) for v in image)
        net.obs_dim = net.img_c * net.img_h * net.img_w
    rc = lib.dtqn_net_init(ctypes.byref(net))
    if rc != 0:
        raise NotImplementedError(
            f"dtqn_net_init rc={rc}: this DTQN variant/shape is outside the gfx950 kernels' coverage "
            f"(D={inner_embed_size}, H={num_heads}, L={history_len}, gate={gate}, dropout={dropout}, bag_size={bag_size}); see DESIGN.md")
    return net


def param_table(net: DtqnNet, width: int = 0) -> Dict[str, Tuple[int, Tuple[int, ...]]]:
    """state_dict key -> (offset into theta, shape), using the reference's key names
    (SURVEY.md section 8b).  Shared GRU gates appear under every layer prefix with the same offset.
    The shapes are those of the buffer (d_model wide; for a width-padded network that is the padded width); `width` = net.d_real
    gives the shapes the reference's state_dict has for the same keys (same offsets: see pad_param / unpad_param)."""
    D, L, A, a = (width or net.d_model), net.ctx_len, net.num_actions, net.action_dim
    tab: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
    if a > 0:
        tab["action_embedding.embedding.0.weight"] = (net.off_act_emb, (A, a))
    if net.img_c > 0:        # representations.py:99-129: nn.Sequential indices 0, 2, 4, 6, 8 are the convolutions, 11 the Linear
        chans = [(net.img_c, 64), (64, 64), (64, 64), (64, 128), (128, 128)]
        offs = [(net.off_cw0, net.off_cb0), (net.off_cw1, net.off_cb1), (net.off_cw2, net.off_cb2), (net.off_cw3, net.off_cb3), (net.off_cw4, net.off_cb4)]
        for i, ((ci, co), (ow, ob)) in enumerate(zip(chans, offs)):
            tab[f"obs_embedding.observation_embedding.{2 * i}.weight"] = (ow, (co, ci, 3, 3))
            tab[f"obs_embedding.observation_embedding.{2 * i}.bias"] = (ob, (co,))
        tab["obs_embedding.observation_embedding.11.weight"] = (net.off_obs_w, (D - a, net.ke))
        tab["obs_embedding.observation_embedding.11.bias"] = (net.off_obs_b, (D - a,))
    elif net.discrete:
        tab["obs_embedding.observation_embedding.0.weight"] = (net.off_obs_tab, (net.vocab, net.embed_per_obs))
        tab["obs_embedding.observation_embedding.2.weight"] = (net.off_obs_w, (D - a, net.ke))
        tab["obs_embedding.observation_embedding.2.bias"] = (net.off_obs_b, (D - a,))
    else:
        tab["obs_embedding.observation_embedding.weight"] = (net.off_obs_w, (D - a, net.ke))
        tab["obs_embedding.observation_embedding.bias"] = (net.off_obs_b, (D - a,))
    tab["position_embedding.position_encoding"] = (net.off_pos, (1, L, D))
    gate_names = [("w_r.weight", net.go_w_r, (D, D)), ("u_r.weight", net.go_u_r, (D, D)),
                  ("w_z.weight", net.go_w_z, (D, D)), ("w_z.bias", net.go_b_z, (D,)),
                  ("u_z.weight", net.go_u_z, (D, D)), ("w_g.weight", net.go_w_g, (D, D)),
                  ("u_g.weight", net.go_u_g, (D, D))]
    for l in range(net.num_layers):
        base = net.off_layer0 + l * net.layer_stride
        pre = f"transformer_layers.{l}."
        tab[pre + "layernorm1.weight"] = (base + net.lo_ln1_w, (D,))
        tab[pre + "layernorm1.bias"] = (base + net.lo_ln1_b, (D,))
        tab[pre + "layernorm2.weight"] = (base + net.lo_ln2_w, (D,))
        tab[pre + "layernorm2.bias"] = (base + net.lo_ln2_b, (D,))
        tab[pre + "attention.in_proj_weight"] = (base + net.lo_in_w, (3 * D, D))
        tab[pre + "attention.in_proj_bias"] = (base + net.lo_in_b, (3 * D,))
        tab[pre + "attention.out_proj.weight"] = (base + net.lo_out_w, (D, D))
        tab[pre + "attention.out_proj.bias"] = (base + net.lo_out_b, (D,))
        tab[pre + "ffn.0.weight"] = (base + net.lo_f1_w, (4 * D, D))
        tab[pre + "ffn.0.bias"] = (base + net.lo_f1_b, (4 * D,))
        tab[pre + "ffn.2.weight"] = (base + net.lo_f2_w, (D, 4 * D))
        tab[pre + "ffn.2.bias"] = (base + net.lo_f2_b, (D,))
        if net.gate == GATES["gru"]:
            for gname, goff in (("attn_gate", net.off_gate_attn), ("mlp_gate", net.off_gate_mlp)):
                for nm, off, shp in gate_names:
                    tab[pre + f"{gname}.{nm}"] = (goff + off, shp)
    if net.bag_size > 0:       # dtqn.py:134-139: registered after the transformer layers, before the head
        tab["bag_attention.in_proj_weight"] = (net.off_bag_in_w, (3 * D, D))
        tab["bag_attention.in_proj_bias"] = (net.off_bag_in_b, (3 * D,))
        tab["bag_attention.out_proj.weight"] = (net.off_bag_out_w, (D, D))
        tab["bag_attention.out_proj.bias"] = (net.off_bag_out_b, (D,))
    tab["ffn.0.weight"] = (net.off_head1_w, (D, 2 * D if net.bag_size > 0 else D))
    tab["ffn.0.bias"] = (net.off_head1_b, (D,))
    tab["ffn.2.weight"] = (net.off_head2_w, (A, D))
    tab["ffn.2.bias"] = (net.off_head2_b, (A,))
    return tab


def ws_lite(net: DtqnNet) -> bool:
    """dtqn_limits.h dtqn_ws_lite: whole-sequence shapes that exist as four-slice (latency-mode) kernels only."""
    return (not net.tiled) and net.d_model == 64 and (net.head_dim == 32 or (net.d_real > 0 and net.head_dim in (8, 16)))


# ---- width-padded networks (include/dtqn_hip.h, DtqnNet.d_real) ---------------------------------------------------------------
# theta holds every tensor at the padded shape with the real entries in front -- except along a head-structured axis, where each
# head's real entries sit in front of that head's block: attention.in_proj_* rows are [q | k | v][head][width], attention.out_proj
# columns are [head][width].  The two functions below move one tensor between the reference's shape and that layout.
def _grids(net: DtqnNet, key: str, real_shape, padded_shape):
    """(real grid, padded grid): the tensor's shape with its head-structured axis split into (blocks, heads, width); the real tensor is
    the leading corner of the padded one in that form."""
    H, Hp = net.heads_real or net.num_heads, net.num_heads
    hd, hdp = net.hd_real or net.head_dim, net.head_dim
    if key.endswith("attention.in_proj_weight") or key.endswith("attention.in_proj_bias"):
        return (3, H, hd) + tuple(real_shape[1:]), (3, Hp, hdp) + tuple(padded_shape[1:])
    if key.endswith("attention.out_proj.weight"):
        return (real_shape[0], H, hd), (padded_shape[0], Hp, hdp)
    return tuple(real_shape), tuple(padded_shape)


def unpad_param(net: DtqnNet, key: str, padded, real_shape):
    """The reference-shaped part of a buffer-shaped tensor (torch or numpy; a view where the layout allows, else a copy)."""
    if not net.d_real or tuple(padded.shape) == tuple(real_shape):
        return padded
    rg, pg = _grids(net, key, real_shape, padded.shape)
    return padded.reshape(pg)[tuple(slice(0, r) for r in rg)].reshape(tuple(real_shape))


def pad_param(net: DtqnNet, key: str, real, padded_shape):
    """A reference-shaped tensor laid out at the buffer's shape, zeros in the padding (numpy in, numpy out; torch in, torch out)."""
    if not net.d_real or tuple(real.shape) == tuple(padded_shape):
        return real
    import numpy as _np
    out = _np.zeros(padded_shape, dtype=real.dtype) if isinstance(real, _np.ndarray) else real.new_zeros(tuple(padded_shape))
    rg, pg = _grids(net, key, real.shape, padded_shape)
    out.reshape(pg)[tuple(slice(0, r) for r in rg)] = real.reshape(rg)        # reshape of the fresh contiguous `out` is a view
    return out
