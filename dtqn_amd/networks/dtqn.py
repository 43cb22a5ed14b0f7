"""DTQN network object with the reference's constructor, forward signature and state_dict layout
(dtqn/networks/dtqn.py:16-218), backed by ONE flat fp32 device buffer that the gfx950 kernels
read directly.

The module is a parameter CONTAINER plus an inference `forward`: every nn.Parameter is a view
into `self.flat` (layout: include/dtqn_hip.h, DtqnNet), so `state_dict()` / `load_state_dict()` /
`parameters()` behave like the reference's -- a policy / target `state_dict` saved by either implementation loads
in the other (the FULL training checkpoints do not: the reference pickles joblib / torch-optimizer objects, dtqn_amd
writes plain arrays) --, while the engine
sees a single contiguous theta.  Training does not go through autograd: DtqnAgent.train() runs the
fused HIP update on these same buffers.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Union

import numpy as np
import torch
import torch.nn as nn

from .. import _binding as B
from .. import engine


class _Node(nn.Module):
    """Anonymous container used to reproduce the reference's dotted state_dict keys."""


def _attach(root: nn.Module, dotted: str, param: nn.Parameter) -> None:
    parts = dotted.split(".")
    mod = root
    for name in parts[:-1]:
        if name not in mod._modules:
            mod.add_module(name, _Node())
        mod = mod._modules[name]
    mod.register_parameter(parts[-1], param)


def _img_check_every() -> int:
    import os
    try:
        return max(0, int(os.environ.get("DTQN_IMG_CHECK_EVERY", "64")))
    except ValueError:
        return 64


class DTQN(nn.Module):
    """Deep Transformer Q-Network.  Arguments as in the reference (dtqn.py:19-59); `pos` defaults
    to "learned" (the reference's default value 1 is rejected by its own PosEnum)."""

    def __init__(self, obs_dim: int, num_actions: int, embed_per_obs_dim: int, action_dim: int,
                 inner_embed_size: int, num_heads: int, num_layers: int, history_len: int, dropout: float = 0.0,
                 gate: str = "res", identity: bool = False, pos: Union[str, int] = "learned", discrete: bool = False,
                 vocab_sizes: Optional[Union[np.ndarray, int]] = None, bag_size: int = 0, _test_lib=None, **kwargs):
        super().__init__()
        image = tuple(int(v) for v in obs_dim) if isinstance(obs_dim, (tuple, list, torch.Size)) else None
        if image is not None and len(image) == 2:
            image = (1,) + image                      # representations.py:88-92: H x W means one channel
        if not 0.0 <= dropout < 1.0:
            raise ValueError(f"dropout probability has to be between 0 and 1, but got {dropout}")     # nn.Dropout's own check
        if pos not in B.POS:
            raise ValueError(f"{pos!r} is not a valid PosEnum")        # PosEnum(pos) in the reference (dtqn.py:101)
        self._lib = _test_lib if _test_lib is not None else engine.get_lib()
        self.obs_dim, self.discrete, self.history_len, self.bag_size = (image if image is not None else obs_dim), discrete, history_len, bag_size
        self.image = image
        self.dropout_p = float(dropout)
        self.num_actions = num_actions
        self.net = B.make_net(self._lib, obs_dim=1 if image is not None else obs_dim, image=image, num_actions=num_actions, embed_per_obs_dim=embed_per_obs_dim,
                              action_dim=action_dim, inner_embed_size=inner_embed_size, num_heads=num_heads,
                              num_layers=num_layers, history_len=history_len, gate=gate, identity=identity, pos=pos,
                              discrete=discrete, vocab_sizes=int(vocab_sizes) if discrete else 0, dropout=dropout,
                              bag_size=bag_size)
        net = self.net
        flat = np.zeros(net.n_theta, dtype=np.float32)
        self._lib.dtqn_net_fill_frozen(ctypes.byref(net), flat.ctypes.data_as(ctypes.c_void_p))
        self.flat = torch.from_numpy(flat)
        self._table = B.param_table(net)
        # width-padded network (DtqnNet.d_real): the buffer holds every tensor at the padded width; state_dict() / load_state_dict()
        # speak the reference's shapes (`_real_shapes`), parameters() are the buffer's own views
        self._real_shapes = {k: shp for k, (_, shp) in B.param_table(net, net.d_real).items()} if net.d_real else None
        self._views = {}
        seen = {}
        for key, (off, shape) in self._table.items():
            trainable = off < net.n_trainable
            if off in seen:                    # shared GRU gate: one Parameter under every layer prefix
                p = seen[off]
            else:
                p = nn.Parameter(self.flat[off:off + int(np.prod(shape))].view(shape), requires_grad=trainable)
                seen[off] = p
                self._views[key] = (p, off, shape)
            _attach(self, key, p)
        # the reference keeps the causal mask as a frozen Parameter per layer (transformer.py:49-53);
        # the kernels never read it, it exists for state_dict compatibility
        mask = torch.triu(torch.ones(history_len, history_len), diagonal=1)
        mask[mask.bool()] = -float("inf")
        for l in range(num_layers):
            _attach(self, f"transformer_layers.{l}.attn_mask", nn.Parameter(mask.clone(), requires_grad=False))
        self.reset_parameters()

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def reset_parameters(self) -> None:
        """utils/torch_utils.py:4-15 as applied by DTQN.__init__ (dtqn.py:156): N(0, 0.02) weights of
        Linear / Embedding / MultiheadAttention, zero biases, LayerNorm (1, 0); the learned position
        table is a bare Parameter and stays 0 (position_encodings.py:41-43)."""
        for key, (p, off, shape) in self._views.items():
            if key == "position_embedding.position_encoding":
                continue
            if self.image is not None and key.startswith("obs_embedding.observation_embedding.") and int(key.split(".")[2]) <= 8:
                # nn.Conv2d is not touched by init_weights (utils/torch_utils.py:4-15): it keeps torch's default,
                # kaiming_uniform(a = sqrt 5) = U(-1 / sqrt(fan_in), 1 / sqrt(fan_in)) for weight and bias alike
                w_shape = self._views[key.rsplit(".", 1)[0] + ".weight"][2]
                bound = 1.0 / float(np.sqrt(w_shape[1] * 9))
                p.uniform_(-bound, bound)
            elif "layernorm" in key:
                p.fill_(1.0 if key.endswith("weight") else 0.0)
            elif key.endswith("bias"):
                p.zero_()
            else:
                p.normal_(mean=0.0, std=0.02)
        self._zero_padding()

    @torch.no_grad()
    def _zero_padding(self) -> None:
        """Width-padded network: everything outside the reference-shaped part of each tensor is zero (and stays zero under training:
        include/dtqn_hip.h, d_real)."""
        if self._real_shapes is None:
            return
        for key, (p, off, shape) in self._views.items():
            real_shape = self._real_shapes[key]
            if tuple(real_shape) != tuple(shape):
                real = B.unpad_param(self.net, key, p.data, real_shape).clone()
                p.data.zero_()
                p.data.copy_(B.pad_param(self.net, key, real, tuple(shape)))

    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        if self._real_shapes is not None:           # the reference's shapes (slices of the buffer; the stacked in_proj tensors as copies)
            prefix = kwargs.get("prefix", args[1] if len(args) > 1 else "")
            for key, real_shape in self._real_shapes.items():
                if prefix + key in sd:
                    sd[prefix + key] = B.unpad_param(self.net, key, sd[prefix + key], real_shape)
        return sd

    def load_state_dict(self, state_dict, *args, **kwargs):
        if self._real_shapes is not None:
            state_dict = dict(state_dict)
            for key, (off, shape) in self._table.items():
                v = state_dict.get(key)
                if v is not None and tuple(v.shape) == tuple(self._real_shapes[key]) and tuple(v.shape) != tuple(shape):
                    state_dict[key] = B.pad_param(self.net, key, v.detach(), tuple(shape))
        return super().load_state_dict(state_dict, *args, **kwargs)

    def _apply(self, fn, recurse=True):
        """Module.to()/cuda()/float(): move the flat buffer once and re-point every view at it."""
        super()._apply(fn, recurse)
        self.flat = fn(self.flat)
        if not self.flat.is_contiguous():
            self.flat = self.flat.contiguous()
        for key, (p, off, shape) in self._views.items():
            p.data = self.flat[off:off + int(np.prod(shape))].view(shape)
        return self

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _forward_images(self, obss: torch.Tensor, dev, _train_dropout=None) -> torch.Tensor:
        """obss [B, seq, C, H, W] (uint8 pixels; the reference casts them to float unscaled) -> Q [B, seq, A]: the convolutional
        embedding (dtqn_img_encode) in front of the row-block forward on precomputed embeddings."""
        from ..image import ImageEncoder
        if dev.type != "cuda" and not getattr(self, "_allow_cpu", False):
            raise engine.EngineUnavailable("DTQN.forward runs on the gfx950 engine only: move the module to a ROCm device")
        Bn, seq = int(obss.size(0)), int(obss.size(1))
        if obss.dtype != torch.uint8:
            # the encoder's first layer gathers uint8 pixels (the reference's replay and context hold images as uint8 and the
            # network sees their float VALUES, dtqn/agents/dtqn.py:199): integral values in 0..255 of any dtype are accepted,
            # anything else (normalised 0..1 images, negative values) would be silently truncated or wrapped by a cast
            # Checked where it costs nothing to ask: host-side inputs every time (no device round trip), device-side inputs on the
            # first such call only (the check is a full pass over the pixels and a blocking read-back: not for the actor's hot path)
            # Device-side inputs afterwards: every DTQN_IMG_CHECK_EVERY-th call (default 64; 1 = every call, 0 = first call only), so a
            # caller that starts feeding normalised images later is still told, at a bounded cost
            n_seen = getattr(self, "_img_range_calls", 0)
            every = _img_check_every()
            if obss.device.type == "cpu" or n_seen == 0 or (every > 0 and n_seen % every == 0):
                o = obss.to(torch.float32)
                if not bool(((o >= 0) & (o <= 255) & (o == o.round())).all()):
                    raise ValueError("image observations must be integral pixel values in 0..255 (uint8 in the replay and the context)")
            if obss.device.type != "cpu":
                self._img_range_calls = n_seen + 1
        imgs = obss.to(device=dev).to(torch.uint8).reshape(Bn * seq, -1).contiguous()
        enc = getattr(self, "_img_enc", None)
        if enc is None or enc.device != dev:
            enc = self._img_enc = ImageEncoder(self._lib, self.net, dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else None
        tokens = Bn * seq
        idx = torch.arange(tokens, dtype=torch.int32, device=dev)
        xemb = torch.empty(tokens, self.net.d_model, dtype=torch.float32, device=dev)
        enc.prep(self.flat, stream)
        enc.encode(self.flat, imgs, idx, tokens, enc.act_buffer(tokens, "infer"), xemb, idx, stream=stream)
        need = self._lib.dtqn_forward_workspace_floats(ctypes.byref(self.net), Bn)
        ws = getattr(self, "_tiled_ws", None)
        if ws is None or ws.numel() < need or ws.device != dev:
            ws = self._tiled_ws = torch.empty(need, dtype=torch.float32, device=dev)
        q = torch.empty((Bn, seq, self.num_actions), dtype=torch.float32, device=dev)
        td_ = _train_dropout if (_train_dropout is not None and self.dropout_p > 0.0) else None
        cp = lambda t: ctypes.c_void_p(t.data_ptr())
        rc = self._lib.dtqn_forward_tiled_pre(ctypes.byref(self.net), cp(self.flat), cp(xemb), None, Bn, seq, cp(q), cp(ws), 1 if td_ else 0,
                                              int(td_[0]) & 0xFFFFFFFF if td_ else 0, int(td_[1]) & 0xFFFFFFFF if td_ else 0, stream)
        if rc != 0:
            raise RuntimeError(f"dtqn_forward_tiled_pre failed with DTQN status {rc}")
        return q

    @torch.no_grad()
    def forward(self, obss: torch.Tensor, actions: Optional[torch.Tensor] = None,
                bag_obss: Optional[torch.Tensor] = None, bag_actions: Optional[torch.Tensor] = None,
                _train_dropout: Optional[tuple] = None) -> torch.Tensor:
        """obss [B, seq, obs_dim] (float, or integer tokens for discrete envs), actions [B, seq, 1]
        -> Q [B, seq, num_actions].  Inference only (no autograd graph)."""
        seq = obss.size(1)
        assert seq <= self.history_len, "Cannot forward, history is longer than expected."
        # images: obs_dim is the (C, H, W) shape of one observation (dtqn.py:175-179)
        obs_dim = tuple(obss.size()[2:]) if obss.dim() > 3 else obss.size(2)
        if self.image is not None and obss.dim() == 4 and self.image[0] == 1:
            obs_dim = (1,) + obs_dim                  # an H x W observation is one channel (representations.py:88-92)
        assert obs_dim == self.obs_dim, f"Obs dim is incorrect. Expected {self.obs_dim} got {obs_dim}"
        dev = self.flat.device
        if self.image is not None:
            return self._forward_images(obss, dev, _train_dropout)
        if dev.type != "cuda" and not getattr(self, "_allow_cpu", False):
            raise engine.EngineUnavailable("DTQN.forward runs on the gfx950 engine only: move the module to a ROCm device")
        o = obss.to(device=dev, dtype=torch.float32).contiguous()
        a = None
        if self.net.action_dim > 0:
            a = actions.to(device=dev).reshape(obss.size(0), seq).to(torch.uint8).contiguous()
        q = torch.empty((obss.size(0), seq, self.num_actions), dtype=torch.float32, device=dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else None
        if self.net.tiled:      # contexts / widths beyond one workgroup's LDS (and bag networks): row-block tiled kernels + a workspace
            need = self._lib.dtqn_forward_workspace_floats(ctypes.byref(self.net), int(obss.size(0)))
            ws = getattr(self, "_tiled_ws", None)
            if ws is None or ws.numel() < need or ws.device != dev:
                ws = self._tiled_ws = torch.empty(need, dtype=torch.float32, device=dev)
            cp = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
            if self.bag_size > 0:
                # bag_obss [B, bag_size, obs_dim], bag_actions [B, bag_size, 1] (dtqn.py:158-164,201-214)
                assert bag_obss is not None and bag_obss.size(1) == self.bag_size, "bag_obss must be [B, bag_size, obs_dim]"
                bo = bag_obss.to(device=dev, dtype=torch.float32).contiguous()
                ba = None
                if self.net.action_dim > 0:
                    ba = bag_actions.to(device=dev).reshape(obss.size(0), self.bag_size).to(torch.uint8).contiguous()
                # _train_dropout = (seed, step): a train-mode forward of the agent (acting, bag eviction) with dropout > 0
                td_ = _train_dropout if (_train_dropout is not None and self.dropout_p > 0.0) else None
                rc = self._lib.dtqn_forward_bag(ctypes.byref(self.net), cp(self.flat), cp(o), cp(a), cp(bo), cp(ba), int(obss.size(0)), int(seq),
                                                cp(q), cp(ws), 1 if td_ else 0, int(td_[0]) & 0xFFFFFFFF if td_ else 0,
                                                int(td_[1]) & 0xFFFFFFFF if td_ else 0, stream)
                if rc != 0:
                    raise RuntimeError(f"dtqn_forward_bag failed with DTQN status {rc}")
                return q
            rc = self._lib.dtqn_forward_tiled(ctypes.byref(self.net), cp(self.flat), cp(o), cp(a), int(obss.size(0)), int(seq), cp(q), cp(ws), stream)
            if rc != 0:
                raise RuntimeError(f"dtqn_forward_tiled failed with DTQN status {rc}")
            return q
        rc = self._lib.dtqn_forward(ctypes.byref(self.net), ctypes.c_void_p(self.flat.data_ptr()),
                                    ctypes.c_void_p(o.data_ptr()), None if a is None else ctypes.c_void_p(a.data_ptr()),
                                    int(obss.size(0)), int(seq), ctypes.c_void_p(q.data_ptr()), stream)
        if rc != 0:
            raise RuntimeError(f"dtqn_forward failed with DTQN status {rc}")
        return q
