"""Host side of the image observation embedding (dtqn_amd/csrc/dtqn_image.hip; reference: dtqn/networks/representations.py:77-130).

`ImageEncoder` owns the scratch the encoder kernels need -- the per-parameter-version transposed weight copies, the NHWC
feature maps of a token list, gradient ping-pong buffers -- as torch tensors (torch is the allocator) and issues the C-ABI calls.
Sized for 288 GB of HBM: the feature maps of a whole training batch stay resident between forward and backward (6 GB at
32 x 51 tokens of 144 x 144 pixels)."""
from __future__ import annotations

import ctypes

import torch

_vp = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())


class ImageEncoder:
    def __init__(self, lib, net, device):
        self.lib, self.net, self.device = lib, net, torch.device(device)
        self._net_ref = ctypes.byref(net)
        self.wprep = torch.empty(int(lib.dtqn_img_prep_floats(self._net_ref)), dtype=torch.float32, device=self.device)
        self._acts = {}
        self._gact = None
        self._wpart = None

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with DTQN status {rc}")

    def prep(self, theta: torch.Tensor, stream) -> None:
        """theta -> transposed weight copies (call once per parameter version)."""
        self._check(self.lib.dtqn_img_prep(self._net_ref, _vp(theta), _vp(self.wprep), stream), "dtqn_img_prep")

    def act_buffer(self, tokens: int, tag: str) -> torch.Tensor:
        need = int(self.lib.dtqn_img_act_floats(self._net_ref, tokens))
        buf = self._acts.get(tag)
        if buf is None or buf.numel() < need:
            buf = self._acts[tag] = torch.empty(need, dtype=torch.float32, device=self.device)
        return buf

    def encode(self, theta, images_u8, img_index, tokens, act, out0, dst0, out1=None, dst1=None, stream=None) -> None:
        self._check(self.lib.dtqn_img_encode(self._net_ref, _vp(theta), _vp(self.wprep), _vp(images_u8), _vp(img_index), int(tokens),
                                             _vp(act), _vp(out0), _vp(dst0), _vp(out1), _vp(dst1), stream), "dtqn_img_encode")

    def backward(self, theta, images_u8, img_index, tokens, act, dxemb_base, dsrc, grad_out, stream=None) -> None:
        need = int(self.lib.dtqn_img_gact_floats(self._net_ref, tokens))
        if self._gact is None or self._gact.numel() < need:
            self._gact = torch.empty(need, dtype=torch.float32, device=self.device)
        if self._wpart is None:
            self._wpart = torch.empty(int(self.lib.dtqn_img_wpart_floats(self._net_ref)), dtype=torch.float32, device=self.device)
        self._check(self.lib.dtqn_img_backward(self._net_ref, _vp(theta), _vp(self.wprep), _vp(images_u8), _vp(img_index), int(tokens),
                                               _vp(act), _vp(dxemb_base), _vp(dsrc), _vp(self._gact), _vp(self._wpart), _vp(grad_out),
                                               stream), "dtqn_img_backward")
