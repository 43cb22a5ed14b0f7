"""Random image shapes through the convolutional observation embedding (dtqn_image.hip) on the emulation against the oracle (itself pinned to the
reference's network by G11): `python tests/hunt/image_random_shapes.py <seed> <trials>` -- C in 1..3, H and W in 5..40 (odd sizes, sizes
that are not multiples of the 16-pixel patches, strides 2 / 1 / 2 / 1 / 2), d_model 64 / 128 / 256, forward over full contexts and prefixes.  End of round 4: 126 random shapes, no failure."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from dtqn_amd import _binding as B
from dtqn_amd.networks.dtqn import DTQN
from emu import emu_build
from oracle import dtqn_oracle as O

emu = B.load_library(emu_build.build())
rng = np.random.default_rng(int(sys.argv[1]))
ok = ref = 0
for trial in range(int(sys.argv[2])):
    C, Hh, Ww = int(rng.integers(1, 4)), int(rng.integers(5, 41)), int(rng.integers(5, 41))
    D = int(rng.choice([64, 128, 256])); H = int(rng.choice([4, 8])); L = int(rng.choice([2, 5, 9])); A = int(rng.integers(2, 7))
    kw = dict(obs_dim=C * Hh * Ww, image=(C, Hh, Ww), num_actions=A, inner_embed_size=D, num_heads=H, num_layers=1, history_len=L,
              gate=str(rng.choice(["res", "gru"])), pos=str(rng.choice(["learned", "sin"])))
    cfg = O.NetCfg(**kw)
    try:
        m = DTQN((C, Hh, Ww), A, 8, 0, D, H, 1, L, gate=kw["gate"], pos=kw["pos"], _test_lib=emu)
    except NotImplementedError:
        ref += 1
        continue
    m._allow_cpu = True
    params = O.init_params(cfg, seed=int(rng.integers(0, 1000)), perturb=True)
    m.load_state_dict({k: v.clone() for k, v in params.items()})
    try:
        for n in sorted({1, L}):
            obs = torch.from_numpy(rng.integers(0, 256, (2, n, C, Hh, Ww)).astype(np.uint8))
            with torch.no_grad():
                want = O.forward(params, cfg, obs.float()).numpy()
            got = m(obs, torch.zeros(2, n, 1, dtype=torch.long)).numpy()
            err = np.abs(got - want).max()
            assert err <= 1e-4 * max(1.0, np.abs(want).max()), (n, err, np.abs(want).max())
        ok += 1
    except Exception as e:
        print("FAIL", kw, type(e).__name__, str(e)[:300], flush=True)
print("ok", ok, "refused", ref)
