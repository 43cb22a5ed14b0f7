"""Random-configuration hunt of the batched actor entry point (dtqn_actor_forward_batch, ragged prefixes) on the emulation against the oracle:
`python tests/hunt/actor_random_configs.py <seed> <trials>`.  End of round 4: 110 accepted configurations, no failure."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
from dtqn_amd import _binding as B
from emu import emu_build
from test_vector_parity import check_batched_actor_vs_oracle
emu = B.load_library(emu_build.build())
rng = np.random.default_rng(int(sys.argv[1]))
ok = ref = 0
for trial in range(int(sys.argv[2])):
    D = int(rng.choice([16, 32, 40, 48, 64, 80, 96, 128]))
    H = int(rng.choice([h for h in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16) if D % h == 0]))
    L = int(rng.choice([3, 8, 16, 17, 30, 50, 64, 70]))
    kw = dict(obs_dim=int(rng.integers(1, 5)), num_actions=int(rng.integers(2, 6)), inner_embed_size=D, num_heads=H, num_layers=int(rng.integers(1, 3)),
              history_len=L, gate=str(rng.choice(["res", "gru"])), identity=bool(rng.integers(0, 2)), pos=str(rng.choice(["learned", "sin", "none"])),
              action_dim=int(rng.choice([0, 0, 4])))
    if rng.integers(0, 2):
        kw.update(discrete=True, vocab_sizes=int(rng.integers(3, 12)))
    try:
        check_batched_actor_vs_oracle(emu, kw, (int(rng.integers(1, 5)),), rounds=1)
        ok += 1
    except NotImplementedError:
        ref += 1
    except Exception as e:
        print("FAIL", kw, type(e).__name__, str(e)[:300], flush=True)
print("ok", ok, "refused", ref)
