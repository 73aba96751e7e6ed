"""CPU diagnostic (oracle-side): how much of the adapted-logit deviation is the sign-like first AdamW step?

The reference fixture gradients are perturbed by Gaussian noise of a given size (relative to each tensor max), stepped with the
closed-form first AdamW update, and the adapted 1-view logits of an otherwise EXACT fp32 forward are compared with the fixture.

    python tools/adamw_sign_sensitivity.py b16_n64_k200_qkvo b16_n64_k200_ent0
"""
import sys, os, numpy as np
ROOT = os.getcwd(); sys.path[:0] = [ROOT, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd"), os.path.join(ROOT, "tests")]
from oracle import ttl_oracle as O
from helpers import load_case, episode_kwargs, max_rel, adamw_first_step
for name in sys.argv[1:]:
    g, cfg, W, x, lora0, tf = load_case(name)
    kw = episode_kwargs(g)
    names = O.trainable_names(cfg)
    ref1 = {k: g["lora1/" + k] for k in names}
    lora_ref = dict(lora0); lora_ref.update(ref1)
    net = O.VitOracle(cfg, W, lora_ref, "fp32")
    z_ref = net.logits(net.forward(x[:1]), tf)
    print(name, "oracle fp32 adapted logits from the reference's weights vs the fixture:", max_rel(z_ref, g["logits1"]))
    rng = np.random.default_rng(0)
    for rel in (1e-3, 3e-3):
        out = []
        for trial in range(3):
            lp = dict(lora0)
            for k in names:
                gr = g["grad/" + k]
                if np.abs(gr).max() == 0:
                    lp[k] = ref1[k]; continue
                noisy = gr + rng.standard_normal(gr.shape).astype(np.float32) * rel * np.abs(gr).max() / 3   # max error ~ rel of the tensor max
                lp[k] = adamw_first_step(lora0[k], noisy, kw["lr"]).astype(np.float32)
            n2 = O.VitOracle(cfg, W, lp, "fp32")
            z = n2.logits(n2.forward(x[:1]), tf)
            out.append(max_rel(z, g["logits1"]))
        print(f"   gradients perturbed by ~{rel:g} of each tensor's max (exact fp32 forward otherwise): adapted logits off by", ["%.2e" % v for v in out])
