"""Per-fixture parity table of the two product builds and the test-only strict (fp32) build on the GPU (logits, adapted logits, worst / median LoRA-gradient deviation vs the
reference-generated fixtures): python tools/parity_per_fixture.py  -> profiles/rNN_parity_per_fixture.txt"""
import sys, os, numpy as np, torch
ROOT = os.getcwd(); sys.path[:0] = [ROOT, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd"), os.path.join(ROOT, "tests")]
from helpers import load_case, episode_kwargs, max_rel
from ttl_amd.engine import TTLEngine
from ttl_amd.config import trainable_names
for name in ["b16_n8_k10", "b16_n64_k200_ent0", "b16_n64_k200_ent1", "b16_n64_k1000_ent0", "b16_n64_k1000_ent1", "b16_n8_k10_qkvo", "b16_n64_k200_qkvo", "l14_n4_k10", "l14_n64_k200",
             "b32_n8_k10", "b16_n8_k10_outliers", "b16_n64_k200_outliers", "b16_n64_k200_outliers_ent1"]:
    g, cfg, W, x, lora0, tf = load_case(name); kw = episode_kwargs(g); names = trainable_names(cfg)
    for prec in ("strict", "fp16", "bf16"):
        eng = TTLEngine(cfg, x.shape[0], tf.shape[0], "cuda:0", prec); eng.load_weights(W)
        eng.set_text_features(torch.from_numpy(tf), float(np.exp(W["logit_scale"])))
        flat = torch.cat([torch.from_numpy(lora0[k]).reshape(-1) for k in names]).cuda().contiguous(); eng.bind_lora(flat)
        l1, l0 = eng.episode(torch.from_numpy(x).cuda(), flat.clone(), torch.zeros_like(flat), torch.zeros_like(flat), n_updates=kw["n_updates"],
                             objective=kw["objective"], mode=1 if kw["mode"] == "topk" else 0, rho=kw["rho"], margin=kw["margin"], lr=kw["lr"], want_logits0=True)
        torch.cuda.synchronize()
        gr, off, errs = eng.grads.cpu().numpy(), 0, {}
        for k in names:
            n = lora0[k].size; ref = g["grad/" + k]
            if np.abs(ref).max() > 0: errs[k] = max_rel(gr[off:off + n].reshape(ref.shape), ref)
            off += n
        w = max(errs, key=errs.get)
        print(f"{name:22s} {prec:6s}: logits {max_rel(l0.cpu().numpy(), g['logits0']):.2e} adapted {max_rel(l1.cpu().numpy(), g['logits1']):.2e} grad max {errs[w]:.2e} ({w.split('layers.')[1].replace('.default.weight','')}) median {np.median(list(errs.values())):.2e}", flush=True)
        eng.close()
