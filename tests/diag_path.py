"""Diagnostic: print HIP-vs-oracle/reference error tables for the golden cases (run on the GPU box)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd"), os.path.join(ROOT, "tests")]
from oracle import ttl_oracle as O
from helpers import load_case, episode_kwargs, max_rel
from test_gpu_path import make_engine, split

PREC = os.environ.get('TTL_PREC', 'bf16')
for name in sys.argv[1:]:
    g, cfg, W, x, lora0, tf = load_case(name)
    kw = episode_kwargs(g)
    eng, flat, names = make_engine(cfg, W, lora0, tf, x.shape[0], PREC)
    snap = flat.clone(); m = torch.zeros_like(flat); v = torch.zeros_like(flat)
    l1, l0 = eng.episode(torch.from_numpy(x).cuda(), snap, m, v, n_updates=kw["n_updates"], objective=kw["objective"],
                         mode=1 if kw["mode"] == "topk" else 0, rho=kw["rho"], margin=kw["margin"], lr=kw["lr"], want_logits0=True)
    torch.cuda.synchronize()
    trace = []
    ob = O.episode(cfg, W, lora0, x, tf, prec=PREC, trace=trace, **kw)
    print(f"== {name}: logits0 vs bf16-oracle {max_rel(l0.cpu().numpy(), ob['logits0']):.2e}  vs ref {max_rel(l0.cpu().numpy(), g['logits0']):.2e}"
          f" | oracle-bf16 vs ref {max_rel(ob['logits0'], g['logits0']):.2e}")
    print(f"   logits1 vs bf16-oracle {max_rel(l1.cpu().numpy(), ob['logits1']):.2e}  vs ref {max_rel(l1.cpu().numpy(), g['logits1']):.2e}")
    lora1 = split(flat, lora0, names); grads = split(eng.grads, lora0, names)
    for k in names:
        gr = g["grad/" + k]; gb = trace[-1]["grads"][k]
        d = np.abs(lora1[k] - g["lora1/" + k])
        print(f"   {k[29:]:45s} grad: vs-bf16 {max_rel(grads[k], gb):.2e} vs-ref {max_rel(grads[k], gr):.2e} (bf16-oracle vs ref {max_rel(gb, gr):.2e})"
              f" | lora1 maxerr {d.max():.2e} frac>1e-4 {float((d > 1e-4).mean()):.3f}")
    eng.close()
