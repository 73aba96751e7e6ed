"""-m gpu: a race screen for the counted-wait pipelines (gemm_big.hip's LDS-DMA ring behind hand-counted vmcnt + raw s_barrier,
attention.hip's prefetch): thousands of episodes with three of them in flight per GPU, every result compared BITWISE with what
one engine computes alone.  A read placed ahead of the wait that retires its DMA "passes reference checks whenever the DMA happens
to land first" (the CDNA guide) — 18 episodes cannot show a 1-in-10^4 event; this is the screen that can.  Reference path being
pinned: HF modeling_clip.py:259-277,333,376 / peft LoRA through clip/custom_clip.py:62-71 — all of it, end to end."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from helpers import load_case
from test_gpu_path import make_engine

pytestmark = pytest.mark.gpu


def _screen(cfg, W, lora0, tf, batches, n_episodes, precision, lora_every=50, n_streams=3):
    """n_episodes through EpisodePipeline(n_streams, use_graph=True) rotating over `batches` (pre-staged, replayed in place as
    bench.py does): adapted logits of EVERY episode and the adapted LoRA buffer of every `lora_every`-th equal the single-engine
    result bit for bit.  -> number of LoRA buffers compared."""
    from ttl_amd.driver import EpisodePipeline
    eng, flat, names = make_engine(cfg, W, lora0, tf, batches[0].shape[0], precision=precision)
    eng.set_concurrency(n_streams)      # the tile choices of a context that shares the GPU (ttl_ctx_set_concurrency): the same kernels as the pipeline's
    snap, m, v = flat.clone(), torch.zeros_like(flat), torch.zeros_like(flat)
    ref_out, ref_lora = [], []
    for xb in batches:
        ref_out.append(eng.episode(xb, snap, m, v, n_updates=1).clone())
        ref_lora.append(flat.clone())
    torch.cuda.synchronize()
    eng.close()
    assert all(torch.isfinite(o).all() for o in ref_out)
    assert not torch.equal(ref_out[0], ref_out[1]) and not torch.equal(ref_lora[0], ref_lora[1]) and not torch.equal(ref_lora[0], snap)
    pipe = EpisodePipeline(cfg, W, names, lora0, torch.from_numpy(tf), float(np.exp(W["logit_scale"])), "cuda:0",
                           n_streams=n_streams, max_views=batches[0].shape[0], precision=precision, use_graph=True)
    rng = np.random.default_rng(0)
    order = rng.integers(0, len(batches), n_episodes)          # irregular: neighbours in flight differ from episode to episode
    outs, loras = [], []
    for i, j in enumerate(order):
        slot = pipe.slots[pipe._next]
        outs.append(pipe.submit(batches[j], persistent_input=True, n_updates=1))
        if i % lora_every == 0:
            with torch.cuda.stream(slot["stream"]):
                loras.append((int(j), slot["flat"].clone()))
    pipe.synchronize()
    torch.cuda.synchronize()
    bad = [i for i, (j, o) in enumerate(zip(order, outs)) if not torch.equal(o, ref_out[j])]
    assert not bad, (precision, f"{len(bad)} of {n_episodes} episodes differ from the single-engine result", bad[:10])
    badl = [i for i, (j, f) in enumerate(loras) if not torch.equal(f, ref_lora[j])]
    assert not badl, (precision, "adapted LoRA buffers differ", badl[:10])
    assert all(len(sl["graphs_in_place"]) == len(batches) for sl in pipe.slots)       # really graph replays, one graph per input buffer
    pipe.close()
    return len(loras)


def _batches(x, n=4):
    x0 = torch.from_numpy(x).cuda()
    return [x0] + [(torch.roll(x0, 3 * k, dims=0) * (1.0 - 0.04 * k)).contiguous() for k in range(1, n)]


def test_1500_episodes_three_in_flight_bitwise_bf16():
    g, cfg, W, x, lora0, tf = load_case("b16_n64_k200_ent0")
    assert _screen(cfg, W, lora0, tf, _batches(x), 1500, "bf16") == 30


def test_300_episodes_three_in_flight_bitwise_fp16():
    g, cfg, W, x, lora0, tf = load_case("b16_n64_k200_ent0")
    assert _screen(cfg, W, lora0, tf, _batches(x), 300, "fp16", lora_every=25) == 12


def test_100_episodes_three_in_flight_bitwise_qkvo():
    """adapters on q, k, v and out_proj: the out_proj K-extension, dK in the first trained layer, 8 weight-gradient products"""
    from ttl_amd import synth
    g, cfg, W, x, lora0, tf = load_case("b16_n64_k200_ent0")
    cfg = cfg.replace(lora_targets=("q_proj", "k_proj", "v_proj", "out_proj"))
    lora0 = synth.lora_init(cfg, 0)
    assert _screen(cfg, W, lora0, tf, _batches(x), 100, "bf16", lora_every=10) == 10


def test_100_episodes_three_in_flight_bitwise_vit_l14():
    """ViT-L/14: T = 257 takes attn_fwd_w_kernel (five key tiles do not fit the persistent kernel), D = 1024 / F = 4096 GEMM shapes"""
    from ttl_amd import synth
    from ttl_amd.config import get_config
    cfg = get_config("ViT-L/14")
    W = synth.vision_weights(cfg, 0)
    lora0 = synth.lora_init(cfg, 0)
    tf = synth.text_features(200, cfg.embed)
    x = synth.views(cfg, 32, 5)
    assert _screen(cfg, W, lora0, tf, _batches(x, 3), 100, "bf16", lora_every=10) == 10
