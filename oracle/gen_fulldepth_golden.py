"""One-off FULL-DEPTH, REAL-WIDTH golden: the reference's own model_forward(inference=True) at the
headline-bench geometry — CLIP-ViT-L/14@336 (23 blocks), LLaMA-7B (32 layers, S=640), OWL-ViT-B/16@768 (12 layers), SAM head —
on crops of the EXACT batch bench.py scores (vstar_amd.synthetic.bench_inputs, B=32) with the bench's weights
(random_state_dict(seed=0, bf16, share_layers=True), upcast to fp32 for the reference's fp32 run).

TEST INFRASTRUCTURE.  Run once in the build container (needs /root/reference, ~45 GB RAM, ~6 min on 8 cores):
    python -m oracle.gen_fulldepth_golden
    python -m oracle.gen_fulldepth_golden --image-size 224 --crops 0,11,21,31      # the geometry the reference really runs
Writes tests/golden/full7b_{336,224}.npz (outputs only: fp32 reference + the reference in bf16 as the noise yardstick).
Round 3: eight crops of the 336^2 bench batch and four of the 224^2 / S = 320 batch
(CLIP-L/14@224, 256 image tokens: VisualSearch/model/VSM.py:230-234,466-473 hard-code that geometry).
The GPU test (tests/test_fulldepth_gpu.py) scores the whole 32-crop batch and compares the recorded crops: this is what pins
error growth over 32+23+12 layers and arg-max / top-k stability at the bench shape.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import ref_shim  # noqa: E402
from vstar_amd.config import VSMConfig  # noqa: E402
from vstar_amd.synthetic import bench_inputs  # noqa: E402
from vstar_amd.weights import random_state_dict, template_chain, trained_like_state_dict  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
CROPS = (0, 4, 9, 13, 17, 22, 26, 31)          # indices into the bench batch (0 and 17 were the round-2 pair)
B, T = 32, 64


def build_reference(cfg, loc_id, sd):
    """7B random-init is slow and pointless (every used parameter is overwritten): build with the initialisers patched out."""
    import torch.nn as nn
    saved = {c: c.reset_parameters for c in (nn.Linear, nn.Embedding, nn.Conv2d, nn.LayerNorm)}
    for c in saved:
        c.reset_parameters = lambda self: None
    inits = {n: getattr(nn.init, n) for n in ("normal_", "trunc_normal_", "uniform_", "kaiming_uniform_", "xavier_uniform_")}
    for n in inits:
        setattr(nn.init, n, lambda t, *a, **k: t)
    try:
        _, model = ref_shim.load_reference(cfg, loc_id)
    finally:
        for c, f in saved.items():
            c.reset_parameters = f
        for n, f in inits.items():
            setattr(nn.init, n, f)
    with torch.no_grad():                      # parameters the engine never reads (IoU head, masks 1-3, ...) -> finite values
        for p in model.parameters():
            if not torch.isfinite(p).all() or p.abs().max() > 1e3:
                p.zero_()
    missing = ref_shim.load_state(model, sd)
    assert not missing, missing[:5]
    return model


def run(model, cfg, loc_id, clip, owl, ids, verify, dtype, crops=CROPS):
    P = cfg.n_img_tokens
    rec = {}
    for ci in crops:
        # fresh CLIP tower per call (transformers 5.x harness artefact, see gen_golden.py; verified to reproduce the tiny goldens)
        from transformers import CLIPVisionModel
        vt = model.get_model().get_vision_tower()
        ccfg = vt.vision_tower.config
        clip_sd = {k: v for k, v in vt.vision_tower.state_dict().items()}
        vt.vision_tower = CLIPVisionModel(ccfg).eval().to(dtype)
        vt.vision_tower.load_state_dict(clip_sd)
        taps = {}
        hooks = [
            model.model.text_hidden_fcs_det[0].register_forward_hook(
                lambda m, i, o: taps.update(hidden=i[0].detach().float().clone(), det=o.detach().float().clone())),
            model.model.text_hidden_fcs_seg[0].register_forward_hook(lambda m, i, o: taps.update(seg=o.detach().float().clone())),
            model.model.mask_decoder.output_hypernetworks_mlps[0].register_forward_hook(
                lambda m, i, o: taps.update(hyper=o.detach().float().clone())),
            model.model.mask_decoder.output_upscaling.register_forward_hook(lambda m, i, o: taps.update(up=o.detach().float().clone())),
            model.lm_head.register_forward_hook(lambda m, i, o: taps.update(logits=o.detach()[0, torch.as_tensor(verify[ci]).long()].float().clone())),
        ]
        t0 = time.time()
        out = ref_shim.reference_forward(model, clip[ci:ci + 1].to(dtype), owl[ci:ci + 1].to(dtype), torch.from_numpy(ids[ci:ci + 1].astype(np.int64)))
        for h in hooks:
            h.remove()
        pos = int(np.where(ids[ci] == loc_id)[0][-1]) - 1 + (P - 1)
        top2 = torch.topk(taps["logits"], 2, dim=-1)
        r = {"pred_logits": out["pred_logits"][0, :, 0].float().numpy(), "pred_boxes": out["pred_boxes"][0].float().numpy(),
             "low_res_masks": out["pred_masks"][0][0].float().numpy(), "llm_hidden_loc": taps["hidden"][0, pos].numpy(),
             "embed_det": taps["det"][0, pos].numpy(), "embed_seg": taps["seg"][0, pos].numpy(),
             "sam_hyper": taps["hyper"].reshape(-1).numpy(), "sam_upscaled_mean": taps["up"][0].double().mean(dim=(1, 2)).float().numpy(),
             "tf_argmax": top2.indices[:, 0].numpy().astype(np.int32), "tf_top2_gap": (top2.values[:, 0] - top2.values[:, 1]).numpy(),
             "tf_logit_spread": (taps["logits"].max(-1).values - taps["logits"].min(-1).values).numpy()}
        print(f"  crop {ci} [{dtype}] {time.time() - t0:.1f}s  logits max {r['pred_logits'].max():.4f}", flush=True)
        for k, v in r.items():
            rec.setdefault(k, []).append(v)
    return {k: np.stack(v) for k, v in rec.items()}


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--image-size", type=int, default=336, choices=(224, 336))
    ap.add_argument("--crops", type=str, default=None)
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--weights", choices=("random", "trained_like"), default="random",
                    help="trained_like (round 4, VERDICT r3 missing #2): vstar_amd.weights.trained_like_state_dict — outlier residual "
                         "channels, a massive-activation BOS, spread norm gains, peaked attention, and a greedy decode that emits the "
                         "answer template for SyntheticTokenizer prompts; written to full7b_tl_{336,224}.npz")
    ap.add_argument("--out", type=str, default=None, help="output file name under tests/golden/ (round 5: the 32-crop noise study "
                    "full7b_tl_336_x32.npz = every crop of the bench batch, recorded with --crops all --mask-f16)")
    ap.add_argument("--input-rank", type=int, default=0, help="round 6: bench_inputs(rank=...) = the seed of the synthetic crop batch; rank 1 is the "
                    "second 32-crop fixture full7b_tl_336_x32_r1.npz (VERDICT r5 item 9: settle the sign of the mask-offset mean over 64 crops)")
    ap.add_argument("--mask-f16", action="store_true", help="store the 192 x 192 masks as float16 (5e-4 relative, two orders below "
                    "the bf16 noise they are compared with): halves the file")
    a = ap.parse_args()
    tl = a.weights == "trained_like"
    if a.crops == "all":
        a.crops = ",".join(str(i) for i in range(B))
    crops = tuple(int(c) for c in a.crops.split(",")) if a.crops else ((0, 9, 17, 31) if tl else CROPS)     # (round 4: the 336 trained-like file was recorded with --crops 0,4,9,13,17,22,26,31)
    out_path = os.path.join(GOLDEN, a.out or f"full7b_{'tl_' if tl else ''}{a.image_size}.npz")
    assert ref_shim.available(), "reference tree not found"
    torch.set_num_threads(a.threads)
    cfg = VSMConfig.seal_7b(a.image_size, max_batch=B, max_text_len=T + 1)
    loc_id = cfg.llm_vocab - 1
    clip, owl, ids, loc, verify = bench_inputs(cfg, B, T, rank=a.input_rank)
    t0 = time.time()
    if tl:
        from vstar_amd.preprocess import SyntheticTokenizer
        sd16 = trained_like_state_dict(cfg, seed=0, dtype=torch.bfloat16, share_layers=True,
                                       chain=template_chain(SyntheticTokenizer(cfg.llm_vocab)))
    else:
        sd16 = random_state_dict(cfg, seed=0, dtype=torch.bfloat16, share_layers=True)
    sd = {k: v.float() for k, v in sd16.items()}
    del sd16
    model = build_reference(cfg, loc_id, sd)
    del sd
    print(f"reference built + loaded in {time.time() - t0:.0f}s", flush=True)
    f32 = run(model, cfg, loc_id, clip, owl, ids, verify, torch.float32, crops)
    model = model.bfloat16()
    b16 = run(model, cfg, loc_id, clip, owl, ids, verify, torch.bfloat16, crops)
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64)))  # noqa: E731
    print("reference-bf16 vs reference-fp32 rel-L2:", {k: "%.2e" % rel(b16[k], f32[k]) for k in f32 if k.startswith(("pred", "low", "llm", "embed", "sam"))})
    if a.mask_f16:
        f32["low_res_masks"] = f32["low_res_masks"].astype(np.float16)
        b16["low_res_masks"] = b16["low_res_masks"].astype(np.float16)
    np.savez_compressed(out_path, crops=np.asarray(crops), batch=B, text_tokens=T, weight_seed=0, image_size=a.image_size, input_rank=a.input_rank,
                        weights=a.weights,
                        **{k: v for k, v in f32.items()}, **{"bf16_" + k: v for k, v in b16.items()})
    print("->", out_path, os.path.getsize(out_path) // 1024, "KiB")


if __name__ == "__main__":
    main()
