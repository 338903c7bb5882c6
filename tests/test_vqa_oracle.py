"""VQA-LLM, CPU side: the oracle restatement against the golden vectors produced by the REFERENCE's own
LlavaSearchLlamaForCausalLM (tests/golden/vqa_*.npz, oracle/gen_vqa_golden.py), and the host logic of vstar_amd.vqa."""
import ast
import glob
import os

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import vqa_oracle as O
from oracle.gen_vqa_golden import make_inputs
from vstar_amd import vqa
from vstar_amd.config import VQAConfig
from vstar_amd.preprocess import SyntheticTokenizer
from vstar_amd.weights import random_state_dict

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "vqa_*.npz")))


def load_case(path):
    z = np.load(path)
    cfg = VQAConfig.tiny(**ast.literal_eval(str(z["cfg_kw"])))
    n_obj = int(z["n_obj"])
    lens = z["opt_lens"].tolist()
    pix, ids, opts = make_inputs(cfg, int(z["input_seed"]), n_obj, len(z["ids"]), lens)
    assert ids == z["ids"].tolist() and np.concatenate(opts).tolist() == z["opts"].tolist()
    np.testing.assert_allclose([float(pix.double().sum()), float(pix.double().abs().sum())], z["in_checksum"], rtol=1e-12)
    il = None if z["images_long"][0] < 0 else [bool(b) for b in z["images_long"]]
    ol = None if z["objects_long"][0] < 0 else [bool(b) for b in z["objects_long"]]
    return z, cfg, pix, ids, opts, n_obj, il, ol


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_matches_reference_golden(path):
    z, cfg, pix, ids, opts, n_obj, il, ol = load_case(path)
    sd = random_state_dict(cfg, int(z["weight_seed"]), torch.float32)
    img_long, img_short = O.encode_images(sd, cfg, pix[:1])
    obj_long, obj_short = O.encode_images(sd, cfg, pix[1:]) if n_obj else (None, None)
    np.testing.assert_allclose(img_long[0].numpy(), z["image_long"].astype(np.float32), atol=2e-3, rtol=2e-3)
    np.testing.assert_allclose(img_short[0].numpy(), z["image_short"].astype(np.float32), atol=2e-3, rtol=2e-3)
    if n_obj:
        np.testing.assert_allclose(obj_short.numpy(), z["obj_short"].astype(np.float32), atol=2e-3, rtol=2e-3)
    emb = O.splice(sd, ids, img_long, img_short, obj_long, obj_short, il, ol)
    q_logits, past = O.llama_forward(sd, cfg, emb)
    np.testing.assert_allclose(q_logits[-1].numpy(), z["q_logits_last"], atol=2e-4)
    np.testing.assert_allclose(q_logits[::16].numpy(), z["q_logits_rows"].astype(np.float32), atol=5e-3)
    losses, pick = O.multiple_choice(sd, cfg, emb, opts)
    np.testing.assert_allclose(losses.numpy(), z["losses"], atol=1e-4)
    assert pick == int(np.argmin(z["losses"]))
    gen = O.greedy_generate(sd, cfg, emb, len(z["gen"]), eos_id=-1)
    assert gen == z["gen"].tolist()


def test_v1_prompt_and_marker_tokenisation():
    tok = SyntheticTokenizer(320)
    p = vqa.v1_prompt("<image>\nWhat colour is the mug?")
    assert p.startswith("A chat between a curious user") and p.endswith("ASSISTANT:")
    assert vqa.v1_prompt("q", "red").endswith("USER: q ASSISTANT: red</s>")
    prompt = vqa.v1_prompt("<image>\nfocus: mug <object> at [0,0,1,1]; cup <object> at [0,0,1,1].\nWhich?")
    ids = vqa.tokenizer_image_object_token(prompt, tok)
    assert ids[0] == tok.bos_token_id and ids.count(tok.bos_token_id) == 1
    assert ids.count(-200) == 1 and ids.count(-300) == 2 and ids.index(-200) < ids.index(-300)
    # the text pieces are tokenised independently of the markers
    flat = [t for t in ids if t >= 0]
    pieces = prompt.replace("<object>", "<image>").split("<image>")
    assert flat == [tok.bos_token_id] + [t for pc in pieces for t in tok(pc).input_ids[1:]]
    # option ids are the suffix of the full prompt's ids (vstar_bench_eval.py:145-146)
    q = vqa.tokenizer_image_object_token(vqa.v1_prompt("<image>\nWhich?"), tok)
    full = vqa.tokenizer_image_object_token(vqa.v1_prompt("<image>\nWhich?", "the red one"), tok)
    assert full[:len(q)] == q and len(full) > len(q)


def test_get_patch_matches_reference_arithmetic():
    v = vqa.VQA_LLM.__new__(vqa.VQA_LLM)
    # vstar_bench_eval.py:49-70 on hand-computed cases
    assert v.get_patch([100, 50, 30, 20], 640, 480) == [3, 0, 227, 224]
    assert v.get_patch([100, 50, 30.2, 20.7], 640, 480, patch_scale=1.2) == [97, 48, 134, 73]
    assert v.get_patch([600, 440, 60, 60], 640, 480, patch_scale=2.0) == [570, 410, 640, 480]


def test_image_processor_matches_hf_clip_processor():
    from transformers import CLIPImageProcessor
    try:
        hf = CLIPImageProcessor(size={"shortest_edge": 224}, crop_size={"height": 224, "width": 224})
    except Exception as e:  # pragma: no cover
        pytest.skip(f"HF processor unavailable: {e}")
    rng = np.random.default_rng(3)
    proc = vqa._ImageProcessor(224)
    for (w, h) in [(500, 500), (640, 360), (231, 517)]:
        img = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
        ours = proc.preprocess(img)["pixel_values"][0].numpy()
        theirs = np.asarray(hf.preprocess(img, return_tensors="np")["pixel_values"][0])
        np.testing.assert_array_equal(ours, theirs)


def test_expand_ids_long_short_selection():
    from vstar_amd.vqa_engine import VqaEngine
    eng = VqaEngine.__new__(VqaEngine)
    eng.cfg = VQAConfig.tiny()
    P, L = eng.cfg.n_img_tokens, eng.cfg.pcv_latents
    ids = [1, 7, -200, 8, -300, 9, -300]
    # images_long None => long image; objects_long None => short objects (llava_search_arch.py:137,176)
    rows = eng.expand_ids(ids, [0], [1, 2], None, None)
    assert len(rows) == 4 + P + 2 * L
    assert rows[2] == -1 and rows[2 + P - 1] == -P                      # slot 0, long rows 0..P-1
    assert rows[2 + P + 1] == -(1 + (P + L) + P)                          # slot 1, first short row
    rows = eng.expand_ids(ids, [0], [1, 2], [False], [True, False])
    assert len(rows) == 4 + L + P + L
    assert rows[2] == -(1 + P)                                            # slot 0, first short row


def test_vqa_checkpoint_dir_roundtrip(tmp_path):
    """HF-style directory (config.json + safetensors shard with the vision tower inside) -> engine key space."""
    import json
    from safetensors.torch import save_file
    from vstar_amd.weights import load_vqa_checkpoint_dir, vqa_config_from_dir, vqa_state_dict_spec
    cfg = VQAConfig.tiny(projector_type=1)
    sd = random_state_dict(cfg, 3, torch.float16)
    hf = {("model.vision_tower.vision_tower." + k[len("clip."):] if k.startswith("clip.") else k): v for k, v in sd.items()}
    save_file(hf, str(tmp_path / "model.safetensors"))
    json.dump({"hidden_size": cfg.llm_hidden, "num_attention_heads": cfg.llm_heads, "intermediate_size": cfg.llm_mlp,
               "num_hidden_layers": cfg.llm_layers, "vocab_size": cfg.llm_vocab, "rms_norm_eps": 1e-5,
               "mm_projector_type": "mlp2x_gelu", "mm_vision_select_layer": -2}, open(tmp_path / "config.json", "w"))
    got = load_vqa_checkpoint_dir(str(tmp_path))
    assert set(got) == set(vqa_state_dict_spec(cfg))
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    c2 = vqa_config_from_dir(str(tmp_path), clip_hidden=cfg.clip_hidden, clip_heads=cfg.clip_heads, clip_mlp=cfg.clip_mlp,
                             clip_layers=cfg.clip_layers)
    assert (c2.llm_hidden, c2.llm_layers, c2.projector_type, c2.llm_rms_eps) == (cfg.llm_hidden, cfg.llm_layers, 1, 1e-5)
