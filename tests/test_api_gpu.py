"""Model-level loading API (north_star: "keeps ... the SEAL VQA-LLM + VSM model-loading API so it is a drop-in for the repo's
search loop"): the reference's OWN wrapper bodies — visual_search.py::VSM.inference (:174-225) and
vstar_bench_eval.py::VQA_LLM.{free_form,multiple_choices}_inference (:78-165), transcribed below statement by statement with
only the `.cuda()` calls dropped — run against `vstar_amd.api.VSMForCausalLM` / `vstar_amd.api.load_pretrained_model` and must
return what the batched drop-in classes (`vstar_amd.vsm.VSM`, `vstar_amd.vqa.VQA_LLM`, themselves golden-tested) return."""
import numpy as np
import pytest
import torch
from PIL import Image

from test_template_fallback_gpu import CFG, EOS, LOC, QUESTION, TOK, _tid, bigram_state_dict
from vstar_amd import preprocess as pp
from vstar_amd.api import VSMForCausalLM, load_pretrained_model
from vstar_amd.config import VQAConfig
from vstar_amd.synthetic import synthetic_image
from vstar_amd.vsm import VSM
from vstar_amd.weights import random_state_dict

pytestmark = pytest.mark.gpu


# ---- visual_search.py:174-225, with `self.model` = the facade ----
def reference_vsm_inference(model, tokenizer, image, question, mode):
    prompt = pp.build_prompt(question, True)                                   # conv_templates["llava_v1"] + <im_start><image><im_end>
    clip_processor = model.get_model().get_vision_tower().image_processor
    image_clip = clip_processor.preprocess(pp.expand2square(image, pp.background_color()), return_tensors="pt")["pixel_values"][0].unsqueeze(0)
    image_clip = image_clip.bfloat16()
    arr = np.array(image)
    original_size_list = [arr.shape[:2]]
    owl = torch.from_numpy(pp.owl_preprocess(image, 768))[None].bfloat16()    # OwlViTProcessor(images=np.array(image))
    resize_list = [owl.shape[:2]]
    input_ids = torch.tensor(pp.tokenizer_image_token(prompt, tokenizer)).unsqueeze(0)
    output_ids, pred_masks, det_result = model.inference(image_clip, owl, input_ids, resize_list, original_size_list,
                                                         max_new_tokens=100, tokenizer=tokenizer, mode=mode)
    if mode == "segmentation":
        pred_mask = torch.clamp(pred_masks[0], min=0)
        return pred_mask[-1]
    if mode == "vqa":
        return output_ids[0, input_ids.shape[1]:].tolist()
    pred_mask = torch.clamp(pred_masks[0], min=0)
    return det_result["pred_boxes"][0].cpu(), det_result["pred_logits"][0].sigmoid().cpu(), pred_mask[-1]


@pytest.mark.parametrize("chain_kind", ["template", "other_wording"])
def test_vsm_model_level_api_equals_the_drop_in_class(cuda, chain_kind):
    colon, sure, comma, dot, okay, bang = (_tid(p) for p in (":", "Sure", ",", ".", "Okay", "!"))
    chain = [(colon, sure), (sure, comma), (comma, LOC), (LOC, dot), (dot, EOS)] if chain_kind == "template" else \
        [(colon, okay), (okay, LOC), (LOC, bang), (bang, EOS)]
    sd = bigram_state_dict(chain)
    model = VSMForCausalLM.from_pretrained(None, low_cpu_mem_usage=True, vision_tower=None, loc_token_idx=LOC,
                                           torch_dtype=torch.bfloat16, device_map="cuda", is_eval=True, cfg=CFG, state_dict=sd)
    model.get_model().initialize_vision_modules(model.get_model().config)
    assert model.eval() is model and model.config.vision_tower
    vsm = VSM(None, engine=model.engine, tokenizer=TOK, strict_template=True)
    img = synthetic_image(420, 310, 6)
    boxes, scores, heat = reference_vsm_inference(model, TOK, img, QUESTION, "detection")
    b2, s2, h2 = vsm.inference(img, QUESTION, mode="detection")
    assert scores.dtype == torch.bfloat16 and boxes.dtype == torch.bfloat16          # what the reference's tensors are
    assert torch.equal(boxes.float(), b2) and torch.equal(scores, s2) and torch.equal(heat, h2)
    seg = reference_vsm_inference(model, TOK, img, QUESTION, "segmentation")
    assert torch.equal(seg, h2)
    new = reference_vsm_inference(model, TOK, img, pp.CUE_QUESTION.format("kite"), "vqa")
    assert new == vsm.generate_ids(img, pp.CUE_QUESTION.format("kite"), 100)
    model.engine.close()


def test_vsm_model_level_api_no_loc_is_the_references_indexerror(cuda):
    colon, okay, bang = _tid(":"), _tid("Okay"), _tid("!")
    model = VSMForCausalLM.from_pretrained(None, loc_token_idx=LOC, cfg=CFG, state_dict=bigram_state_dict([(colon, okay), (okay, bang), (bang, EOS)]))
    with pytest.raises(IndexError):
        reference_vsm_inference(model, TOK, synthetic_image(300, 300, 1), QUESTION, "detection")
    model.engine.close()


def test_vsm_from_pretrained_reads_a_checkpoint_directory(cuda, tmp_path):
    """save_pretrained-layout directories on disk -> from_pretrained -> same outputs as the engine fed the in-memory state dict."""
    from safetensors.torch import save_file
    sd = random_state_dict(CFG, seed=4, dtype=torch.bfloat16)
    vdir, cdir = tmp_path / "vsm", tmp_path / "clip"
    vdir.mkdir()
    cdir.mkdir()
    save_file({k: v for k, v in sd.items() if not k.startswith("clip.")}, str(vdir / "model.safetensors"))
    save_file({k[len("clip."):]: v for k, v in sd.items() if k.startswith("clip.")}, str(cdir / "model.safetensors"))
    a = VSMForCausalLM.from_pretrained(str(vdir), vision_tower=str(cdir), loc_token_idx=LOC, torch_dtype=torch.bfloat16, cfg=CFG)
    b = VSMForCausalLM.from_pretrained(None, loc_token_idx=LOC, cfg=CFG, state_dict=sd)
    g = torch.Generator().manual_seed(2)
    clip = torch.randn(1, 3, 224, 224, generator=g).bfloat16()
    owl = torch.randn(1, 3, 768, 768, generator=g).bfloat16()
    ids = np.asarray([[1, 5, -200, 9, 11, LOC, 4]], np.int32)
    loc = np.asarray([5 - 1 + CFG.n_img_tokens - 1], np.int32)
    ra, rb = a.engine.score_batch(clip, owl, ids, loc), b.engine.score_batch(clip, owl, ids, loc)
    for k in ("pred_logits", "pred_boxes", "low_res_masks"):
        assert np.array_equal(ra[k], rb[k]), k
    with pytest.raises(FileNotFoundError):
        VSMForCausalLM.from_pretrained(str(vdir), vision_tower="openai/clip-vit-large-patch14", loc_token_idx=LOC, cfg=CFG)
    a.engine.close()
    b.engine.close()


# ---- vstar_bench_eval.py:78-165 with `self.model`, `self.tokenizer`, `self.image_processor` from load_pretrained_model ----
def reference_multiple_choices(tokenizer, model, image_processor, image, question, options, object_crops, images_long, objects_long):
    from vstar_amd import vqa
    qs = "<image>\n" + question
    prompt = vqa.v1_prompt(qs)
    question_input_ids = torch.tensor(vqa.tokenizer_image_object_token(prompt, tokenizer)).unsqueeze(0)
    image_tensor = image_processor.preprocess(image, return_tensors="pt")["pixel_values"][0]
    output_question = model(question_input_ids, use_cache=True, images=image_tensor.unsqueeze(0).half(),
                            object_features=object_crops.half() if object_crops is not None else None,
                            images_long=images_long, objects_long=objects_long)
    question_logits = output_question.logits
    question_past_key_values = output_question.past_key_values
    loss_list = []
    for option in options:
        full_prompt = vqa.v1_prompt(qs, option)
        full_input_ids = torch.tensor(vqa.tokenizer_image_object_token(full_prompt, tokenizer)).unsqueeze(0)
        option_answer_input_ids = full_input_ids[:, question_input_ids.shape[1]:]
        output_option = model(input_ids=option_answer_input_ids, use_cache=True,
                              attention_mask=torch.ones(1, question_logits.shape[1] + option_answer_input_ids.shape[1]),
                              past_key_values=question_past_key_values)
        logits = torch.cat([question_logits[:, -1:], output_option.logits[:, :-1]], 1)
        logits = logits.view(-1, model.config.vocab_size)
        labels = option_answer_input_ids.view(-1)
        loss_list.append(torch.nn.CrossEntropyLoss()(logits.float(), labels).to(torch.float16))
    return loss_list, torch.stack(loss_list).argmin().cpu().item()


def test_load_pretrained_model_serves_the_references_vqa_llm_bodies(cuda):
    from vstar_amd import vqa
    cfg = VQAConfig.tiny()
    sd = random_state_dict(cfg, seed=0, dtype=torch.float16)
    tokenizer, model, image_processor, context_len = load_pretrained_model("seal_vqa_7b", None, "seal_vqa_7bllava", cfg=cfg, state_dict=sd)
    assert context_len == 2048 and model.config.vocab_size == cfg.llm_vocab
    llm = vqa.VQA_LLM(cfg=cfg, engine=model.engine)                                  # the batched drop-in class on the SAME engine
    rng = np.random.default_rng(5)
    image = Image.fromarray(rng.integers(0, 256, (300, 420, 3), dtype=np.uint8))
    crops = torch.stack([llm.get_object_crop(image, [40, 30, 50, 60], patch_scale=1.2),
                         llm.get_object_crop(image, [200, 100, 90, 40], patch_scale=1.2)], 0)
    question = "Additional visual information to focus on: mug <object> at location [0.1,0.1,0.2,0.3]; cup <object> at " \
               "location [0.5,0.3,0.7,0.5].\nWhat is the colour of the mug?"
    options = ["The colour of the mug is red.", "The colour of the mug is blue.", "green", "The mug is yellow and white."]
    losses, pick = reference_multiple_choices(tokenizer, model, image_processor, image, question, options, crops, [False], [True, True])
    want = llm.option_losses(image, question, options, crops, images_long=[False], objects_long=[True, True])
    assert [float(x) for x in losses] == [float(x) for x in want]                    # same engine calls: bit-identical
    assert pick == llm.multiple_choices_inference(image, question, options, crops, images_long=[False], objects_long=[True, True])
    # free-form: model.generate(...) echoes the prompt ids and appends the greedy continuation (vstar_bench_eval.py:91-107)
    prompt = vqa.v1_prompt("<image>\n" + "What is in the picture?")
    input_ids = torch.tensor(vqa.tokenizer_image_object_token(prompt, tokenizer)).unsqueeze(0)
    image_tensor = image_processor.preprocess(image, return_tensors="pt")["pixel_values"][0]
    out = model.generate(input_ids, images=image_tensor.unsqueeze(0).half(), object_features=None, images_long=None,
                         objects_long=None, do_sample=False, num_beams=1, temperature=0, top_p=None, max_new_tokens=6,
                         use_cache=True, stopping_criteria=[object()])
    assert (input_ids != out[:, :input_ids.shape[1]]).sum().item() == 0
    llm.free_form_inference(image, "What is in the picture?", max_new_tokens=6)
    assert out[0, input_ids.shape[1]:].tolist() == list(llm.generated_ids[0])
    with pytest.raises(NotImplementedError):
        load_pretrained_model("x", None, "x", load_8bit=True)
