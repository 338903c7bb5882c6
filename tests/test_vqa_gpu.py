"""VQA-LLM engine on the MI355X, through the C-ABI of include/vstar_vqa.h.

(1) against the golden vectors produced by the REFERENCE's own LlavaSearchLlamaForCausalLM (tests/golden/vqa_*.npz):
    long/short image and object features, the question's last-row logits, every option continuation's logits scored
    against the forked question cache, the CrossEntropy option losses and the argmin, and the greedy decode.
(2) internal consistency that holds at any size: forked-prefix option scoring == re-prefilling question+option;
    KV-cached decode == cache-less re-prefill; batched decode == one sequence at a time; the weight-streaming GEMM ==
    the MFMA tile GEMM == torch on the same fp16 operands.

Tolerance: the reference computes in fp16 end to end and so does the engine (fp16 storage, fp32 accumulate, HF's rounding
points).  Gate: rel_L2(engine, golden fp32) <= max(5e-3, 3 x rel_L2(fp16 reference algorithm on torch-CPU, golden)); the
noise floor is measured in the test with the oracle on an fp16 state dict and printed.  Tokens are compared while the
reference's own top-2 logit margin exceeds 0.05 (below that fp16 rounding may legitimately flip the arg-max).
"""
import ctypes
import glob
import os

import numpy as np
import pytest
import torch

from oracle import vqa_oracle as O
from tests.test_vqa_oracle import load_case
from vstar_amd import _lib
from vstar_amd.config import VQAConfig
from vstar_amd.vqa_engine import Seq, VqaEngine
from vstar_amd.weights import random_state_dict

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "vqa_*.npz")))


def rel_l2(got, ref):
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    ref = np.asarray(ref, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))


_ENGINES = {}


def engine_for(cfg, wseed):
    key = (cfg.projector_type, wseed)
    if key not in _ENGINES:
        eng = VqaEngine(cfg, 0)
        eng.load_state_dict(random_state_dict(cfg, seed=wseed, dtype=torch.float16))
        _ENGINES[key] = eng
    return _ENGINES[key]


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_engine_matches_reference_golden(cuda, path):
    z, cfg, pix, ids, opts, n_obj, il, ol = load_case(path)
    wseed = int(z["weight_seed"])
    eng = engine_for(cfg, wseed)
    # ---- fp16 noise floor of the reference algorithm itself (torch CPU, fp16 state dict) ----
    sd16 = random_state_dict(cfg, wseed, torch.float16)
    n_long, n_short = O.encode_images(sd16, cfg, pix.half())
    emb16 = O.splice(sd16, ids, n_long[:1], n_short[:1], n_long[1:], n_short[1:], il, ol)
    q16, past16 = O.llama_forward(sd16, cfg, emb16)
    noise = {"image_long": rel_l2(n_long[0].float(), z["image_long"]), "image_short": rel_l2(n_short[0].float(), z["image_short"]),
             "q_logits_last": rel_l2(q16[-1].float(), z["q_logits_last"])}
    # ---- engine ----
    eng.encode_images(pix, 0)
    got = {}
    long0, short0 = eng.features(0)
    got["image_long"], got["image_short"] = long0, short0
    report = {}
    for k in ("image_long", "image_short"):
        report[k] = rel_l2(got[k], z[k])
    for j in range(n_obj):
        lj, sj = eng.features(1 + j)
        report[f"obj{j}_long"] = rel_l2(lj, z["obj_long"][j])
        report[f"obj{j}_short"] = rel_l2(sj, z["obj_short"][j])
        noise[f"obj{j}_long"], noise[f"obj{j}_short"] = noise["image_long"], noise["image_short"]
    rows = eng.expand_ids(ids, [0], list(range(1, 1 + n_obj)), il, ol)
    S = len(rows)
    want = [(0, r) for r in range(0, S, 16)] + [(0, -1)]
    q_logits, q_arg = eng.forward([Seq(rows, kv_slot=0)], want)
    report["q_logits_last"] = rel_l2(q_logits[-1], z["q_logits_last"])
    report["q_logits_rows"] = rel_l2(q_logits[:-1], z["q_logits_rows"])
    noise["q_logits_rows"] = rel_l2(q16[::16].float(), z["q_logits_rows"])
    # ---- options: forked prefix (slot 0) -> slots 1.. ; logits of every option row ----
    seqs = [Seq(o, kv_slot=1 + j, past_len=S, prefix_slot=0) for j, o in enumerate(opts)]
    o_logits, _ = eng.forward(seqs, [(j, t) for j, o in enumerate(opts) for t in range(len(o))])
    report["opt_logits"] = rel_l2(o_logits, z["opt_logits"])
    o16 = torch.cat([O.llama_forward(sd16, cfg, sd16["model.embed_tokens.weight"][torch.tensor(o)], past16)[0] for o in opts], 0)
    noise["opt_logits"] = rel_l2(o16.float(), z["opt_logits"])
    losses, k = [], 0
    for o in opts:
        lg = torch.cat([torch.from_numpy(q_logits[-1:]), torch.from_numpy(o_logits[k:k + len(o) - 1])], 0)
        k += len(o)
        losses.append(float(torch.nn.functional.cross_entropy(lg.float(), torch.tensor(o))))
    print(os.path.basename(path), {k: "%.2e (noise %.2e)" % (v, noise[k]) for k, v in report.items()})
    print("  losses", [round(x, 4) for x in losses], "reference", z["losses"].round(4).tolist())
    for k, v in report.items():
        assert v <= max(5e-3, 3 * noise[k]), (k, v, noise[k])
    np.testing.assert_allclose(losses, z["losses"], atol=0.02)
    ref_sorted = np.sort(z["losses"])
    if ref_sorted[1] - ref_sorted[0] > 0.05:
        assert int(np.argmin(losses)) == int(np.argmin(z["losses"]))
    # ---- greedy decode with the KV cache ----
    cur, pos, gen = int(q_arg[-1]), S, []
    for step in range(len(z["gen"])):
        gen.append(cur)
        _, nxt = eng.forward([Seq([cur], kv_slot=0, past_len=pos)], [(0, 0)], logits=False)
        cur, pos = int(nxt[0]), pos + 1
    n_cmp = 0
    for g, r, m in zip(gen, z["gen"], z["gen_margin"]):
        if m < 0.05:
            break
        assert g == int(r)
        n_cmp += 1
    print("  greedy tokens compared:", n_cmp, "of", len(gen), gen, z["gen"].tolist())
    assert n_cmp >= 1


def test_fork_equals_reprefill_and_cache_equals_no_cache(cuda):
    cfg = VQAConfig.tiny()
    eng = engine_for(cfg, 0)
    g = torch.Generator().manual_seed(11)
    eng.encode_images(torch.randn(1, 3, 224, 224, generator=g), 0)
    q = [1] + torch.randint(3, 300, (9,), generator=g).tolist()
    q[2] = -200
    rows = eng.expand_ids(q, [0], [], None, None)          # 256 long rows + 9 text rows
    S = len(rows)
    opt = torch.randint(3, 300, (7,), generator=g).tolist()
    # (a) question prefill into slot 0, option forked into slot 1
    eng.forward([Seq(rows, kv_slot=0)], [])
    a_logits, _ = eng.forward([Seq(opt, kv_slot=1, past_len=S, prefix_slot=0)], [(0, t) for t in range(len(opt))])
    # (b) one prefill of question + option in slot 2 (flash-attention path)
    b_logits, _ = eng.forward([Seq(rows + opt, kv_slot=2)], [(0, S + t) for t in range(len(opt))])
    err = rel_l2(a_logits, b_logits)
    print("fork vs re-prefill rel_l2 %.2e" % err)
    assert err < 5e-3
    # (c) token-by-token with the cache (slot 3 continues a copy of the question) == (b)
    eng.forward([Seq(rows, kv_slot=3)], [])
    c_rows = []
    for t, tok in enumerate(opt):
        lg, _ = eng.forward([Seq([tok], kv_slot=3, past_len=S + t)], [(0, 0)])
        c_rows.append(lg[0])
    err = rel_l2(np.stack(c_rows), b_logits)
    print("decode-with-cache vs prefill rel_l2 %.2e" % err)
    assert err < 5e-3


def test_batched_ragged_decode_equals_single(cuda):
    cfg = VQAConfig.tiny()
    eng = engine_for(cfg, 0)
    g = torch.Generator().manual_seed(12)
    eng.encode_images(torch.randn(2, 3, 224, 224, generator=g), 0)
    prompts = []
    for i, n in enumerate((6, 11)):
        ids = [1] + torch.randint(3, 300, (n,), generator=g).tolist()
        ids[1] = -200
        prompts.append(eng.expand_ids(ids, [i], [], [i == 0], None))   # sample 0 long (256 rows), sample 1 short (32 rows)
    singles = []
    for i, rows in enumerate(prompts):
        lg, _ = eng.forward([Seq(rows, kv_slot=4 + i)], [(0, -1)])
        lg2, _ = eng.forward([Seq([5 + i], kv_slot=4 + i, past_len=len(rows))], [(0, 0)])
        singles.append((lg[0], lg2[0]))
    lg, _ = eng.forward([Seq(r, kv_slot=i) for i, r in enumerate(prompts)], [(0, -1), (1, -1)])       # ragged prefill batch
    lg2, _ = eng.forward([Seq([5 + i], kv_slot=i, past_len=len(r)) for i, r in enumerate(prompts)], [(0, 0), (1, 0)])
    for i in range(2):
        assert rel_l2(lg[i], singles[i][0]) < 2e-3, rel_l2(lg[i], singles[i][0])
        assert rel_l2(lg2[i], singles[i][1]) < 2e-3, rel_l2(lg2[i], singles[i][1])


@pytest.mark.parametrize("M,N,K,epi", [(1, 4096, 4096, 0), (7, 1000, 1024, 0), (16, 512, 11008, 0), (33, 768, 256, 2),
                                         (64, 22016, 4096, 4), (48, 320, 256, 0), (5, 4096, 4096, 0)])
def test_weight_streaming_gemm(cuda, lib, M, N, K, epi):
    """gemm_skinny_kernel vs torch fp32 on the same fp16 operands, with bias/residual; and vs the MFMA tile kernel."""
    g = torch.Generator().manual_seed(M * 1000 + N)
    Npad = (N + 255) // 256 * 256
    n_out = N // 2 if epi == 4 else N
    A = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    W = torch.zeros(Npad, K, dtype=torch.float16)
    W[:N] = (torch.randn(N, K, generator=g) / K ** 0.5).half()
    W = W.cuda()
    use_bias = epi != 4
    bias = (torch.randn(Npad, generator=g) * 0.1).half().cuda() if use_bias else None
    res = (torch.randn(M, n_out, generator=g) * 0.5).half().cuda() if M % 2 else None
    outs = []
    for kernel in (1, 2):
        C = torch.zeros(M, n_out, dtype=torch.float16, device="cuda")
        P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        rc = lib.vstar_vqa_op_gemm(P(A), P(W), P(bias), P(res), P(C), M, N, K, epi, kernel, None, 0.0)
        assert rc == 0, lib.vstar_vqa_last_error(None)
        outs.append(C.float().cpu())
    ref = A.float().cpu() @ W[:N].float().cpu().T
    if use_bias:
        ref = ref + bias[:N].float().cpu()
    if epi == 2:
        ref = torch.nn.functional.gelu(ref.half().float())
    if epi == 4:   # packed rows: blocks of 16 gate rows followed by 16 up rows
        r = ref.view(M, N // 32, 2, 16)
        ref = (torch.nn.functional.silu(r[:, :, 0].half().float()).half().float() * r[:, :, 1].half().float()).reshape(M, n_out)
    if res is not None:
        ref = ref.half().float() + res.float().cpu()
    scale = float(ref.abs().max())
    for o in outs:
        assert float((o - ref).abs().max()) <= 2e-3 * scale + 1e-3, float((o - ref).abs().max())
    assert float((outs[0] - outs[1]).abs().max()) <= 1e-3 * scale + 1e-3


@pytest.mark.parametrize("M,N,K,epi,norm,use_res", [
    (1, 4096, 4096, 0, False, True),       # o_proj + residual, batch 1
    (1, 12288, 4096, 0, True, False),      # qkv with the fused RMSNorm
    (1, 22016, 4096, 4, True, False),      # gate|up, SiLU(gate)*up, fused RMSNorm
    (1, 4096, 11008, 0, False, True),      # down_proj: 172 double steps = 21.5 per wave
    (4, 1024, 512, 0, False, False),       # one double step per wave
    (16, 1000, 1088, 0, True, True),       # 17 double steps: some waves 3, some 2; N tail
    (7, 512, 1536, 4, False, False),       # 3 per wave, NT = 2 (ring depth 2)
    (3, 256, 6144, 0, True, False),        # 12 per wave: three steady rounds
])
def test_weight_streaming_ring_kernel_is_bit_identical(cuda, lib, M, N, K, epi, norm, use_res):
    """gemm_skinny_ring_kernel (weights through per-wave LDS rings, M <= 16: every decode step) against gemm_skinny_kernel
    (weights through registers) on the same operands: the same MFMA operands in the same order -> the same bits."""
    g = torch.Generator().manual_seed(M * 7 + N + K)
    Npad = (N + 255) // 256 * 256
    n_out = N // 2 if epi == 4 else N
    A = (torch.randn(M, K, generator=g) * 1.5).half().cuda()
    W = torch.zeros(Npad, K, dtype=torch.float16)
    W[:N] = (torch.randn(N, K, generator=g) / K ** 0.5).half()
    W = W.cuda()
    gain = (1 + 0.1 * torch.randn(K, generator=g)).half().cuda() if norm else None
    bias = (torch.randn(Npad, generator=g) * 0.1).half().cuda() if epi != 4 and not norm else None
    res = (torch.randn(M, n_out, generator=g) * 0.5).half().cuda() if use_res else None
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    outs = []
    for kernel in (1, 3):
        C = torch.full((M, n_out), float("nan"), dtype=torch.float16, device="cuda")
        rc = lib.vstar_vqa_op_gemm(P(A), P(W), P(bias), P(res), P(C), M, N, K, epi, kernel, P(gain), 1e-5)
        assert rc == 0, lib.vstar_vqa_last_error(None)
        outs.append(C)
    assert not torch.isnan(outs[0].float()).any()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))


@pytest.mark.parametrize("M,N,K,epi", [(1, 768, 4096, 0), (9, 512, 1024, 0), (32, 2048, 4096, 4)])
def test_weight_streaming_gemm_with_fused_rmsnorm(cuda, lib, M, N, K, epi):
    """LlamaRMSNorm fused into the operand load == HF's two-step form (fp32 statistics, fp16 rounding points) + GEMM."""
    g = torch.Generator().manual_seed(M + N)
    Npad = (N + 255) // 256 * 256
    n_out = N // 2 if epi == 4 else N
    A = (torch.randn(M, K, generator=g) * 3.0).half().cuda()
    gain = (1 + 0.1 * torch.randn(K, generator=g)).half().cuda()
    W = torch.zeros(Npad, K, dtype=torch.float16)
    W[:N] = (torch.randn(N, K, generator=g) / K ** 0.5).half()
    W = W.cuda()
    C = torch.zeros(M, n_out, dtype=torch.float16, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    rc = lib.vstar_vqa_op_gemm(P(A), P(W), None, None, P(C), M, N, K, epi, 1, P(gain), 1e-5)
    assert rc == 0, lib.vstar_vqa_last_error(None)
    x = A.float().cpu()
    xn = (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5)).half()
    y = (gain.cpu() * xn).float()                                  # fp16 * fp16 -> fp16, as LlamaRMSNorm returns
    ref = y @ W[:N].float().cpu().T
    if epi == 4:
        r = ref.view(M, N // 32, 2, 16)
        ref = (torch.nn.functional.silu(r[:, :, 0].half().float()).half().float() * r[:, :, 1].half().float()).reshape(M, n_out)
    scale = float(ref.abs().max())
    assert float((C.float().cpu() - ref).abs().max()) <= 2e-3 * scale + 1e-3


# ------------------------------------------------------------------------------------------------------------
# the drop-in class (vstar_amd.vqa.VQA_LLM == reference VQA_LLM, vstar_bench_eval.py:38-165) and the evaluation loop
# ------------------------------------------------------------------------------------------------------------
def _vqa_llm(wseed=0):
    from vstar_amd.vqa import VQA_LLM
    cfg = VQAConfig.tiny()
    return VQA_LLM(cfg=cfg, engine=engine_for(cfg, wseed)), cfg


def _oracle_sample(llm, cfg, sd, image, question, crops, images_long, objects_long, answer=None):
    """What the reference computes for one sample, from the same host preprocessing, on the fp32 oracle."""
    from vstar_amd import vqa
    pix = [llm.image_processor.preprocess(image)["pixel_values"][0]] + ([c for c in crops] if crops is not None else [])
    pix = torch.stack(pix, 0).half().float()
    lo, sh = O.encode_images(sd, cfg, pix)
    ids = vqa.tokenizer_image_object_token(vqa.v1_prompt("<image>\n" + question, answer), llm.tokenizer)
    return ids, O.splice(sd, ids, lo[:1], sh[:1], lo[1:], sh[1:], images_long, objects_long)


def test_vqa_llm_class_against_oracle(cuda):
    from PIL import Image
    llm, cfg = _vqa_llm(0)
    sd = random_state_dict(cfg, 0, torch.float32)
    rng = np.random.default_rng(5)
    image = Image.fromarray(rng.integers(0, 256, (300, 420, 3), dtype=np.uint8))
    crops = torch.stack([llm.get_object_crop(image, [40, 30, 50, 60], patch_scale=1.2),
                         llm.get_object_crop(image, [200, 100, 90, 40], patch_scale=1.2)], 0)
    question = "Additional visual information to focus on: mug <object> at location [0.1,0.1,0.2,0.3]; cup <object> at " \
               "location [0.5,0.3,0.7,0.5].\nWhat is the colour of the mug?"
    options = ["The colour of the mug is red.", "The colour of the mug is blue.", "green", "The mug is yellow and white."]
    losses = llm.option_losses(image, question, options, crops, images_long=[False], objects_long=[True, True])
    chosen = llm.multiple_choices_inference(image, question, options, crops, images_long=[False], objects_long=[True, True])
    q_ids, emb = _oracle_sample(llm, cfg, sd, image, question, crops, [False], [True, True])
    opt_ids = []
    for o in options:
        full, _ = _oracle_sample(llm, cfg, sd, image, question, crops, [False], [True, True], answer=o)
        assert full[:len(q_ids)] == q_ids
        opt_ids.append(full[len(q_ids):])
    ref_losses, ref_pick = O.multiple_choice(sd, cfg, emb, opt_ids)
    print("losses", [round(float(x), 4) for x in losses], "oracle", [round(float(x), 4) for x in ref_losses])
    np.testing.assert_allclose([float(x) for x in losses], ref_losses.numpy(), atol=0.03)
    srt = np.sort(ref_losses.numpy())
    if srt[1] - srt[0] > 0.06:
        assert chosen == ref_pick
    # free-form greedy answer, plain image (long features), checked token by token against the oracle's arg-max
    text = llm.free_form_inference(image, "What is in the picture?", max_new_tokens=5)
    assert isinstance(text, str)
    _, emb2 = _oracle_sample(llm, cfg, sd, image, "What is in the picture?", None, None, None)
    table = sd["model.embed_tokens.weight"]
    logits, past = O.llama_forward(sd, cfg, emb2)
    for tok in llm.generated_ids[0]:
        top = logits[-1].topk(2)
        if float(top.values[0] - top.values[1]) < 0.05:
            break
        assert tok == int(top.indices[0])
        logits, past = O.llama_forward(sd, cfg, table[torch.tensor([tok])], past)
    # batched decode == one by one
    samples = [dict(image=image, question="What is in the picture?"),
               dict(image=image, question=question, object_crops=crops, images_long=[False], objects_long=[True, True])]
    texts = llm.free_form_batch(samples, max_new_tokens=4)
    batch_ids = [list(x) for x in llm.generated_ids]
    assert texts[0] == llm.free_form_inference(image, "What is in the picture?", max_new_tokens=4)
    assert batch_ids[0] == llm.generated_ids[0]


def test_eval_loop_end_to_end_on_synthetic_benchmark(cuda, tmp_path):
    """vstar_bench_eval.py:168-280 with BOTH models on the HIP engines (tiny widths): free-form answer -> (forced) missing
    objects -> visual search -> object crops + focus prompt -> multiple choice; checks the result file's schema."""
    import json
    from types import SimpleNamespace
    from PIL import Image
    from vstar_amd import bench_eval
    from vstar_amd.config import VSMConfig
    from vstar_amd.vsm import VSM
    llm, _ = _vqa_llm(0)
    vcfg = VSMConfig.tiny(max_text_len=128)
    vsm = VSM(SimpleNamespace(version="synthetic", vision_tower="synthetic", conv_type="llava_v1", use_mm_start_end=True,
                              model_max_length=512), cfg=vcfg, synthetic_seed=0)
    rng = np.random.default_rng(9)
    for split, n in (("direct_attributes", 2), ("relative_position", 1)):
        d = tmp_path / split
        d.mkdir()
        for i in range(n):
            Image.fromarray(rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)).save(d / f"img{i}.jpg")
            json.dump({"question": "What is the colour of the mug?", "options": ["red", "blue", "green", "black"]},
                      open(d / f"img{i}.json", "w"))
    calls = {"n": 0}
    real_free_form = llm.free_form_inference

    def free_form(image, question, **kw):
        real_free_form(image, question, max_new_tokens=3)      # exercise the decode path; random weights => random text
        calls["n"] += 1
        return bench_eval.MISSING_MSG + " mug, table." if calls["n"] % 2 else "It is red."
    llm.free_form_inference = free_form
    # random weights make the detector fire on many boxes; the tiny engine's feature table holds 8 images -> max_found_objects
    args = SimpleNamespace(benchmark_folder=str(tmp_path), output_path=str(tmp_path / "eval_result.json"),
                           minimum_size_scale=4.0, minimum_size=224, vsm_model_path="synthetic", max_found_objects=5)
    results = bench_eval.eval_model(args, llm, vsm)              # all searches in one cross-image lock-step stream
    out = json.load(open(args.output_path))
    # the reference's schedule (one image at a time) gives the same file
    calls["n"] = 0
    args1 = SimpleNamespace(**{**vars(args), "search_window": 1, "output_path": str(tmp_path / "eval_result_w1.json")})
    assert bench_eval.eval_model(args1, llm, vsm) == results
    assert set(out) == {"direct_attributes", "relative_position"}
    rec = out["direct_attributes"][0]
    assert set(rec) == {"question", "options", "image", "prediction_freeform", "missing_objects", "search_result",
                        "option_chosen", "correct"}
    searched = [r for s in out.values() for r in s if r["missing_objects"]]
    assert searched and all(len(r["search_result"]) >= 1 and 0 <= r["option_chosen"] < 4 for r in searched)
    assert results == out


def test_vqa_engine_reports_errors(cuda):
    """Limits are enforced with messages; nothing falls back or clamps silently."""
    from vstar_amd._lib import VstarError
    cfg = VQAConfig.tiny()
    eng = engine_for(cfg, 0)
    with pytest.raises(VstarError, match="max_ctx"):
        eng.forward([Seq([5] * 10, kv_slot=0, past_len=cfg.max_ctx - 4)], [(0, -1)])
    with pytest.raises(VstarError, match="slot"):
        eng.forward([Seq([5, 6], kv_slot=cfg.max_slots)], [(0, -1)])
    with pytest.raises(VstarError, match="prefix"):
        eng.forward([Seq([5, 6], kv_slot=1, prefix_slot=0, past_len=0)], [(0, -1)])
    with pytest.raises(VstarError, match="max_rows"):
        eng.forward([Seq([5] * 600, kv_slot=i) for i in range(4)], [(0, -1)])      # 4 x 600 padded rows > 2048
    with pytest.raises(VstarError, match="slot range"):
        eng.encode_images(torch.zeros(2, 3, 224, 224), cfg.max_images - 1)
    # a sequence of one row and an empty want list are legal
    lg, arg = eng.forward([Seq([7], kv_slot=2)], [])
    assert lg is None and len(arg) == 0


@pytest.mark.parametrize("M,N,K,epi", [(1024, 512, 256, 0), (2048, 4096, 4096, 0), (1280, 8192, 1024, 4)])
def test_fp16_gemm4w_equals_gemm256(M, N, K, epi):
    """Round 6: the fp16 instantiation of the 4-wave / AGPR 256^2 kernel (v_mfma_f32_16x16x32_f16 in the generated K loop) against the
    8-wave gemm256 on the same operands: bit-identical, three repetitions (the VQA-LLM's prefill linears take it at M % 256 == 0)."""
    import ctypes
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K + epi)
    n_out = N // 2 if epi == 4 else N
    A = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    res = (torch.randn(M, n_out, generator=g) * 0.5).half().cuda() if epi == 0 else None
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None      # noqa: E731
    outs = {}
    for kernel in (5, 4, 4, 4):
        C = torch.full((M, n_out), float("nan"), dtype=torch.float16, device="cuda")
        rc = lib.vstar_vqa_op_gemm(P(A), P(W), None, P(res), P(C), M, N, K, epi, kernel, None, 0.0)
        assert rc == 0, lib.vstar_vqa_last_error(None)
        if kernel == 5:
            outs[5] = C
        else:
            assert torch.equal(C.view(torch.int16), outs[5].view(torch.int16))
    assert not torch.isnan(outs[5].float()).any()
