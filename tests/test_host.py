"""Host-side logic: preprocessing vs the HF processors the reference calls, prompt/tokenisation, config/FLOP tables,
record (de)serialisation, and the world_size-2 all-gather of result records over gloo."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
from PIL import Image

from vstar_amd import preprocess as pp
from vstar_amd.config import VSMConfig
from vstar_amd.engine import VstarEngine, loc_positions
from vstar_amd.weights import dense_pe, random_state_dict, state_dict_spec


def _img(w, h, seed):
    rng = np.random.default_rng(seed)
    return Image.fromarray(rng.integers(0, 255, size=(h, w, 3), dtype=np.uint8))


@pytest.mark.parametrize("w,h", [(640, 480), (300, 900), (224, 224), (1000, 1000)])
def test_preprocess_matches_hf_processors(w, h):
    from transformers import CLIPImageProcessor, OwlViTImageProcessor
    img = _img(w, h, w + h)
    clip_proc = CLIPImageProcessor(size={"shortest_edge": 224}, crop_size={"height": 224, "width": 224})
    ref = clip_proc.preprocess(pp.expand2square(img, pp.background_color()), return_tensors="np")["pixel_values"][0]
    got = pp.clip_preprocess(img, 224)
    assert got.shape == (3, 224, 224)
    assert np.array_equal(got, ref)                     # bit-identical to the HF processor
    owl_proc = OwlViTImageProcessor()
    ref2 = owl_proc(images=np.array(img), return_tensors="np")["pixel_values"][0]
    got2 = pp.owl_preprocess(img, 768)
    assert np.array_equal(got2, ref2)


@pytest.mark.parametrize("w,h,ow,oh", [(640, 480, 336, 336), (300, 900, 768, 768), (1000, 700, 224, 224), (200, 150, 768, 768),
                                        (768, 768, 768, 768), (1920, 1080, 336, 336)])
def test_pil_resize_oracle_is_bit_exact(w, h, ow, oh):
    """oracle/pil_resize_oracle.py (the algorithm the GPU preprocessing kernels implement) == Pillow, bit for bit."""
    from oracle.pil_resize_oracle import resize_u8
    img = np.asarray(_img(w, h, w * 3 + h))
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), resample=Image.BICUBIC))
    assert np.array_equal(resize_u8(img, ow, oh), ref)


def test_expand2square_top_left():
    img = _img(30, 10, 0)
    sq = pp.expand2square(img)
    assert sq.size == (30, 30)
    assert np.array_equal(np.asarray(sq)[:10], np.asarray(img))
    assert tuple(np.asarray(sq)[20, 5]) == (122, 116, 104)


def test_prompt_and_tokenisation():
    tok = pp.SyntheticTokenizer(32004)
    q = pp.LOCATE_QUESTION.format("red cup")
    prompt = pp.build_prompt(q)
    assert prompt.startswith(pp.LLAVA_V1_SYSTEM + " USER: <im_start><image><im_end>\nPlease locate the red cup")
    assert prompt.endswith(" ASSISTANT:")
    ids = pp.tokenizer_image_token(prompt, tok)
    assert ids[0] == tok.bos_token_id and ids.count(-200) == 1 and ids.count(tok.bos_token_id) == 1
    i = ids.index(-200)
    assert ids[i - 1] == tok.special["<im_start>"] and ids[i + 1] == tok.special["<im_end>"]
    full = pp.tokenizer_image_token(pp.build_prompt(q, answer=pp.ANSWER_TEMPLATE), tok)
    assert full[: len(ids)] == ids and full.count(tok.special["[LOC]"]) == 1
    # the hidden state that predicts [LOC] sits at idx([LOC]) - 1 + (P - 1) of the spliced sequence (VSM.py:230-234)
    loc = loc_positions(np.asarray([full]), tok.special["[LOC]"], 256)
    assert int(loc[0]) == full.index(tok.special["[LOC]"]) - 1 + 255


def test_config_flops_match_baseline_tables():
    f336 = VSMConfig.seal_7b(336).flops_per_crop(64)
    f224 = VSMConfig.seal_7b(224).flops_per_crop(64)
    assert abs(f336["core"] / 8.768e12 - 1) < 0.01
    assert abs(f224["core"] / 4.329e12 - 1) < 0.01
    assert abs(f336["full"] / 9.37e12 - 1) < 0.01
    assert abs(f336["owl_tower"] / 590.1e9 - 1) < 0.01


def test_state_dict_spec_and_dense_pe():
    cfg = VSMConfig.tiny()
    spec = state_dict_spec(cfg)
    sd = random_state_dict(cfg, 3, torch.float32)
    assert list(sd.keys()) == list(spec.keys())
    assert all(tuple(sd[k].shape) == tuple(v) for k, v in spec.items())
    sub = random_state_dict(cfg, 3, torch.float32, keys=["lm_head.weight"])
    assert torch.equal(sub["lm_head.weight"], sd["lm_head.weight"])
    from oracle import vsm_oracle
    pe = dense_pe(sd["model.prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"])
    ref = vsm_oracle.dense_pe(sd)[0].permute(1, 2, 0).reshape(2304, 256)
    assert torch.allclose(pe, ref, atol=1e-6)


def test_result_record_unpack_roundtrip():
    from vstar_amd import _lib
    rec = np.arange(2 * _lib.RESULT_FLOATS, dtype=np.float32).reshape(2, -1)
    out = VstarEngine.unpack(rec, 3)
    assert out["pred_logits"].shape == (2, 2304, 1) and out["pred_boxes"].shape == (2, 2304, 4)
    assert out["low_res_masks"].shape == (2, 1, 192, 192) and out["tf_argmax"].shape == (2, 3)
    assert out["pred_boxes"][1, 0, 0] == rec[1, 2304]
    assert out["low_res_masks"][0, 0, 0, 0] == rec[0, 2304 * 5]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dist_worker(rank, world, port, n_items, q):
    import torch.distributed as dist
    from vstar_amd import dist as vd
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    R = 7
    full = torch.arange(n_items * R, dtype=torch.float32).reshape(n_items, R)      # record i = row i
    mine = vd.shard_indices(n_items, rank, world)
    per = vd.pad_count(n_items, world)
    local = torch.zeros(per, R)
    local[: len(mine)] = full[mine]                                               # this rank "scored" its shard
    got = vd.allgather_records(local, n_items)
    q.put((rank, torch.equal(got, full)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [8, 5, 1])
def test_allgather_records_world2_gloo(n_items):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


# ---- data-parallel VSM.inference_batch / visual_search over 2 ranks (gloo), with a deterministic fake engine ----
from _fake_vsm import FakeEngine as _FakeEngine  # noqa: E402


def _search_once():
    import warnings
    from vstar_amd.synthetic import synthetic_image
    from vstar_amd.search import smallest_size_for, visual_search
    from vstar_amd.vsm import VSM
    eng = _FakeEngine()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vsm = VSM(None, engine=eng, tokenizer=pp.SyntheticTokenizer(eng.cfg.llm_vocab), strict_template=False)
        img = synthetic_image(1280, 720, 5)
        stats = {}
        step, n, ok, _ = visual_search(vsm, img, "kite", None, smallest_size_for(1280, 720), confidence_high=2.0,
                                       target_cue_threshold=-1.0, target_cue_threshold_minimum=-1.0, stats=stats)
    return ([int(v) for v in step["bbox"]], [float(v) for v in step["detection_result"]], int(n), bool(ok),
            [p["bbox"] for p in stats["search_path"]], sum(eng.calls))


def _dp_search_worker(rank, world, port, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    q.put((rank, _search_once()))
    dist.destroy_process_group()


def test_data_parallel_search_world2_equals_single_process():
    import torch.multiprocessing as mp
    single = _search_once()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_search_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    for r in (0, 1):
        assert res[r][:5] == single[:5]                    # same path, boxes and decisions on every rank
    assert res[0][5] + res[1][5] == single[5]              # the crops were split between the ranks, none scored twice
    assert min(res[0][5], res[1][5]) >= single[5] // 2 - 2


def test_bench_multiprocess_launch_path_on_cpu():
    """The driver's N>1 launch line (torch.distributed.run, one rank per GPU) against bench.py's own collectives and timing
    protocol, with the engine stubbed out (--fake-engine, gloo): rank 0 prints ONE JSON line, n_gpus = world, the gather is in
    rank order and the slowest rank's time is the one reported."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--fake-engine"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["ms_per_step"] >= 20.0          # rank 1 sleeps 20 ms per step: max over ranks, not rank 0's 10 ms
    for k in ("metric", "value", "unit", "higher_is_better", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d
    # the best-first stream legs at N = 2: crop sharding (every rank walks the same searches; each step's crops
    # dealt over the ranks, records all-gathered) and sample sharding (searches dealt over the ranks) must reach exactly the outcomes
    # of the single-process run — same searches, same visited nodes, same final boxes
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--fake-engine"],
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    ref = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])["search_stream"]
    for key in ("search_stream", "search_stream_shard_samples"):
        leg = d[key]
        assert "error" not in leg, leg
        assert leg["ranks"] == 2 and leg["searches"] == ref["searches"] == 12
        assert leg["useful_crops"] == ref["useful_crops"] and leg["outcome_digest"] == ref["outcome_digest"], (key, leg, ref)
        assert leg["wasted_crop_frac"] >= 0.0 and leg["searches_per_s"] > 0
    assert d["search_stream"]["shard"] == "crops" and d["search_stream_shard_samples"]["shard"] == "samples"
    assert d["search_stream"]["per_rank_crops_per_step"] <= d["search_stream"]["mean_crops_per_step"] / 2 + 1e-9


def test_bench_plain_launch_with_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` WITHOUT a launcher (the form of the driver's N = 1 command) used to die
    on an assertion.  It must start its own two ranks (torch.distributed.run on 127.0.0.1) and still print ONE JSON line, last on
    stdout, with n_gpus = 2; a launcher whose world size contradicts --gpus is refused with a message, not an AssertionError."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--fake-engine"],
                         capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    nonempty = [l for l in out.stdout.splitlines() if l.strip()]
    assert nonempty[-1].startswith("{") and sum(l.startswith("{") for l in nonempty) == 1, out.stdout[-2000:]
    d = json.loads(nonempty[-1])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["search_stream"]["ranks"] == 2 and d["search_stream_shard_samples"]["ranks"] == 2
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--fake-engine"], capture_output=True, text=True,
                         timeout=120, cwd=root, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "--gpus 4" in bad.stderr and "AssertionError" not in bad.stderr


def _make_bench_folder(root):
    """A miniature V*Bench tree (two splits, image + JSON annotation with bbox / target_object lists, vstar_bench layout)."""
    import json
    from vstar_amd.synthetic import synthetic_image
    k = 0
    for split, sizes in (("direct_attributes", [(900, 600), (640, 480)]), ("relative_position", [(1000, 700)])):
        d = os.path.join(root, split)
        os.makedirs(d)
        for j, (w, h) in enumerate(sizes):
            synthetic_image(w, h, 50 + k).save(os.path.join(d, f"img{j}.jpg"))
            targets = ["kite", "dog"] if split == "relative_position" else ["umbrella"]
            json.dump({"target_object": targets, "bbox": [[10 + 5 * t, 20, 80, 60] for t in range(len(targets))],
                       "question": "?", "options": ["a", "b"]}, open(os.path.join(d, f"img{j}.json"), "w"))
            k += 1


@pytest.mark.parametrize("shard", ["crops", "samples"])
def test_visual_search_entry_point_under_torchrun_world2(tmp_path, shard):
    """BASELINE configs 3/4 launch path: the REAL visual_search.py entry point under torch.distributed.run with two ranks
    (gloo on CPU, engine stubbed through --vsm-factory; the VSM class, its crop sharding and the record all-gather are the real
    ones): process group set up and torn down, rank 0 alone prints, and the metrics equal the single-process run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    folder = str(tmp_path / "bench")
    _make_bench_folder(folder)
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "tests") + os.pathsep + root)
    common = ["--benchmark-folder", folder, "--vsm-factory", "_fake_vsm:make", "--confidence_high", "2.0", "--confidence_low", "0.0",
              "--target_cue_threshold", "-1", "--target_cue_threshold_minimum", "-1"]
    single_json = str(tmp_path / "single.json")
    out1 = subprocess.run([sys.executable, os.path.join(root, "visual_search.py"), *common, "--output_path", single_json],
                          capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out1.returncode == 0, out1.stderr[-2000:]
    port = _free_port()
    multi_json = str(tmp_path / "multi.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "visual_search.py"), *common, "--shard", shard, "--output_path", multi_json]
    out2 = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out2.returncode == 0, out2.stderr[-2000:]
    metric_lines = lambda text: [l for l in text.splitlines() if l.startswith(("Avg search path length", "Top 1 Acc"))]  # noqa: E731
    assert len(metric_lines(out2.stdout)) == 2                       # rank 0 only
    assert metric_lines(out2.stdout) == metric_lines(out1.stdout)
    a, b = json.load(open(single_json)), json.load(open(multi_json))
    assert b["world_size"] == 2 and b["shard"] == shard
    assert a["hits"] == b["hits"] and a["path_lengths"] == b["path_lengths"] and len(a["hits"]) == 4


def test_visual_search_entry_point_visualization_writes_the_references_file_set(tmp_path):
    """--visualization (visual_search.py:339-376, 531-548): one directory <output_path>/<split>/<image stem>_<k> per (image, target)
    with whole_image.jpg, step_k.jpg, step_k_heatmap.jpg for expanded nodes, final_patch_image.jpg + search_result.jpg for the node
    that carries the detection, context_cue.txt — rendered with PIL (cv2/matplotlib are absent); metrics unchanged by rendering."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    folder = str(tmp_path / "bench")
    _make_bench_folder(folder)
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "tests") + os.pathsep + root)
    common = ["--benchmark-folder", folder, "--vsm-factory", "_fake_vsm:make", "--confidence_high", "2.0", "--confidence_low", "0.0",
              "--target_cue_threshold", "-1", "--target_cue_threshold_minimum", "-1"]
    plain = subprocess.run([sys.executable, os.path.join(root, "visual_search.py"), *common, "--output_path", str(tmp_path / "p.json")],
                           capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert plain.returncode == 0, plain.stderr[-2000:]
    out_dir = str(tmp_path / "vis")
    vis = subprocess.run([sys.executable, os.path.join(root, "visual_search.py"), *common, "--visualization", "--output_path", out_dir],
                         capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert vis.returncode == 0, vis.stderr[-2000:]
    a, b = json.load(open(tmp_path / "p.json")), json.load(open(os.path.join(out_dir, "results.json")))
    assert a["hits"] == b["hits"] and a["path_lengths"] == b["path_lengths"]
    dirs = {"direct_attributes": ["img0_0", "img1_0"], "relative_position": ["img0_0", "img0_1"]}
    from PIL import Image
    for split, names in dirs.items():
        assert sorted(os.listdir(os.path.join(out_dir, split))) == names
        for nm in names:
            files = set(os.listdir(os.path.join(out_dir, split, nm)))
            assert {"whole_image.jpg", "step_1.jpg", "context_cue.txt", "final_patch_image.jpg", "search_result.jpg"} <= files, files
            assert "step_1_heatmap.jpg" in files                   # the root is larger than smallest_size: it was expanded
            im = Image.open(os.path.join(out_dir, split, nm, "step_1_heatmap.jpg"))
            whole = Image.open(os.path.join(out_dir, split, nm, "whole_image.jpg"))
            assert im.size == whole.size                           # the root patch is the whole image


def test_trained_like_weights_are_deterministic_keep_the_key_set_and_answer_the_template_on_the_oracle():
    """vstar_amd.weights.trained_like_state_dict (round 4): same keys / shapes as the random set, reproducible, layer aliasing of
    share_layers preserved through the dtype conversion, feature switches independent of the draw order — and, through the CPU
    oracle at the tiny geometry, a greedy continuation of a locate prompt that IS "Sure, [LOC]." followed by EOS, with outlier
    channels in the residual stream (max / rms of the final hidden state well above the random set's)."""
    from oracle import vsm_oracle
    from vstar_amd.weights import random_state_dict, template_chain, trained_like_state_dict
    cfg = VSMConfig.tiny()
    tok = pp.SyntheticTokenizer(cfg.llm_vocab)
    chain = template_chain(tok)
    assert len(chain) == 5 and chain[-1][1] == tok.eos_token_id and chain[2][1] == cfg.llm_vocab - 1
    a = trained_like_state_dict(cfg, seed=3, chain=chain)
    b = trained_like_state_dict(cfg, seed=3, chain=chain)
    ref = random_state_dict(cfg, seed=3)
    assert list(a) == list(ref) and all(a[k].shape == ref[k].shape and a[k].dtype == torch.bfloat16 for k in a)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert a["model.layers.1.mlp.down_proj.weight"] is a["model.layers.0.mlp.down_proj.weight"]        # share_layers aliasing survives
    only_attn = trained_like_state_dict(cfg, seed=3, chain=chain, features=("attn",))
    assert torch.equal(only_attn["model.layers.0.mlp.down_proj.weight"], ref["model.layers.0.mlp.down_proj.weight"])
    assert not torch.equal(only_attn["model.layers.0.self_attn.k_proj.weight"], ref["model.layers.0.self_attn.k_proj.weight"])
    # greedy decode on the oracle: prompt + teacher-forced template, arg-max at the answer positions
    q = pp.LOCATE_QUESTION.format("green bottle")
    ids_p = pp.tokenizer_image_token(pp.build_prompt(q), tok)
    ids_f = pp.tokenizer_image_token(pp.build_prompt(q, answer=pp.ANSWER_TEMPLATE), tok) + [tok.eos_token_id]
    P = cfg.n_img_tokens
    pos = [c - 1 + (P - 1) for c in range(len(ids_p), len(ids_f))]
    clip = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    sd32 = {k: v.float() for k, v in a.items()}
    peaks, orig = [], vsm_oracle.rms_norm

    def spy(x, w, eps):                      # the RAW residual stream as every RMSNorm sees it: worst token's max / rms
        peaks.append(float((x.abs().amax(-1) / x.pow(2).mean(-1).sqrt()).max()))
        return orig(x, w, eps)

    vsm_oracle.rms_norm = spy
    try:
        with torch.no_grad():
            out = vsm_oracle.vsm_forward(sd32, cfg, clip, None, torch.tensor([ids_f[:-1]]), cfg.llm_vocab - 1, verify_pos=torch.tensor([pos[:-1]]))
            peak_tl = max(peaks)
            peaks.clear()
            vsm_oracle.vsm_forward({k: v.float() for k, v in ref.items()}, cfg, clip, None, torch.tensor([ids_f[:-1]]), cfg.llm_vocab - 1)
            peak_rnd = max(peaks)
    finally:
        vsm_oracle.rms_norm = orig
    assert out["tf_argmax"][0].tolist() == ids_f[len(ids_p):-1]
    top2 = out["tf_logits"][0].topk(2, -1).values
    assert float((top2[:, 0] - top2[:, 1]).min()) > 10.0                          # decided by a wide margin, not by luck
    assert peak_tl > 2.0 * peak_rnd and peak_tl > 10.0, (peak_tl, peak_rnd)        # outlier channels / the massive-activation BOS


def test_scores_keep_the_references_bf16_sigmoid_ties():
    """visual_search.py:225 returns det_result['pred_logits'][0].sigmoid() of a BF16 tensor: the rounding makes distinct logits
    tie, and the scheduler's argmax() (first maximum) / `> confidence` comparisons act on the rounded values.  The drop-in keeps
    that dtype so that the selected box is the reference's even among near-equal detections."""
    from vstar_amd.vsm import _scores
    logits = np.asarray([[1.0], [3.015625], [3.03125], [2.0]], np.float32)       # bf16-representable, as the engine emits them
    s = _scores(logits)
    assert s.dtype == torch.bfloat16 and s.shape == (4, 1)
    assert float(s[1]) == float(s[2]) == 0.953125                                 # a tie after rounding ...
    assert int(s.view(-1).argmax()) == 1                                          # ... resolved like the reference: first maximum
    assert int(torch.from_numpy(logits).sigmoid().view(-1).argmax()) == 2         # fp32 scores would have picked the other box
    # threshold semantics on rounded values: sigmoid(0.002) = 0.5005 rounds to 0.5 and is NOT > 0.5
    assert not bool(_scores(np.asarray([0.002], np.float32))[0] > 0.5)


def test_vsm_checkpoint_dir_roundtrip(tmp_path):
    """A HF `save_pretrained`-layout VSM directory (two safetensors shards; a stray vision tower copy and non-engine keys inside,
    as in craigwu/seal_vsm_7b) + a separate CLIP directory (openai/clip-vit-large-patch14: vision_model.* next to text_model.*)
    -> the engine's key space, tensor for tensor (vstar_amd.weights.load_checkpoint_dir; visual_search.py:157-161)."""
    from safetensors.torch import save_file
    from vstar_amd.config import VSMConfig
    from vstar_amd.weights import load_checkpoint_dir, random_state_dict, state_dict_spec
    cfg = VSMConfig.tiny()
    sd = random_state_dict(cfg, seed=3, dtype=torch.bfloat16)
    vsm_dir, clip_dir = tmp_path / "seal_vsm", tmp_path / "clip"
    vsm_dir.mkdir()
    clip_dir.mkdir()
    own = {k: v for k, v in sd.items() if not k.startswith("clip.")}
    keys = sorted(own)
    half = len(keys) // 2
    shard_a = {k: own[k] for k in keys[:half]}
    shard_b = {k: own[k] for k in keys[half:]}
    shard_b["model.vision_tower.vision_tower.vision_model.embeddings.class_embedding"] = torch.zeros(4)   # ignored copy
    shard_b["model.owlvit.text_model.embeddings.token_embedding.weight"] = torch.zeros(2, 2)              # not on the path
    save_file(shard_a, str(vsm_dir / "model-00001-of-00002.safetensors"))
    save_file(shard_b, str(vsm_dir / "model-00002-of-00002.safetensors"))
    clip = {k[len("clip."):]: v for k, v in sd.items() if k.startswith("clip.")}
    clip["text_model.embeddings.token_embedding.weight"] = torch.zeros(2, 2)
    clip["visual_projection.weight"] = torch.zeros(2, 2)
    torch.save(clip, str(clip_dir / "pytorch_model.bin"))                                                 # .bin shard form
    got = load_checkpoint_dir(str(vsm_dir), str(clip_dir))
    for k in state_dict_spec(cfg):
        assert k in got, k
        assert torch.equal(got[k], sd[k]), k
    assert not any(".vision_tower." in k for k in got)
    with pytest.raises(FileNotFoundError):
        load_checkpoint_dir(str(vsm_dir), str(tmp_path / "missing"))


def test_bench_eval_vsm_factory_requires_a_local_vision_tower(tmp_path):
    """Round-1 advisor finding: eval_model hard-coded the hub name of the CLIP tower; with a real checkpoint directory that ended
    in an os.listdir FileNotFoundError deep in the loader.  Now --vision-tower is honoured and its absence is reported clearly."""
    import types
    from vstar_amd.bench_eval import make_vsm
    (tmp_path / "vsm").mkdir()
    args = types.SimpleNamespace(vsm_model_path=str(tmp_path / "vsm"), vision_tower=None)
    with pytest.raises(FileNotFoundError, match="--vision-tower"):
        make_vsm(args)


def test_prompts_match_the_references_conversation_templates():
    """build_prompt for both --conv_type values vs the strings the reference's Conversation.get_prompt() returns
    (tests/golden/prompts.json, recorded from VisualSearch/model/llava/conversation.py by oracle/gen_prompt_golden.py)."""
    import json
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "prompts.json")))
    assert {g["conv_type"] for g in gold} == {"llava_v1", "llava_llama_2"}
    for g in gold:
        got = pp.build_prompt(g["question"], g["use_mm_start_end"], g["answer"] or None, conv_type=g["conv_type"])
        if g["answer"]:      # the reference closes a given answer with sep2; the teacher-forced prompt stops after the answer
            got += " </s>" if g["conv_type"] == "llava_llama_2" else "</s>"
        assert got == g["prompt"], (g["conv_type"], g["use_mm_start_end"], g["answer"])
    with pytest.raises(ValueError):
        pp.build_prompt("q", True, None, conv_type="mpt")


# ---------------- grouped scoring bookkeeping (VSM._score_boxes_grouped) on a CPU stand-in for the engine ----------------
import warnings  # noqa: E402

from vstar_amd.synthetic import synthetic_image  # noqa: E402
class _GroupedFakeEngine(_FakeEngine):
    """Adds the on-device entry points: crops travel as boxes; a record is a function of (box, the prompt's token ids), whichever
    entry point produced it — so any slip in the grouping's index bookkeeping shows up as a wrong record in the caller's order."""

    def __init__(self, max_batch=4, max_text_len=96):
        super().__init__(max_batch)
        from vstar_amd.config import VSMConfig
        self.cfg = VSMConfig.tiny(max_batch=max_batch, max_text_len=max_text_len)
        self.grouped_calls, self.plain_calls, self._boxes = [], [], None

    def set_image(self, image):
        self.image = image

    @staticmethod
    def _rec(box, ids):
        import zlib
        from vstar_amd import _lib
        rec = np.zeros(_lib.RESULT_FLOATS, np.float32)
        seed = zlib.crc32(repr((tuple(int(v) for v in box), tuple(int(t) for t in ids if t != 0))).encode()) % (2 ** 31)
        rec[:16] = np.random.default_rng(seed).standard_normal(16)
        return rec

    def score_boxes(self, xyxy, ids, loc, verify_pos=None, raw=False, out_dev=None, share_prefix=None):
        self.plain_calls.append(len(xyxy))
        assert len(xyxy) <= self.cfg.max_batch
        out = np.stack([self._rec(b, row) for b, row in zip(np.asarray(xyxy), np.asarray(ids))])
        return out if raw else self.unpack(out, 0)

    def preprocess_boxes(self, xyxy):
        self._boxes = np.asarray(xyxy)

    def score_grouped(self, clip, owl, prefix_ids, suffix_ids, loc_in_suffix, verify_in_suffix=None, raw=False, internal_pixels=False):
        G, T, Ls = suffix_ids.shape
        assert internal_pixels and len(self._boxes) == G and G * T <= self.cfg.max_batch and Ls <= 32
        self.grouped_calls.append((G, T))
        out = np.stack([self._rec(self._boxes[g], list(prefix_ids) + list(suffix_ids[g, t])) for g in range(G) for t in range(T)])
        return out if raw else self.unpack(out, 0)


@pytest.mark.parametrize("mode", [True, "always"])
def test_grouped_scoring_bookkeeping_returns_records_in_the_callers_order(mode):
    from vstar_amd.vsm import VSM
    eng = _GroupedFakeEngine(max_batch=4)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vsm = VSM(None, engine=eng, tokenizer=pp.SyntheticTokenizer(eng.cfg.llm_vocab), strict_template=False)
        vsm.set_image(synthetic_image(400, 300, 1))
        vsm.group_prompts = mode
        qs = [pp.LOCATE_QUESTION.format(n) for n in ("kite", "red umbrella", "dog")] + ["What is shown here?"]
        boxes = [[0, 0, 400, 300], [0, 0, 200, 150], [200, 0, 200, 150], [0, 150, 200, 150]]
        # box 0: three template prompts + one foreign question; box 1: two prompts; box 2: one prompt; box 3: the foreign question only
        pairs = [(0, 0), (1, 1), (0, 1), (2, 2), (0, 2), (1, 0), (0, 3), (3, 3)]
        out = vsm.inference_boxes([boxes[b] for b, _ in pairs], [qs[q] for _, q in pairs], mode="detection", upsample=False)
    want = []
    for b, q in pairs:
        ids = vsm._ids(qs[q])[0]
        x, y, w, h = boxes[b]
        want.append(_GroupedFakeEngine._rec([x, y, x + w, y + h], ids))
    got = np.stack([o[0].numpy().ravel() for o in out])      # pred_boxes of each result
    from vstar_amd.engine import VstarEngine
    ref = VstarEngine.unpack(np.stack(want), 0)["pred_boxes"].reshape(len(pairs), -1)
    assert np.array_equal(got, ref)
    # the foreign question never takes the grouped entry point; template prompts of multi-prompt crops always do
    assert (1, 3) in eng.grouped_calls or any(T == 3 for _, T in eng.grouped_calls)
    if mode == "always":
        assert any(T == 1 for _, T in eng.grouped_calls)            # box 2's single prompt goes through the grouped entry too
        assert sum(eng.plain_calls) == 2                            # only the two foreign-question pairs stay on the plain path
    else:
        assert all(T >= 2 for _, T in eng.grouped_calls)
        assert sum(eng.plain_calls) == 3


def test_grouped_always_batches_crops_with_different_single_prompts_into_one_call():
    """visual_search_stream switches a grouping VSM to group_prompts = "always"; a window of searches for
    DIFFERENT objects on different crops must then still share engine calls (crops are bucketed by their number of prompts, the
    suffix ids are per crop) — round 3 made one call per distinct prompt."""
    from vstar_amd.vsm import VSM
    eng = _GroupedFakeEngine(max_batch=4)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vsm = VSM(None, engine=eng, tokenizer=pp.SyntheticTokenizer(eng.cfg.llm_vocab), strict_template=False)
        vsm.set_image(synthetic_image(400, 300, 1))
        vsm.group_prompts = "always"
        names = ["kite", "red umbrella", "dog", "boat", "traffic light", "cup"]
        boxes = [[10 * k, 5 * k, 100 + k, 80 + k] for k in range(6)]
        out = vsm.inference_boxes(boxes, [pp.LOCATE_QUESTION.format(n) for n in names], mode="detection", upsample=False)
    # two calls (the activation-row budget of this tiny configuration admits 3 one-prompt crops per call), not six
    assert eng.grouped_calls == [(3, 1), (3, 1)] and eng.plain_calls == []
    assert vsm.timers["engine_calls"] == 2 and vsm.timers["crops"] == 6
    for k, o in enumerate(out):
        x, y, w, h = boxes[k]
        want = _GroupedFakeEngine._rec([x, y, x + w, y + h], vsm._ids(pp.LOCATE_QUESTION.format(names[k]))[0])
        from vstar_amd.engine import VstarEngine
        assert np.array_equal(o[0].numpy().ravel(), VstarEngine.unpack(want[None], 0)["pred_boxes"].ravel())


def test_calibrate_step_ms_feeds_the_speculation_policy():
    """VSM.calibrate_step_ms measures t(B) on the engine at hand (here the CPU stand-in) and visual_search_stream's default policy
    picks the table up (`step_ms_table`)."""
    from vstar_amd.search import SpeculationPolicy
    from vstar_amd.vsm import VSM
    eng = _FakeEngine(max_batch=4)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vsm = VSM(None, engine=eng, tokenizer=pp.SyntheticTokenizer(eng.cfg.llm_vocab), strict_template=False)
    table = vsm.calibrate_step_ms(batches=(1, 2, 4, 8), repeats=1)
    assert list(table) == [1, 2, 4] and all(v > 0 for v in table.values())      # 8 > max_batch is skipped
    assert vsm.step_ms_table is table
    pol = SpeculationPolicy(vsm.step_ms_table, cap=8)
    assert pol.step_ms(1) == table[1] and pol.step_ms(3) == pytest.approx((table[2] + table[4]) / 2)


def test_choose_shard_policy():
    """`--shard auto` (vstar_amd.dist.choose_shard): priced with the measured step-time table — whole searches per rank when every rank
    can be kept busy, crops of a step when there are fewer searches than ranks (a single search cannot use a second GPU otherwise)."""
    from vstar_amd.dist import choose_shard, step_ms
    assert step_ms(1) == 18.6 and step_ms(32) == 232.0 and 25.1 < step_ms(3) < 39.0 and step_ms(64) == 464.0
    assert choose_shard(1, 8, 1) == "crops"            # one search, eight GPUs: only crop sharding spreads its speculative crops
    assert choose_shard(4, 8, 1) == "crops"            # fewer searches than ranks
    assert choose_shard(1, 8, 1, crops_per_search_step=1) == "samples"      # the reference's schedule has one crop per step: nothing to deal
    assert choose_shard(191, 8, 4) == "samples"        # the V*Bench split: plenty of searches, no collective on the data path
    assert choose_shard(64, 8, 8) == "samples"
    assert choose_shard(10, 1, 4) == "samples"         # one rank: nothing to deal
    # a table in which small batches are as efficient as large ones removes crop sharding's only cost advantage
    flat = {1: 7.0, 32: 224.0}
    assert choose_shard(191, 8, 4, flat) == "samples"


def test_entry_points_default_to_the_engine_collective_with_fallback():
    """Round 6: --engine-comm defaults to `auto` on both entry points (and bench.py); the bare flag still parses as `on`."""
    import visual_search
    import vstar_bench_eval
    a = visual_search.parse_args(["--benchmark-folder", "x"])
    assert a.engine_comm == "auto" and a.shard == "crops"
    assert visual_search.parse_args(["--benchmark-folder", "x", "--engine-comm"]).engine_comm == "on"
    assert visual_search.parse_args(["--benchmark-folder", "x", "--engine-comm", "off", "--shard", "auto"]).shard == "auto"
    assert vstar_bench_eval.parse_args([]).engine_comm == "auto"


def test_w8a8_prompt_padding_makes_full_batches_whole_tiles():
    """vstar_amd/vsm.py::w8a8_padded_len: S = L - 1 + P becomes a multiple of 8 (so 32 x S and 64 x S are multiples of 256: the
    block-scaled W8A8 chain's domain), never beyond max_text_len, never shorter."""
    from vstar_amd.vsm import w8a8_padded_len
    for P in (256, 576):
        for L in range(20, 70):
            Lp = w8a8_padded_len(L, P, 80)
            assert L <= Lp <= L + 7 and (Lp - 1 + P) % 8 == 0 and (32 * (Lp - 1 + P)) % 256 == 0
    assert w8a8_padded_len(64, 576, 65) == 65          # S = 639 -> 640
    assert w8a8_padded_len(66, 576, 66) == 66          # no room: unchanged
