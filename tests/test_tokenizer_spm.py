"""The tokenizer-dependent host logic on a REAL sentencepiece / LLaMA tokenizer (VERDICT r2 missing #5, weak #3).

tests/golden/spm_llama/ is a tiny BPE model with byte fallback trained by oracle/gen_spm_tokenizer.py (LLaMA recipe: BOS 1,
EOS 2, dummy prefix) with [LOC], <im_start>, <im_end> added like the checkpoint's tokenizer; tests/golden/spm_prompts.json holds
the ids the REFERENCE's own tokenizer_image_token + conversation templates produce with it (mm_utils.py:19-44,
visual_search.py:176-196).  Checked here, without a GPU: the `VSM.__init__` AutoTokenizer branch (visual_search.py:148-156), the
prompt ids, the prompt / teacher-forced "Sure, [LOC]." boundary and verify tokens of `VSM._ids`, and the template-prefix split that
`group_prompts = "always"` relies on — including object names that do NOT keep the prefix stable."""
import json
import os
import types
import warnings

import numpy as np
import pytest

from test_host import _GroupedFakeEngine
from vstar_amd import preprocess as pp
from vstar_amd.synthetic import synthetic_image
from vstar_amd.vsm import VSM

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOK_DIR = os.path.join(GOLD, "spm_llama")


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(GOLD, "spm_prompts.json")))


def _vsm(conv_type="llava_v1", max_batch=4, group=True):
    eng = _GroupedFakeEngine(max_batch=max_batch, max_text_len=160)
    args = types.SimpleNamespace(version=TOK_DIR, vision_tower=None, conv_type=conv_type, use_mm_start_end=True, model_max_length=512)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vsm = VSM(args, engine=eng)            # tokenizer=None + a real directory -> the AutoTokenizer branch
    vsm.group_prompts = group
    return vsm, eng


def test_autotokenizer_branch_loads_a_llama_tokenizer(gold):
    vsm, _ = _vsm()
    tok = vsm.vsm_tokenizer
    assert not isinstance(tok, pp.SyntheticTokenizer)
    assert tok.pad_token == tok.unk_token and tok.padding_side == "right" and tok.model_max_length == 512
    assert vsm.loc_token_idx == gold["loc_token_idx"] and tok.bos_token_id == gold["bos"] and tok.eos_token_id == gold["eos"]
    assert vsm.strict_template is True                     # a real checkpoint directory: template mismatches are not tolerated
    # [LOC] is ONE id wherever it stands; decode round trip drops specials like the reference's batch_decode call
    ids = tok("Sure, [LOC].", add_special_tokens=False).input_ids
    assert ids.count(vsm.loc_token_idx) == 1
    assert tok.batch_decode([tok("Please locate the dog.").input_ids], skip_special_tokens=True)[0] == "Please locate the dog."


@pytest.mark.parametrize("conv_type", ["llava_v1", "llava_llama_2"])
def test_prompt_ids_and_answer_boundary_match_the_reference(gold, conv_type):
    vsm, _ = _vsm(conv_type)
    P = vsm.cfg.n_img_tokens
    cases = [c for c in gold["cases"] if c["conv_type"] == conv_type]
    assert len(cases) >= 15
    for c in cases:
        q = c["question"]
        got_p = pp.tokenizer_image_token(pp.build_prompt(q, True, conv_type=conv_type), vsm.vsm_tokenizer)
        assert got_p == c["prompt_ids"], q
        ids, loc_pos, ver_pos, ver_tok = vsm._ids(q)
        assert ids.tolist() == c["full_ids"], q
        assert got_p.count(pp.IMAGE_TOKEN_INDEX) == 1 and got_p[0] == gold["bos"]
        # the answer tokens start exactly where the prompt ends; the verify tokens are the answer up to and including [LOC]
        n_p = len(c["prompt_ids"])
        assert c["full_ids"][:n_p] == c["prompt_ids"]
        loc_col = c["full_ids"].index(gold["loc_token_idx"])
        assert loc_pos == loc_col - 1 + (P - 1)
        assert ver_tok == c["full_ids"][n_p:loc_col + 1] and ver_tok[-1] == gold["loc_token_idx"]
        assert ver_pos == [col - 1 + (P - 1) for col in range(n_p, loc_col + 1)]
        # what greedy decoding must emit for the single-prefill shortcut to be exact: the ids of " Sure, [LOC]." up to [LOC]
        k = gold["answer_ids"].index(gold["loc_token_idx"])
        assert ver_tok == gold["answer_ids"][:k + 1]


def test_template_prefix_is_stable_under_sentencepiece_merges(gold):
    """`group_prompts` splits every locate prompt at the TEMPLATE's common prefix.  Under BPE the ids of '... locate the <name>'
    could in principle merge across the boundary; sentencepiece never merges across a space, so for names that start a new word the
    prefix is stable — checked on 17 names incl. byte-fallback, digits, punctuation and leading spaces — and any prompt that does not
    start with the template ids must be left to the plain path (the `ok` guard of _score_boxes_grouped)."""
    vsm, _ = _vsm()
    tpl = vsm._template_lcp()
    assert pp.IMAGE_TOKEN_INDEX in tpl
    # the split point is right after "... Please locate the": the shared ids end with the piece(s) of " the"
    tail = vsm.vsm_tokenizer.convert_ids_to_tokens(tpl[-3:])
    assert tail[-1] == "▁the", tail
    n_stable = 0
    for c in gold["cases"]:
        if c["conv_type"] != "llava_v1" or not c["question"].startswith("Please locate the "):
            continue
        ids = c["full_ids"]
        stable = ids[:len(tpl)] == tpl
        n_stable += stable
        assert stable, c["question"]          # every whitespace-separated name keeps the prefix
        assert len(ids) - len(tpl) >= 1
    assert n_stable >= 15
    # a question that is not the locate template shares less than the template prefix
    other = vsm._ids("Where is the dog?")[0].tolist()
    assert other[:len(tpl)] != tpl


@pytest.mark.parametrize("mode", [True, "always"])
def test_grouped_bookkeeping_with_the_real_tokenizer(mode):
    """The grouped entry point's index bookkeeping (records in the caller's order, template vs foreign prompts) on sentencepiece
    ids: a record of the stand-in engine is a function of (box, the prompt's token ids), so a wrong prefix split shows up."""
    vsm, eng = _vsm(group=mode)
    vsm.strict_template = False
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vsm.set_image(synthetic_image(400, 300, 1))
        names = ("kite", "person in a yellow coat", "7up can", "naïve sign")
        qs = [pp.LOCATE_QUESTION.format(n) for n in names] + ["What is shown here?"]
        boxes = [[0, 0, 400, 300], [0, 0, 200, 150], [200, 0, 200, 150], [0, 150, 200, 150]]
        pairs = [(0, 0), (1, 1), (0, 1), (2, 2), (0, 2), (1, 0), (0, 4), (3, 4), (2, 3), (0, 3)]
        out = vsm.inference_boxes([boxes[b] for b, _ in pairs], [qs[q] for _, q in pairs], mode="detection", upsample=False)
    want = []
    for b, q in pairs:
        x, y, w, h = boxes[b]
        want.append(_GroupedFakeEngine._rec([x, y, x + w, y + h], vsm._ids(qs[q])[0]))
    from vstar_amd.engine import VstarEngine
    ref = VstarEngine.unpack(np.stack(want), 0)["pred_boxes"].reshape(len(pairs), -1)
    got = np.stack([o[0].numpy().ravel() for o in out])
    assert np.array_equal(got, ref)
    # which prompts may take the grouped entry point: template prefix + at most 32 suffix tokens (this toy vocabulary spells
    # "ASSISTANT:" in ~10 pieces, so long names fall back to the plain path — with the same records, as asserted above)
    tpl = vsm._template_lcp()
    fits = [vsm._ids(q)[0].tolist()[:len(tpl)] == tpl and len(vsm._ids(q)[0]) - len(tpl) <= 32 for q in qs]
    assert fits[0] and not fits[4]
    n_plain_pairs = sum(1 for _, q in pairs if not fits[q])
    if mode == "always":
        assert sum(eng.plain_calls) == n_plain_pairs
    else:
        assert sum(eng.plain_calls) >= n_plain_pairs
    assert eng.grouped_calls and all(T <= sum(fits) for _, T in eng.grouped_calls)
