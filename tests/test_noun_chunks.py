"""Noun-chunk extraction of the contextual-cue branch (reference: visual_search.py:54-112).

tests/golden/noun_chunks.json was recorded by running the REFERENCE's own extract_noun_chunks on hand-annotated dependency parses
(oracle/gen_noun_chunk_golden.py; spaCy itself is not in this image).  Checked here:
  * `chunks_from_parse` (the product's restatement of the walk, used with real spaCy parses) returns the same chunks on the same
    parses;
  * the rule-based fallback reaches the same DECISION visual_search.py:437-440 takes (one chunk -> that chunk, else "region ...")
    and, when there is one chunk, the same text;
  * falling back is announced with a warning."""
import json
import os
import warnings

import pytest

from vstar_amd import noun_chunks as nc

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "noun_chunks.json")))


class Tok:
    def __init__(self, i, text, pos, dep):
        self.i, self.text, self.pos_, self.dep_, self.children = i, text, pos, dep, []


def make_doc(parse):
    toks = [Tok(i, w, pos, dep) for i, (w, pos, dep, _) in enumerate(parse)]
    for i, (_, _, _, head) in enumerate(parse):
        if head != i:
            toks[head].children.append(toks[i])
    return toks


def text_of(toks, a, b):
    out = ""
    for t in toks[a:b + 1]:
        out += t.text if (not out or t.text in ",.;") else " " + t.text
    return out


@pytest.mark.parametrize("g", GOLD, ids=[g["sentence"][:30] for g in GOLD])
def test_walk_matches_reference_on_the_same_parse(g):
    doc = make_doc(g["parse"])
    spans = nc.chunks_from_parse(doc)
    assert [text_of(doc, a, b) for a, b in spans] == g["chunks"]


@pytest.mark.parametrize("g", GOLD, ids=[g["sentence"][:30] for g in GOLD])
def test_rule_based_fallback_takes_the_references_decision(g):
    chunks = nc.rule_based_noun_chunks(g["sentence"])
    phrase = chunks[0] if len(chunks) == 1 else "region {}".format(g["sentence"])
    assert phrase == g["phrase"], (chunks, g["chunks"])
    if len(g["chunks"]) != 1:
        assert len(chunks) != 1


def test_fallback_is_announced_once():
    nc._warned = False
    try:
        import spacy  # noqa: F401
        pytest.skip("spaCy is installed here: no fallback")
    except ImportError:
        pass
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        f = nc.get_noun_chunker()
        nc.get_noun_chunker()
    assert f is nc.rule_based_noun_chunks
    assert len([x for x in w if "rule-based noun-chunk fallback" in str(x.message)]) == 1


def test_reference_cue_sentence_end_to_end():
    """The string handling around the chunker in the cue branch (visual_search.py:430-440) on a typical VQA answer."""
    target = "mug"
    vqa = "The mug is most likely to appear on the wooden table near the window."
    phrase = vqa.split("most likely to appear")[-1].strip()
    if phrase.endswith("."):
        phrase = phrase[:-1]
    phrase = phrase.split(target)[-1]
    chunks = nc.rule_based_noun_chunks(phrase)
    assert chunks == ["wooden table near the window"]
