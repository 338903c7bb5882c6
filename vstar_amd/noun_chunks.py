"""Noun-phrase extraction for the contextual-cue branch (reference: visual_search.py:54-112, spaCy en_core_web_sm).

Two layers, as in the reference:
  * the PARSER (spaCy `en_core_web_sm`, a third-party model: POS tags + dependency arcs) — used when it is installed;
  * the WALK over the parse (`chunks_from_parse`): for every NOUN/PRON token extend left over amod/compound/poss children and
    right over relcl/prep children (whole sub-trees), then keep the longest non-overlapping spans in text order.  This part is
    reference logic and is pinned to the reference's own functions on hand-annotated parses
    (oracle/gen_noun_chunk_golden.py -> tests/golden/noun_chunks.json, tests/test_noun_chunks.py).
spaCy is absent from the build image, so `get_noun_chunker()` falls back — with a warning — to a rule-based chunker that
strips leading prepositions/determiners and splits coordinations.  It is not a parser; on the phrasing the VSM emits ("... is
most likely to appear on the wooden table near the window") it reproduces the walk's decision (one chunk -> that chunk, else
"region <phrase>", visual_search.py:437-440), which the same fixtures check.  Pass `noun_chunker=` to visual_search() to use
another parser.
"""
from __future__ import annotations

import re
import warnings
from typing import Callable, List, Sequence, Tuple

# ---------------- the walk (visual_search.py:54-112) over any parse exposing .i, .pos_, .dep_, .children ----------------


def _subtree_span(token) -> Tuple[int, int]:
    lo = hi = token.i
    for c in token.children:
        a, b = _subtree_span(c)
        lo, hi = min(lo, a), max(hi, b)
    return lo, hi


def _chunk_of(token) -> Tuple[int, int]:
    left = [c for c in token.children if c.i < token.i]
    right = [c for c in token.children if c.i >= token.i]
    start = end = token.i
    for c in reversed(left):
        if c.dep_ not in ("amod", "compound", "poss"):
            break
        start, _ = _subtree_span(c)
    for c in right:
        if c.dep_ not in ("relcl", "prep"):
            break
        _, end = _subtree_span(c)
    return start, end


def chunks_from_parse(doc: Sequence) -> List[Tuple[int, int]]:
    """Inclusive token spans of the noun chunks of a parsed sentence, in text order."""
    chunks = [_chunk_of(t) for t in doc if t.pos_ in ("NOUN", "PRON")]
    chunks.sort(key=lambda c: c[1] - c[0], reverse=True)          # stable, like the reference's sorted()
    kept: List[Tuple[int, int]] = []
    for c in chunks:
        if all(min(k[1], c[1]) - max(k[0], c[0]) < 0 for k in kept):
            kept.append(c)
    kept.sort(key=lambda c: c[0])
    return kept


def spacy_noun_chunks_factory() -> Callable[[str], List[str]]:
    import spacy
    nlp = spacy.load("en_core_web_sm")

    def extract(expression: str) -> List[str]:
        doc = nlp(expression)
        return [doc[a:b + 1].text for a, b in chunks_from_parse(doc)]

    return extract


# ---------------- rule-based fallback (no parser) ----------------
# single-word prepositions / adverbial leads that spaCy attaches ABOVE the noun (so the walk never includes them); multi-word
# forms whose middle word is itself a noun ("on top of X", "in front of X") are deliberately absent: the walk yields
# "top of X" / "front of X" there, and stripping just "on" / "in" reproduces that
_LEAD = ("next to", "close to", "on", "in", "at", "near", "by", "beside", "behind", "under", "above", "around", "inside", "within",
         "along", "against", "towards", "toward", "to", "of", "with", "over", "below", "beneath", "across", "between", "among",
         "onto", "into", "from", "atop", "underneath", "outside")
_DET = ("the", "a", "an", "this", "that", "these", "those", "some", "any", "each", "every", "another")     # dep_ = det
_NUM = ("one", "two", "three", "four", "five", "six", "seven", "eight", "nine", "ten")                    # dep_ = nummod


def _strip(words: List[str]) -> List[str]:
    changed = True
    while words and changed:
        changed = False
        low = " ".join(words).lower()
        for lead in _LEAD:
            if low.startswith(lead + " "):
                words = words[len(lead.split()):]
                changed = True
                break
        if words and (words[0].lower() in _DET or words[0].lower() in _NUM or words[0].isdigit()):
            words = words[1:]
            changed = True
    return words


def rule_based_noun_chunks(expression: str) -> List[str]:
    parts = re.split(r",|;|\band\b|\bor\b", expression)
    out = []
    for p in parts:
        w = _strip(p.strip().strip(".").split())
        if w:
            out.append(" ".join(w))
    return out


_warned = False


def get_noun_chunker() -> Callable[[str], List[str]]:
    """spaCy-based extractor when spaCy + en_core_web_sm are importable; otherwise the rule-based fallback, with ONE warning:
    the number of chunks decides between the 'noun_chunks[0]' and 'region {phrase}' prompt forms, so the user must know when
    the contextual-cue branch is not running the reference's parser."""
    global _warned
    try:
        return spacy_noun_chunks_factory()
    except Exception as exc:     # ImportError (no spaCy) or OSError (model not installed)
        if not _warned:
            warnings.warn("spaCy / en_core_web_sm unavailable (%s: %s): the contextual-cue branch uses the rule-based noun-chunk "
                          "fallback (vstar_amd/noun_chunks.py), which is not the reference's parser" % (type(exc).__name__, exc))
            _warned = True
        return rule_based_noun_chunks
