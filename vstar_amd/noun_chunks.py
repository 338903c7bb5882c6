"""Noun-phrase extraction for the contextual-cue branch (reference: visual_search.py:54-112, spaCy en_core_web_sm).

`get_noun_chunker()` returns the spaCy-based extractor when spaCy and its English model are installed (same dependency
walk as the reference: NOUN/PRON heads, left children amod/compound/poss, right children relcl/prep, longest
non-overlapping spans).  spaCy is absent from the build image, so the default is a small rule-based fallback that strips
leading prepositions/determiners and splits coordinated phrases; it covers the phrasing the VSM is trained to emit
("... is most likely to appear on the table near the window") but is NOT a parser — pass your own `noun_chunker` to
`visual_search(...)` when spaCy is available.
"""
from __future__ import annotations

import re
from typing import Callable, List

_LEAD = ("on top of", "in front of", "next to", "close to", "on", "in", "at", "near", "by", "beside", "behind", "under",
         "above", "around", "inside", "within", "along", "against", "towards", "to", "of", "with")
_DET = ("the", "a", "an", "this", "that", "these", "those", "its", "their", "his", "her")


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
        if words and words[0].lower() in _DET:
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


def spacy_noun_chunks_factory() -> Callable[[str], List[str]]:
    import spacy
    nlp = spacy.load("en_core_web_sm")

    def span(token):
        lo = hi = token.i
        for c in token.children:
            a, b = span(c)
            lo, hi = min(lo, a), max(hi, b)
        return lo, hi

    def chunk_of(token):
        left = [c for c in token.children if c.i < token.i]
        right = [c for c in token.children if c.i >= token.i]
        start = end = token.i
        for c in reversed(left):
            if c.dep_ not in ("amod", "compound", "poss"):
                break
            start, _ = span(c)
        for c in right:
            if c.dep_ not in ("relcl", "prep"):
                break
            _, end = span(c)
        return start, end

    def extract(expression: str) -> List[str]:
        doc = nlp(expression)
        chunks = [chunk_of(t) for t in doc if t.pos_ in ("NOUN", "PRON")]
        chunks.sort(key=lambda c: c[1] - c[0], reverse=True)
        kept = []
        for c in chunks:
            if all(min(k[1], c[1]) - max(k[0], c[0]) < 0 for k in kept):
                kept.append(c)
        kept.sort(key=lambda c: c[0])
        return [doc[a:b + 1].text for a, b in kept]

    return extract


def get_noun_chunker() -> Callable[[str], List[str]]:
    try:
        return spacy_noun_chunks_factory()
    except Exception:
        return rule_based_noun_chunks
