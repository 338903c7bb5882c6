"""Pins the noun-chunk WALK to the reference (visual_search.py:54-112): runs the reference's own `extract_noun_chunks`
(imported from /root/reference/visual_search.py through oracle.search_oracle.load_reference_search) with its `nlp` replaced by
a table of HAND-ANNOTATED parses (spaCy and en_core_web_sm are absent from this image; the parses follow en_core_web_sm's
conventions: det/amod/compound/poss/nummod on the left, prep->pobj, relcl, cc/conj) and records the chunks it returns.

TEST INFRASTRUCTURE.  Run:  python -m oracle.gen_noun_chunk_golden   -> tests/golden/noun_chunks.json
The sentences are the location phrases the contextual-cue branch feeds it (visual_search.py:430-437: the VQA answer after
"most likely to appear", without the final period)."""
from __future__ import annotations

import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "noun_chunks.json")

# sentence -> [(text, POS, dep, head index)] ; head == own index marks the root
PARSES = {
    "on the wooden table near the window": [
        ("on", "ADP", "ROOT", 0), ("the", "DET", "det", 3), ("wooden", "ADJ", "amod", 3), ("table", "NOUN", "pobj", 0),
        ("near", "ADP", "prep", 3), ("the", "DET", "det", 6), ("window", "NOUN", "pobj", 4)],
    "in the sky": [("in", "ADP", "ROOT", 0), ("the", "DET", "det", 2), ("sky", "NOUN", "pobj", 0)],
    "on the kitchen counter": [
        ("on", "ADP", "ROOT", 0), ("the", "DET", "det", 3), ("kitchen", "NOUN", "compound", 3), ("counter", "NOUN", "pobj", 0)],
    "on the table and the chair": [
        ("on", "ADP", "ROOT", 0), ("the", "DET", "det", 2), ("table", "NOUN", "pobj", 0), ("and", "CCONJ", "cc", 2),
        ("the", "DET", "det", 5), ("chair", "NOUN", "conj", 2)],
    "near the person who is holding the umbrella": [
        ("near", "ADP", "ROOT", 0), ("the", "DET", "det", 2), ("person", "NOUN", "pobj", 0), ("who", "PRON", "nsubj", 5),
        ("is", "AUX", "aux", 5), ("holding", "VERB", "relcl", 2), ("the", "DET", "det", 7), ("umbrella", "NOUN", "dobj", 5)],
    "on the left side of the street": [
        ("on", "ADP", "ROOT", 0), ("the", "DET", "det", 3), ("left", "ADJ", "amod", 3), ("side", "NOUN", "pobj", 0),
        ("of", "ADP", "prep", 3), ("the", "DET", "det", 6), ("street", "NOUN", "pobj", 4)],
    "on a shelf in the living room": [
        ("on", "ADP", "ROOT", 0), ("a", "DET", "det", 2), ("shelf", "NOUN", "pobj", 0), ("in", "ADP", "prep", 2),
        ("the", "DET", "det", 6), ("living", "NOUN", "compound", 6), ("room", "NOUN", "pobj", 3)],
    "beside the red car or behind the tree": [
        ("beside", "ADP", "ROOT", 0), ("the", "DET", "det", 3), ("red", "ADJ", "amod", 3), ("car", "NOUN", "pobj", 0),
        ("or", "CCONJ", "cc", 0), ("behind", "ADP", "conj", 0), ("the", "DET", "det", 7), ("tree", "NOUN", "pobj", 5)],
    "next to the large brown dog": [
        ("next", "ADV", "ROOT", 0), ("to", "ADP", "prep", 0), ("the", "DET", "det", 5), ("large", "ADJ", "amod", 5),
        ("brown", "ADJ", "amod", 5), ("dog", "NOUN", "pobj", 1)],
    "on top of the refrigerator": [
        ("on", "ADP", "ROOT", 0), ("top", "NOUN", "pobj", 0), ("of", "ADP", "prep", 1), ("the", "DET", "det", 4),
        ("refrigerator", "NOUN", "pobj", 2)],
    "in front of the building": [
        ("in", "ADP", "ROOT", 0), ("front", "NOUN", "pobj", 0), ("of", "ADP", "prep", 1), ("the", "DET", "det", 4),
        ("building", "NOUN", "pobj", 2)],
    "on its back": [("on", "ADP", "ROOT", 0), ("its", "PRON", "poss", 2), ("back", "NOUN", "pobj", 0)],
    "around two chairs": [("around", "ADP", "ROOT", 0), ("two", "NUM", "nummod", 2), ("chairs", "NOUN", "pobj", 0)],
    "in the water": [("in", "ADP", "ROOT", 0), ("the", "DET", "det", 2), ("water", "NOUN", "pobj", 0)],
    "on the road, the sidewalk, and the grass": [
        ("on", "ADP", "ROOT", 0), ("the", "DET", "det", 2), ("road", "NOUN", "pobj", 0), (",", "PUNCT", "punct", 2),
        ("the", "DET", "det", 5), ("sidewalk", "NOUN", "conj", 2), (",", "PUNCT", "punct", 5), ("and", "CCONJ", "cc", 5),
        ("the", "DET", "det", 9), ("grass", "NOUN", "conj", 5)],
    "it": [("it", "PRON", "ROOT", 0)],
}


class Tok:
    def __init__(self, i, text, pos, dep):
        self.i, self.text, self.pos_, self.dep_, self.children = i, text, pos, dep, []


class Span:
    def __init__(self, toks):
        self.toks = toks

    @property
    def text(self):          # spaCy keeps the source spacing: no blank before punctuation
        out = ""
        for t in self.toks:
            out += t.text if (not out or t.text in ",.;") else " " + t.text
        return out


class Doc(list):
    def __getitem__(self, k):
        return Span(list.__getitem__(self, k)) if isinstance(k, slice) else list.__getitem__(self, k)


def make_doc(parse) -> Doc:
    toks = [Tok(i, w, pos, dep) for i, (w, pos, dep, _) in enumerate(parse)]
    for i, (_, _, _, head) in enumerate(parse):
        if head != i:
            toks[head].children.append(toks[i])
    for t in toks:
        t.children.sort(key=lambda c: c.i)
    return Doc(toks)


def main():
    from oracle.search_oracle import load_reference_search
    ref = load_reference_search()
    ref.nlp = lambda text: make_doc(PARSES[text])          # the reference's module-level parser (visual_search.py:23)
    out = []
    for sent, parse in PARSES.items():
        chunks = ref.extract_noun_chunks(sent)
        # what visual_search.py:437-440 does with them
        phrase = chunks[0] if len(chunks) == 1 else "region {}".format(sent)
        out.append({"sentence": sent, "parse": [list(p) for p in parse], "chunks": chunks, "phrase": phrase})
        print(f"{sent!r:55s} -> {chunks}")
    json.dump(out, open(OUT, "w"), indent=1)
    print("->", OUT)


if __name__ == "__main__":
    main()
