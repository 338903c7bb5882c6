"""The V* crop scheduler (visual_search.py:227-283, 378-516) re-driven in batches.

`visual_search(...)` keeps the reference's signature and return tuple.  The reference recurses one crop at a time
(`visual_search_queue`, batch 1); here the same best-first order is an explicit loop and the VSM
is asked for MANY nodes per call: every node's bbox is a pure function of its parent's bbox (`get_sub_patches`), so the
children of the current node and of the best queue entries can be scored speculatively in the same engine batch.
Results are cached by bbox and only CONSUMED when the loop reaches that node, in exactly the reference's pop order
(same `queue.PriorityQueue`, same `Prioritize` ordering, same float32 numpy reductions), so paths / boxes / answers are
identical to a batch-1 run and the speculation only costs bounded extra crops.
"""
from __future__ import annotations

import copy
import functools
from queue import PriorityQueue
from typing import Sequence,  Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from .preprocess import CUE_QUESTION, LOCATE_QUESTION


# ---------------- geometry / score helpers (visual_search.py:227-283) ----------------
# NOTE: refine_bbox .. Prioritize below restate the reference's small decision helpers statement by statement ON PURPOSE: their
# float32 / Python-int semantics (int(w // n) vs int(h / n), numpy pairwise sums, `>=` vs `>`, heapq tie order through a
# priority-only comparison) decide which crop is visited next, and the scheduler goldens (tests/golden/search_paths.json, recorded
# from the reference's own visual_search.py) require them bit for bit.  Everything else in this file — the generator form of the
# loop, speculation, the lock-step / streaming drivers, the lazy exact tie-break — is this repository's own design.
def refine_bbox(bbox, image_width, image_height):
    bbox[0] = max(0, bbox[0])
    bbox[1] = max(0, bbox[1])
    bbox[2] = min(bbox[2], image_width - bbox[0])
    bbox[3] = min(bbox[3], image_height - bbox[1])
    return bbox


def split_4subpatches(current_patch_bbox) -> Tuple[int, int]:
    hw_ratio = current_patch_bbox[3] / current_patch_bbox[2]
    if hw_ratio >= 2:
        return 1, 4
    if hw_ratio <= 0.5:
        return 4, 1
    return 2, 2


def get_sub_patches(current_patch_bbox, num_of_width_patches: int, num_of_height_patches: int):
    width_stride = int(current_patch_bbox[2] // num_of_width_patches)
    height_stride = int(current_patch_bbox[3] / num_of_height_patches)
    sub_patches = []
    for j in range(num_of_height_patches):
        for i in range(num_of_width_patches):
            w = current_patch_bbox[2] - i * width_stride if i == num_of_width_patches - 1 else width_stride
            h = current_patch_bbox[3] - j * height_stride if j == num_of_height_patches - 1 else height_stride
            sub_patches.append([current_patch_bbox[0] + i * width_stride, current_patch_bbox[1] + j * height_stride, w, h])
    return sub_patches, width_stride, height_stride


def get_subpatch_scores(score_heatmap: np.ndarray, current_patch_bbox, sub_patches) -> List:
    area = current_patch_bbox[2] * current_patch_bbox[3]
    total_sum = (score_heatmap / area).sum()
    sub_scores = []
    for sp in sub_patches:
        x, y, w, h = sp[0] - current_patch_bbox[0], sp[1] - current_patch_bbox[1], sp[2], sp[3]
        score = (score_heatmap[y:y + h, x:x + w] / area).sum()
        if total_sum > 0:
            score /= total_sum
        else:
            score *= 0
        sub_scores.append(score)
    return sub_scores


def normalize_score(score_heatmap: torch.Tensor) -> torch.Tensor:
    max_score = score_heatmap.max()
    min_score = score_heatmap.min()
    if max_score != min_score:
        return (score_heatmap - min_score) / (max_score - min_score)
    return score_heatmap * 0


def iou(bbox1, bbox2):
    x1 = max(bbox1[0], bbox2[0])
    y1 = max(bbox1[1], bbox2[1])
    x2 = min(bbox1[0] + bbox1[2], bbox2[0] + bbox2[2])
    y2 = min(bbox1[1] + bbox1[3], bbox2[1] + bbox2[3])
    inter_area = max(0, x2 - x1) * max(0, y2 - y1)
    return inter_area / (bbox1[2] * bbox1[3] + bbox2[2] * bbox2[3] - inter_area)


@functools.total_ordering
class Prioritize:
    """Heap entry comparing ONLY the priority (visual_search.py:378-389): ties resolve by heapq sift order."""

    def __init__(self, priority, item):
        self.priority = priority
        self.item = item

    def __eq__(self, other):
        return self.priority == other.priority

    def __lt__(self, other):
        return self.priority < other.priority


class LazyExactPrioritize(Prioritize):
    """Queue entry of the device-reductions path.  `priority` is the negated child score from the fp64 rectangle-sum algebra,
    which agrees with the reference's float32 numpy reductions to ~1e-5 relative — enough to order entries unless two of them
    are nearly tied.  In that case (and only then) both entries compute their EXACT reference score — the float32 path on
    materialised heat maps, `exact()` — and are ordered by it, so the pop order is the reference's in every case."""
    REL_TOL = 1e-4          # >= 10 x the measured disagreement between the two arithmetic paths
    n_exact = 0             # diagnostics: how often the exact path was needed

    def __init__(self, priority, item, exact, exact_zero: bool = False):
        super().__init__(priority, item)
        self._exact_fn, self._exact = exact, None
        # the score is 0 in the REFERENCE's float32 arithmetic too, not just nearly: every contribution was a sum over pixels that
        # are all exactly zero (clamped heat map, min 0) or one of the reference's own "return 0" cases.  Two such entries are an
        # exact tie in both arithmetics: no materialised heat map is needed to order them.  (A trained segmentation head is negative
        # on the background, so clamped heat maps are zero over most children: with trained-like weights 160 of 336 pushes of the
        # config-2 search leg took the exact path, 18 ms each, until round 4.)
        self.exact_zero = bool(exact_zero)

    def exact(self):
        if self._exact is None:
            self._exact = -self._exact_fn()
            LazyExactPrioritize.n_exact += 1
        return self._exact

    def _near(self, other):
        if self.exact_zero and getattr(other, "exact_zero", False):
            return False
        a, b = float(self.priority), float(other.priority)
        return abs(a - b) <= self.REL_TOL * max(abs(a), abs(b)) + 1e-12

    def __eq__(self, other):
        if isinstance(other, LazyExactPrioritize) and self._near(other):
            return self.exact() == other.exact()
        return self.priority == other.priority

    def __lt__(self, other):
        if isinstance(other, LazyExactPrioritize) and self._near(other):
            return self.exact() < other.exact()
        return self.priority < other.priority


def smallest_size_for(image_width: int, image_height: int, minimum_size_scale: float = 4.0, minimum_size: int = 224) -> int:
    """visual_search.py:545."""
    return max(int(np.ceil(min(image_width, image_height) / minimum_size_scale)), minimum_size)


def _crop(image, bbox):
    return image.crop((int(bbox[0]), int(bbox[1]), int(bbox[0] + bbox[2]), int(bbox[1] + bbox[3])))


class _NodeScorer:
    """Caches detection-mode VSM results per bbox and fills the cache in speculative batches."""

    def __init__(self, vsm, image, question: str, smallest_size: int, batch_size: Optional[int], speculate: bool,
                 gpu_preprocess: bool = True, device_reductions: Optional[bool] = None, upload_image: bool = True, slot: int = 0):
        self.vsm, self.image, self.question = vsm, image, question
        self.slot = slot                    # image slot of this search's image (cross-image batches: visual_search_stream)
        self.stream = False                 # True: `missing` asks for the node alone; the stream driver adds speculation itself
        self._last = None                   # (bbox, queue) of the last request, for `candidates`
        # on-device heat-map statistics whenever the VSM offers them; None = automatic, False = host reductions
        can = bool(getattr(vsm, "supports_device_reductions", hasattr(vsm, "heatmap_stats"))) and hasattr(vsm, "inference_batch")
        self.device_reductions = can if device_reductions is None else (bool(device_reductions) and can)
        self.smallest_size = smallest_size
        self.batched = hasattr(vsm, "inference_batch")
        # device-side crop/resize when the VSM offers it: the full image is uploaded once, crops travel as boxes
        self.on_device = gpu_preprocess and bool(getattr(vsm, "supports_gpu_preprocess", False))
        if self.on_device and upload_image:       # (visual_search_many: the targets of one image share ONE upload)
            vsm.set_image(image) if slot == 0 else vsm.set_image(image, slot)
        world = vsm._dist()[0] if hasattr(vsm, "_dist") else 1    # one engine batch per rank and step
        self.batch_size = batch_size or (getattr(getattr(vsm, "cfg", None), "max_batch", 1) * world if self.batched else 1)
        self.speculate = speculate and self.batched and self.batch_size > 1
        self.cache: Dict[Tuple, Tuple] = {}
        self.n_scored = 0
        self.n_batches = 0

    def _children(self, bbox):
        if min(bbox[2], bbox[3]) <= self.smallest_size:
            return []
        return get_sub_patches(bbox, *split_4subpatches(bbox))[0]

    def plan(self, bbox, queue: PriorityQueue) -> List[list]:
        """The crops one engine step would score for this node: the node itself plus, when speculating, breadth-first its
        children, the queue's entries best-first and their children, ... up to one batch."""
        key = tuple(bbox)
        todo = [list(bbox)]
        if self.speculate:
            frontier = self._children(bbox)
            # ordered by the float priority only: comparing the entries themselves would make LazyExactPrioritize materialise
            # full-resolution heat maps for every near-tied pair just to rank SPECULATION candidates
            frontier += [e.item["bbox"] for e in sorted(queue.queue, key=lambda e: float(e.priority))]
            seen = {key}
            while frontier and len(todo) < self.batch_size:
                nxt = []
                for b in frontier:
                    k = tuple(b)
                    if k in seen:
                        continue
                    seen.add(k)
                    if k not in self.cache:
                        todo.append(list(b))
                        if len(todo) >= self.batch_size:
                            break
                    nxt += self._children(b)
                frontier = nxt
        return todo

    def store(self, todo: List[list], res: List, sizes: List[Tuple[int, int]]) -> None:
        for b, sz, r in zip(todo, sizes, res):
            self.cache[tuple(b)] = [r, sz, not self.batched]     # [result, (w, h), heatmap already full-res?]
        self.n_scored += len(todo)

    def missing(self, bbox, queue: PriorityQueue) -> Optional[List[list]]:
        """None when `bbox` is cached, else the crops one engine step should score now (`plan`)."""
        if tuple(bbox) in self.cache:
            return None
        if self.stream:
            self._last = (list(bbox), queue)
            return [list(bbox)]
        return self.plan(bbox, queue)

    def candidates(self, prior: "SpeculationPolicy") -> List[Tuple[float, list]]:
        """Speculation candidates for the node this search is waiting on: (visit probability, bbox), uncached boxes only.
        Boxes are pure functions of committed boxes: the node's children, the queue's entries best-first, and THEIR children."""
        if self._last is None:
            return []
        bbox, queue = self._last
        out, seen = [], {tuple(bbox)}

        def add(p, b):
            k = tuple(b)
            if k not in seen:
                seen.add(k)
                if k not in self.cache:
                    out.append((p, list(b)))

        for c in self._children(bbox):
            add(prior.p_child, c)
        for rank, e in enumerate(sorted(queue.queue, key=lambda e: float(e.priority))[:prior.max_queue_rank]):
            pq = prior.p_queue * prior.queue_decay ** rank
            add(pq, e.item["bbox"])
            for c in self._children(e.item["bbox"]):
                add(pq * prior.p_child * getattr(prior, "queue_child_factor", 1.0), c)
        return out

    def score(self, todo: List[list]) -> List:
        """One engine step for the crops `todo` (this scorer's question)."""
        # defer_mismatch: a crop whose template check fails is only re-decoded (and can only raise the reference's
        # IndexError) when the best-first order really visits it — speculative crops the reference never evaluates cannot
        # abort a search the reference would complete
        kw = {"defer_mismatch": True} if getattr(self.vsm, "supports_deferred_mismatch", False) else {}
        if self.on_device:
            if self.slot:
                kw["slots"] = [self.slot] * len(todo)
            return self.vsm.inference_boxes(todo, self.question, mode="detection", upsample=False, **kw)
        crops = [_crop(self.image, b) for b in todo]
        if self.batched:
            return self.vsm.inference_batch(crops, self.question, mode="detection", upsample=False, **kw)
        return [self.vsm.inference(copy.deepcopy(c), self.question, mode="detection") for c in crops]

    def accept(self, todo: List[list], res: List) -> None:
        """Results of one engine step for `todo` (from `score`, or from a multi-target step of visual_search_many)."""
        if self.on_device:
            sizes = [(int(b[0] + b[2]) - int(b[0]), int(b[1] + b[3]) - int(b[1])) for b in todo]
        else:
            sizes = [_crop(self.image, b).size for b in todo]
        self.store(todo, res, sizes)
        self.n_batches += 1

    def take(self, bbox):
        key = tuple(bbox)
        if hasattr(self.cache[key][0], "resolve"):              # DeferredMismatch: the node is being consumed NOW
            self.cache[key][0] = self.cache[key][0].resolve()
        (boxes, scores, heat), (w, h), full = self.cache[key]
        if not full and self.device_reductions:
            return boxes, scores, heat                          # 192x192 low-res logits; statistics are taken on the GPU
        if not full:
            heat = self.vsm.upsample_heatmap(heat, h, w)        # full-resolution heatmap only for COMMITTED nodes
            self.cache[key] = [(boxes, scores, heat), (w, h), True]
        return boxes, scores, heat

    def get(self, bbox, queue: PriorityQueue):
        todo = self.missing(bbox, queue)
        if todo is not None:
            self.accept(todo, self.score(todo))
        return self.take(bbox)


def visual_search(vsm, image, target_object_name, target_bbox, smallest_size, confidence_high=0.5, confidence_low=0.3,
                  target_cue_threshold=6.0, target_cue_threshold_decay=0.7, target_cue_threshold_minimum=3.0,
                  visualize=False, save_path=None, *, batch_size: Optional[int] = None, speculate: bool = True,
                  noun_chunker: Optional[Callable[[str], List[str]]] = None, stats: Optional[dict] = None,
                  gpu_preprocess: bool = True, device_reductions: Optional[bool] = None,
                  _scorer: Optional["_NodeScorer"] = None):
    """Same contract as the reference's visual_search (visual_search.py:484-516): returns
    (final_step, path_length, search_successful, all_valid_boxes)."""
    steps = _visual_search_steps(vsm, image, target_object_name, target_bbox, smallest_size, confidence_high, confidence_low,
                                 target_cue_threshold, target_cue_threshold_decay, target_cue_threshold_minimum, visualize,
                                 save_path, batch_size=batch_size, speculate=speculate, noun_chunker=noun_chunker, stats=stats,
                                 gpu_preprocess=gpu_preprocess, device_reductions=device_reductions, _scorer=_scorer)
    try:
        req = next(steps)
        while True:
            req = steps.send(req[1].score(req[2]) if req[0] == "score" else _serve_stats(vsm, req[1]))
    except StopIteration as done:
        return done.value


def _serve_stats(vsm, reqs):
    """Heat-map statistics for [(low_res, h, w, rects), ...]: one engine call when the VSM offers the batched form."""
    if hasattr(vsm, "heatmap_stats_batch"):
        return vsm.heatmap_stats_batch(reqs)
    return [vsm.heatmap_stats(m, h, w, r) for m, h, w, r in reqs]


def _visual_search_steps(vsm, image, target_object_name, target_bbox, smallest_size, confidence_high=0.5, confidence_low=0.3,
                         target_cue_threshold=6.0, target_cue_threshold_decay=0.7, target_cue_threshold_minimum=3.0,
                         visualize=False, save_path=None, *, batch_size: Optional[int] = None, speculate: bool = True,
                         noun_chunker: Optional[Callable[[str], List[str]]] = None, stats: Optional[dict] = None,
                         gpu_preprocess: bool = True, device_reductions: Optional[bool] = None,
                         _scorer: Optional["_NodeScorer"] = None):
    """The search as a generator: whenever it needs crops scored it yields ("score", scorer, crops) and is sent the engine's
    results for them; whenever it needs heat-map statistics (device reductions) it yields ("stats", [(low_res, h, w, rects), ...])
    and is sent one statistics vector per item; its return value is visual_search's.  `visual_search` drives one such generator, `visual_search_many` several in lock
    step.  Everything else — the decision math of visual_search.py:390-516 — is unchanged."""
    if visualize:
        assert save_path is not None                         # visual_search.py:485-486
        device_reductions = False                            # the rendered heat maps need the full-resolution maps on the host
        if _scorer is not None:                              # a prebuilt scorer (stream / many drivers) must follow suit, or the
            _scorer.device_reductions = False                # step_k_heatmap.jpg files would silently be missing
    init_patch = {"bbox": [0, 0, image.width, image.height], "scale_level": 1, "score": None, "parent_index": -1}
    search_path = [init_patch]
    queue: PriorityQueue = PriorityQueue()
    question = LOCATE_QUESTION.format(target_object_name)
    scorer = _scorer or _NodeScorer(vsm, image, question, smallest_size, batch_size, speculate, gpu_preprocess,
                                    device_reductions)
    on_dev = scorer.device_reductions

    search_successful, all_valid_boxes = False, None
    current_patch = init_patch
    while True:
        bbox = current_patch["bbox"]
        level = current_patch["scale_level"]
        pw, ph = int(bbox[0] + bbox[2]) - int(bbox[0]), int(bbox[1] + bbox[3]) - int(bbox[1])
        todo = scorer.missing(bbox, queue)
        if todo is not None:
            scorer.accept(todo, (yield "score", scorer, todo))
        pred_bboxes, pred_logits, target_cue_heatmap = scorer.take(bbox)
        expand = True
        if len(pred_logits) > 0:
            top_index = pred_logits.view(-1).argmax()
            top_logit = pred_logits.view(-1).max()
            final_bbox = pred_bboxes[top_index].view(4)
            final_bbox = final_bbox * torch.Tensor([pw, ph, pw, ph])
            final_bbox[:2] -= final_bbox[2:] / 2
            if top_logit > confidence_high:
                search_path[-1]["detection_result"] = final_bbox
                if len(search_path) == 1:   # multiple instances are only returned for the whole image
                    all_valid_boxes = pred_bboxes[pred_logits.view(-1) > 0.5].view(-1, 4)
                    all_valid_boxes = all_valid_boxes * torch.Tensor([[pw, ph, pw, ph]])
                    all_valid_boxes[:, :2] -= all_valid_boxes[:, 2:] / 2
                search_successful = True
                break
            search_path[-1]["temp_detection_result"] = (top_logit, final_bbox)

        if min(bbox[2], bbox[3]) <= smallest_size:
            expand = False                                    # already the smallest unit: no children
        if expand and on_dev:
            # ---- the heat map never leaves the GPU.  One kernel returns min / max / sum and the sums over
            # the four child rectangles of clamp(bilinear(low_res)); min-max normalisation is applied algebraically:
            #   sum_rect (H - mn)/(mx - mn) = (sum_rect H - mn*|rect|)/(mx - mn).   fp64 accumulation (the reference sums
            # float32 with numpy's pairwise order; results agree to ~1e-7 relative, so only exact ties could reorder). ----
            basic_sub_patches, _, _ = get_sub_patches(bbox, *split_4subpatches(bbox))
            current_patch_index = len(search_path) - 1
            threshold = max(target_cue_threshold_minimum, target_cue_threshold * target_cue_threshold_decay ** (level - 1))
            rel = lambda owner: [[sp[0] - owner[0], sp[1] - owner[1], sp[2], sp[3]] for sp in basic_sub_patches]  # noqa: E731
            # this node's statistics and every ancestor's (over THIS node's child rectangles) in one request
            stat_reqs = [(target_cue_heatmap, bbox[3], bbox[2], rel(bbox))]
            anc = current_patch
            while anc["parent_index"] != -1:
                anc = search_path[anc["parent_index"]]
                tb = anc["bbox"]
                stat_reqs.append((anc["heat_stats"]["low_res"], tb[3], tb[2], rel(tb)))
            stat_res = yield "stats", stat_reqs
            st, anc_stats = stat_res[0], list(stat_res[1:])
            low = target_cue_heatmap
            if not st[1] > threshold:
                patch = _crop(image, bbox)
                vqa_results = vsm.inference(copy.deepcopy(patch), CUE_QUESTION.format(target_object_name), mode="vqa")
                phrase = vqa_results.split("most likely to appear")[-1].strip()
                if phrase.endswith("."):
                    phrase = phrase[:-1]
                phrase = phrase.split(target_object_name)[-1]
                if noun_chunker is None:
                    from .noun_chunks import get_noun_chunker
                    noun_chunker = get_noun_chunker()
                noun_chunks = noun_chunker(phrase)
                phrase = noun_chunks[0] if len(noun_chunks) == 1 else "region {}".format(phrase)
                low = vsm.inference_batch([patch], LOCATE_QUESTION.format(phrase), mode="segmentation", upsample=False)[0]
                st = (yield "stats", [(low, bbox[3], bbox[2], rel(bbox))])[0]
                search_path[current_patch_index]["context_cue"] = vqa_results + "#" + phrase
            current_patch["heat_stats"] = {"low_res": low, "min": st[0], "max": st[1], "sum": st[2]}

            def rect_scores(owner, stats):
                hs = owner["heat_stats"]
                mn, mx = hs["min"], hs["max"]
                nsub = len(basic_sub_patches)
                if not mx != mn:
                    return [0.0] * nsub, [True] * nsub
                area = owner["bbox"][2] * owner["bbox"][3]
                total = (hs["sum"] - mn * area) / (mx - mn)
                if not total > 0:
                    return [0.0] * nsub, [True] * nsub
                # a rectangle whose fp64 sum of the (non-negative) clamped map is exactly 0 holds only zeros: with min == 0 its
                # normalised float32 sum is exactly 0 in the reference as well
                zero = [mn == 0.0 and stats[3 + k] == 0.0 for k in range(nsub)]
                return [((stats[3 + k] - mn * sp[2] * sp[3]) / (mx - mn)) / total for k, sp in enumerate(basic_sub_patches)], zero

            basic_sub_scores = [0.0] * len(basic_sub_patches)
            sub_exact_zero = [True] * len(basic_sub_patches)
            tmp_patch, tmp_stats = current_patch, st
            while True:
                sc, zr = rect_scores(tmp_patch, tmp_stats)
                sub_exact_zero = [a and b for a, b in zip(sub_exact_zero, zr)]
                basic_sub_scores = [basic_sub_scores[i] + sc[i] / (4 ** tmp_patch["scale_level"]) for i in range(len(sc))]
                if tmp_patch["parent_index"] == -1:
                    break
                tmp_patch = search_path[tmp_patch["parent_index"]]
                tmp_stats = anc_stats.pop(0)

            def exact_scores(node=current_patch, subs=basic_sub_patches, memo={}):  # noqa: B006  (per-node memo on purpose)
                """The reference's own float32 arithmetic for THIS node's children (visual_search.py:445-462): materialise the
                normalised heat map of the node and of every ancestor and reduce with numpy.  Only evaluated on near-ties."""
                if "v" not in memo:
                    acc = [0] * len(subs)
                    t = node
                    while True:
                        if "final_heatmap" not in t:
                            tb2 = t["bbox"]
                            hm = vsm.upsample_heatmap(t["heat_stats"]["low_res"], tb2[3], tb2[2]).view(tb2[3], tb2[2], 1)
                            t["final_heatmap"] = normalize_score(hm).cpu().numpy()
                        part = get_subpatch_scores(t["final_heatmap"], t["bbox"], subs)
                        acc = [acc[i] + part[i] / (4 ** t["scale_level"]) for i in range(len(acc))]
                        if t["parent_index"] == -1:
                            break
                        t = search_path[t["parent_index"]]
                    memo["v"] = acc
                return memo["v"]

            for k, (sub_patch, sub_score) in enumerate(zip(basic_sub_patches, basic_sub_scores)):
                info = {"bbox": sub_patch, "scale_level": level + 1, "score": np.float32(sub_score),
                        "parent_index": current_patch_index}
                queue.put(LazyExactPrioritize(-info["score"], info, lambda k=k, f=exact_scores: f()[k],
                                              exact_zero=sub_exact_zero[k] and float(sub_score) == 0.0))
        elif expand:
            heat = target_cue_heatmap.view(bbox[3], bbox[2], 1)
            score_max = heat.max().item()
            threshold = max(target_cue_threshold_minimum, target_cue_threshold * target_cue_threshold_decay ** (level - 1))
            current_patch_index = len(search_path) - 1
            if score_max > threshold:
                final_heatmap = normalize_score(heat)
            else:
                # contextual-cue branch (visual_search.py:427-443): free-text VQA -> noun phrase -> segmentation
                patch = _crop(image, bbox)
                vqa_results = vsm.inference(copy.deepcopy(patch), CUE_QUESTION.format(target_object_name), mode="vqa")
                phrase = vqa_results.split("most likely to appear")[-1].strip()
                if phrase.endswith("."):
                    phrase = phrase[:-1]
                phrase = phrase.split(target_object_name)[-1]
                if noun_chunker is None:
                    from .noun_chunks import get_noun_chunker
                    noun_chunker = get_noun_chunker()     # spaCy if installed, else the rule-based fallback
                noun_chunks = noun_chunker(phrase)
                phrase = noun_chunks[0] if len(noun_chunks) == 1 else "region {}".format(phrase)
                ctx = vsm.inference(copy.deepcopy(patch), LOCATE_QUESTION.format(phrase), mode="segmentation")
                final_heatmap = normalize_score(ctx.view(bbox[3], bbox[2], 1))
                search_path[current_patch_index]["context_cue"] = vqa_results + "#" + phrase
            search_path[current_patch_index]["final_heatmap"] = final_heatmap.cpu().numpy()

            basic_sub_patches, _, _ = get_sub_patches(bbox, *split_4subpatches(bbox))
            tmp_patch = current_patch
            basic_sub_scores = [0] * len(basic_sub_patches)
            while True:   # accumulate every ancestor's heatmap mass, weighted 1/4^level (visual_search.py:451-462)
                tmp_sub_scores = get_subpatch_scores(tmp_patch["final_heatmap"], tmp_patch["bbox"], basic_sub_patches)
                basic_sub_scores = [basic_sub_scores[i] + tmp_sub_scores[i] / (4 ** tmp_patch["scale_level"])
                                    for i in range(len(basic_sub_scores))]
                if tmp_patch["parent_index"] == -1:
                    break
                tmp_patch = search_path[tmp_patch["parent_index"]]
            for sub_patch, sub_score in zip(basic_sub_patches, basic_sub_scores):
                info = {"bbox": sub_patch, "scale_level": level + 1, "score": sub_score, "parent_index": current_patch_index}
                queue.put(Prioritize(-info["score"], info))

        if queue.empty():
            break
        current_patch = queue.get().item
        search_path.append(current_patch)

    path_length = len(search_path)
    final_step = search_path[-1]
    if not search_successful:
        # no confident detection: fall back to the best temp detection seen on the path (visual_search.py:498-511)
        max_logit = 0
        final_step = None
        path_length = 0
        for i, step in enumerate(search_path):
            if "temp_detection_result" in step and step["temp_detection_result"][0] > max_logit:
                max_logit = step["temp_detection_result"][0]
                final_step = step
                path_length = i + 1
        final_step["detection_result"] = final_step["temp_detection_result"][1]
        if max_logit >= confidence_low:
            search_successful = True
    if stats is not None:
        stats.update(crops_scored=scorer.n_scored, engine_batches=scorer.n_batches, path_visited=len(search_path),
                     search_path=search_path)
    if visualize:                                            # visual_search.py:512-514
        from .visualize import visualize_search_path
        vis_path_length = path_length if search_successful else len(search_path)
        visualize_search_path(image, search_path, vis_path_length, target_bbox, target_object_name, save_path)
    return final_step, path_length, search_successful, all_valid_boxes


class SpeculationPolicy:
    """Cost-aware speculation for the lock-step / streaming drivers.

    The reference evaluates exactly the crops its best-first order visits (average path ~4.65 nodes).  A batched engine
    can score crops AHEAD of the order, but a speculative crop is only worth its cost if it is likely to be visited AND the batch
    it joins is not already efficient: a miss is pure waste (one crop's marginal engine time), a hit saves the search one serial
    engine step.  With the measured step-time table t(B) (ms for a B-crop step on one rank; defaults = this repository's MI355X
    measurements, profiles/r03_small_batch.json; `VSM.step_ms_table` overrides) a candidate with visit probability p joins a
    step that already holds B crops iff

            p * t(1) / n_live   >=   t(B + 1) - t(B)

    (expected serial time saved, shared among the n_live searches that advance in lock step — with many live searches a hit on
    one of them rarely shortens the schedule — against the marginal cost of one more crop), and never beyond `cap` crops.  With a
    window of concurrent searches that fills the batch with MUST crops the right-hand side is the full per-crop cost and nothing
    is speculated; a lone search (latency mode) speculates its children and the head of its queue while batches are small.
    The priors are visit FREQUENCIES measured by replaying best-first searches (tools/calibrate_speculation.py ->
    profiles/r04_speculation_priors.json) in the regime of the reference's reported mean path (~4.65 visited nodes per search,
    SURVEY §6): a child of the node being scored is visited later with probability 0.25, the r-th best queue entry with
    0.75 x 0.7^r, a child of that entry with 0.45 x the product (round 3 guessed 0.45 / 0.5 x 0.5^r / the bare product: children
    over-, the queue under-estimated).  With the MI355X step table a lone search therefore speculates the head of its queue but
    not the four children of the node in flight (expected saving 0.25 x t(1) = 4.2 ms < 6.7 ms marginal cost); `wasted_crop_frac`
    and bench.py's `search_latency` leg are the measured outcome.  The loop is closed on that outcome: `observe_speculation`
    compares the hits of every finished search with what the priors had promised and scales all probabilities by the ratio."""

    DEFAULT_STEP_MS = {1: 16.8, 2: 23.5, 4: 36.5, 8: 63.7, 16: 126.5, 32: 232.7}
    # (mean visited nodes per search, p_child, p_queue, queue_decay): profiles/r04_speculation_priors.json.  How often a candidate
    # is visited depends on how soon searches end, so the priors follow the OBSERVED mean path length of the searches this policy
    # has seen finish (`observe`), starting from the reference's regime
    REGIMES = ((1.77, 0.109, 0.403, 0.32), (2.64, 0.170, 0.595, 0.55), (3.69, 0.219, 0.693, 0.68), (4.41, 0.250, 0.747, 0.75),
               (6.64, 0.338, 0.835, 0.83), (11.91, 0.481, 0.929, 0.93))
    PRIOR_NODES, PRIOR_WEIGHT = 4.65, 4.0
    CAL_WEIGHT = 6.0

    def __init__(self, step_ms: Optional[Dict[int, float]] = None, cap: int = 32, world: int = 1, p_child: float = 0.25,
                 p_queue: float = 0.75, queue_decay: float = 0.7, max_queue_rank: int = 4, enabled: bool = True,
                 queue_child_factor: float = 0.45):
        self.table = dict(sorted((step_ms or self.DEFAULT_STEP_MS).items()))
        self.cap, self.world, self.enabled = int(cap), max(int(world), 1), enabled
        self.p_child, self.p_queue, self.queue_decay, self.max_queue_rank = p_child, p_queue, queue_decay, max_queue_rank
        self.queue_child_factor = queue_child_factor
        self.adaptive = (p_child, p_queue, queue_decay) == (0.25, 0.75, 0.7)      # explicit priors are kept as given
        self._nodes_sum, self._nodes_n = 0.0, 0
        # closed loop on the hit rate: `calibration` = observed hits / predicted hits over the searches that have finished
        # (`observe_speculation`); every candidate's probability is multiplied by it before the cost test.  Starts at 1 with the
        # weight of CAL_WEIGHT predicted hits, clamped to [0.25, 1.5]: priors replayed on a stand-in cannot know a model's
        # detection confidence or how its heat maps rank the queue (first measurement on the engine: 44 % hits where the table
        # promised 64 %)
        self.calibration, self._hits, self._pred = 1.0, 0.0, 0.0
        self.last_selected_p: List[float] = []

    def observe(self, nodes_visited: int) -> None:
        """A search ended after visiting `nodes_visited` nodes: move the priors to the regime the searches are really in.  (Under
        crop sharding every rank sees the same searches end in the same order, so the ranks' policies stay identical.)"""
        if not self.adaptive:
            return
        self._nodes_sum += float(nodes_visited)
        self._nodes_n += 1
        m = (self.PRIOR_NODES * self.PRIOR_WEIGHT + self._nodes_sum) / (self.PRIOR_WEIGHT + self._nodes_n)
        R = self.REGIMES
        if m <= R[0][0]:
            f = max(m - 1.0, 0.0) / (R[0][0] - 1.0)            # a search that always ends at its root visits nothing else
            self.p_child, self.p_queue, self.queue_decay = R[0][1] * f, R[0][2] * f, R[0][3]
            return
        for lo, hi in zip(R, R[1:]):
            if m <= hi[0]:
                t = (m - lo[0]) / (hi[0] - lo[0])
                self.p_child, self.p_queue, self.queue_decay = (lo[k] + t * (hi[k] - lo[k]) for k in (1, 2, 3))
                return
        self.p_child, self.p_queue, self.queue_decay = R[-1][1:]

    def observe_speculation(self, predicted_hits: float, hits: int) -> None:
        """A search ended: of its speculative crops `hits` were visited later, the priors had promised `predicted_hits`."""
        if not self.adaptive:
            return
        self._pred += float(predicted_hits)
        self._hits += float(hits)
        self.calibration = min(1.5, max(0.25, (self._hits + self.CAL_WEIGHT) / (self._pred + self.CAL_WEIGHT)))

    def step_ms(self, n_crops: int) -> float:
        """t(B): one engine step of n_crops crops dealt over `world` ranks (piecewise linear in the per-rank batch)."""
        b = -(-max(n_crops, 0) // self.world)
        if b <= 0:
            return 0.0
        ks = list(self.table)
        if b <= ks[0]:
            return self.table[ks[0]] * b / ks[0]
        for lo, hi in zip(ks, ks[1:]):
            if b <= hi:
                return self.table[lo] + (self.table[hi] - self.table[lo]) * (b - lo) / (hi - lo)
        return self.table[ks[-1]] * b / ks[-1]

    def select(self, n_must: int, cands: List[Tuple[float, object]], n_live: int) -> List[object]:
        """The candidates worth scoring in a step that already holds n_must crops; cands = [(visit probability, item), ...]."""
        if not self.enabled:
            return []
        chosen, B = [], n_must
        self.last_selected_p = []
        base = self.step_ms(n_must)
        cal = getattr(self, "calibration", 1.0)
        for p, item in sorted(cands, key=lambda c: -c[0]):
            if B >= self.cap:
                break
            p = min(p * cal, 1.0)
            if p * self.step_ms(1) / max(n_live, 1) < self.step_ms(B + 1) - self.step_ms(B):
                break
            # bounded downside: whatever the priors say, one step's speculation never costs more than one single-crop step
            if self.step_ms(B + 1) - base > self.step_ms(1):
                break
            chosen.append(item)
            self.last_selected_p.append(p)
            B += 1
        return chosen


def _run_loader(loader):
    """A sample's image loader on the prefetch thread: the file is opened AND decoded there (PIL opens lazily; the decode of a 4K JPEG
    is ~60 ms that the search thread would otherwise spend between two engine steps)."""
    image = loader()
    load = getattr(image, "load", None)
    if callable(load):
        load()
    return image


def _run_loader_upload(vsm, image, slot):
    """Prefetch-thread work for a sample whose image slot could be reserved ahead: load / decode AND start the upload
    (VSM.set_image_async: pinned staging + a copy stream) while the main thread is inside an engine step."""
    image = _run_loader(image) if callable(image) else image
    vsm.set_image_async(image, slot)
    return image


def visual_search_stream(vsm, samples, *, window: Optional[int] = None, policy: Optional[SpeculationPolicy] = None,
                         stats: Optional[dict] = None, prefetch: int = 8, **kw):
    """Cross-image lock-step search: `samples` = iterable of (image, target_object_name, target_bbox, smallest_size) — `image` a
    PIL image or a zero-argument loader returning one (called when the sample enters the window; loaders with the same `.key`, or
    the same loader object, share an image slot; `smallest_size` may then be a callable of the loaded image); returns the
    list of visual_search 4-tuples in sample order — what the reference's outer loops compute one sample at a time
    (visual_search.py:536-560; vstar_bench_eval.py:190-262 calls visual_search per missing object of each question).

    Up to `window` searches are LIVE at once (default: the engine batch x world size); each is a generator
    (`_visual_search_steps`) that yields the crop it needs next.  One engine step scores the MUST crops of every live search —
    crops of different images in the same batch through the engine's image slots (VSM.set_image(image, slot)) — plus whatever
    `policy` (SpeculationPolicy) finds worth speculating on; when a search ends the next sample takes its place, so batches stay
    full of crops the reference's order really visits.  Images are uploaded once per sample window (samples that share an image
    object share its slot).  Per (image, target) the decisions are exactly those of `visual_search`: with batch-invariant records
    (plain batches, or group_prompts = "always") the results are bit-identical to the per-sample loop.

    prefetch: how many of the upcoming samples' images are prepared ahead of the window on a worker thread while the engine is
    busy: the loader runs there (open + decode) and — with an engine that offers asynchronous uploads (VSM.supports_async_upload)
    and a spare image slot to reserve — so does the upload (pinned staging, copy stream; the first preprocessing of the slot waits
    for it on the device), which takes image transfer off the serial path between two engine steps (round 4: it was 7 % of the
    one-GPU stream and would not shrink with more ranks).  0 = everything when the sample enters the window.
    A loader is called once per residency of its image: samples that are live together share one slot and one call; when the
    last of them has ended the slot (and the host copy) is released, and a LATER sample with the same key loads the image again.
    Nothing about the results depends on it.

    stats (optional dict) receives: searches, crops_scored (engine records), useful_crops (nodes the best-first order visited),
    wasted_crop_frac, engine_steps, per_search = [{crops_scored, path_visited, ...}]."""
    samples = list(samples)
    n = len(samples)
    results: List = [None] * n
    if n == 0:
        return results
    on_device = bool(getattr(vsm, "supports_gpu_preprocess", False)) and hasattr(vsm, "inference_boxes") and kw.get("gpu_preprocess", True)
    per_stats: List[dict] = [dict() for _ in range(n)]
    if not on_device:
        # no resident images: the plain per-sample loop (still batched / speculative inside each search)
        for i, (image, name, gt, smallest) in enumerate(samples):
            image = image() if callable(image) else image
            smallest = smallest(image) if callable(smallest) else smallest
            results[i] = visual_search(vsm, image, name, gt, smallest, stats=per_stats[i], **kw)
        _fill_stream_stats(stats, per_stats, 0)
        return results
    world = vsm._dist()[0] if hasattr(vsm, "_dist") else 1
    cap = max(int(getattr(getattr(vsm, "cfg", None), "max_batch", 32)), 1) * world
    if policy is None:
        policy = SpeculationPolicy(getattr(vsm, "step_ms_table", None), cap=cap, world=world, enabled=kw.get("speculate", True))
    n_slots = int(getattr(vsm, "max_image_slots", 64))
    window = max(1, min(window or cap, n))
    plan_bs, plan_spec = kw.get("batch_size"), kw.get("speculate", True)
    want_async = bool(kw.get("async_upload", True))
    kw = {k: v for k, v in kw.items() if k not in ("batch_size", "speculate", "stats", "async_upload")}
    own_plans = bool(getattr(policy, "per_search_plan", False))      # visual_search_many: every search plans its own step

    slot_of: Dict[int, int] = {}                  # id(image) -> slot
    slot_refs: Dict[int, int] = {}                # slot -> live searches using it
    free_slots = list(range(n_slots - 1, -1, -1))
    gens: Dict[int, object] = {}
    scorers: Dict[int, _NodeScorer] = {}
    waiting: Dict[int, List[list]] = {}           # search -> the crop(s) it waits for
    want_stats: Dict[int, List] = {}
    next_sample = 0
    engine_steps = 0
    n_spec: Dict[int, int] = {}                   # search -> speculative crops scored for it
    pred_hits: Dict[int, float] = {}              # search -> sum of the (calibrated) visit probabilities of those crops

    loaded: Dict[object, object] = {}             # slot key -> the PIL image living in that slot
    pending: Dict[object, object] = {}            # slot key -> Future of a loader running ahead (prefetch)
    pool = None
    async_up = bool(getattr(vsm, "supports_async_upload", False)) and want_async
    if prefetch > 0 and (async_up or any(callable(smp[0]) for smp in samples)):
        from concurrent.futures import ThreadPoolExecutor
        # several workers: decoding a 4K image takes 40 - 100 ms of host time (PIL releases the GIL) and a step of a full window
        # consumes 4 - 5 new images; with sharded crops every rank decodes every image while its engine steps shrink with the world
        # size, so the decode rate is what must scale.  The uploads themselves are serialised inside the VSM
        pool = ThreadPoolExecutor(max_workers=4 if async_up else 1, thread_name_prefix="vstar-image-prefetch")
    n_async = 0

    def prefetch_ahead(limit=None):
        # images of the next samples that are not in the window yet, in sample order, at most `limit` (default: `prefetch`) in
        # flight or waiting.  pending[key] = (future, reserved slot or None)
        nonlocal n_async
        if pool is None:
            return
        limit = prefetch if limit is None else limit
        i = next_sample
        while i < n and len(pending) < limit:
            image = samples[i][0]
            key = getattr(image, "key", id(image))
            if key not in slot_of and key not in pending:
                if async_up and free_slots:
                    sl = free_slots.pop()
                    pending[key] = (pool.submit(_run_loader_upload, vsm, image, sl), sl)
                    n_async += 1
                elif callable(image):
                    pending[key] = (pool.submit(_run_loader, image), None)
            i += 1

    def start(i):
        image, name, gt, smallest = samples[i]
        key = getattr(image, "key", id(image))
        if key not in slot_of:
            fut, reserved = pending.get(key, (None, None))
            if reserved is None and not free_slots:
                return False
            if fut is not None:
                del pending[key]
                loaded[key] = fut.result()
            else:
                loaded[key] = image() if callable(image) else image
            if reserved is not None:                # loaded AND already on its way to the slot (prefetch thread)
                slot_of[key] = reserved
                slot_refs[reserved] = 0
            else:
                slot_of[key] = free_slots.pop()
                slot_refs[slot_of[key]] = 0
                vsm.set_image(loaded[key]) if slot_of[key] == 0 else vsm.set_image(loaded[key], slot_of[key])
        image = loaded[key]
        if callable(smallest):
            smallest = smallest(image)
        sl = slot_of[key]
        slot_refs[sl] += 1
        if own_plans:
            sc = _NodeScorer(vsm, image, LOCATE_QUESTION.format(name), smallest, plan_bs, plan_spec, True, kw.get("device_reductions"),
                             upload_image=False, slot=sl)
        else:
            sc = _NodeScorer(vsm, image, LOCATE_QUESTION.format(name), smallest, 1, False, True, kw.get("device_reductions"),
                             upload_image=False, slot=sl)
            sc.stream = True
        scorers[i] = sc
        gens[i] = _visual_search_steps(vsm, image, name, gt, smallest, _scorer=sc, stats=per_stats[i], **kw)
        advance(i, first=True)
        return True

    def finish(i, value):
        results[i] = value
        if hasattr(policy, "observe"):
            policy.observe(int(per_stats[i].get("path_visited", 1)))
        if hasattr(policy, "observe_speculation") and n_spec.get(i):
            wasted = int(per_stats[i].get("crops_scored", 0)) - int(per_stats[i].get("path_visited", 0))
            policy.observe_speculation(pred_hits.get(i, 0.0), max(n_spec[i] - max(wasted, 0), 0))
        sl = scorers[i].slot
        slot_refs[sl] -= 1
        if slot_refs[sl] == 0:                     # last live search of that image: the slot can take another image
            key = next(k for k, v in slot_of.items() if v == sl)
            del slot_of[key]
            loaded.pop(key, None)
            free_slots.append(sl)
            rel = getattr(vsm, "release_image", None)
            if rel is not None:
                rel(sl)
        gens.pop(i).close()

    def advance(i, value=None, first=False):
        try:
            req = next(gens[i]) if first else gens[i].send(value)
            if req[0] == "score":
                waiting[i] = req[2]
            else:
                want_stats[i] = req[1]
        except StopIteration as done:
            finish(i, done.value)

    # One process: a sample whose image is still being prepared by the prefetch thread does not hold up the searches that are
    # ready — the engine step runs with what is live and the sample joins a later step (start-up of a job: the first window's
    # images are decoded and uploaded WHILE the first steps run, instead of all of them before the first step).  With crop sharding
    # every rank must form identical batches, and "is the future done" is a matter of timing: there the refill blocks.
    nonblocking = world == 1 and pool is not None

    def refill():
        nonlocal next_sample
        while next_sample < n and len(gens) < window:
            if nonblocking and gens:
                key = getattr(samples[next_sample][0], "key", id(samples[next_sample][0]))
                if key not in slot_of and key in pending and not pending[key][0].done():
                    break
            if not start(next_sample):
                break                               # every image slot is in use: wait for a search to end
            next_sample += 1
        prefetch_ahead()

    def drain_stats():
        nonlocal want_stats
        while want_stats:
            cur, want_stats = want_stats, {}
            flat_reqs = [r for i in cur for r in cur[i]]
            res = _serve_stats(vsm, flat_reqs)
            k = 0
            for i in cur:
                n_i = len(cur[i])
                advance(i, res[k:k + n_i])
                k += n_i

    grouping = getattr(vsm, "group_prompts", False)
    if grouping:
        # a (crop, prompt) record must not depend on its batch companions.  Crops are batched by their NUMBER of prompts
        # (VSM._score_boxes_grouped), so a window of searches for different objects on different images is still one engine call
        vsm.group_prompts = "always"
    calls0 = int(getattr(vsm, "timers", {}).get("engine_calls", 0)) if isinstance(getattr(vsm, "timers", None), dict) else None
    dkw = {"defer_mismatch": True} if getattr(vsm, "supports_deferred_mismatch", False) else {}
    try:
        prefetch_ahead(limit=max(prefetch, min(window, n_slots // 2)))      # the whole first window's images go to the workers at once
        refill()
        drain_stats()
        refill()
        while waiting or gens:
            if not waiting:                         # every live search ended inside drain_stats / refill
                refill()
                drain_stats()
                if not waiting and not want_stats and next_sample >= n:
                    break
                continue
            reqs, waiting = waiting, {}
            # MUST crops first, then the policy's picks among every live search's candidates
            must = [(i, j, b) for i, boxes in reqs.items() for j, b in enumerate(boxes)]
            cands = [] if own_plans else [(p, (i, b)) for i in reqs for p, b in scorers[i].candidates(policy)]
            extra = policy.select(len(must), cands, len(reqs)) if cands else []
            for (i, _), pv in zip(extra, getattr(policy, "last_selected_p", []) or [0.0] * len(extra)):
                n_spec[i] = n_spec.get(i, 0) + 1
                pred_hits[i] = pred_hits.get(i, 0.0) + float(pv)
            flat = [(i, j, list(b)) for i, j, b in must] + [(i, None, list(b)) for i, b in extra]
            # box-major order inside the step: requests for the same crop (same image slot, same box) next to each other so that
            # a grouping VSM scores that crop's towers once for all its prompts
            order: Dict[Tuple, List[int]] = {}
            for k, (i, _, b) in enumerate(flat):
                order.setdefault((scorers[i].slot,) + tuple(b), []).append(k)
            seq = [k for ks in order.values() for k in ks]
            out: List = [None] * len(flat)
            c0 = 0
            while c0 < len(seq):
                c1 = min(c0 + cap, len(seq))
                while c1 < len(seq) and flat[seq[c1]][2] == flat[seq[c1 - 1]][2] and \
                        scorers[flat[seq[c1]][0]].slot == scorers[flat[seq[c1 - 1]][0]].slot:      # keep one crop's requests together
                    c1 += 1
                part = seq[c0:c1]
                sl_part = [scorers[flat[k][0]].slot for k in part]
                skw = {"slots": sl_part} if any(sl_part) else {}        # (one image in slot 0: the pre-slot call signature)
                res = vsm.inference_boxes([flat[k][2] for k in part], [scorers[flat[k][0]].question for k in part], mode="detection",
                                          upsample=False, **skw, **dkw)
                for k, r in zip(part, res):
                    out[k] = r
                engine_steps += 1
                c0 = c1
            per: Dict[int, List] = {i: [None] * len(boxes) for i, boxes in reqs.items()}
            spec: Dict[int, Tuple[List, List]] = {}
            for (i, j, b), r in zip(flat, out):
                if j is None:
                    spec.setdefault(i, ([], []))
                    spec[i][0].append(b)
                    spec[i][1].append(r)
                else:
                    per[i][j] = r
            for i, (bs, rs) in spec.items():       # speculative records go straight into the search's cache
                scorers[i].accept(bs, rs)
            for i in reqs:                         # in sample order: host decisions of search i, up to its next request
                advance(i, per[i])
            drain_stats()
            refill()
            drain_stats()
    finally:
        if grouping:
            vsm.group_prompts = grouping
        for g in list(gens.values()):
            g.close()
        if pool is not None:
            for f, _ in pending.values():
                f.cancel()
            pool.shutdown(wait=True)
        # prefetched-but-unstarted images (an exception unwound the driver): their reserved slots hold host copies too
        rel = getattr(vsm, "release_image", None)
        if rel is not None:
            for _, reserved in pending.values():
                if reserved is not None:
                    rel(reserved)
    _fill_stream_stats(stats, per_stats, engine_steps)
    if stats is not None and calls0 is not None:
        # launches of an engine scoring entry point (a step can take several: batch cap, prompt-count buckets); engine_steps counts
        # the driver's scoring rounds
        stats["engine_calls"] = int(vsm.timers.get("engine_calls", 0)) - calls0
    if stats is not None:
        stats["async_uploads"] = n_async
    return results


def _fill_stream_stats(stats: Optional[dict], per_stats: List[dict], engine_steps: int) -> None:
    if stats is None:
        return
    scored = sum(int(p.get("crops_scored", 0)) for p in per_stats)
    useful = sum(int(p.get("path_visited", 0)) for p in per_stats)
    stats.update(searches=len(per_stats), crops_scored=scored, useful_crops=useful,
                 wasted_crop_frac=(1.0 - useful / scored) if scored else 0.0, engine_steps=engine_steps,
                 per_search=[{k: v for k, v in p.items() if k != "search_path"} for p in per_stats])
    if stats.get("keep_paths"):              # the visited boxes of every search, in visit order (decision-parity reports)
        stats["visit_orders"] = [[tuple(int(v) for v in p["bbox"]) for p in ps.get("search_path", [])] for ps in per_stats]


def visual_search_many(vsm, image, target_object_names: Sequence[str], target_bboxes=None, smallest_size: int = 224, *,
                       stats: Optional[dict] = None, **kw):
    """Several targets on ONE image (the reference loops `visual_search` per missing object, vstar_bench_eval.py:205-209):
    same per-target results as that loop — each entry is visual_search's 4-tuple — but the searches advance in LOCK STEP: each
    is a generator (`_visual_search_steps`) that yields the crops it needs scored; one engine step scores what ALL live searches
    ask for, all targets of a crop in the same call, so small per-target steps still fill the GPU and a grouping VSM
    (`vsm.group_prompts`) evaluates each crop's towers and shared prompt positions once for every target that wants it.  With
    grouping the VSM is switched to group_prompts = "always" for the call: a (crop, prompt) record then never depends on which
    other targets asked for the crop, so target i's result is a function of (image, target i) alone.  One target, or a VSM
    without on-device crops, is the plain loop.  `stats` (optional) receives the stream statistics plus per-target entries
    (round 3: each target has its OWN statistics dict — they used to overwrite one another).
    This is visual_search_stream with every sample on the same image and the whole set live at once; every search plans its own
    step (`batch_size`, `speculate`: breadth-first like `visual_search` does alone), not the stream's cost model."""
    names = list(target_object_names)
    gts = list(target_bboxes) if target_bboxes is not None else [None] * len(names)
    on_device = bool(getattr(vsm, "supports_gpu_preprocess", False)) and kw.get("gpu_preprocess", True)
    if not (on_device and len(names) > 1):
        per = [dict() for _ in names]
        skw = {k: v for k, v in kw.items() if k != "stats"}
        out = [visual_search(vsm, image, n, gt, smallest_size, stats=st, **skw) for n, gt, st in zip(names, gts, per)]
        _fill_stream_stats(stats, per, 0)
        return out
    pol = _PerSearchPlans()
    return visual_search_stream(vsm, [(image, n, gt, smallest_size) for n, gt in zip(names, gts)], window=len(names), policy=pol,
                                stats=stats, **kw)


class _PerSearchPlans(SpeculationPolicy):
    """visual_search_many's behaviour: each search asks for its own node plus its own breadth-first speculation
    (`_NodeScorer.plan` with the caller's batch_size / speculate), and the driver only merges the requests of a step."""
    per_search_plan = True

    def select(self, n_must, cands, n_live):
        return []
