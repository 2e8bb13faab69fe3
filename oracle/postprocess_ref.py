"""Loop-style restatement of the reference's line pre/post-processing (TEST INFRASTRUCTURE ONLY), written
independently of effocr_amd/postprocess.py so the two can be checked against each other and against
hand-computed cases.  Follows infer_effocr.py:345-367 (en_preprocess), :370-410 (en_postprocess),
:412-418 (jp_preprocess); constants from utils/spell_check_utils.py:60-65 and infer_effocr.py:239-240.
Parity unpinned: the reference has no tests for these functions."""
LARGE = 1_000_000
DISTINCT = list("aenr")
NONDISTINCT = list("wuosvcxz")


def sort_filter(boxes, thresh, vertical):
    key = (lambda x: x[1]) if vertical else (lambda x: x[0])
    return [list(x[:4]) for x in sorted([list(b) for b in boxes], key=key) if x[4] > thresh]


def en_preprocess(char_boxes, word_boxes, score_thresh=0.5, score_thresh_word=0.5, vertical=False):
    chars = sort_filter(char_boxes, score_thresh, vertical)
    words = sort_filter(word_boxes, score_thresh_word, vertical)
    word_end_idx, closest_idx = [], 0
    for wordleft in [w[0] for w in words]:
        prev = LARGE
        for idx, charright in enumerate([c[2] for c in chars]):
            dist = abs(wordleft - charright)
            if dist < prev and charright > wordleft:
                prev, closest_idx = dist, idx
        word_end_idx.append(closest_idx)
    return chars, word_end_idx


def en_postprocess(line_output, word_end_idx, charheights, charbottoms, anchor_margin=None, anchor_multiplier=4):
    assert len(line_output) == len(charheights) == len(charbottoms)
    if any(len(x) == 0 for x in (line_output, word_end_idx, charheights, charbottoms)):
        return None
    out, hs, bs = [], [], []
    for idx, x in enumerate(line_output):
        if idx in word_end_idx:
            out.append(" " + x); hs += [LARGE, charheights[idx]]; bs += [0, charbottoms[idx]]
        else:
            out.append(x); hs.append(charheights[idx]); bs.append(charbottoms[idx])
    bs = bs[1:] if bs[0] == 0 else bs
    hs = hs[1:] if hs[0] == LARGE else hs
    line = "".join(out).strip()
    assert len(hs) == len(line)
    anchors = [i for i, c in enumerate(line) if c in DISTINCT]
    if len(anchors) > 0 and anchor_margin is not None:
        ah = sum(hs[i] for i in anchors) / len(anchors)
        ab = sum(bs[i] for i in anchors) / len(anchors)
        tolower = [i for i in range(len(line)) if abs(hs[i] - ah) < anchor_margin * ah]
        toupper = [i for i in range(len(line)) if hs[i] - ah > anchor_margin * anchor_multiplier * ah]
        toperiod = [i for i, c in enumerate(line) if c == "-" and abs(bs[i] - ab) < anchor_margin * ah]
        line = "".join(c.lower() if i in tolower else c for i, c in enumerate(line))
        line = "".join(c.upper() if i in toupper and c in NONDISTINCT else c for i, c in enumerate(line))
        line = "".join("." if i in toperiod else c for i, c in enumerate(line))
    return line


def infer_ref(image, result, lang, encode, knn_search, candidate_chars, transform, k=10, vertical=False, score_thresh=0.5,
              score_thresh_word=0.5, anchor_margin=None, double_clipped=True):
    """``EffOCR.infer`` from the localizer result on (infer_effocr.py:268-343, kNN branch :310-319), loop for loop over caller-supplied
    oracle stages: ``transform(crop HWC uint8) -> [3,S,S]``, ``encode([n,3,S,S]) -> [n,D]``, ``knn_search(q [n,D] unit rows, k) -> ids [n,k]``.
    Pinned to the reference by tests/golden/ref_run_effocr.json["infer"] (tests/test_ref_golden.py)."""
    import numpy as np
    import torch
    if lang == "en":
        char_bboxes, word_bboxes = result if isinstance(result[0], np.ndarray) else result[0]
        char_bboxes, word_end_idx = en_preprocess(char_bboxes, word_bboxes, score_thresh, score_thresh_word, vertical)
        word_bboxes = [list(w[:5]) for w in word_bboxes]
    else:
        char_bboxes, word_bboxes = sort_filter(result[0][0], score_thresh, vertical), None
    H, W = image.shape[0], image.shape[1]
    crops, heights, bottoms = [], [], []
    for bbox in char_bboxes:
        x0, y0, x1, y1 = map(int, map(round, bbox))
        if double_clipped:
            x0, y0, x1, y1 = (0, y0, W, y1) if vertical else (x0, 0, x1, H)
        crops.append(transform(image[y0:y1, x0:x1, :]))
        heights.append(bbox[3] - bbox[1]); bottoms.append(bbox[3])
    if len(crops) == 0:
        return None, None, None, None
    emb = torch.nn.functional.normalize(encode(torch.stack(crops)), p=2, dim=1)
    index_list = knn_search(emb, k).squeeze(-1).tolist()
    nearest = [[candidate_chars[nn] for nn in nns] for nns in index_list]
    output_nns = ["".join(chars).strip() for chars in nearest]
    output = "".join(x[0] for x in nearest).strip()
    if lang == "en":
        output = en_postprocess(output, word_end_idx, heights, bottoms, anchor_margin=anchor_margin)
    return output, output_nns, char_bboxes, word_bboxes
