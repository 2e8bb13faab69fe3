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
