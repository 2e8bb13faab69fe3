"""TEST INFRASTRUCTURE — ``run_effocr`` (infer_effocr_onnx_multi.py:227-397) restated on the CPU, loop for loop, over the
oracle's own stages (never imported by effocr_amd/).  PARITY UNPINNED like its parts: the reference driver cannot run here
(onnxruntime / faiss / cv2 absent, and as committed it raises TypeError at :489).

    localizer    oracle/yolo_ref.py: load_localizer_img -> yolov5s_forward -> non_max_suppression(...)[0]        (:236-256)
    boxes        labels 0 / 1, en_preprocess / jp_preprocess with Python's ``sorted`` on tensor rows              (:275-292)
    crops        torch.round, int(round(x * W / 640)), numpy slicing of the image                                 (:313-325)
    transform    oracle/crop_transform_ref.paired_transform per crop; a failing crop -> None                      (:326-345)
    recognizer   create_batches (None -> zeros, last batch padded to 64) -> oracle encoder -> F.normalize ->
                 IndexFlatIP.search(k=1) (oracle/flat_ip.c) -> candidate_chars                                     (:347-375)
    post         per-line slices, "".join(x[0] ...).strip(), en_postprocess                                        (:377-392)
``localizer_results`` lets a test substitute the boxes of the device localizer (whose network differs from the CPU one by fp32
summation order), so that every LATER stage can be compared exactly.
"""
import numpy as np
import torch

from . import knn_ref
from . import yolo_ref as Y
from .crop_transform_ref import paired_transform
from .encoders_ref import encoder_forward, l2_normalize
from .postprocess_ref import en_postprocess

LARGE_NUMBER = 1_000_000


def en_preprocess(bboxes_char, bboxes_word, vertical=False):                  # :70-90
    sorted_bboxes_char = sorted(bboxes_char, key=lambda x: x[1] if vertical else x[0])
    sorted_bboxes_word = sorted(bboxes_word, key=lambda x: x[1] if vertical else x[0])
    word_end_idx, closest_idx = [], 0
    rights = [x[2] for x in sorted_bboxes_char]
    for wordleft in [x[0] for x in sorted_bboxes_word]:
        prev_dist = LARGE_NUMBER
        for idx, charright in enumerate(rights):
            dist = abs(wordleft - charright)
            if dist < prev_dist and charright > wordleft:
                prev_dist, closest_idx = dist, idx
        word_end_idx.append(closest_idx)
    return sorted_bboxes_char, word_end_idx


def localize(images, loc_sd, conf_thresh, iou_thresh, input_shape=(640, 640)):
    out = []
    for im in images:
        pre = Y.load_localizer_img(im, input_shape, bgr=False)
        out.append(Y.non_max_suppression(Y.yolov5s_forward(loc_sd, torch.from_numpy(pre)), conf_thresh, iou_thresh, max_det=1000)[0])
    return out


def run_effocr_ref(images, loc_sd, enc_arch, enc_sd, index, candidate_chars, lang, vertical=False, conf_thresh=0.3, iou_thresh=0.01,
                   localizer_results=None, anchor_margin=None, size=224):
    results = localizer_results if localizer_results is not None else localize(images, loc_sd, conf_thresh, iou_thresh)
    char_crops, word_end_idxs, n_chars, charheights, charbottoms = [], [], [], [], []
    for im, result in zip(images, results):
        bboxes, labels = result[:, :4], result[:, -1]
        if lang == "en":
            char_bboxes, word_bboxes = bboxes[labels == 0], bboxes[labels == 1]
            if len(char_bboxes) != 0:
                char_bboxes, word_end_idx = en_preprocess(char_bboxes, word_bboxes)
                n_chars.append(len(char_bboxes)); word_end_idxs.append(word_end_idx)
            else:
                n_chars.append(0); word_end_idxs.append([])
        else:
            char_bboxes = bboxes[labels == 0]
            if len(char_bboxes) != 0:
                char_bboxes = sorted(char_bboxes, key=lambda x: x[1] if vertical else x[0])
                n_chars.append(len(char_bboxes))
            else:
                n_chars.append(0)
        im_height, im_width = im.shape[0], im.shape[1]
        for bbox in char_bboxes:
            x0, y0, x1, y1 = torch.round(bbox)
            if vertical:
                x0, y0, x1, y1 = 0, int(round(y0.item() * im_height / 640)), im_width, int(round(y1.item() * im_height / 640))
            else:
                x0, y0, x1, y1 = int(round(x0.item() * im_width / 640)), 0, int(round(x1.item() * im_width / 640)), im_height
            char_crops.append(im[y0:y1, x0:x1, :])
            if lang == "en":
                charheights.append(float(bbox[3] - bbox[1])); charbottoms.append(float(bbox[3]))
    transformed = []
    for c in char_crops:                                                      # TransformationThread: any failure -> None
        try:
            if c.shape[0] == 0 or c.shape[1] == 0:
                raise ValueError("empty crop")
            transformed.append(torch.from_numpy(np.asarray(paired_transform(c, size=size), dtype=np.float32)))
        except Exception:
            transformed.append(None)
    batches, batch = [], []                                                   # create_batches (:143-158)
    for i, d in enumerate(transformed):
        batch.append(d if d is not None else torch.zeros((3, size, size)))
        if (i + 1) % 64 == 0:
            batches.append(torch.stack(batch)); batch = []
    if batch:
        batches.append(torch.nn.functional.pad(torch.stack(batch), (0, 0, 0, 0, 0, 0, 0, 64 - len(batch))))
    indices = []
    for b in batches:
        emb = l2_normalize(encoder_forward(enc_arch, enc_sd, b)).numpy()
        indices += knn_ref.flat_ip_search(emb, index, 1)[1][:, 0].tolist()
    nn_outputs = [candidate_chars[i] for i in indices]
    idx, outputs = 0, []
    heights, bottoms = [], []
    for l in n_chars:
        outputs.append("".join(x[0] for x in nn_outputs[idx:idx + l]).strip())
        heights.append(charheights[idx:idx + l]); bottoms.append(charbottoms[idx:idx + l])
        idx += l
    if lang == "en":
        return [en_postprocess(outputs[i], word_end_idxs[i], heights[i], bottoms[i], anchor_margin=anchor_margin) for i in range(len(outputs))], results
    return outputs, results
