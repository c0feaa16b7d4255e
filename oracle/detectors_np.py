"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the detector adaptors' host-side tails.

What the reference does to the tensors a detector's ``interpreter.invoke()`` returns, before the
boxes reach the tracker (paths relative to the upstream tree):

  * ``ssd_nms_boxes``      -- tools/ssd_mobilenet.py:59-98   SSDMobileNet.nms_boxes
  * ``ssd_predict_tail``   -- tools/ssd_mobilenet.py:111-150 SSDMobileNet.predict after get_tensor
  * ``ssd_detect_filter``  -- tools/ssd_mobilenet.py:202-213 SSD_MOBILENET.detect_image after predict
  * ``yolov5_detect_tail`` -- tools/yolov5.py:120-146        YOLOV5.detect_image after get_tensor

PINNED by ``tests/golden/ssd_tail.npz`` and ``tests/golden/yolov5_tail.npz``: outputs of the
reference's own classes driven with canned interpreter outputs (``scripts/make_golden_detectors.py``);
``tests/test_oracle_golden.py`` holds this file to them.

The statements keep the reference's order and dtypes (f32 tensors, Python-float image sizes) because
both decide results: e.g. ``output[0][indices][:, reorder] * [w, h, w, h]`` promotes to f64, while the
YOLOv5 tail stays f32 except for the final ``*= np.array([w, h, w, h])`` in-place multiply (f32 result).
"""
import numpy as np


def ssd_nms_boxes(boxes, labels, scores, iou_threshold):
    """ssd_mobilenet.py:59-98 as scalar loops (same IEEE operations per pair, so bit-identical to the reference's
    vector statements).  boxes [K,4] xyxy, labels [K], scores [K] -> per-class lists, classes in the iteration
    order of ``set(labels)`` like the reference.  Three things differ from every other overlap test on the path:
    +1 on the INTERSECTION extents only (:83-84), areas = w * h without it (:72), survivors are ``ovr <= thr`` (:88);
    the right / bottom edges are re-derived as x + w, y + h (:80-81), not read back from the box."""
    per_class = []
    for cls_value in set(labels):                                   # :61
        member = np.flatnonzero(labels == cls_value)
        b, cl, s = boxes[member], labels[member], scores[member]
        left, top = b[:, 0], b[:, 1]
        wid, hei = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]             # :69-70
        area = wid * hei                                            # :72
        # :73 descending score.  Equal scores (common with a uint8 model: scores are multiples of 1/256): the reference's own NumPy (1.19,
        # Dockerfile.rpi-armv7:29-30) sorts the <= 16 rows of a class by insertion -- stably -- so the reversal puts the HIGHER row first;
        # NumPy 2.x on an AVX-512 host uses an unstable vector sort for the same call and orders ties arbitrarily.  Restated as the
        # reference's environment computes it.
        alive = [int(v) for v in s.argsort(kind='stable')[::-1]]
        keep = []
        while alive:
            i = alive.pop(0)
            keep.append(i)
            survivors = []
            for j in alive:
                iw = np.maximum(0.0, np.minimum(left[i] + wid[i], left[j] + wid[j]) - np.maximum(left[i], left[j]) + 1)
                ih = np.maximum(0.0, np.minimum(top[i] + hei[i], top[j] + hei[j]) - np.maximum(top[i], top[j]) + 1)
                inter = iw * ih
                if inter / (area[i] + area[j] - inter) <= iou_threshold:      # :87-88
                    survivors.append(j)
            alive = survivors
        keep = np.array(keep)
        per_class.append((b[keep], cl[keep], s[keep]))
    return [p[0] for p in per_class], [p[1] for p in per_class], [p[2] for p in per_class]


def ssd_predict_tail(output, label_lines, confidence=0.5, iou_threshold=0.5, original_image_size=None, net_size=(300, 300)):
    """ssd_mobilenet.py:111-150.  output = [boxes f32 [N,4] (ymin,xmin,ymax,xmax normalised), classes f32 [N],
    scores f32 [N], count]; label_lines = dict line-number -> text of the label file (:46-48).
    -> (boxes f64 [M,4] xyxy pixels, label names, scores f32 [M]) or ([], [], [])."""
    raw_boxes, raw_cls, raw_scores = (np.array(o, copy=True) for o in output[:3])
    nan_row, nan_col = np.nonzero(np.isnan(raw_boxes))
    raw_scores[np.concatenate([nan_row, nan_col])] = 0          # :111-113: the ROW and the COLUMN numbers of every NaN
    raw_scores[np.isnan(raw_scores)] = 0                        #           are both used as score indices; :115-116
    sel = np.flatnonzero(raw_scores >= confidence)              # :119
    w, h = original_image_size if original_image_size is not None else net_size
    boxes = raw_boxes[sel][:, [1, 0, 3, 2]] * [w, h, w, h]      # :121-127 -> xyxy pixels, f64 from here on
    labels, scores = raw_cls[sel], raw_scores[sel]
    n_boxes, n_labels, n_scores = ssd_nms_boxes(boxes, labels, scores, iou_threshold)
    if not n_boxes:
        return [], [], []
    boxes = np.concatenate(n_boxes)
    labels = np.concatenate(n_labels).astype(np.uint64)         # :139
    scores = np.concatenate(n_scores)
    names = [label_lines[int(li) + 1] for li in labels if 0 <= li < len(label_lines) - 1]      # :142-147
    return boxes, names, scores


def ssd_detect_filter(boxes, labels, scores, wanted_labels, score_threshold):
    """ssd_mobilenet.py:202-213: wanted label, score >= threshold, xyxy -> tlwh."""
    rb, rl, rs = [], [], []
    for i in range(len(boxes)):
        if labels[i] in wanted_labels and scores[i] >= score_threshold:
            box = boxes[i]
            rb.append([box[0], box[1], box[2] - box[0], box[3] - box[1]])
            rl.append(labels[i])
            rs.append(scores[i])
    return rb, rl, rs


def yolov5_detect_tail(output_data, label_lines, wanted_labels, score_threshold, img_size):
    """yolov5.py:120-146.  output_data f32 [1,N,5+C] rows (xc, yc, w, h normalised, obj, cls...); img_size =
    (width, height) of the ORIGINAL image -> (boxes tlwh as lists of np.float32, labels, scores).
    All arithmetic stays f32 (the image-size multiply is an in-place f32 update, :132), rows keep np.where order."""
    rows = np.asarray(output_data, dtype=np.float32).reshape(-1, np.shape(output_data)[-1])
    half_w, half_h = rows[:, 2] / 2, rows[:, 3] / 2
    corners = np.stack([rows[:, 0] - half_w, rows[:, 1] - half_h, rows[:, 0] + half_w, rows[:, 1] + half_h], axis=1)   # :121-125
    cls_conf = rows[:, 5:] * rows[:, 4:5]                                    # :126
    best = np.argmax(cls_conf, axis=1)                                       # :127 first maximum wins
    conf = cls_conf[np.arange(len(rows)), best]
    chosen = np.flatnonzero(conf >= score_threshold)                          # :130
    scale = np.array([img_size[0], img_size[1], img_size[0], img_size[1]])
    rb, rl, rs = [], [], []
    for r in chosen:
        xyxy = (corners[r] * scale).astype(np.float32)                       # :131, f32 storage
        label = label_lines[int(np.float32(best[r]))]
        if label in wanted_labels and conf[r] >= score_threshold:            # :139
            rb.append([xyxy[0], xyxy[1], xyxy[2] - xyxy[0], xyxy[3] - xyxy[1]])      # :140-142
            rl.append(label)
            rs.append(conf[r])
    return rb, rl, rs
