"""oracle/ -- TEST INFRASTRUCTURE ONLY.

A CPU restatement (numpy / torch-CPU / plain C) of the algorithms on DeepDish's
detect -> encode -> track hot path, written from the cited reference lines
(paths are relative to the upstream AdaptiveCity/deepdish tree).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from this package, and only as the checker or
as the timed CPU baseline.  The product (``deepdish_amd``) never imports it and
has no CPU fallback: it raises when the HIP library is missing.

Parity pinning status (see DESIGN.md "Oracle"):
  * deep_sort math, tracker state machine, NMS, count-line logic: PINNED by
    golden vectors generated from the imported reference modules
    (``scripts/make_golden.py`` -> ``tests/golden/*.npz``).
  * crop/bilinear resize (cv2), Lanczos resize (Pillow ANTIALIAS), detector /
    encoder forward (tflite_runtime + absent weight blobs): PARITY UNPINNED
    against the real reference; restated from public algorithm descriptions and
    checked only for self-consistency (Pillow's own LANCZOS is used as a
    third-party pin for the stretch resize).
"""
