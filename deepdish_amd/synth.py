"""Seeded synthetic workloads for tests and bench.py (no datasets, no weights on this box).

Shapes follow BASELINE.json's configs: 640x480 BGR u8 frames with ~20 moving
textured rectangles, integer tlwh detections (true box + U{-2..2} jitter),
tie-free scores in (0.5, 1), 128-d unit appearance features = per-identity
gaussian + 0.05 * per-frame noise.  Boxes stay >= 2 px inside the frame so the
reference's failed-crop branch (tools/generate_detections.py:201-204, unseeded
random patch) can never fire.
"""
import numpy as np


class Scene:
    def __init__(self, seed=0, n_obj=20, width=640, height=480, n_frames=100,
                 jitter=2, p_miss=0.03, churn=True, n_dup=0.25, feat_noise=0.05,
                 wrange=(20, 50), hrange=(60, 120), vmax=2.0):
        rng = np.random.default_rng(seed)
        self.seed, self.W, self.H, self.n_frames = seed, width, height, n_frames
        self.n_obj = n_obj
        self.w = rng.integers(wrange[0], wrange[1], n_obj).astype(np.float64)
        self.h = rng.integers(hrange[0], hrange[1], n_obj).astype(np.float64)
        x0 = rng.uniform(4, width - 4 - self.w)
        y0 = rng.uniform(4, height - 4 - self.h)
        vx = rng.uniform(-vmax, vmax, n_obj)
        vy = rng.uniform(-vmax / 4, vmax / 4, n_obj)
        # make sure a healthy fraction actually crosses the vertical mid-line
        cross = rng.random(n_obj) < 0.6
        left = x0 + self.w / 2 < width / 2
        vx = np.where(cross, np.where(left, np.abs(vx) + 0.8, -np.abs(vx) - 0.8), vx)
        if churn:
            self.t0 = np.where(rng.random(n_obj) < 0.3, rng.integers(0, n_frames // 2, n_obj), 0)
            self.t1 = np.where(rng.random(n_obj) < 0.3,
                               rng.integers(n_frames // 2, n_frames, n_obj), n_frames)
        else:
            self.t0 = np.zeros(n_obj, dtype=np.int64)
            self.t1 = np.full(n_obj, n_frames, dtype=np.int64)
        # trajectories with reflection at a 3 px margin
        self.xy = np.zeros((n_frames, n_obj, 2))
        x, y = x0.copy(), y0.copy()
        for f in range(n_frames):
            self.xy[f, :, 0], self.xy[f, :, 1] = x, y
            x, y = x + vx, y + vy
            lo, hi = 3.0, width - 3.0 - self.w
            bx = (x < lo) | (x > hi)
            vx = np.where(bx, -vx, vx)
            x = np.clip(x, lo, hi)
            lo, hi = 3.0, height - 3.0 - self.h
            by = (y < lo) | (y > hi)
            vy = np.where(by, -vy, vy)
            y = np.clip(y, lo, hi)
        self.ident = rng.standard_normal((n_obj, 128)).astype(np.float32)
        self.tex = rng.integers(0, 256, (n_obj, 16, 8, 3), dtype=np.uint8)
        self.bg = self._background(rng)
        self._frng_seed = int(rng.integers(1 << 31))
        self.jitter, self.p_miss, self.n_dup, self.feat_noise = jitter, p_miss, n_dup, feat_noise

    def _background(self, rng):
        lo = rng.integers(40, 200, (self.H // 32 + 2, self.W // 32 + 2, 3)).astype(np.float32)
        yy = np.arange(self.H)[:, None] / 32.0
        xx = np.arange(self.W)[None, :] / 32.0
        y0, x0 = np.floor(yy).astype(int), np.floor(xx).astype(int)
        fy, fx = (yy - y0)[..., None], (xx - x0)[..., None]
        img = (lo[y0, x0] * (1 - fy) * (1 - fx) + lo[y0 + 1, x0] * fy * (1 - fx)
               + lo[y0, x0 + 1] * (1 - fy) * fx + lo[y0 + 1, x0 + 1] * fy * fx)
        return np.clip(img, 0, 255).astype(np.uint8)

    def alive(self, f):
        return np.nonzero((self.t0 <= f) & (f < self.t1))[0]

    def frame(self, f):
        """BGR u8 [H, W, 3] -- background + one textured rectangle per live object."""
        img = self.bg.copy()
        for i in self.alive(f):
            x, y = int(round(self.xy[f, i, 0])), int(round(self.xy[f, i, 1]))
            w, h = int(self.w[i]), int(self.h[i])
            ty = (np.arange(h) * 16 // h)[:, None]
            tx = (np.arange(w) * 8 // w)[None, :]
            img[y:y + h, x:x + w] = self.tex[i][ty, tx]
        return img

    def detections(self, f):
        """-> boxes int64 [K,4] tlwh, scores f64 [K] (tie-free), ident int64 [K], features f32 [K,128].

        Includes a few lower-scored, heavily overlapping duplicates so NMS has work to do.
        """
        rng = np.random.default_rng((self._frng_seed, f))
        ids = self.alive(f)
        ids = ids[rng.random(len(ids)) >= self.p_miss]
        boxes, scores, who = [], [], []
        for i in ids:
            j = rng.integers(-self.jitter, self.jitter + 1, 4) if self.jitter else np.zeros(4, np.int64)
            x = int(round(self.xy[f, i, 0])) + j[0]
            y = int(round(self.xy[f, i, 1])) + j[1]
            w = int(self.w[i]) + j[2]
            h = int(self.h[i]) + j[3]
            x = min(max(x, 2), self.W - 2 - w)
            y = min(max(y, 2), self.H - 2 - h)
            boxes.append((x, y, w, h)); scores.append(rng.uniform(0.6, 1.0)); who.append(i)
            if rng.random() < self.n_dup:
                boxes.append((min(x + 1, self.W - 2 - w), y, w, h))
                scores.append(rng.uniform(0.5, 0.6)); who.append(i)
        boxes = np.array(boxes, dtype=np.int64).reshape(-1, 4)
        scores = np.array(scores, dtype=np.float64) + 1e-6 * np.arange(len(scores))
        who = np.array(who, dtype=np.int64)
        feats = self.ident[who] + self.feat_noise * rng.standard_normal((len(who), 128)).astype(np.float32)
        feats = (feats / np.linalg.norm(feats, axis=1, keepdims=True)).astype(np.float32)
        return boxes, scores, who, feats

    def countline(self):
        """Default vertical mid-line, deepdish.py:739-741."""
        return np.array([[self.W / 2, 0], [self.W / 2, self.H]], dtype=int).astype(float)


def tracker_scene(seed=0, n_obj=256, n_frames=60):
    """Config 4: T = D = 256 on a 4000x3000 canvas so boxes rarely overlap."""
    return Scene(seed=seed, n_obj=n_obj, width=4000, height=3000, n_frames=n_frames,
                 p_miss=0.02, churn=False, n_dup=0.0)
