"""CPU restatement of OpenCV's BackgroundSubtractorMOG2 -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).

The reference calls ``cv2.createBackgroundSubtractorMOG2()`` / ``backSub.apply(frame)`` (deepdish.py:889,922)
and tests ``np.count_nonzero(fgMask[y:y+h, x:x+w]) >= ratio * w * h`` per box (deepdish.py:957).  The arithmetic
is third party: OpenCV (``opencv-python``, unpinned in the reference's requirements; absent from the reference
tree and from this image).  **Parity unpinned**: this file restates the published algorithm -- Z. Zivkovic,
"Improved adaptive Gaussian mixture model for background subtraction", ICPR 2004, and Zivkovic & van der
Heijden, PRL 27(7) 2006, as implemented in OpenCV 4.x ``modules/video/src/bgfg_gaussmix2.cpp``
(``MOG2Invoker::operator()``, ``detectShadowGMM``, ``BackgroundSubtractorMOG2Impl::apply``) -- statement by
statement in f32, vectorised over pixels with masks instead of per-pixel loops.  The reference holds no test or
golden vector for this step.
"""
import numpy as np

F = np.float32
NMIX = 5
FLT_EPSILON = F(1.1920929e-07)


class MOG2:
    """One subtractor (one stream).  State arrays are mode-major: w, var [5, P]; mu [5, 3, P]; nmodes [P]."""

    def __init__(self, history=500, varThreshold=16, detectShadows=True):
        self.history = int(history)
        self.Tb = F(varThreshold)
        self.TB = F(0.9)
        self.Tg = F(3.0 * 3.0)
        self.var_init = F(15.0)
        self.var_min = F(4.0)
        self.var_max = F(5 * 15.0)
        self.fCT = F(0.05)
        self.tau = F(0.5)
        self.shadow_val = 127
        self.detect_shadows = bool(detectShadows)
        self.nframes = 0
        self.shape = None

    def _init(self, h, w):
        n = h * w
        self.shape = (h, w)
        self.w = np.zeros((NMIX, n), F)
        self.var = np.zeros((NMIX, n), F)
        self.mu = np.zeros((NMIX, 3, n), F)
        self.nmodes = np.zeros(n, np.int32)
        self.nframes = 0

    def _swap(self, i, sel):
        """std::swap(gmm[i], gmm[i-1]) and the matching means, for the pixels in sel."""
        for a in (self.w, self.var):
            t = a[i, sel].copy(); a[i, sel] = a[i - 1, sel]; a[i - 1, sel] = t
        t = self.mu[i][:, sel].copy(); self.mu[i][:, sel] = self.mu[i - 1][:, sel]; self.mu[i - 1][:, sel] = t

    def apply(self, image, learningRate=-1):
        image = np.asarray(image)
        assert image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] == 3
        h, w = image.shape[:2]
        if self.shape != (h, w):
            self._init(h, w)
        self.nframes += 1
        lr = learningRate if (learningRate >= 0 and self.nframes > 1) else 1.0 / min(2 * self.nframes, self.history)
        alphaT = F(lr)
        alpha1 = F(1.0) - alphaT
        prune = -alphaT * self.fCT
        n = h * w
        data = image.reshape(n, 3).astype(F).T.copy()               # [3, P]
        nmodes = self.nmodes
        background = np.zeros(n, bool)
        fits = np.zeros(n, bool)
        total = np.zeros(n, F)
        with np.errstate(all='ignore'):
            for m in range(NMIX):
                act = m < nmodes                                    # nmodes shrinks inside the loop, as upstream
                weight = alpha1 * self.w[m] + prune
                pos = np.full(n, m, np.int32)
                look = act & ~fits
                var = self.var[m].copy()
                e = self.mu[m] - data
                dist2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2]
                background |= look & (total < self.TB) & (dist2 < self.Tb * var)
                fit = look & (dist2 < self.Tg * var)
                fits |= fit
                weight = np.where(fit, weight + alphaT, weight)
                k = alphaT / weight
                for c in range(3):
                    self.mu[m, c] = np.where(fit, self.mu[m, c] - k * e[c], self.mu[m, c])
                vn = var + k * (dist2 - var)
                vn = np.where(vn < self.var_min, self.var_min, vn)
                vn = np.where(vn > self.var_max, self.var_max, vn)
                self.var[m] = np.where(fit, vn, self.var[m])
                going = fit.copy()
                for i in range(m, 0, -1):
                    going &= ~(weight < self.w[i - 1])
                    sel = np.nonzero(going)[0]
                    if sel.size:
                        self._swap(i, sel)
                        pos[sel] = i - 1
                pr = act & (weight < -prune)
                weight = np.where(pr, F(0), weight)
                nmodes = nmodes - pr.astype(np.int32)
                sel = np.nonzero(act)[0]
                self.w[pos[sel], sel] = weight[sel]
                total = np.where(act, total + weight, total)
            inv = np.where(np.abs(total) > FLT_EPSILON, F(1.0) / total, F(0))
            for m in range(NMIX):
                self.w[m] = np.where(m < nmodes, self.w[m] * inv, self.w[m])

            new = ~fits & (alphaT > 0)
            full = nmodes == NMIX
            mode = np.where(full, NMIX - 1, nmodes)
            nmodes = np.where(new & ~full, nmodes + 1, nmodes)
            sel = np.nonzero(new)[0]
            first = nmodes[sel] == 1
            for i in range(NMIX):
                other = new & (nmodes != 1) & (i < nmodes - 1) & (mode != i)
                self.w[i] = np.where(other, self.w[i] * alpha1, self.w[i])
            self.w[mode[sel], sel] = np.where(first, F(1.0), alphaT)
            self.var[mode[sel], sel] = self.var_init
            for c in range(3):
                self.mu[mode[sel], c, sel] = data[c, sel]
            going = new.copy()
            for i in range(NMIX - 1, 0, -1):
                inside = i <= nmodes - 1
                going &= ~(inside & (alphaT < self.w[i - 1]))
                s2 = np.nonzero(going & inside)[0]
                if s2.size:
                    self._swap(i, s2)
            self.nmodes = nmodes.astype(np.int32)

            mask = np.where(background, 0, 255).astype(np.uint8)
            if self.detect_shadows:
                tw = np.zeros(n, F)
                undecided = ~background
                shadow = np.zeros(n, bool)
                for m in range(NMIX):
                    a_ = undecided & (m < nmodes)
                    mu = self.mu[m]
                    num = np.zeros(n, F); den = np.zeros(n, F)
                    for c in range(3):
                        num = num + data[c] * mu[c]
                        den = den + mu[c] * mu[c]
                    zero = a_ & (den == 0)
                    undecided &= ~zero
                    a_ &= ~zero
                    cand = a_ & (num <= den) & (num >= self.tau * den)
                    a = num / den
                    d2a = np.zeros(n, F)
                    for c in range(3):
                        q = a * mu[c] - data[c]
                        d2a = d2a + q * q
                    hit = cand & (d2a < self.Tb * self.var[m] * a * a)
                    shadow |= hit
                    undecided &= ~hit
                    a_ &= ~hit
                    tw = np.where(a_, tw + self.w[m], tw)
                    undecided &= ~(a_ & (tw > self.TB))
                mask[shadow] = self.shadow_val
        return mask.reshape(h, w)


def motion_box_filter(fgmask, boxes_xywh, ratio):
    """deepdish.py:957: keep box k iff count_nonzero(fgMask[y:y+h, x:x+w]) >= ratio * w * h."""
    keep = []
    for (x, y, w, h) in boxes_xywh:
        keep.append(bool(np.count_nonzero(fgmask[y:y + h, x:x + w]) >= ratio * w * h))
    return keep


def live_state(m):
    """(w, var, mu, nmodes) of a MOG2 with everything past a pixel's mode count zeroed (those slots are stale)."""
    live = np.arange(NMIX)[:, None] < m.nmodes[None, :]
    return (np.where(live, m.w, F(0)), np.where(live, m.var, F(0)), np.where(live[:, None, :], m.mu, F(0)),
            m.nmodes.astype(np.uint8))


class MOG2Scalar:
    """The same statements as one Python loop per pixel, in the order bgfg_gaussmix2.cpp runs them -- slow; used by
    the tests to check the vectorised restatement above on small frames."""

    def __init__(self, history=500, varThreshold=16, detectShadows=True):
        self.p = MOG2(history, varThreshold, detectShadows)

    def apply(self, image, learningRate=-1):
        p = self.p
        h, w = image.shape[:2]
        if p.shape != (h, w):
            p._init(h, w)
        p.nframes += 1
        lr = learningRate if (learningRate >= 0 and p.nframes > 1) else 1.0 / min(2 * p.nframes, p.history)
        alphaT = F(lr); alpha1 = F(1.0) - alphaT; prune = -alphaT * p.fCT
        img = image.reshape(-1, 3)
        mask = np.zeros(h * w, np.uint8)
        with np.errstate(all='ignore'):
            for x in range(h * w):
                data = img[x].astype(F)
                gw = [p.w[k, x] for k in range(NMIX)]; gv = [p.var[k, x] for k in range(NMIX)]
                mean = [p.mu[k, :, x].copy() for k in range(NMIX)]
                nmodes = int(p.nmodes[x])
                background = fits = False
                total = F(0)
                mode = 0
                while mode < nmodes:
                    weight = alpha1 * gw[mode] + prune
                    swap_count = 0
                    if not fits:
                        var = gv[mode]
                        d = mean[mode] - data
                        dist2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
                        if total < p.TB and dist2 < p.Tb * var:
                            background = True
                        if dist2 < p.Tg * var:
                            fits = True
                            weight = weight + alphaT
                            k = alphaT / weight
                            for c in range(3):
                                mean[mode][c] = mean[mode][c] - k * d[c]
                            vn = var + k * (dist2 - var)
                            vn = p.var_min if vn < p.var_min else vn
                            gv[mode] = p.var_max if vn > p.var_max else vn
                            i = mode
                            while i > 0:
                                if weight < gw[i - 1]:
                                    break
                                swap_count += 1
                                gw[i], gw[i - 1] = gw[i - 1], gw[i]
                                gv[i], gv[i - 1] = gv[i - 1], gv[i]
                                mean[i], mean[i - 1] = mean[i - 1], mean[i]
                                i -= 1
                    if weight < -prune:
                        weight = F(0)
                        nmodes -= 1
                    gw[mode - swap_count] = weight
                    total = total + weight
                    mode += 1
                inv = F(1.0) / total if abs(total) > FLT_EPSILON else F(0)
                for k in range(nmodes):
                    gw[k] = gw[k] * inv
                if not fits and alphaT > 0:
                    if nmodes == NMIX:
                        mode = NMIX - 1
                    else:
                        mode = nmodes
                        nmodes += 1
                    if nmodes == 1:
                        gw[mode] = F(1.0)
                    else:
                        gw[mode] = alphaT
                        for i in range(nmodes - 1):
                            gw[i] = gw[i] * alpha1
                    mean[mode] = data.copy()
                    gv[mode] = p.var_init
                    i = nmodes - 1
                    while i > 0:
                        if alphaT < gw[i - 1]:
                            break
                        gw[i], gw[i - 1] = gw[i - 1], gw[i]
                        gv[i], gv[i - 1] = gv[i - 1], gv[i]
                        mean[i], mean[i - 1] = mean[i - 1], mean[i]
                        i -= 1
                for k in range(NMIX):
                    p.w[k, x] = gw[k]; p.var[k, x] = gv[k]; p.mu[k, :, x] = mean[k]
                p.nmodes[x] = nmodes
                if background:
                    mask[x] = 0
                elif p.detect_shadows and self._shadow(data, nmodes, gw, gv, mean):
                    mask[x] = p.shadow_val
                else:
                    mask[x] = 255
        return mask.reshape(h, w)

    def _shadow(self, data, nmodes, gw, gv, mean):
        p = self.p
        tw = F(0)
        for mode in range(nmodes):
            num = F(0); den = F(0)
            for c in range(3):
                num = num + data[c] * mean[mode][c]
                den = den + mean[mode][c] * mean[mode][c]
            if den == 0:
                return False
            if num <= den and num >= p.tau * den:
                a = num / den
                d2a = F(0)
                for c in range(3):
                    q = a * mean[mode][c] - data[c]
                    d2a = d2a + q * q
                if d2a < p.Tb * gv[mode] * a * a:
                    return True
            tw = tw + gw[mode]
            if tw > p.TB:
                return False
        return False
