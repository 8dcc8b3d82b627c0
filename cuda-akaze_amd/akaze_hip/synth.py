"""Seeded synthetic test scenes: the reference's data/img1.png / img2.png are not
in its checkout, so every benchmark / parity input at 1080p, 4K and 720p is generated here.

scene(w, h, seed)  -> uint8 image: smooth background gradient + random rectangles / discs / line
                     segments of random contrast + Gaussian noise, lightly blurred.
pair(w, h, seed)   -> (img1, img2): img2 = img1 warped by a fixed small homography
                     (3 deg rotation, 1.05 scale, 20 px shift), bilinear.
to_float(u8)       -> float32 in [0,1] exactly as main.cpp:149 (convertTo(CV_32FC1, 1.0/255.0)).
random_descriptors -> AkazePoint arrays for the 10k x 10k matcher config.
"""
import numpy as np


def scene(w, h, seed, nshapes=None):
    rng = np.random.default_rng(seed)
    if nshapes is None:
        nshapes = max(25, int(175 * (w * h) / (1920.0 * 1080.0)))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = 0.35 + 0.25 * (xx / w) + 0.15 * (yy / h)
    for _ in range(nshapes):
        kind = rng.integers(0, 3)
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        amp = rng.uniform(0.08, 0.45) * (1 if rng.random() < 0.5 else -1)
        if kind == 0:       # rectangle
            rw, rh = rng.uniform(8, 0.12 * w), rng.uniform(8, 0.12 * h)
            x0, x1 = int(max(0, cx - rw / 2)), int(min(w, cx + rw / 2))
            y0, y1 = int(max(0, cy - rh / 2)), int(min(h, cy + rh / 2))
            img[y0:y1, x0:x1] += amp
        elif kind == 1:     # disc
            r = rng.uniform(5, 0.06 * min(w, h) + 6)
            x0, x1 = int(max(0, cx - r)), int(min(w, cx + r + 1))
            y0, y1 = int(max(0, cy - r)), int(min(h, cy + r + 1))
            if x1 > x0 and y1 > y0:
                sub = (xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2 <= r * r
                img[y0:y1, x0:x1] += amp * sub
        else:               # thick line segment
            ang = rng.uniform(0, np.pi)
            ln = rng.uniform(20, 0.15 * w)
            th = rng.uniform(1.5, 5.0)
            dx, dy = np.cos(ang), np.sin(ang)
            r = ln / 2 + th
            x0, x1 = int(max(0, cx - r)), int(min(w, cx + r + 1))
            y0, y1 = int(max(0, cy - r)), int(min(h, cy + r + 1))
            if x1 > x0 and y1 > y0:
                px = xx[y0:y1, x0:x1] - cx
                py = yy[y0:y1, x0:x1] - cy
                along = px * dx + py * dy
                across = -px * dy + py * dx
                img[y0:y1, x0:x1] += amp * ((np.abs(along) <= ln / 2) & (np.abs(across) <= th))
    img += rng.normal(0.0, 0.01, size=img.shape).astype(np.float32)
    # light 3x3 binomial blur (camera-like edges)
    p = np.pad(img, 1, mode="edge")
    img = (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:] + 2 * p[1:-1, :-2] + 4 * p[1:-1, 1:-1] + 2 * p[1:-1, 2:] +
           p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) / 16.0
    return np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8)


def warp(u8, angle_deg=3.0, scale=1.05, shift=(20.0, 20.0)):
    h, w = u8.shape
    a = np.deg2rad(angle_deg)
    ca, sa = np.cos(a) * scale, np.sin(a) * scale
    cx, cy = w / 2.0, h / 2.0
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    # inverse map: destination pixel -> source coordinate
    X, Y = xx - cx - shift[0], yy - cy - shift[1]
    det = ca * ca + sa * sa
    sx = (ca * X + sa * Y) / det + cx
    sy = (-sa * X + ca * Y) / det + cy
    sx = np.clip(sx, 0, w - 1.001)
    sy = np.clip(sy, 0, h - 1.001)
    x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    fx, fy = sx - x0, sy - y0
    f = u8.astype(np.float64)
    out = (f[y0, x0] * (1 - fx) * (1 - fy) + f[y0, x0 + 1] * fx * (1 - fy) +
           f[y0 + 1, x0] * (1 - fx) * fy + f[y0 + 1, x0 + 1] * fx * fy)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def pair(w, h, seed, nshapes=None):
    a = scene(w, h, seed, nshapes)
    return a, warp(a)


def to_float(u8, pitch=None):
    """uint8 -> float32 [0,1] (main.cpp:149), optionally padded to `pitch` columns."""
    f = (u8.astype(np.float64) * (1.0 / 255.0)).astype(np.float32)
    if pitch is None or pitch == u8.shape[1]:
        return np.ascontiguousarray(f)
    out = np.zeros((u8.shape[0], pitch), np.float32)
    out[:, :u8.shape[1]] = f
    return out


def random_descriptors(n, seed, dtype, planted_from=None, nplanted=0, maxflip=40):
    """n AkazePoints with uniformly random 486-bit descriptors (byte 60: top 2 bits zero)."""
    rng = np.random.default_rng(seed)
    pts = np.zeros(n, dtype)
    pts["features"] = rng.integers(0, 256, size=(n, 61), dtype=np.uint8)
    pts["features"][:, 60] &= 0x3F
    pts["x"] = rng.uniform(0, 1920, n).astype(np.float32)
    pts["y"] = rng.uniform(0, 1080, n).astype(np.float32)
    pts["match"] = -7
    if planted_from is not None and nplanted > 0:
        src = rng.choice(len(planted_from), nplanted, replace=False)
        dst = rng.choice(n, nplanted, replace=False)
        f = planted_from["features"][src].copy()
        for i in range(nplanted):
            bits = rng.choice(486, rng.integers(0, maxflip + 1), replace=False)
            for b in bits:
                f[i, b >> 3] ^= np.uint8(1 << (b & 7))
        pts["features"][dst] = f
    return pts
