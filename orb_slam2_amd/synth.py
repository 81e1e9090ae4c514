"""Seeded synthetic grayscale frames for tests and bench (datasets are not available).

image i = 3-octave value noise (uniform u8 grids at 1/8, 1/16, 1/32 resolution, bilinearly upsampled, weights
0.5/0.3/0.2), over-painted by 400 random shapes (axis-aligned rectangles, ellipses, rotated rectangles; side
6..60 px; uniform grey), plus uniform integer noise in [-6, 6], clipped to u8.  Consecutive frames of a
sequence are the same scene translated by (3, 1) px with fresh noise, so the frame-to-frame matcher finds matches.
"""
import numpy as np

BASE_SEED = 20260921


def _upsample(grid, h, w):
    gh, gw = grid.shape
    ys = (np.arange(h) + 0.5) * gh / h - 0.5
    xs = (np.arange(w) + 0.5) * gw / w - 0.5
    y0 = np.clip(np.floor(ys).astype(int), 0, gh - 1)
    x0 = np.clip(np.floor(xs).astype(int), 0, gw - 1)
    y1 = np.clip(y0 + 1, 0, gh - 1)
    x1 = np.clip(x0 + 1, 0, gw - 1)
    fy = np.clip(ys - np.floor(ys), 0, 1)[:, None]
    fx = np.clip(xs - np.floor(xs), 0, 1)[None, :]
    g = grid.astype(np.float64)
    top = g[y0][:, x0] * (1 - fx) + g[y0][:, x1] * fx
    bot = g[y1][:, x0] * (1 - fx) + g[y1][:, x1] * fx
    return top * (1 - fy) + bot * fy


def scene(width, height, seed=0, margin=64, nshapes=400):
    """Noise-free scene, (height+2*margin) x (width+2*margin) float64, so translated crops stay in-bounds."""
    rng = np.random.default_rng(BASE_SEED + seed)
    H, W = height + 2 * margin, width + 2 * margin
    img = np.zeros((H, W))
    for div, wgt in ((8, 0.5), (16, 0.3), (32, 0.2)):
        grid = rng.integers(0, 256, size=(max(H // div, 2), max(W // div, 2)))
        img += wgt * _upsample(grid, H, W)
    n = int(nshapes * (H * W) / (1241.0 * 376.0 + 1e-9) * 0.75) if (width, height) != (1241, 376) else nshapes
    n = max(n, 40)
    for _ in range(n):
        kind = rng.integers(0, 3)
        cx, cy = rng.integers(0, W), rng.integers(0, H)
        a, b = rng.integers(6, 61) / 2.0, rng.integers(6, 61) / 2.0
        grey = float(rng.integers(0, 256))
        r = int(np.ceil(np.hypot(a, b))) + 1
        x0, x1 = max(cx - r, 0), min(cx + r + 1, W)
        y0, y1 = max(cy - r, 0), min(cy + r + 1, H)
        if x0 >= x1 or y0 >= y1:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1]
        dx, dy = xx - cx, yy - cy
        if kind == 0:
            m = (np.abs(dx) <= a) & (np.abs(dy) <= b)
        elif kind == 1:
            m = (dx / a) ** 2 + (dy / b) ** 2 <= 1.0
        else:
            th = rng.uniform(0, np.pi)
            c, s = np.cos(th), np.sin(th)
            u, v = dx * c + dy * s, -dx * s + dy * c
            m = (np.abs(u) <= a) & (np.abs(v) <= b)
        img[y0:y1, x0:x1][m] = grey
    return img


def frame_from_scene(sc, width, height, t=0, seed=0, margin=64, shift=(3, 1)):
    """Frame t of the sequence: crop translated by t*shift, plus fresh uniform noise in [-6, 6]."""
    rng = np.random.default_rng((BASE_SEED + seed) * 1000003 + 7919 * t + 1)
    ox = margin + (t * shift[0]) % (2 * margin - 1) - margin // 2
    oy = margin + (t * shift[1]) % (2 * margin - 1) - margin // 2
    crop = sc[oy:oy + height, ox:ox + width]
    noise = rng.integers(-6, 7, size=(height, width))
    return np.clip(np.rint(crop) + noise, 0, 255).astype(np.uint8)


def frame(width, height, seed=0, t=0):
    return frame_from_scene(scene(width, height, seed), width, height, t=t, seed=seed)


def sequence(width, height, nframes, seed=0):
    sc = scene(width, height, seed)
    return [frame_from_scene(sc, width, height, t=t, seed=seed) for t in range(nframes)]


def stereo_sequence(width, height, nframes, fx, bf, seed=0, disp_near=24, disp_far=8, step=(0.25, 0.125), return_depth=False):
    """A rectified stereo stream of a two-layer scene with a known camera trajectory (the input of a front-end loop).

    Two fronto-parallel textured planes: a far one (disparity disp_far px, depth bf/disp_far) and, in front of it over about a third of
    the view, a near one (disparity disp_near px).  The camera translates by step = (x, y) baselines per frame without rotating, so a
    plane of disparity d moves by exactly d*step px per frame (disparities divisible by 8 keep every shift an integer: each image is an
    exact crop of the layer textures, composited near-over-far, plus fresh sensor noise).  Returns (lefts, rights, Tcw, Tpred):
    Tcw[k] = the true world-to-camera pose of frame k, float32 4x4 (world = camera 0); Tpred[k] = the constant-velocity prediction
    Tcw[k-1] * inv(Tcw[k-2]) * Tcw[k-1] a motion model would hand the matcher (k >= 2; Tpred[1] = Tcw[0], Tpred[0] = Tcw[0])."""
    assert disp_near % 8 == 0 and disp_far % 8 == 0
    margin = int(disp_near * (nframes * max(step) + 1)) + 16
    far = scene(width, height, seed=seed, margin=margin)
    near = scene(width, height, seed=seed + 1, margin=margin)
    rng = np.random.default_rng(BASE_SEED + 555 + seed)
    H, W = far.shape
    mask = np.zeros((H, W), bool)                       # where the near plane exists, in its own texture coordinates
    for _ in range(max(4, (W * H) // 90000)):
        bw, bh = int(rng.integers(W // 10, W // 4)), int(rng.integers(H // 6, H // 2))
        x0, y0 = int(rng.integers(0, W - bw)), int(rng.integers(0, H - bh))
        mask[y0:y0 + bh, x0:x0 + bw] = True
    b = bf / fx
    lefts, rights, poses, depths = [], [], [], []
    for k in range(nframes):
        for cam, out in ((0.0, lefts), (1.0, rights)):
            px, py = k * step[0] + cam, k * step[1]     # camera position in baselines
            def crop(tex, d):
                ox, oy = margin // 2 + int(round(d * px)), margin // 2 + int(round(d * py))
                return tex[oy:oy + height, ox:ox + width]
            img = np.where(crop(mask, disp_near), crop(near, disp_near), crop(far, disp_far))
            nrng = np.random.default_rng((BASE_SEED + seed) * 1000003 + 7919 * k + 31 * int(cam) + 5)
            out.append(np.clip(np.rint(img) + nrng.integers(-6, 7, size=(height, width)), 0, 255).astype(np.uint8))
            if cam == 0.0:                              # what the left camera sees, in metres along the optical axis: bf / disparity of the visible plane
                depths.append(np.where(crop(mask, disp_near), np.float32(bf / disp_near), np.float32(bf / disp_far)).astype(np.float32))
        T = np.eye(4, dtype=np.float32)
        T[0, 3], T[1, 3] = np.float32(-k * step[0] * b), np.float32(-k * step[1] * b)
        poses.append(T)
    pred = [poses[0], poses[0]] + [(poses[k - 1] @ np.linalg.inv(poses[k - 2]) @ poses[k - 1]).astype(np.float32) for k in range(2, nframes)]
    if return_depth:                                    # (+ the left camera's true depth maps: the depth sensor of an RGB-D sequence, the scene knowledge of a monocular one)
        return lefts, rights, poses, pred[:nframes], depths
    return lefts, rights, poses, pred[:nframes]


# degenerate cases the parity tests cover
def zeros(width, height):
    return np.zeros((height, width), np.uint8)


def checkerboard(width, height, cell=8, lo=40, hi=210):
    yy, xx = np.mgrid[0:height, 0:width]
    return np.where(((yy // cell) + (xx // cell)) % 2 == 0, lo, hi).astype(np.uint8)


def ramp(width, height):
    xx = np.arange(width)[None, :] * 255.0 / max(width - 1, 1)
    yy = np.arange(height)[:, None] * 64.0 / max(height - 1, 1)
    return np.clip(xx * 0.75 + yy, 0, 255).astype(np.uint8)


def low_texture(width, height, seed=3):
    """Smooth image with a few weak blobs: exercises the minThFAST fallback and < nfeatures outputs."""
    rng = np.random.default_rng(BASE_SEED + 77 + seed)
    img = np.full((height, width), 120.0)
    for _ in range(12):
        cx, cy = rng.integers(30, width - 30), rng.integers(30, height - 30)
        s = rng.integers(3, 9)
        img[cy - s:cy + s, cx - s:cx + s] += rng.integers(9, 30)
    return np.clip(img, 0, 255).astype(np.uint8)


def descriptor_db(n_keyframes, per_kf, seed=7):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(n_keyframes * per_kf, 32), dtype=np.uint8)


def descriptor_query(db, nq, seed=7, planted_frac=0.5, max_flips=20):
    """Query set with a fraction of planted near-duplicates of DB rows (<= max_flips bit flips)."""
    rng = np.random.default_rng(seed + 1)
    q = rng.integers(0, 256, size=(nq, 32), dtype=np.uint8)
    planted = rng.random(nq) < planted_frac
    src = rng.integers(0, len(db), size=nq)
    for i in np.nonzero(planted)[0]:
        d = db[src[i]].copy()
        for bit in rng.integers(0, 256, size=rng.integers(0, max_flips + 1)):
            d[bit >> 3] ^= np.uint8(1 << (bit & 7))
        q[i] = d
    return q
