"""Oracle: the strong / weak image augmentation of the USB CV datasets (TEST INFRASTRUCTURE, see oracle/__init__.py).

numpy restatement, on uint8 HWC arrays, of
  semilearn/datasets/augmentation/randaugment.py   the 14 RandAugment ops (:16-113), Cutout (:116-146), RandAugment.__call__ (:189-196)
  semilearn/datasets/cv_datasets/cifar.py:34-49     transform_weak / transform_strong: RandomCrop(reflect padding) + RandomHorizontalFlip
                                                    [+ RandAugment(3, 5)] + ToTensor + Normalize
The ops are thin wrappers of Pillow (third-party; requirements.txt leaves it unpinned, the build container has 12.2.0), whose integer
algorithms are restated here: ImageOps.autocontrast / equalize / posterize / solarize (lookup tables), ImageEnhance.* = Image.blend with a
degenerate image (float32 interpolation, truncation), ImageFilter.SMOOTH (3x3 / 13, borders copied), Image.rotate / transform(AFFINE) with
NEAREST resampling (16.16 fixed-point walk of Geometry.c), ImageDraw.rectangle.  Pinned bit for bit by tests/golden/augment.npz, produced by
calling the reference's own functions.  torchvision is absent from the build container: RandomCrop / RandomHorizontalFlip / ToTensor /
Normalize are restated from their documented semantics (np.pad(mode='reflect') crop window, x[..., ::-1], /255, (x - mean) / std) -- that
part is parity-unpinned.  Randomness (op choice, magnitudes, crop offsets, flips, cutout position) is an INPUT everywhere.
"""
import math

import numpy as np

OPS = ["AutoContrast", "Brightness", "Color", "Contrast", "Equalize", "Identity", "Posterize", "Rotate", "Sharpness", "ShearX", "ShearY",
       "Solarize", "TranslateX", "TranslateY"]                       # augment_list() order (:149-166)
RANGES = [(0, 1), (0.05, 0.95), (0.05, 0.95), (0.05, 0.95), (0, 1), (0, 1), (4, 8), (-30, 30), (0.05, 0.95), (-0.3, 0.3), (-0.3, 0.3),
          (0, 256), (-0.3, 0.3), (-0.3, 0.3)]
CUTOUT_COLOR = (125, 123, 114)


def _lut(img, luts):
    return np.stack([luts[c][img[..., c]] for c in range(img.shape[-1])], axis=-1).astype(np.uint8)


def autocontrast(img):
    luts = []
    for c in range(3):
        h = np.bincount(img[..., c].ravel(), minlength=256)
        nz = np.nonzero(h)[0]
        lo, hi = int(nz[0]), int(nz[-1])
        if hi <= lo:
            luts.append(np.arange(256))
        else:
            scale = 255.0 / (hi - lo)
            offset = -lo * scale
            luts.append(np.array([min(255, max(0, int(ix * scale + offset))) for ix in range(256)]))
    return _lut(img, luts)


def equalize(img):
    luts = []
    for c in range(3):
        h = np.bincount(img[..., c].ravel(), minlength=256).tolist()
        histo = [v for v in h if v]
        step = (sum(histo) - histo[-1]) // 255 if len(histo) > 1 else 0
        if not step:
            luts.append(np.arange(256))
        else:
            n, l = step // 2, []
            for i in range(256):
                l.append(min(255, n // step))          # Image.point clips the table to 8 bits (bins above the last occupied one reach 256)
                n += h[i]
            luts.append(np.array(l))
    return _lut(img, luts)


def posterize(img, v):
    bits = max(1, int(v))
    return img & np.uint8(~(2 ** (8 - bits) - 1) & 0xFF)


def solarize(img, v):
    return np.where(img < v, img, 255 - img).astype(np.uint8)


def blend(deg, img, alpha):
    """Image.blend(deg, img, alpha) of Pillow's Blend.c for 0 <= alpha <= 1: (UINT8)((int)a + alpha * ((int)b - (int)a)) in float32."""
    a, b, al = deg.astype(np.float32), img.astype(np.float32), np.float32(alpha)
    return (a + al * (b - a)).astype(np.float32).astype(np.int32).astype(np.uint8)


def gray(img):
    r, g, b = (img[..., c].astype(np.int64) for c in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def brightness(img, v):
    return blend(np.zeros_like(img), img, v)


def color(img, v):
    return blend(np.repeat(gray(img)[..., None], 3, axis=-1), img, v)


def contrast(img, v):
    g = gray(img)
    mean = int(float(g.astype(np.int64).sum()) / g.size + 0.5)
    return blend(np.full_like(img, mean), img, v)


def smooth(img):
    """ImageFilter.SMOOTH: 3x3 kernel (1,1,1,1,5,1,1,1,1) / 13, float32 accumulation row by row (top, middle, bottom) starting from 0.5,
    truncation; the 1-pixel border is copied."""
    k = (np.array([1, 1, 1, 1, 5, 1, 1, 1, 1], dtype=np.float32) / np.float32(13.0)).astype(np.float32)
    out = img.copy()
    f = img.astype(np.float32)
    H, W = img.shape[:2]
    ss = np.full((H - 2, W - 2, 3), np.float32(0.5), dtype=np.float32)
    for r, dy in enumerate((1, 0, -1)):                      # Filter.c walks in1 (y+1), in0 (y), in_1 (y-1) with kernel rows 0, 1, 2
        row = f[1 + dy:H - 1 + dy]
        t = (row[:, 0:W - 2] * k[3 * r] + row[:, 1:W - 1] * k[3 * r + 1]).astype(np.float32)
        t = (t + row[:, 2:W] * k[3 * r + 2]).astype(np.float32)
        ss = (ss + t).astype(np.float32)
    out[1:H - 1, 1:W - 1] = np.clip(ss, 0, 255).astype(np.int32).astype(np.uint8)
    return out


def sharpness(img, v):
    return blend(smooth(img), img, v)


def _fix(v):
    return int(math.floor(v * 65536.0 + 0.5))


def affine_nearest(img, a):
    """img.transform(size, AFFINE, a) with NEAREST resampling, fill 0: Geometry.c affine_fixed (16.16 fixed point), or ImagingScaleAffine
    when there is no shear / rotation (a[1] == a[3] == 0)."""
    H, W = img.shape[:2]
    out = np.zeros_like(img)
    if a[1] == 0 and a[3] == 0:
        xo = a[2] + a[0] * 0.5
        yo = a[5] + a[4] * 0.5
        # ImagingScaleAffine accumulates xo += a[0] (float64)
        xi, v = [], xo
        for _ in range(W):
            xi.append(-1 if v < 0 else int(v)); v += a[0]
        yi, v = [], yo
        for _ in range(H):
            yi.append(-1 if v < 0 else int(v)); v += a[4]
        for y in range(H):
            if 0 <= yi[y] < H:
                for x in range(W):
                    if 0 <= xi[x] < W:
                        out[y, x] = img[yi[y], xi[x]]
        return out
    a0, a1, a3, a4 = _fix(a[0]), _fix(a[1]), _fix(a[3]), _fix(a[4])
    a2 = _fix(a[2] + a[0] * 0.5 + a[1] * 0.5)
    a5 = _fix(a[5] + a[3] * 0.5 + a[4] * 0.5)
    for y in range(H):
        xx, yy = a2, a5
        for x in range(W):
            xin, yin = xx >> 16, yy >> 16
            if 0 <= xin < W and 0 <= yin < H:
                out[y, x] = img[yin, xin]
            xx += a0; yy += a3
        a2 += a1; a5 += a4
    return out


def rotate_matrix(v, W, H):
    """Image.rotate(angle): the inverse affine matrix about the image centre (Image.py), entries rounded to 15 decimals."""
    ang = v % 360.0
    ang = -math.radians(ang)
    m = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0, round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]
    cx, cy = W / 2.0, H / 2.0
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2]
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5]
    m[2] += cx; m[5] += cy
    return m


def rotate(img, v):
    H, W = img.shape[:2]
    ang = v % 360.0
    if ang == 0:
        return img.copy()
    return affine_nearest(img, rotate_matrix(v, W, H))      # (magnitudes are drawn from (-30, 30): Image.rotate's 90-degree shortcuts never fire)


def affine_op(name, img, v):
    H, W = img.shape[:2]
    m = {"ShearX": (1, v, 0, 0, 1, 0), "ShearY": (1, 0, 0, v, 1, 0), "TranslateX": (1, 0, v * W, 0, 1, 0), "TranslateY": (1, 0, 0, 0, 1, v * H)}[name]
    return affine_nearest(img, m)


def cutout(img, v, ux, uy):
    """Cutout (:116-146): v in [0, 0.5] (fraction of the width); ux, uy = the two np.random.uniform(w) / uniform(h) draws."""
    if v <= 0.0:
        return img
    H, W = img.shape[:2]
    v = v * W
    x0, y0 = int(max(0, ux - v / 2.0)), int(max(0, uy - v / 2.0))
    x1, y1 = min(W, x0 + v), min(H, y0 + v)
    out = img.copy()
    # ImageDraw.rectangle((x0, y0, x1, y1), fill): integer coordinates truncated, both ends inclusive, clipped to the image
    xa, ya, xb, yb = int(x0), int(y0), int(x1), int(y1)
    out[max(ya, 0):min(yb, H - 1) + 1, max(xa, 0):min(xb, W - 1) + 1] = np.array(CUTOUT_COLOR, dtype=np.uint8)
    return out


def apply_op(op, img, v):
    name = OPS[op] if isinstance(op, (int, np.integer)) else op
    if name == "AutoContrast":
        return autocontrast(img)
    if name == "Brightness":
        return brightness(img, v)
    if name == "Color":
        return color(img, v)
    if name == "Contrast":
        return contrast(img, v)
    if name == "Equalize":
        return equalize(img)
    if name == "Identity":
        return img
    if name == "Posterize":
        return posterize(img, v)
    if name == "Rotate":
        return rotate(img, v)
    if name == "Sharpness":
        return sharpness(img, v)
    if name == "Solarize":
        return solarize(img, v)
    return affine_op(name, img, v)


def crop_flip(img, pad, size, i, j, flip):
    """RandomCrop(size, padding=pad, padding_mode='reflect') at offset (i, j) of the padded image, then RandomHorizontalFlip."""
    p = np.pad(img, ((pad, pad), (pad, pad), (0, 0)), mode="reflect")
    c = p[i:i + size, j:j + size]
    return np.ascontiguousarray(c[:, ::-1] if flip else c)


def to_tensor_normalize(img, mean, std):
    """ToTensor + Normalize: float32 CHW, (x / 255 - mean) / std."""
    x = img.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    return ((x - np.asarray(mean, np.float32)[:, None, None]) / np.asarray(std, np.float32)[:, None, None]).astype(np.float32)


def strong(img, pad, size, i, j, flip, ops, vals, cut_v, ux, uy, mean, std):
    """transform_strong (cifar.py:42-49) with every random draw given: ops / vals = the n RandAugment picks and magnitudes."""
    x = crop_flip(img, pad, size, i, j, flip)
    for o, v in zip(ops, vals):
        x = apply_op(int(o), x, float(v))
    x = cutout(x, cut_v, ux, uy)
    return to_tensor_normalize(x, mean, std)


def weak(img, pad, size, i, j, flip, mean, std):
    return to_tensor_normalize(crop_flip(img, pad, size, i, j, flip), mean, std)
