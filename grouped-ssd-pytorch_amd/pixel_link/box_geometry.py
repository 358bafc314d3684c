"""The box tail of ``mask_to_box`` (ssd_liverdet/pixel_link/postprocess.py:124-160) as a documented, testable definition.

The reference hands each labelled component to OpenCV: nearest-neighbour resize of the label map to the image size, ``findContours`` ->
``contours[0]``, ``minAreaRect`` (:45-51), ``boxPoints`` + ``np.int0`` + clamping (:53-80), then the axis-aligned bounds of the four
corners, and the mean of the bilinearly resized positive-class probability over the component's pixels as its score.  OpenCV is not in
this image and the reference holds no fixture of that tail, so nothing can pin cv2's tie-breaking; what CAN be pinned is the geometry those
calls compute, and that is what this module defines (tests/test_pixellink_cpu.py holds closed-form cases and a brute-force cross-check):

* ``upscale_nearest``   -- cv2.INTER_NEAREST: destination pixel (y, x) reads source (floor(y * h / H), floor(x * w / W)).
* ``upscale_bilinear``  -- cv2.INTER_LINEAR: source coordinate (x + 0.5) * w / W - 0.5, clamped to the border, fp32 weights.
* ``min_area_rect``     -- the minimum-area enclosing rectangle of a set of pixel CENTRES (what minAreaRect returns for the outer contour of a
  component: the contour's points are boundary pixel coordinates and the hull of the boundary is the hull of the component).  Convex hull by
  Andrew's monotone chain on the integer coordinates, then rotating calipers: one candidate rectangle per hull edge, the smallest area wins,
  the FIRST such edge in hull order (monotone chain: it starts at the leftmost-then-topmost point) breaks ties.  Degenerate sets: one point ->
  a 0 x 0 rectangle; collinear points -> zero height (the reference's ``min_height`` / ``min_area`` filters drop both).
* ``rect_corners_int``  -- the corners truncated toward zero like ``np.int0(cv2.boxPoints(rect))`` (after rounding to 1e-6 so that corners which
  are integers in exact arithmetic do not lose a pixel to a 1e-16 rounding error; cv2's float32 trigonometry has no such guarantee -- this is the
  one place where the definition is deliberately cleaner than the library) and clamped into the image (:58-69).
Host code, numpy only: the reference runs this tail on the CPU too; the connected components themselves come from the device
(gssd_pixellink_decode_f32)."""
import numpy as np


def upscale_nearest(a, out_hw):
    h, w = a.shape
    H, W = out_hw
    ys = np.minimum((np.arange(H) * (h / H)).astype(np.int64), h - 1)
    xs = np.minimum((np.arange(W) * (w / W)).astype(np.int64), w - 1)
    return a[ys[:, None], xs[None, :]]


def upscale_bilinear(a, out_hw):
    a = np.asarray(a, np.float32)
    h, w = a.shape
    H, W = out_hw

    def axis(n_in, n_out):
        s = (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * np.float32(n_in / n_out) - np.float32(0.5)
        i0 = np.floor(s).astype(np.int64)
        f = (s - i0).astype(np.float32)
        lo, hi = np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1)
        f = np.where(i0 < 0, np.float32(0.0), f)
        return lo, hi, f
    y0, y1, fy = axis(h, H)
    x0, x1, fx = axis(w, W)
    top = a[y0][:, x0] * (1 - fx)[None, :] + a[y0][:, x1] * fx[None, :]
    bot = a[y1][:, x0] * (1 - fx)[None, :] + a[y1][:, x1] * fx[None, :]
    return (top * (1 - fy)[:, None] + bot * fy[:, None]).astype(np.float32)


def convex_hull(points):
    """Counter-clockwise hull (in image coordinates with y down: the signed area is positive for this orientation) of integer points [n, 2]
    as (x, y); collinear boundary points dropped; starts at the lowest-x-then-lowest-y point."""
    pts = np.unique(np.asarray(points, np.int64).reshape(-1, 2), axis=0)         # sorted by x, then y
    if len(pts) <= 2:
        return pts

    def half(seq):
        out = []
        for p in seq:
            while len(out) >= 2:
                (ax, ay), (bx, by) = out[-2], out[-1]
                if (bx - ax) * (p[1] - ay) - (by - ay) * (p[0] - ax) <= 0:
                    out.pop()
                else:
                    break
            out.append((int(p[0]), int(p[1])))
        return out
    lower, upper = half(pts), half(pts[::-1])
    hull = lower[:-1] + upper[:-1]
    return np.asarray(hull, np.int64)


def min_area_rect(points):
    """-> dict(center (cx, cy), size (w, h) with w along the chosen edge, corners float64 [4, 2], area)."""
    hull = convex_hull(points).astype(np.float64)
    n = len(hull)
    if n == 0:
        raise ValueError('empty point set')
    if n == 1:
        return dict(center=(hull[0, 0], hull[0, 1]), size=(0.0, 0.0), corners=np.repeat(hull, 4, axis=0), area=0.0)
    best = None
    for i in range(n if n > 2 else 1):
        p, q = hull[i], hull[(i + 1) % n]
        e = q - p
        u = e / np.hypot(e[0], e[1])
        v = np.array([-u[1], u[0]])
        a, b = hull @ u, hull @ v
        a0, a1, b0, b1 = a.min(), a.max(), b.min(), b.max()
        area = (a1 - a0) * (b1 - b0)
        if best is None or area < best[0] - 1e-9 * max(1.0, best[0]):
            best = (area, u, v, a0, a1, b0, b1)
    area, u, v, a0, a1, b0, b1 = best
    corners = np.stack([a0 * u + b0 * v, a1 * u + b0 * v, a1 * u + b1 * v, a0 * u + b1 * v])
    c = 0.5 * (a0 + a1) * u + 0.5 * (b0 + b1) * v
    return dict(center=(float(c[0]), float(c[1])), size=(float(a1 - a0), float(b1 - b0)), corners=corners, area=float(area))


def rect_corners_int(corners, image_shape):
    h, w = image_shape[:2]
    c = np.trunc(np.round(np.asarray(corners, np.float64), 6)).astype(np.int64)
    c[:, 0] = np.clip(c[:, 0], 0, w - 1)
    c[:, 1] = np.clip(c[:, 1], 0, h - 1)
    return c


def component_boxes(labels, prob, img_shape, min_height, min_area):
    """postprocess.py:124-160 for ONE image: ``labels`` int [h, w] (0 = background), ``prob`` fp32 [h, w] positive-class probability ->
    ([[min_x, min_y, max_x, max_y], ...], [score, ...]) in image pixels, components in label order, the reference's filters applied."""
    res = upscale_nearest(np.asarray(labels), img_shape)
    score_map = upscale_bilinear(prob, img_shape)
    boxes, scores = [], []
    for i in range(1, int(res.max()) + 1):
        ys, xs = np.nonzero(res == i)
        if len(ys) == 0:
            continue
        r = min_area_rect(np.stack([xs, ys], 1))
        w_, h_ = r['size']
        if min(w_, h_) < min_height or r['area'] < min_area:
            continue
        c = rect_corners_int(r['corners'], img_shape)
        boxes.append([int(c[:, 0].min()), int(c[:, 1].min()), int(c[:, 0].max()), int(c[:, 1].max())])
        scores.append(float(score_map[ys, xs].mean()))
    return boxes, scores
