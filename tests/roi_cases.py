"""Hand-derived known answers for torchvision.ops.RoIAlign(output_size=(8, 8), spatial_scale, sampling_ratio=0),
aligned=False, as the reference constructs and calls it (model/rcnn_discriminator_app.py:98-99,131-146).

Nothing here calls the oracle or the HIP kernel: every expected value is a closed form written out below (constants,
bin-centre values of linear ramps, literal numbers worked by hand), or -- for the randomised cross-check -- the
"tent function" statement of bilinear interpolation evaluated in float64, which shares no code with
oracle/model.py::roi_align (floor/lerp gather) or csrc/roi_align.hip.

The published algorithm (torchvision/csrc/ops/cpu/roi_align_kernel.cpp, aligned=False) in words:
  roi = box * scale (no half-pixel shift); roi_w = max(x2 - x1, 1), roi_h likewise; bin = roi / 8;
  grid = ceil(roi / 8) samples per bin and axis (sampling_ratio = 0);
  sample iy of bin ph sits at y = y1 + ph * bin_h + (iy + 0.5) * bin_h / grid_h;
  a sample with y < -1 or y > H (x likewise) contributes 0 but still counts in the mean;
  otherwise y is clamped to [0, H - 1] and the feature is bilinearly interpolated there;
  bin value = sum of the grid_h * grid_w samples / (grid_h * grid_w).
"""
import math

import numpy as np

P = 8


def _ramp(H, W, ax, ay, c0):
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    return ax * x + ay * y + c0


def _case(name, feat, rois, scale, expected):
    return dict(name=name, feat=np.asarray(feat, np.float32), rois=np.asarray(rois, np.float32), scale=scale,
                expected=np.asarray(expected, np.float64))


def known_answer_cases():
    cases = []

    # 1. constant map, interior ROI: every sample reads the constant -> every bin is the constant.
    cases.append(_case("constant", np.full((1, 2, 16, 16), 3.25), [[0, 8, 12, 40, 52]], 0.25, np.full((1, 2, P, P), 3.25)))

    # 2. linear ramps, every sample inside [0, 31]^2: bilinear interpolation reproduces a linear function exactly and
    #    a bin's samples are symmetric about its centre, so bin (ph, pw) = f(y1 + (ph + .5) bin_h, x1 + (pw + .5) bin_w).
    #    Box (10, 20)-(73, 99) px at 1/4: x1 = 2.5, y1 = 5, roi_w = 15.75 (grid 2), roi_h = 19.75 (grid 3).
    f0, f1 = _ramp(32, 32, 2.0, 3.0, 1.0), _ramp(32, 32, -1.0, 0.5, 0.0)
    x1, y1, bw, bh = 2.5, 5.0, 15.75 / 8, 19.75 / 8
    xc = x1 + (np.arange(P) + 0.5) * bw
    yc = y1 + (np.arange(P) + 0.5) * bh
    exp = np.stack([2.0 * xc[None, :] + 3.0 * yc[:, None] + 1.0, -xc[None, :] + 0.5 * yc[:, None]])[None]
    cases.append(_case("ramp_interior", np.stack([f0, f1])[None], [[0, 10, 20, 73, 99]], 0.25, exp))

    # 3. ROI narrower than one feature pixel: 1.2 x 2 px at 1/4 = 0.3 x 0.5 -> both clamped to 1 (NOT 0.3 / 0.5):
    #    bin = 1/8, grid = ceil(1/8) = 1, sample pw at x1 + (pw + .5) / 8. x1 = 20 * .25 = 5, y1 = 7.
    xc = 5.0 + (np.arange(P) + 0.5) / 8
    yc = 7.0 + (np.arange(P) + 0.5) / 8
    exp = (2.0 * xc[None, :] + 3.0 * yc[:, None] + 1.0)[None, None]
    cases.append(_case("roi_below_one_pixel", f0[None, None], [[0, 20, 28, 21.2, 30]], 0.25, exp))

    # 4. left edge, scale 1, f = x + 10 on a 16 x 16 map. Box x in [-2, 6]: roi_w = 8, bin 1, grid 1, samples at
    #    x = -1.5 (< -1: contributes 0), -0.5 (in (-1, 0): clamped to 0 -> 10), 0.5, 1.5, ..., 5.5.
    fx = _ramp(16, 16, 1.0, 0.0, 10.0)
    row = np.array([0.0, 10.0, 10.5, 11.5, 12.5, 13.5, 14.5, 15.5])
    cases.append(_case("left_edge_drop_and_clamp", fx[None, None], [[0, -2, 4, 6, 12]], 1.0, np.tile(row, (P, 1))[None, None]))
    #    Box x in [-1.5, 6.5]: samples at x = -1 exactly (NOT < -1: kept, clamped to 0 -> 10), 0, 1, ..., 6.
    row = np.array([10.0, 10.0, 11.0, 12.0, 13.0, 14.0, 15.0, 16.0])
    cases.append(_case("left_edge_exactly_minus_one", fx[None, None], [[0, -1.5, 4, 6.5, 12]], 1.0, np.tile(row, (P, 1))[None, None]))

    # 5. bottom edge, scale 1, f = y + 1, H = 16. Box y in [12.5, 20.5]: samples y = 13, 14, 15, 16, 17, ...:
    #    13 -> 14, 14 -> 15, 15 -> 16, y = 16 == H is NOT > H: kept and clamped to H - 1 -> 16, y >= 17 > H -> 0.
    fy = _ramp(16, 16, 0.0, 1.0, 1.0)
    col = np.array([14.0, 15.0, 16.0, 16.0, 0.0, 0.0, 0.0, 0.0])
    cases.append(_case("bottom_edge_y_equals_H_kept", fy[None, None], [[0, 3, 12.5, 11, 20.5]], 1.0, np.tile(col[:, None], (1, P))[None, None]))
    #    Box y in [12, 20]: samples 12.5, 13.5, 14.5, 15.5 (in (H-1, H): clamped to 15 -> 16), 16.5 > H -> 0, ...
    col = np.array([13.5, 14.5, 15.5, 16.0, 0.0, 0.0, 0.0, 0.0])
    cases.append(_case("bottom_edge_fraction_clamped", fy[None, None], [[0, 3, 12, 11, 20]], 1.0, np.tile(col[:, None], (1, P))[None, None]))

    # 6. adaptive grid. 100-px box at 1/8: x1 = 1, roi = 12.5, bin = 1.5625, grid = ceil(12.5 / 8) = 2.
    #    Feature = 1 in column x = 2, else 0 (constant in y, so the y interpolation sums to 1).
    #    bin 0 samples: x = 1 + .25 * 1.5625 = 1.390625 -> weight of column 2 = .390625; x = 1 + .75 * 1.5625 = 2.171875
    #    -> 1 - .171875 = .828125; mean .609375  (a grid of 1 would sample 1.78125 -> .78125).
    #    bin 1 samples: 2.953125 -> .046875; 3.734375 -> 0; mean .0234375. bins 2..7: 0.
    d = np.zeros((16, 16))
    d[:, 2] = 1.0
    row = np.array([0.609375, 0.0234375, 0, 0, 0, 0, 0, 0])
    cases.append(_case("grid2_for_100px_at_eighth", d[None, None], [[0, 8, 8, 108, 108]], 0.125, np.tile(row, (P, 1))[None, None]))
    #    9-px box at 1/4: x1 = 1, roi = 2.25, bin = .28125, grid = ceil(2.25 / 8) = 1: bin pw = f(1 + (pw + .5) * .28125).
    fr = _ramp(32, 32, 1.0, 0.0, 0.0)
    row = 1.0 + (np.arange(P) + 0.5) * 0.28125      # 1.140625, 1.421875, ...
    assert row[0] == 1.140625
    cases.append(_case("grid1_for_9px_at_quarter", fr[None, None], [[0, 4, 4, 13, 13]], 0.25, np.tile(row, (P, 1))[None, None]))

    # 7. batch index selects the image
    two = np.stack([np.full((1, 8, 8), 1.0), np.full((1, 8, 8), 5.0)])
    cases.append(_case("batch_index", two, [[1, 4, 4, 20, 20], [0, 4, 4, 20, 20]], 0.25,
                       np.stack([np.full((1, P, P), 5.0), np.full((1, P, P), 1.0)])))
    return cases


# --------------------------------------------------------------------------- tent-function statement (float64)
def tent_weights(lo, hi, L, scale):
    """(P, L) matrix A with bin[p] = sum_i A[p, i] * f[i] along one axis: mean over the bin's samples of the tent
    max(0, 1 - |c - i|) at the clamped sample position c, zero for dropped samples."""
    a, b = float(lo) * scale, float(hi) * scale
    roi = max(b - a, 1.0)
    g = int(math.ceil(roi / P))
    binw = roi / P
    A = np.zeros((P, L), np.float64)
    for p in range(P):
        for s in range(g):
            c = a + p * binw + (s + 0.5) * binw / g
            if c < -1.0 or c > L:
                continue
            c = min(max(c, 0.0), L - 1.0)
            for i in range(L):
                A[p, i] += max(0.0, 1.0 - abs(c - i))
    return A / g


def expected_by_tents(feat, rois, scale):
    feat = np.asarray(feat, np.float64)
    out = np.zeros((len(rois), feat.shape[1], P, P))
    for k, r in enumerate(np.asarray(rois, np.float64)):
        Ay = tent_weights(r[2], r[4], feat.shape[2], scale)
        Ax = tent_weights(r[1], r[3], feat.shape[3], scale)
        out[k] = np.einsum("pi,cij,qj->cpq", Ay, feat[int(r[0])], Ax)
    return out


def random_case(seed, H=16, W=16, C=4, B=2, K=12, scale=0.25):
    rng = np.random.default_rng(seed)
    feat = rng.standard_normal((B, C, H, W)).astype(np.float32)
    size = H / scale
    x1 = rng.uniform(-0.3 * size, 0.9 * size, K)
    y1 = rng.uniform(-0.3 * size, 0.9 * size, K)
    w = rng.uniform(0.5, 0.9 * size, K)
    h = rng.uniform(0.5, 0.9 * size, K)
    rois = np.stack([rng.integers(0, B, K).astype(np.float64), x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
    return dict(name=f"random{seed}", feat=feat, rois=rois, scale=scale, expected=expected_by_tents(feat, rois, scale))


# routing of model/rcnn_discriminator_app.py:131: fine map iff width < 64 AND height < 64 (image pixels)
ROUTING = [  # (x1, y1, x2, y2), expected "small"
    ((10.0, 10.0, 74.0, 30.0), False),    # exactly 64.0 wide -> coarse (1/8) map
    ((10.0, 10.0, 73.9, 73.9), True),     # 63.9 x 63.9 -> fine (1/4) map
    ((10.0, 10.0, 73.9, 74.0), False),    # 64.0 tall -> coarse
    ((0.0, 0.0, 12.0, 9.0), True),
    ((0.0, 0.0, 128.0, 128.0), False),
]
