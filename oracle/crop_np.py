"""ORACLE (test infrastructure): numpy restatement of the crop the reference makes per detection:
generate_patch_image_cv -> cv2.getAffineTransform -> cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT) on uint8 ->
ToTensor -> Normalize  (pocolib/utils/vibe_image_utils.py:49-107,233-266,343-351; called per detection from
pocolib/core/tester.py:182-203).

The arithmetic of the two cv2 calls lives in the un-vendored dependency opencv-python==4.5.5.64
(/root/reference/requirements.txt:4), absent from this image: PARITY UNPINNED against cv2 itself.  What is restated
here is OpenCV 4.5.5's published algorithm for exactly this call (modules/imgproc/src/imgwarp.cpp, modules/core/src/
matrix_decomp.cpp), step by step, so that the result is BYTE-exact by construction rather than "within one grey level":

  getAffineTransform   6x6 system in double, cv::solve(DECOMP_LU) = LUImpl<double>: partial pivoting on |a|, row
                       elimination with d = -1/pivot, alpha = a*d, a += alpha*a_i, back substitution s/a_ii
  warpAffine           M inverted in double (D = 1/det, b = -A^-1 t); AB_BITS = 10: adelta[x] = cvRound(M0*x*1024),
                       bdelta[x] = cvRound(M3*x*1024), per row X0 = cvRound((M1*y + M2)*1024) + 16 (round_delta =
                       1024/32/2), X = (X0 + adelta[x]) >> 5: integer pixel = saturate_short(X >> 5), 5-bit fraction X & 31
  remapBilinear (8u)   weight table BilinearTab_i[32*32][2][2] of int16 = saturate_short((1-fy|fy)*(1-fx|fx)*32768), sum
                       forced to 32768 (entry (0,0) = {32767,0,0,1}); value = (sum_k w_k*v_k + 2^14) >> 15, pixels outside
                       the image = borderValue 0
  cvRound              round-half-to-even (SSE2 cvtsd2si) = np.rint

`tools/validate_assets.py --crop` compares this file (and the HIP kernel) with the real cv2 where it is importable.
The numpy dtype chain of gen_trans_from_patch_cv is the reference's pinned numpy==1.18.1 (requirements.txt:1): a float32
scalar times a Python float is float64 there (NumPy 2 would keep float32), so the restatement computes in float64 and
rounds to float32 exactly where the reference stores into float32 arrays."""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)

AB_BITS, INTER_BITS, COEF_BITS = 10, 5, 15
TAB = 1 << INTER_BITS


def lu_solve_cv(A, b):
    """cv::hal::LU64f / LUImpl<double> (matrix_decomp.cpp) on a copy: solves A x = b for one right-hand side."""
    A = np.array(A, np.float64)
    b = np.array(b, np.float64)
    m = A.shape[0]
    for i in range(m):
        k = i
        for j in range(i + 1, m):
            if abs(A[j, i]) > abs(A[k, i]):
                k = j
        if abs(A[k, i]) < np.finfo(np.float64).eps * 100:
            return None
        if k != i:
            A[[i, k], i:] = A[[k, i], i:]
            b[[i, k]] = b[[k, i]]
        d = -1.0 / A[i, i]
        for j in range(i + 1, m):
            alpha = A[j, i] * d
            for c in range(i + 1, m):
                A[j, c] = A[j, c] + alpha * A[i, c]
            b[j] = b[j] + alpha * b[i]
    for i in range(m - 1, -1, -1):
        s = b[i]
        for c in range(i + 1, m):
            s = s - A[i, c] * b[c]
        b[i] = s / A[i, i]
    return b


def get_affine_transform_cv(src, dst):
    """cv2.getAffineTransform(src[3,2] float32, dst[3,2] float32) -> 2x3 float64 (imgwarp.cpp)."""
    a = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    for i in range(3):
        a[2 * i, 0:3] = (src[i][0], src[i][1], 1.0)
        a[2 * i + 1, 3:6] = (src[i][0], src[i][1], 1.0)
        b[2 * i], b[2 * i + 1] = dst[i][0], dst[i][1]
    x = lu_solve_cv(a, b)
    if x is None:
        x = np.zeros(6)
    return x.reshape(2, 3)


def patch_points(c_x, c_y, bb_w, bb_h, res, scale):
    """the two float32 point triples gen_trans_from_patch_cv(..., rot=0) hands to cv2.getAffineTransform
    (vibe_image_utils.py:58-87) with numpy-1.18 promotion: src_w = width*scale in float64; the half extents and the points
    are stored as float32.  Pinned to the reference's own function (cv2 stubbed by a recorder): tests/golden/ops.npz crop_*."""
    src_w, src_h = np.float64(bb_w) * np.float64(scale), np.float64(bb_h) * np.float64(scale)
    cx, cy = np.float64(c_x), np.float64(c_y)
    down = np.float32(src_h * 0.5)
    right = np.float32(src_w * 0.5)
    src = np.zeros((3, 2), np.float32)
    src[0] = (cx, cy)
    src[1] = (cx + np.float64(np.float32(0.0)), cy + np.float64(down))
    src[2] = (cx + np.float64(right), cy + np.float64(np.float32(0.0)))
    half = np.float32(res * 0.5)
    dst = np.array([[half, half], [half, half + half], [half + half, half]], np.float32)
    return src, dst


def gen_trans_from_patch(c_x, c_y, bb_w, bb_h, res, scale):
    """gen_trans_from_patch_cv(..., rot=0, inv=False) (vibe_image_utils.py:58-92)."""
    return get_affine_transform_cv(*patch_points(c_x, c_y, bb_w, bb_h, res, scale))


def invert_affine_cv(M):
    """the inversion cv::warpAffine applies without WARP_INVERSE_MAP (imgwarp.cpp)."""
    M = np.array(M, np.float64).reshape(6).copy()
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11
    M[1] = M[1] * -D
    M[3] = M[3] * -D
    M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return M


def bilinear_tab_i():
    """BilinearTab_i (initInterTab2D, imgwarp.cpp): [32*32][4] int16 weights, index fy*32 + fx, order (y0x0, y0x1, y1x0, y1x1)."""
    t1 = np.zeros((TAB, 2), np.float32)
    for i in range(TAB):
        x = np.float32(i) * np.float32(1.0 / TAB)
        t1[i] = (np.float32(1.0) - x, x)
    tab = np.zeros((TAB * TAB, 4), np.int64)
    for i in range(TAB):
        for j in range(TAB):
            v = np.array([t1[i, 0] * t1[j, 0], t1[i, 0] * t1[j, 1], t1[i, 1] * t1[j, 0], t1[i, 1] * t1[j, 1]], np.float32)
            it = np.clip(np.rint(v.astype(np.float64) * (1 << COEF_BITS)), -32768, 32767).astype(np.int64)
            s = int(it.sum())
            if s != (1 << COEF_BITS):
                # ksize = 2: the search window of the original (k1, k2 in [1, 3)) only has the entry's own element 3 in range;
                # the others read the next, not yet initialised (zero) table entries and never win for diff < 0
                diff = s - (1 << COEF_BITS)
                assert diff < 0
                it[3] -= diff
            tab[i * TAB + j] = it
    return tab


_TAB_I = None


def warp_affine_u8(img, M, res):
    """cv2.warpAffine(img uint8 [H,W,C], M 2x3 float64, (res,res), INTER_LINEAR, BORDER_CONSTANT 0) -> uint8 [res,res,C]."""
    global _TAB_I
    if _TAB_I is None:
        _TAB_I = bilinear_tab_i()
    H, W = img.shape[:2]
    Mi = invert_affine_cv(M)
    AB = float(1 << AB_BITS)
    xs = np.arange(res, dtype=np.float64)
    adelta = np.rint(Mi[0] * xs * AB).astype(np.int64)
    bdelta = np.rint(Mi[3] * xs * AB).astype(np.int64)
    rd = (1 << AB_BITS) // TAB // 2
    X0 = np.rint((Mi[1] * xs + Mi[2]) * AB).astype(np.int64) + rd          # indexed by y
    Y0 = np.rint((Mi[4] * xs + Mi[5]) * AB).astype(np.int64) + rd
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    w = _TAB_I[(Y & (TAB - 1)) * TAB + (X & (TAB - 1))]                     # [res,res,4]
    f = img.astype(np.int64)

    def px(xx, yy):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = f[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return np.where(ok[..., None], v, 0)

    acc = w[..., 0:1] * px(sx, sy) + w[..., 1:2] * px(sx + 1, sy) + w[..., 2:3] * px(sx, sy + 1) + w[..., 3:4] * px(sx + 1, sy + 1)
    return np.clip((acc + (1 << (COEF_BITS - 1))) >> COEF_BITS, 0, 255).astype(np.uint8)


def crop_u8_np(frame_u8, boxes, bbox_scale=1.0, res=224):
    """generate_patch_image_cv per box: uint8 crops [N,res,res,3] (`raw_image` of get_single_image_crop_demo)."""
    out = np.zeros((len(boxes), res, res, frame_u8.shape[2]), np.uint8)
    for n, (cx, cy, bw, bh) in enumerate(np.asarray(boxes)):
        out[n] = warp_affine_u8(frame_u8, gen_trans_from_patch(cx, cy, bw, bh, res, bbox_scale), res)
    return out


def normalize_np(crops_u8):
    """ToTensor -> Normalize in float32 (vibe_image_utils.py:343-351): (p/255 - mean)/std, NCHW."""
    p = crops_u8.astype(np.float32) / np.float32(255.0)
    return ((p - MEAN) / STD).transpose(0, 3, 1, 2).astype(np.float32)


def crop_normalize_np(frame_u8, boxes, bbox_scale=1.0, res=224):
    return normalize_np(crop_u8_np(frame_u8, boxes, bbox_scale, res))


def crop_normalize_float_np(frame_u8, boxes, bbox_scale=1.0, res=224):
    """the same crop with EXACT float bilinear weights (no 1/32-px quantisation): only used by the tests to bound how far
    cv2's fixed-point result is from the ideal one (<= 1 grey level + the 1/64 px coordinate rounding)."""
    H, W, _ = frame_u8.shape
    out = np.zeros((len(boxes), res, res, 3), np.float64)
    f = frame_u8.astype(np.float64)
    ys, xs = np.meshgrid(np.arange(res, dtype=np.float64), np.arange(res, dtype=np.float64), indexing="ij")
    for n, (cx, cy, bw, bh) in enumerate(np.asarray(boxes, np.float64)):
        sx = cx + (xs - 0.5 * res) * (bw * bbox_scale / res)
        sy = cy + (ys - 0.5 * res) * (bh * bbox_scale / res)
        x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
        fx, fy = sx - x0, sy - y0

        def px(xx, yy):
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            v = f[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
            return np.where(ok[..., None], v, 0.0)

        out[n] = ((1 - fy) * (1 - fx))[..., None] * px(x0, y0) + ((1 - fy) * fx)[..., None] * px(x0 + 1, y0) \
            + (fy * (1 - fx))[..., None] * px(x0, y0 + 1) + (fy * fx)[..., None] * px(x0 + 1, y0 + 1)
    return out
