"""ORACLE (test infrastructure): numpy restatement of the crop the reference makes per detection:
generate_patch_image_cv -> cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT) -> ToTensor -> Normalize
(pocolib/utils/vibe_image_utils.py:94-107,233-266,343-351) with exact float bilinear weights (cv2 itself
quantises the weights to 1/32 px, so agreement with real cv2 is within 1 grey level; cv2 is absent here:
parity unpinned against cv2 itself)."""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)


def crop_normalize_np(frame_u8, boxes, bbox_scale=1.0, res=224):
    H, W, _ = frame_u8.shape
    out = np.zeros((len(boxes), 3, res, res), np.float32)
    f = frame_u8.astype(np.float32)
    ys, xs = np.meshgrid(np.arange(res, dtype=np.float32), np.arange(res, dtype=np.float32), indexing="ij")
    for n, (cx, cy, bw, bh) in enumerate(np.asarray(boxes, np.float32)):
        sx = cx + (xs - np.float32(0.5 * res)) * (bw * np.float32(bbox_scale) / np.float32(res))
        sy = cy + (ys - np.float32(0.5 * res)) * (bh * np.float32(bbox_scale) / np.float32(res))
        x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
        fx, fy = sx - x0, sy - y0

        def px(xx, yy):
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            v = f[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
            return np.where(ok[..., None], v, 0.0)

        p = ((1 - fy) * (1 - fx))[..., None] * px(x0, y0) + ((1 - fy) * fx)[..., None] * px(x0 + 1, y0) \
            + (fy * (1 - fx))[..., None] * px(x0, y0 + 1) + (fy * fx)[..., None] * px(x0 + 1, y0 + 1)
        p = np.clip(np.rint(p), 0, 255) / 255.0
        out[n] = ((p - MEAN) / STD).transpose(2, 0, 1)
    return out
