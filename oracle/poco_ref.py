"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU fp32 restatement of the reference's per-crop regressor  POCO.forward
(/root/reference/pocolib/models/poco.py:99-129) and everything it calls, written functionally over
a flat {reference state_dict key -> tensor} dict so that the very same arrays can be loaded into
the reference's nn.Modules (oracle/gen_golden.py does that and pins this file against them) and
into the HIP engine (which looks tensors up by the same names).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Pinned against: the reference modules imported in the build container (gen_golden.py), outputs
committed under tests/golden/.  The SMPL linear-blend-skinning step lives in the un-vendored
dependency smplx==0.1.28 (requirements.txt:7) -> PARITY UNPINNED at that boundary; it is restated
here from the published algorithm and cross-checked against the float64 numpy
version in oracle/smpl_np.py plus invariants (tests/test_oracle_smpl.py).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
BN_EPS = 1e-5

# --------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------


def conv_bn(sd: SD, x, conv: str, bn: str | None, stride=1, relu=True, pad=None):
    """nn.Conv2d -> nn.BatchNorm2d(eval) -> ReLU, e.g. hrnet.py:45-47."""
    w = sd[conv + ".weight"]
    if pad is None:
        pad = (w.shape[-1] - 1) // 2
    y = F.conv2d(x, w, sd.get(conv + ".bias"), stride=stride, padding=pad)
    if bn is not None:
        y = F.batch_norm(y, sd[bn + ".running_mean"], sd[bn + ".running_var"], sd[bn + ".weight"],
                         sd[bn + ".bias"], False, 0.0, BN_EPS)
    return F.relu(y) if relu else y


def linear(sd: SD, x, name: str):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def basic_block(sd: SD, x, p: str):
    """hrnet.py:42-58 (no downsample inside HRNet branches)."""
    y = conv_bn(sd, x, p + ".conv1", p + ".bn1")
    y = conv_bn(sd, y, p + ".conv2", p + ".bn2", relu=False)
    return F.relu(y + x)


def bottleneck(sd: SD, x, p: str, stride=1):
    """hrnet.py:79-99 / resnet.py:101-121 (stride on the 3x3, torchvision v1.5)."""
    y = conv_bn(sd, x, p + ".conv1", p + ".bn1")
    y = conv_bn(sd, y, p + ".conv2", p + ".bn2", stride=stride)
    y = conv_bn(sd, y, p + ".conv3", p + ".bn3", relu=False)
    r = x
    if (p + ".downsample.0.weight") in sd:
        r = conv_bn(sd, x, p + ".downsample.0", p + ".downsample.1", stride=stride, relu=False)
    return F.relu(y + r)


# --------------------------------------------------------------------------------------------
# HRNet (hrnet.py:248-266, 466-528 ; hrnet_cls.py:438-486)
# --------------------------------------------------------------------------------------------


def hr_module(sd: SD, xs: List[torch.Tensor], p: str) -> List[torch.Tensor]:
    nb = len(xs)
    xs = list(xs)
    for i in range(nb):
        for k in range(4):
            xs[i] = basic_block(sd, xs[i], f"{p}.branches.{i}.{k}")
    outs = []
    for i in range(nb):
        y = None
        for j in range(nb):
            if j == i:
                t = xs[j]
            elif j > i:
                t = conv_bn(sd, xs[j], f"{p}.fuse_layers.{i}.{j}.0", f"{p}.fuse_layers.{i}.{j}.1", relu=False)
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
            else:
                t = xs[j]
                for k in range(i - j):
                    q = f"{p}.fuse_layers.{i}.{j}.{k}"
                    t = conv_bn(sd, t, q + ".0", q + ".1", stride=2, relu=(k < i - j - 1))
            y = t if y is None else y + t
        outs.append(F.relu(y))
    return outs


def hrnet_trunk(sd: SD, img, p: str, modules=(1, 4, 3)) -> List[torch.Tensor]:
    x = conv_bn(sd, img, p + "conv1", p + "bn1", stride=2)
    x = conv_bn(sd, x, p + "conv2", p + "bn2", stride=2)
    for k in range(4):
        x = bottleneck(sd, x, f"{p}layer1.{k}")
    ys = [x]
    for s, nmod in enumerate(modules):
        stage, nb = s + 2, s + 2
        t = f"{p}transition{s + 1}"
        xs = []
        for i in range(nb):
            if i < len(ys) and (f"{t}.{i}.0.weight") in sd:          # same-resolution channel change
                xs.append(conv_bn(sd, ys[i], f"{t}.{i}.0", f"{t}.{i}.1"))
            elif i < len(ys):
                xs.append(ys[i])
            else:                                                     # new lower-resolution branch
                z = ys[-1]
                k = 0
                while (f"{t}.{i}.{k}.0.weight") in sd:
                    z = conv_bn(sd, z, f"{t}.{i}.{k}.0", f"{t}.{i}.{k}.1", stride=2)
                    k += 1
                xs.append(z)
        for m in range(nmod):
            xs = hr_module(sd, xs, f"{p}stage{stage}.{m}")
        ys = xs
    return ys


def hrnet_w32(sd: SD, img, p="backbone."):
    """PoseHighResolutionNet(downsample=False, use_conv=True): hrnet.py:515-519 -> [B,480,56,56]."""
    x = hrnet_trunk(sd, img, p)
    outs = [x[0]]
    for b, nlayers in ((1, 1), (2, 2), (3, 3)):
        y = x[b]
        for t in range(nlayers):
            y = F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=True)
            q = f"{p}upsample_stage_{b + 1}"
            y = conv_bn(sd, y, f"{q}.{1 + 4 * t}", f"{q}.{2 + 4 * t}")
        outs.append(y)
    return torch.cat(outs, 1)


def hrnet_w48_cls(sd: SD, img, p="backbone."):
    """HighResolutionNet + classification head, hrnet_cls.py:471-486 -> [B,2048]."""
    ys = hrnet_trunk(sd, img, p)
    y = bottleneck(sd, ys[0], f"{p}incre_modules.0.0")
    for i in range(3):
        d = conv_bn(sd, y, f"{p}downsamp_modules.{i}.0", f"{p}downsamp_modules.{i}.1", stride=2)
        y = bottleneck(sd, ys[i + 1], f"{p}incre_modules.{i + 1}.0") + d
    y = conv_bn(sd, y, f"{p}final_layer.0", f"{p}final_layer.1")
    return y.mean(dim=(2, 3))


def resnet50(sd: SD, img, p="backbone."):
    """resnet.py:201-217 -> [B,2048,7,7]."""
    x = conv_bn(sd, img, p + "conv1", p + "bn1", stride=2, pad=3)
    x = F.max_pool2d(x, 3, 2, 1)
    for li, nblk in enumerate((3, 4, 6, 3)):
        for k in range(nblk):
            x = bottleneck(sd, x, f"{p}layer{li + 1}.{k}", stride=2 if (k == 0 and li > 0) else 1)
    return x


# --------------------------------------------------------------------------------------------
# geometry (utils/geometry.py:247-261, 447-463, 480-508 ; smplcam_head.py:99-139)
# --------------------------------------------------------------------------------------------


def rot6d_to_rotmat(x):
    x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-1)


def weak_persp_to_persp(cam, focal=5000.0, img_res=224):
    return torch.stack([cam[:, 1], cam[:, 2], 2 * focal / (img_res * cam[:, 0] + 1e-9)], dim=-1)


def project(points, translation, fx, cx, cy):
    """perspective_projection with identity rotation; fx per-sample or scalar."""
    p = points + translation[:, None, :]
    p = p / p[:, :, 2:3]
    fx = torch.as_tensor(fx, dtype=p.dtype).reshape(-1, 1)
    u = fx * p[:, :, 0] + torch.as_tensor(cx, dtype=p.dtype).reshape(-1, 1)
    v = fx * p[:, :, 1] + torch.as_tensor(cy, dtype=p.dtype).reshape(-1, 1)
    return torch.stack([u, v], -1)


def full_img_cam(cam, bbox_height, center, img_w, img_h, focal):
    """convert_pare_to_full_img_cam, smplcam_head.py:123-139."""
    s, tx, ty = cam[:, 0], cam[:, 1], cam[:, 2]
    r = bbox_height / 224
    tz = 2 * focal / (r * 224 * s)
    cx = 2 * (center[:, 0] - img_w / 2.0) / (s * bbox_height)
    cy = 2 * (center[:, 1] - img_h / 2.0) / (s * bbox_height)
    return torch.stack([tx + cx, ty + cy, tz], dim=-1)


# --------------------------------------------------------------------------------------------
# SMPL (smplx.lbs.lbs restated, SURVEY.md 3.5 ; wrapper smpl_head.py:22-34)
# --------------------------------------------------------------------------------------------


def smpl_lbs(smpl: Dict[str, torch.Tensor], betas, rotmat):
    """betas [B,10], rotmat [B,24,3,3] -> vertices [B,V,3], joints49 [B,49,3]."""
    B = betas.shape[0]
    vt, sdirs, pdirs = smpl["v_template"], smpl["shapedirs"], smpl["posedirs"]
    V = vt.shape[0]
    v_shaped = vt[None] + torch.einsum("bl,mkl->bmk", betas, sdirs)
    J = torch.einsum("bik,ji->bjk", v_shaped, smpl["J_regressor"])
    ident = torch.eye(3, dtype=betas.dtype)
    pose_feat = (rotmat[:, 1:] - ident).reshape(B, 207)
    v_posed = v_shaped + (pose_feat @ pdirs).view(B, V, 3)
    parents = smpl["parents"].tolist()
    rel = J.clone()
    rel[:, 1:] = J[:, 1:] - J[:, parents[1:]]
    T = torch.zeros(B, 24, 4, 4, dtype=betas.dtype)
    T[:, :, :3, :3] = rotmat
    T[:, :, :3, 3] = rel
    T[:, :, 3, 3] = 1.0
    G = [T[:, 0]]
    for i in range(1, 24):
        G.append(G[parents[i]] @ T[:, i])
    G = torch.stack(G, 1)
    posed_joints = G[:, :, :3, 3]
    Jh = torch.cat([J, torch.zeros(B, 24, 1, dtype=betas.dtype)], 2).unsqueeze(-1)
    A = G - F.pad(G @ Jh, [3, 0])
    Tv = (smpl["lbs_weights"] @ A.view(B, 24, 16)).view(B, V, 4, 4)
    vh = torch.cat([v_posed, torch.ones(B, V, 1, dtype=betas.dtype)], 2).unsqueeze(-1)
    verts = (Tv @ vh)[:, :, :3, 0]
    j45 = torch.cat([posed_joints, verts[:, smpl["extra_vertex_ids"].long()]], 1)
    extra = torch.einsum("bik,ji->bjk", verts, smpl["J_regressor_extra"])
    j54 = torch.cat([j45, extra], 1)
    return verts, j54[:, smpl["joint_map"].long()]


# --------------------------------------------------------------------------------------------
# heads
# --------------------------------------------------------------------------------------------


def keypoint_attention(feat, heat):
    """layers/keypoint_attention.py:34-48 (softmax over pixels, then heat . feat^T)."""
    B, J = heat.shape[:2]
    w = F.softmax(heat.reshape(B, J, -1), dim=-1)
    f = feat.reshape(B, feat.shape[1], -1)
    return torch.matmul(w, f.transpose(2, 1)).transpose(2, 1)      # [B,C,J]


def pare_head(sd: SD, feats, p="head."):
    """pare_head.forward default-flag path, pare_head.py:669-752, 896-928."""
    B = feats.shape[0]

    def branch(name):
        y = conv_bn(sd, feats, f"{p}{name}.0", f"{p}{name}.1")
        return conv_bn(sd, y, f"{p}{name}.3", f"{p}{name}.4")

    part_feats = branch("keypoint_deconv_layers")
    heat = conv_bn(sd, part_feats, p + "keypoint_final_layer", None, relu=False)
    smpl_feats = branch("smpl_deconv_layers")
    cam_shape = conv_bn(sd, smpl_feats, p + "smpl_final_layer", None, relu=False)
    attn = heat[:, 1:]
    local = keypoint_attention(smpl_feats, attn)                    # [B,128,24]
    cs = keypoint_attention(cam_shape, attn)                        # [B,64,24]
    w = sd[p + "pose_mlp.weight"][0, :, :, :, 0, 0]                 # [6,128,24]
    pose6d = torch.einsum("bcj,ocj->boj", local, w).transpose(2, 1)  # [B,24,6]
    flat = cs.flatten(1)
    cam = linear(sd, flat, p + "cam_mlp")
    shape = linear(sd, flat, p + "shape_mlp")
    return {
        "pred_pose": rot6d_to_rotmat(pose6d).reshape(B, 24, 3, 3),
        "pred_pose6d": pose6d, "pred_cam": cam, "pred_shape": shape,
        "pred_segm_mask": heat, "uncert_feat": local.reshape(B, -1),
    }


def cliff_head(sd: SD, feats, bbox_info, p="head.", n_iter=3):
    """cliff_head.py:74-127."""
    B = feats.shape[0]
    if feats.dim() > 2:
        feats = feats.mean(dim=(2, 3))
    pose = sd[p + "init_pose"].expand(B, -1)
    shape = sd[p + "init_shape"].expand(B, -1)
    cam = sd[p + "init_cam"].expand(B, -1)
    xc = None
    for _ in range(n_iter):
        xc = torch.cat([feats, bbox_info, pose, shape, cam], 1)
        xc = linear(sd, linear(sd, xc, p + "fc1"), p + "fc2")
        pose = linear(sd, xc, p + "decpose") + pose
        shape = linear(sd, xc, p + "decshape") + shape
        cam = linear(sd, xc, p + "deccam") + cam
    return {
        "pred_pose": rot6d_to_rotmat(pose).view(B, 24, 3, 3), "pred_cam": cam, "pred_shape": shape,
        "pred_pose_6d": pose, "uncert_feat": feats, "body_feat2": xc,
    }


def poco_head(sd: SD, uncert_feat, rotmat, p="uncert_head."):
    """poco_head.py:116-148 (sigmoid activations, dropout = identity in eval)."""
    B = rotmat.shape[0]
    pose = rotmat.reshape(B, -1)
    if (p + "uncert_fc_poseNet.weight") in sd:                      # 'feat-pose-net' (POCO-CLIFF)
        a = torch.sigmoid(linear(sd, uncert_feat, p + "uncert_fc_featNet"))
        b = torch.sigmoid(linear(sd, pose, p + "uncert_fc_poseNet"))
        x = torch.cat([a, b], 1)
    else:                                                           # 'feat-pose' (POCO-PARE)
        x = torch.cat([uncert_feat, pose], 1)
    k = 1
    while (f"{p}uncert_fc{k}.weight") in sd:
        x = torch.sigmoid(linear(sd, x, f"{p}uncert_fc{k}"))
        k += 1
    return x


def _flow_mlp(sd: SD, x, q: str, tanh: bool):
    x = F.leaky_relu(linear(sd, x, q + ".0"), 0.01)
    x = F.leaky_relu(linear(sd, x, q + ".2"), 0.01)
    x = linear(sd, x, q + ".4")
    return torch.tanh(x) if tanh else x


def realnvp_backward(sd: SD, x, cond, p="flow_head.flow."):
    """RealNVP.backward_p, layers/real_nvp.py:40-53 -> (z, log_det)."""
    mask = sd[p + "mask"]
    z = x
    logdet = x.new_zeros(x.shape[0])
    for i in reversed(range(mask.shape[0])):
        m = mask[i]
        z_ = m * z
        inp = torch.cat([z_, cond], 1) if cond is not None else z_
        s = _flow_mlp(sd, inp, f"{p}s.{i}", True) * (1 - m)
        t = _flow_mlp(sd, inp, f"{p}t.{i}", False) * (1 - m)
        z = (1 - m) * (z - t) * torch.exp(-s) + z_
        logdet = logdet - s.sum(1)
    return z, logdet


def realnvp_log_prob(sd: SD, x, cond, p="flow_head.flow."):
    """RealNVP.log_prob with the N(0,I) prior, real_nvp.py:55-65."""
    z, logdet = realnvp_backward(sd, x, cond, p)
    d = z.shape[1]
    return -0.5 * (z * z).sum(1) - 0.5 * d * np.log(2 * np.pi) + logdet


def realnvp_forward(sd: SD, z, cond, p="flow_head.flow."):
    """RealNVP.forward_p (sampling direction), real_nvp.py:25-38."""
    mask = sd[p + "mask"]
    x = z
    for i in range(mask.shape[0]):
        m = mask[i]
        x_ = x * m
        inp = torch.cat([x_, cond], 1) if cond is not None else x_
        s = _flow_mlp(sd, inp, f"{p}s.{i}", True) * (1 - m)
        t = _flow_mlp(sd, inp, f"{p}t.{i}", False) * (1 - m)
        x = x_ + (1 - m) * (x * torch.exp(s) + t)
    return x


# --------------------------------------------------------------------------------------------
# uncertainty post-processing (utils/poco_utils.py:21-25, 50-60 ; core/tester.py:242-245)
# --------------------------------------------------------------------------------------------
SMPL_SKELETON_CHILD_ORDER = list(range(1, 24))


def kinematic_uncert(var: np.ndarray, parents) -> np.ndarray:
    var = var.copy()
    for i in range(1, 24):
        var[:, i] += var[:, parents[i]]
    return var


def global_uncert(var: np.ndarray, variant: str, thr=0.40) -> np.ndarray:
    var = var.copy()
    if "cliff" in variant:
        var[var[:, 0] > 2 * thr] = 1.0
        g = var[:, 0]
    else:
        var[var[:, 0] > thr] = 1.0
        g = var.mean(-1)
    return np.clip(g, 0, 0.99)


# --------------------------------------------------------------------------------------------
# full model
# --------------------------------------------------------------------------------------------
BACKBONES = {"hrnet_w32": hrnet_w32, "hrnet_w48_cls": hrnet_w48_cls, "resnet50": resnet50}


@torch.no_grad()
def poco_forward(variant: str, sd: SD, smpl: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor]):
    """variant e.g. 'hrnet_w32-pare', 'hrnet_w48_cls-cliff', 'resnet50-cliff' (poco.py:41)."""
    bname, hname = variant.split("-")
    feats = BACKBONES[bname](sd, batch["img"])
    if hname == "cliff":
        out = cliff_head(sd, feats, batch["bbox_info"])
    else:
        out = pare_head(sd, feats)
    verts, j49 = smpl_lbs(smpl, out["pred_shape"], out["pred_pose"])
    out["smpl_vertices"], out["smpl_joints3d"] = verts, j49
    cam = out["pred_cam"]
    out["pred_cam_t"] = weak_persp_to_persp(cam)
    if hname == "cliff":
        img_h, img_w = batch["orig_shape"][:, 0], batch["orig_shape"][:, 1]
        focal = batch["focal_length"]
        t_full = full_img_cam(cam, batch["scale"] * 200.0, batch["center"], img_w, img_h, focal)
        out["pred_fullimg_cam_t"] = t_full
        out["smpl_joints2d"] = project(j49, t_full, focal, img_w / 2.0, img_h / 2.0)
    else:
        out["smpl_joints2d"] = project(j49, out["pred_cam_t"], 5000.0, 0.0, 0.0) / 112.0
    out["var_pose"] = poco_head(sd, out["uncert_feat"], out["pred_pose"])
    out["log_phi"] = None          # nf_head.py:129-136: the flow is not evaluated at inference
    out["gt_pose_cond_idx"] = []
    return out


def to_torch(d: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}
