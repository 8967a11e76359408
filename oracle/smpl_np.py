"""ORACLE (test infrastructure).  float64 numpy restatement of smplx.lbs.lbs as called by
pocolib/models/head/smpl_head.py:22-34,53-58 (pose2rot=False) + the 49-joint wrapper.

smplx==0.1.28 is an un-vendored dependency (requirements.txt:7): PARITY UNPINNED against it; this
file restates the published algorithm with explicit loops so that it is an
independent check of oracle/poco_ref.smpl_lbs and of the HIP kernel.
"""
import numpy as np


def smpl_lbs_np(smpl, betas, rotmat):
    f8 = np.float64
    vt = smpl["v_template"].astype(f8)
    V = vt.shape[0]
    B = betas.shape[0]
    betas = betas.astype(f8)
    R = rotmat.astype(f8)
    parents = [int(p) for p in smpl["parents"]]
    v_shaped = vt[None] + np.tensordot(betas, smpl["shapedirs"].astype(f8), axes=([1], [2]))   # [B,V,3]
    J = np.einsum("jv,bvk->bjk", smpl["J_regressor"].astype(f8), v_shaped)
    pose_feat = (R[:, 1:] - np.eye(3)).reshape(B, 207)
    v_posed = v_shaped + (pose_feat @ smpl["posedirs"].astype(f8)).reshape(B, V, 3)
    G = np.zeros((B, 24, 4, 4))
    for b in range(B):
        for i in range(24):
            T = np.eye(4)
            T[:3, :3] = R[b, i]
            T[:3, 3] = J[b, i] - (J[b, parents[i]] if i > 0 else 0.0)
            G[b, i] = T if i == 0 else G[b, parents[i]] @ T
    posed_joints = G[:, :, :3, 3].copy()
    A = G.copy()
    for b in range(B):
        for i in range(24):
            A[b, i, :3, 3] -= G[b, i, :3, :3] @ J[b, i]
    Tv = np.einsum("vj,bjmn->bvmn", smpl["lbs_weights"].astype(f8), A)
    verts = np.einsum("bvmn,bvn->bvm", Tv[:, :, :3, :3], v_posed) + Tv[:, :, :3, 3]
    j45 = np.concatenate([posed_joints, verts[:, smpl["extra_vertex_ids"]]], 1)
    extra = np.einsum("jv,bvk->bjk", smpl["J_regressor_extra"].astype(f8), verts)
    j54 = np.concatenate([j45, extra], 1)
    return verts, j54[:, smpl["joint_map"]]
