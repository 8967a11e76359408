"""Convert the licensed SMPL model (data/smpl/SMPL_NEUTRAL.pkl, chumpy/scipy-sparse pickle) and
data/J_regressor_extra.npy into the plain .npz the engine loads.  To be run by
the end user who holds the SMPL licence; needs scipy (and chumpy only if the pickle contains chumpy
arrays - they are unwrapped via their `.r` attribute).

    python tools/convert_smpl.py data/smpl/SMPL_NEUTRAL.pkl data/J_regressor_extra.npy data/smpl/SMPL_NEUTRAL.npz
"""
import pickle
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from poco_amd.synth import JOINT_MAP_49, SMPL_EXTRA_VERTEX_IDS  # noqa: E402


def dense(x):
    if hasattr(x, "toarray"):
        x = x.toarray()
    if hasattr(x, "r"):
        x = x.r
    return np.asarray(x)


def main(pkl, extra, out):
    with open(pkl, "rb") as f:
        d = pickle.load(f, encoding="latin1")
    V = dense(d["v_template"]).shape[0]
    posedirs = dense(d["posedirs"]).reshape(V * 3, -1).T            # smplx: [207, V*3]
    parents = dense(d["kintree_table"])[0].astype(np.int64)
    parents[0] = -1
    np.savez_compressed(
        out, v_template=dense(d["v_template"]).astype(np.float32),
        shapedirs=dense(d["shapedirs"])[:, :, :10].astype(np.float32), posedirs=posedirs.astype(np.float32),
        J_regressor=dense(d["J_regressor"]).astype(np.float32), lbs_weights=dense(d["weights"]).astype(np.float32),
        J_regressor_extra=np.load(extra).astype(np.float32), parents=parents.astype(np.int32),
        extra_vertex_ids=SMPL_EXTRA_VERTEX_IDS, joint_map=JOINT_MAP_49, faces=dense(d["f"]).astype(np.int32))
    print("wrote", out)


if __name__ == "__main__":
    main(*sys.argv[1:4])
