"""Host-side mirror of the reference's model interface for the MI355X engine.

`POCO(backbone=..., **cfg.POCO, pretrained=ckpt)` + `model(batch) -> dict` with the same ctor
kwargs, batch fields and output keys as pocolib/models/poco.py:13-129, so that it drops in at the
reference's seam `output = self.model(batch)` (pocolib/core/tester.py:213,408).

This file is plumbing only: it moves names, shapes and raw pointers across the C ABI
(include/poco_hip.h).  All arithmetic runs in libpoco_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch

from ._lib import PocoHipError, check, lib

SMPL_KEYS = ("v_template", "shapedirs", "posedirs", "J_regressor", "J_regressor_extra", "lbs_weights",
             "parents", "extra_vertex_ids", "joint_map")


class _SizedStruct(C.Structure):
    """poco_inputs_t / poco_outputs_t (include/poco_hip.h, ABI 4): `struct_size` first, then the pointers, passed by keyword or
    in header order.  The size word is always filled in here, so a field list shorter than the library's is read as NULLs
    instead of as whatever lies behind the struct (tests/test_engine_cpu.py ties both field lists to the header)."""

    def __init__(self, *ptrs, **named):
        super().__init__(C.sizeof(type(self)), *ptrs, **named)


class _Inputs(_SizedStruct):
    _fields_ = [("struct_size", C.c_uint64)] + [(n, C.c_void_p) for n in (
        "img", "bbox_info", "focal_length", "scale", "center", "orig_shape")]


class _Outputs(_SizedStruct):
    _fields_ = [("struct_size", C.c_uint64)] + [(n, C.c_void_p) for n in (
        "pred_pose", "pred_pose6d", "pred_shape", "pred_cam", "pred_cam_t", "pred_fullimg_cam_t", "smpl_vertices",
        "smpl_joints3d", "smpl_joints2d", "var_pose", "uncert_feat", "pred_segm_mask", "body_feat2", "backbone_feat", "record")]


ABI_VERSION = 4          # include/poco_hip.h POCO_ABI_VERSION this binding was written against


def _bind():
    L = lib()
    if not hasattr(L, "poco_abi_version") or L.poco_abi_version() != ABI_VERSION:
        raise PocoHipError(f"libpoco_hip.so has ABI version {L.poco_abi_version() if hasattr(L, 'poco_abi_version') else '< 3'}, "
                           f"this binding needs {ABI_VERSION}: rebuild with `python -m poco_amd.build`")
    L.poco_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.poco_create_ex.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
    L.poco_destroy.argtypes = [C.c_void_p]
    L.poco_destroy.restype = None
    L.poco_num_tensors.argtypes = [C.c_void_p]
    L.poco_tensor_info.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.poco_load_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]
    L.poco_finalize.argtypes = [C.c_void_p]
    L.poco_forward.argtypes = [C.c_void_p, C.c_int, C.POINTER(_Inputs), C.POINTER(_Outputs), C.c_void_p]
    L.poco_status.argtypes = [C.c_void_p]
    L.poco_num_ops.argtypes = [C.c_void_p]
    L.poco_op_sched.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
    L.poco_op_info.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.poco_profile_ops.argtypes = [C.c_void_p, C.c_int, C.POINTER(_Inputs), C.POINTER(_Outputs), C.c_int,
                                   C.POINTER(C.c_float), C.c_int, C.c_void_p]
    L.poco_workspace_bytes.argtypes = [C.c_void_p]
    L.poco_workspace_bytes.restype = C.c_size_t
    L.poco_uncert_feat_dim.argtypes = [C.c_void_p]
    L.poco_set_conv_cfg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.poco_get_conv_desc.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.poco_get_conv_cfg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.poco_set_num_lanes.argtypes = [C.c_void_p, C.c_int]
    L.poco_smpl_lbs.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
    L.poco_realnvp.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.poco_realnvp_rep.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    return L


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class POCO:
    """Drop-in for pocolib.models.POCO (inference only).

    Extra kwargs on top of the reference ctor: `max_batch` (workspace is planned once for it),
    `smpl` (dict or .npz path of the SMPL-shaped body model; the reference reads data/smpl),
    `device`.
    """

    def __init__(self, backbone="resnet50-cliff", img_res=224, uncert_layer="diff_branch", activation_type="sigmoid",
                 uncert_type=("pose",), uncert_inp_type="feat", loss_ver="norm_flow_res_gaus", num_neurons="1024-512",
                 num_flow_layers=3, sigma_dim=1, num_nf_rv=9, mask_params_id="", nflow_mask_type="alter",
                 exclude_uncert_idx="", use_dropout=False, use_iter_feats=False, cond_nflow=True, context_dim=512,
                 gt_pose_cond=False, gt_pose_cond_ds="h36m", gt_pose_cond_ratio=0.25, pretrained=None,
                 inf_model="best", is_test=True, *, max_batch=64, smpl=None, device="cuda:0", keep_state_dict=None,
                 engine_options=None, max_graphs=8):
        if img_res != 224:
            raise ValueError("the engine is built for 224x224 crops (configs/demo_poco_*.yaml DATASET.IMG_RES)")
        if uncert_layer != "diff_branch" or activation_type != "sigmoid" or sigma_dim != 1 or num_nf_rv != 9:
            raise ValueError("only the shipped POCO configuration is supported "
                             "(UNCERT_LAYER diff_branch, ACTIVATION_TYPE sigmoid, SIGMA_DIM 1, NUM_NF_RV 9)")
        self.backbone_name, self.head_name = backbone.split("-")
        self.variant = backbone
        self.max_batch = int(max_batch)
        self.num_flow_layers = int(num_flow_layers)
        self.inf_model = inf_model
        self.device = torch.device(device)
        self._L = _bind()
        self._h = C.c_void_p()
        # engine_options: dict or "k=v,k=v" string of poco_create_ex build options (include/poco_hip.h); the library reads no
        # environment variable
        if isinstance(engine_options, dict):
            engine_options = ",".join(f"{k}={int(v) if isinstance(v, bool) else v}" for k, v in engine_options.items())
        self.engine_options = engine_options or ""
        self.max_graphs = int(max_graphs)
        check(self._L.poco_create_ex(backbone.encode(), self.max_batch, self.num_flow_layers, self.engine_options.encode(),
                                     C.byref(self._h)), "poco_create_ex")
        self._finalized = False
        self._loaded = set()
        # host copies for state_dict() (nn.Module.state_dict() always works in the reference: checkpoint re-save, weight diffing).
        # The engine itself keeps only BN-folded, fragment-packed copies on the device; the originals are ~300 MB of host memory
        # for HRNet-W48.  keep_state_dict = None (default): kept only for tensors that cannot be re-read - a model built from a
        # `pretrained` file drops them and state_dict() re-reads that file on demand, a model fed through
        # load_state_dict() keeps them so that state_dict() works like nn.Module's.  True / False force either way.
        self._state = None if (keep_state_dict is False or (keep_state_dict is None and pretrained is not None)) else {}
        self._pretrained_path = None
        if smpl is not None:
            self.load_smpl(smpl)
        if pretrained is not None:
            self.load_pretrained(pretrained)

    # ---- reference-compatible no-ops ---------------------------------------------------------
    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._L.poco_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    # ---- tensors -------------------------------------------------------------------------------
    def expected_tensors(self) -> List[Tuple[str, Tuple[int, ...], bool]]:
        out = []
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 6)()
        rank, req = C.c_int(), C.c_int()
        for i in range(self._L.poco_num_tensors(self._h)):
            check(self._L.poco_tensor_info(self._h, i, name, 256, shape, C.byref(rank), C.byref(req)), "poco_tensor_info")
            out.append((name.value.decode(), tuple(shape[k] for k in range(rank.value)), bool(req.value)))
        return out

    def _load_one(self, name: str, arr) -> None:
        if isinstance(arr, torch.Tensor):
            arr = arr.detach().cpu().numpy()
        orig_dtype, orig_shape = np.asarray(arr).dtype, np.asarray(arr).shape
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        shp = (C.c_int64 * max(1, arr.ndim))(*arr.shape)
        check(self._L.poco_load_tensor(self._h, name.encode(), C.c_void_p(arr.ctypes.data), shp, arr.ndim),
              f"poco_load_tensor({name})")
        self._loaded.add(name)
        if self._state is not None and not name.startswith("smpl."):
            self._state[name] = (arr.reshape(orig_shape), orig_dtype)

    def load_state_dict(self, state_dict: Dict[str, object], strict: bool = True):
        """Keys as in the reference checkpoint after `model.` stripping: backbone.*, head.*,
        uncert_head.*, flow_head.*  (pocolib/utils/train_utils.py:69-90).  strict: every key must
        be known to the engine and (at finalize) every required tensor present."""
        known = {n for n, _, _ in self.expected_tensors()}
        unexpected = []
        for k, v in state_dict.items():
            k = k[len("model."):] if k.startswith("model.") else k
            if k not in known:
                unexpected.append(k)
                continue
            self._load_one(k, v)
        if unexpected and strict:
            raise PocoHipError(f"unexpected keys in state_dict: {unexpected[:8]}{' ...' if len(unexpected) > 8 else ''}")
        return unexpected

    def state_dict(self) -> "Dict[str, torch.Tensor]":
        """The loaded parameters under the reference's state_dict keys (nn.Module.state_dict of pocolib.models.POCO,
        poco.py:13-42), in the engine's declaration order, as CPU tensors - from the host copies (models fed through
        load_state_dict) or re-read from the `pretrained` file (see `keep_state_dict`).
        Differences from the reference's state_dict(): the `smpl.*` buffers are not included (the body model is loaded
        separately, load_smpl); entries the engine tolerates but never reads (num_batches_tracked, backbone.final_layer, ...)
        appear only if they were loaded; values come back in the dtype they were loaded with (an int64
        num_batches_tracked stays int64; it crosses the C ABI as float32, exact up to 2^24)."""
        from collections import OrderedDict
        if self._state is None:
            if self._pretrained_path is None:
                raise PocoHipError("state_dict(): the model was built with keep_state_dict=False and without a `pretrained` file "
                                   "to re-read the parameters from")
            from .checkpoint import read_checkpoint
            sd = {(k[len("model."):] if k.startswith("model.") else k): v
                  for k, v in read_checkpoint(self._pretrained_path, self.inf_model).items()}
            return OrderedDict((n, torch.as_tensor(sd[n])) for n, _, _ in self.expected_tensors() if n in sd)
        out = OrderedDict()
        for name, _, _ in self.expected_tensors():
            if name in self._state:
                arr, dt = self._state[name]
                out[name] = torch.from_numpy(arr if dt == np.float32 else arr.astype(dt))
        return out

    def load_smpl(self, smpl) -> None:
        if isinstance(smpl, (str, Path)):
            smpl = dict(np.load(str(smpl)))
        for k in SMPL_KEYS:
            self._load_one("smpl." + k, np.asarray(smpl[k], dtype=np.float32))

    def load_pretrained(self, file) -> None:
        """pocolib/models/poco.py:131-154 (torch checkpoint -> state_dict -> per-part prefixes)."""
        from .checkpoint import read_checkpoint
        self.load_state_dict(read_checkpoint(file, self.inf_model), strict=True)
        self._pretrained_path = file

    def finalize(self) -> "POCO":
        if not self._finalized:
            torch.cuda.set_device(self.device)
            check(self._L.poco_finalize(self._h), "poco_finalize")
            self._finalized = True
        return self

    def _apply_tuned(self, B: int) -> None:
        """Measured conv tile table (poco_amd/tuned/gfx950.json, produced by poco_amd.tune) for this
        batch size; shapes without an entry keep the built-in heuristic."""
        done = self.__dict__.setdefault("_tuned_B", set())
        if B in done:
            return
        done.add(B)
        from . import tune
        table = self.__dict__.setdefault("_tune_table", tune.load_table())
        if table:
            tune.apply_table(self, B, table)

    # ---- forward ---------------------------------------------------------------------------------
    def _alloc_outputs(self, B: int, want_segm: bool, want_backbone_feat: bool = False) -> Dict[str, torch.Tensor]:
        d = self.device
        f = torch.float32
        ufd = self._L.poco_uncert_feat_dim(self._h)
        o = {
            "smpl_vertices": torch.empty(B, 6890, 3, device=d, dtype=f),
            "smpl_joints3d": torch.empty(B, 49, 3, device=d, dtype=f),
            "smpl_joints2d": torch.empty(B, 49, 2, device=d, dtype=f),
            "pred_cam_t": torch.empty(B, 3, device=d, dtype=f),
            "pred_pose": torch.empty(B, 24, 3, 3, device=d, dtype=f),
            "pred_cam": torch.empty(B, 3, device=d, dtype=f),
            "pred_shape": torch.empty(B, 10, device=d, dtype=f),
            "uncert_feat": torch.empty(B, ufd, device=d, dtype=f),
            "var_pose": torch.empty(B, 24, device=d, dtype=f),
            "record": torch.empty(B, 254, device=d, dtype=f),     # packed [pose 216 | betas 10 | cam 3 | var 24 | confidence 1]
        }
        if self.head_name == "cliff":
            o["pred_fullimg_cam_t"] = torch.empty(B, 3, device=d, dtype=f)
            o["pred_pose_6d"] = torch.empty(B, 144, device=d, dtype=f)
            o["body_feat2"] = torch.empty(B, 1024, device=d, dtype=f)
        else:
            o["pred_pose6d"] = torch.empty(B, 24, 6, device=d, dtype=f)
            if want_segm:
                o["pred_segm_mask"] = torch.empty(B, 25, 56, 56, device=d, dtype=f)
            if want_backbone_feat:      # parity checks only: the HRNet-W32 output map (hrnet.py:515-519)
                o["backbone_feat"] = torch.empty(B, 480, 56, 56, device=d, dtype=f)
        return o

    def _pack_io(self, batch, out):
        def dp(t):
            if t is None:
                return None
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise PocoHipError("batch tensors must be contiguous float32 CUDA tensors")
            return t.data_ptr()

        img = batch["img"]
        B = img.shape[0]
        if tuple(img.shape[1:]) != (3, 224, 224):
            raise PocoHipError(f"img must be [B,3,224,224], got {tuple(img.shape)}")
        if self.head_name == "cliff":
            orig = batch["orig_shape"].to(torch.float32).contiguous()
            ins = _Inputs(dp(img), dp(batch["bbox_info"].contiguous()), dp(batch["focal_length"].to(torch.float32).contiguous()),
                          dp(batch["scale"].to(torch.float32).contiguous()), dp(batch["center"].to(torch.float32).contiguous()),
                          dp(orig))
            keep = (orig,)
        else:
            ins = _Inputs(dp(img), None, None, None, None, None)
            keep = ()
        g = out.get
        outs = _Outputs(dp(g("pred_pose")), dp(g("pred_pose6d", g("pred_pose_6d"))), dp(g("pred_shape")), dp(g("pred_cam")),
                        dp(g("pred_cam_t")), dp(g("pred_fullimg_cam_t")), dp(g("smpl_vertices")), dp(g("smpl_joints3d")),
                        dp(g("smpl_joints2d")), dp(g("var_pose")), dp(g("uncert_feat")), dp(g("pred_segm_mask")),
                        dp(g("body_feat2")), dp(g("backbone_feat")), dp(g("record")))
        return B, ins, outs, keep

    @torch.no_grad()
    def __call__(self, batch: Dict[str, torch.Tensor], out: Optional[Dict[str, torch.Tensor]] = None,
                 want_segm: bool = True) -> Dict[str, object]:
        self.finalize()
        B = batch["img"].shape[0]
        if out is None:
            out = self._alloc_outputs(B, want_segm)
        B, ins, outs, _keep = self._pack_io(batch, out)
        self._apply_tuned(B)
        check(self._L.poco_forward(self._h, B, C.byref(ins), C.byref(outs), _stream()), "poco_forward")
        res = dict(out)
        res["log_phi"] = None            # nf_head.py:129-136: not evaluated at inference
        res["gt_pose_cond_idx"] = []     # poco_head.py:100,150
        return res

    forward = __call__

    def graph_forward(self, batch: Dict[str, torch.Tensor], out: Dict[str, torch.Tensor]) -> Dict[str, object]:
        """Replay the forward as a hipGraph (captured on first use for these exact input/output tensors:
        the kernels read/write the same device addresses on every replay, so refill `batch` in place).
        Removes ~375 launches + the fork/join events of the lanes from the host critical path.
        At most `max_graphs` graphs are kept (least recently used first out; `release_graphs()` drops them all): a caller that
        hands over fresh tensors every time would otherwise grow the cache without bound."""
        from collections import OrderedDict
        # address AND shape of every tensor: a view of a captured tensor (x[:8] of a batch captured at 16 crops) has the same
        # data_ptr and must get its own graph - replaying the 16-crop graph would overwrite rows 8-15 of the caller's outputs
        ident = lambda d: tuple(sorted((k, v.data_ptr(), tuple(v.shape)) for k, v in d.items()))
        key = (ident(batch), ident(out))
        cache = self.__dict__.setdefault("_graphs", OrderedDict())
        if key in cache:
            cache.move_to_end(key)
        else:
            if len(cache) >= max(1, self.max_graphs):
                torch.cuda.synchronize()              # a graph about to be dropped may still be replaying on the stream
            while len(cache) >= max(1, self.max_graphs):
                cache.popitem(last=False)
            self(batch, out=out)                      # warm-up on the capture stream's pool (tuning table, attributes)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self(batch, out=out)
                s.synchronize()
                with torch.cuda.graph(g, stream=s):
                    self(batch, out=out)
            torch.cuda.current_stream().wait_stream(s)
            cache[key] = g
        cache[key].replay()
        res = dict(out)
        res["log_phi"] = None
        res["gt_pose_cond_idx"] = []
        return res

    def check_status(self, sync: bool = False) -> None:
        """poco_status (include/poco_hip.h): raises PocoHipError if a bounded in-kernel wait (grid barrier of the fused regressor,
        stream-K hand-off) timed out in a forward since the last call - the outputs of those forwards are invalid; the engine is
        re-armed and usable again afterwards.  Call it after the stream / event the forward (or graph replay) ran on has been
        synchronised and before its outputs are used; `sync=True` synchronises the current stream first.  `model(batch)` and
        `graph_forward` only enqueue, exactly like the reference's `self.model(batch)` on a CUDA device, so they cannot check
        themselves; tester.py and stream.py call this before results are written."""
        if sync:
            torch.cuda.current_stream().synchronize()
        if self._finalized:
            check(self._L.poco_status(self._h), "poco_status")

    def release_graphs(self) -> None:
        """Drop every captured hipGraph of graph_forward."""
        self.__dict__.pop("_graphs", None)

    # ---- introspection / stand-alone ops -----------------------------------------------------------
    def ops(self):
        name = C.create_string_buffer(256)
        fl, ty = C.c_double(), C.c_int()
        out = []
        for i in range(self._L.poco_num_ops(self._h)):
            check(self._L.poco_op_info(self._h, i, name, 256, C.byref(fl), C.byref(ty)), "poco_op_info")
            out.append((name.value.decode(), fl.value, ty.value))
        return out

    def op_sched(self, op_index: int):
        """(phase, lane, wait_mask, reads, writes) of an op - the schedule the engine enqueues (include/poco_hip.h poco_op_sched; no
        GPU needed); reads / writes are (activation id, first channel, end channel) triples."""
        v = (C.c_int * 512)()          # the fused regressor (OP_MLP) lists the accesses of its ~20 sub-ops
        check(self._L.poco_op_sched(self._h, op_index, v, 512), "poco_op_sched")
        nr = v[3]
        rd = tuple((v[4 + 3 * k], v[5 + 3 * k], v[6 + 3 * k]) for k in range(nr))
        p = 4 + 3 * nr
        wr = tuple((v[p + 1 + 3 * k], v[p + 2 + 3 * k], v[p + 3 + 3 * k]) for k in range(v[p]))
        return v[0], v[1], v[2], rd, wr

    def profile_ops(self, batch, iters=5):
        self.finalize()
        B = batch["img"].shape[0]
        out = self._alloc_outputs(B, True)
        B, ins, outs, _keep = self._pack_io(batch, out)
        self._apply_tuned(B)
        n = self._L.poco_num_ops(self._h)
        ms = (C.c_float * n)()
        check(self._L.poco_profile_ops(self._h, B, C.byref(ins), C.byref(outs), iters, ms, n, _stream()), "poco_profile_ops")
        return [(nm, fl, ty, ms[i]) for i, (nm, fl, ty) in enumerate(self.ops())]

    def conv_desc(self, op_index: int):
        d = (C.c_int * 8)()
        rc = self._L.poco_get_conv_desc(self._h, op_index, d)
        return None if rc != 0 else tuple(d)

    def conv_cfg(self, op_index: int, B: int):
        """(MT,NT,WM,WN,R,NI,ALG) conv op `op_index` runs with at batch size B."""
        self._apply_tuned(B)
        c = (C.c_int * 7)()
        check(self._L.poco_get_conv_cfg(self._h, op_index, B, c), "poco_get_conv_cfg")
        return tuple(c)

    @staticmethod
    def kernel_symbol(desc, cfg) -> str:
        """Name of the HIP kernel template instance a conv op launches (as rocprofv3 prints it)."""
        MT, NT, WM, WN, R, NI, ALG = cfg
        ks, st = desc[4], desc[5]
        return {0: f"conv_mfma_kernel<{ks}, {st}, {MT}, {NT}>", 1: f"conv_dma_kernel<{ks}, {st}, {MT}, {NT}>",
                2: f"conv_dma_persist_kernel<{ks}, {st}, {MT}, {NT}>", 3: f"conv_wino_kernel<{NT}>",
                4: f"conv_wino2_kernel<{NT}, {3 if MT == 3 else 2}>", 5: "linear_mfma_kernel" if ks == 1 else "conv3x3_splitk_kernel", 6: f"gemm1x1_kernel<{MT}, {NT}, {R}", 7: f"conv_wino4_kernel<{NT}>", 8: f"conv_wino4p_kernel<{NT}, {0 if NI else (1 if R == 4 else 2)}>",
                9: f"gemm1x1t_kernel<{MT}, {NT}", 10: f"gemm3x3_kernel<{MT}, {NT}, {R}",
                11: f"wg_gemm_kernel<{MT}, {NT}, {R}>", 13: f"conv_wino4w_kernel<{NT}, {0 if NI else 1}>",
                14: f"gemm1x1sk_kernel<{MT}, {NT}, {R}"}[ALG]

    def set_conv_cfg(self, op_index: int, B: int, cfg) -> None:
        arr = (C.c_int * 7)(*(tuple(cfg) + (0,) * (7 - len(cfg))))
        check(self._L.poco_set_conv_cfg(self._h, op_index, B, arr), "poco_set_conv_cfg")

    def set_num_lanes(self, n: int) -> None:
        """1 = single stream; 4 (default) = independent branches on forked HIP streams."""
        check(self._L.poco_set_num_lanes(self._h, int(n)), "poco_set_num_lanes")

    def workspace_bytes(self) -> int:
        return int(self._L.poco_workspace_bytes(self._h))

    def smpl_lbs(self, betas: torch.Tensor, rotmat: torch.Tensor):
        self.finalize()
        B = betas.shape[0]
        verts = torch.empty(B, 6890, 3, device=self.device, dtype=torch.float32)
        j49 = torch.empty(B, 49, 3, device=self.device, dtype=torch.float32)
        check(self._L.poco_smpl_lbs(self._h, B, betas.contiguous().data_ptr(), rotmat.contiguous().data_ptr(),
                                    verts.data_ptr(), j49.data_ptr(), _stream()), "poco_smpl_lbs")
        return verts, j49

    def _realnvp(self, x: torch.Tensor, ctx: torch.Tensor, rep: int, forward: int) -> torch.Tensor:
        self.finalize()
        N = x.shape[0]
        rows = (N + rep - 1) // rep
        if tuple(x.shape) != (N, 9) or tuple(ctx.shape) != (rows, 512):
            raise PocoHipError(f"realnvp: x must be [N,9] and ctx [ceil(N/rep),512], got {tuple(x.shape)} / {tuple(ctx.shape)} (rep {rep})")
        out = torch.empty((N, 9) if forward else (N,), device=self.device, dtype=torch.float32)
        check(self._L.poco_realnvp_rep(self._h, N, x.contiguous().data_ptr(), ctx.contiguous().data_ptr(), int(rep), out.data_ptr(),
                                       forward, _stream()), "poco_realnvp")
        return out

    def realnvp_log_prob(self, x: torch.Tensor, ctx: torch.Tensor, rep: int = 1) -> torch.Tensor:
        """RealNVP.log_prob(x [N,9], x_cond) (real_nvp.py:55-65).  rep = 1: ctx is [N,512]; rep = 24: ctx is the per-crop
        [N/24,512] context of nf_head.py:105-110 (its repeat_interleave is not materialised)."""
        return self._realnvp(x, ctx, rep, 0)

    def realnvp_forward(self, z: torch.Tensor, ctx: torch.Tensor, rep: int = 1) -> torch.Tensor:
        """RealNVP.forward_p(z [N,9], x_cond) (real_nvp.py:25-38)."""
        return self._realnvp(z, ctx, rep, 1)
