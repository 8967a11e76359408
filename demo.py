"""demo.py - the reference's demo CLI (demo.py:219-311) on the MI355X engine.

    python demo.py --cfg configs/demo_poco_cliff.yaml --ckpt data/poco_cliff.pt \\
                   --mode folder --image_folder <dir> --output_folder out --smpl data/smpl/SMPL_NEUTRAL.npz

Same flags as the reference where they concern the regressor (--cfg --ckpt --mode --image_folder
--vid_file --output_folder --batch_size --no_render --no_kinematic_uncert --inf_model).  Detector /
tracker / renderer are third-party and out of scope: person boxes come from
--detections (json {image name: [[cx,cy,w,h],...]}, the format multi_person_tracker produces) or
default to one centred box; results are written as .npz next to what the reference would render.
--mode video expects --vid_file to be a folder of extracted frames (the reference shells out to
ffmpeg first, demo.py:71; ffmpeg/cv2 are not part of this image).
"""
import argparse
import json
import os
import sys


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--cfg", type=str, required=True, help="config file that defines model hyperparams")
    p.add_argument("--ckpt", type=str, required=True, help="checkpoint path (.pt/.ckpt/.pth or run dir)")
    p.add_argument("--inf_model", type=str, default="best")
    p.add_argument("--mode", default="folder", choices=["video", "folder", "directory", "webcam"])
    p.add_argument("--vid_file", type=str, help="folder of extracted video frames")
    p.add_argument("--image_folder", type=str, help="input image folder")
    p.add_argument("--output_folder", type=str, default="out", help="output folder to write results")
    p.add_argument("--batch_size", type=int, default=64, help="batch size of POCO")
    p.add_argument("--tracker_batch_size", type=int, default=12)
    p.add_argument("--detector", type=str, default="yolo")
    p.add_argument("--no_render", action="store_true", help="(rendering is out of scope; always off)")
    p.add_argument("--no_kinematic_uncert", action="store_false",
                   help="Do not use SMPL Kinematic for uncert (same store_false semantics as the reference)")
    p.add_argument("--smooth", action="store_true", help="one-euro smoothing of each track (video mode)")
    p.add_argument("--min_cutoff", type=float, default=0.004)
    p.add_argument("--beta", type=float, default=1.5)
    p.add_argument("--tracking", type=str, default=None,
                   help="video mode: json or the reference's tracking_results_<method>.pkl {person_id: {'bbox': [[cx,cy,w,h],...], "
                        "'frames': [idx,...]}} (multi_person_tracker output); default = one centred track over all frames")
    p.add_argument("--skip_frame", type=int, default=1)
    p.add_argument("--save_obj", action="store_true", help="save results as .obj files (meshes/<image|person>/<idx>.obj)")
    p.add_argument("--detections", type=str, default=None,
                   help="json {image name: [[cx,cy,w,h],...]} or the reference's detection_results.pkl (per-image list)")
    p.add_argument("--gpus", type=int, default=1,
                   help="video mode: one process per GPU, whole tracks sharded across the ranks, ONE all-gather of the "
                        "packed SMPL records (RCCL over xGMI); `demo.py --gpus N` starts the N ranks itself")
    p.add_argument("--dist_backend", default="nccl", choices=["nccl", "gloo"],
                   help="nccl = RCCL (one GPU per rank); gloo only to exercise the multi-rank path on fewer devices")
    p.add_argument("--smpl", type=str, default="data/smpl/SMPL_NEUTRAL.npz",
                   help="SMPL body model as .npz (tools/convert_smpl.py converts the licensed .pkl)")
    return p.parse_args(argv)


def _spawn_ranks(args) -> int:
    """`demo.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess
    import torch
    if args.dist_backend == "nccl" and torch.cuda.device_count() < args.gpus:
        sys.exit(f"demo.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible; RCCL needs one per rank")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))


def main(args):
    if args.mode in ("webcam",):
        sys.exit("webcam mode needs a capture device + renderer: out of scope")
    if args.gpus > 1:
        if args.mode != "video":
            sys.exit("--gpus N shards whole tracks: video mode only (folder mode images are independent - run N demos)")
        if "WORLD_SIZE" not in os.environ:
            sys.exit(_spawn_ranks(args))
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
            torch.cuda.set_device(dev)
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    from poco_amd.tester import POCOTester, load_detections
    folder = args.image_folder if args.mode in ("folder", "directory") else args.vid_file
    if not folder or not os.path.isdir(folder):
        sys.exit(f"input folder not found: {folder}")
    tester = POCOTester(args)
    out_dir = os.path.join(args.output_folder, os.path.basename(os.path.normpath(folder)) + "_")
    if args.mode == "video":
        stats = tester.run_on_video_folder(folder, args.tracking, out_dir)
    else:
        stats = tester.run_on_image_folder(folder, load_detections(args.detections), out_dir)
    if "fps" in stats:                                                    # rank 0 (the reference logs 'poco FPS', demo.py:136-145)
        print(json.dumps({"poco_fps": round(stats["fps"], 2), **stats}))
    if args.gpus > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main(parse_args())
