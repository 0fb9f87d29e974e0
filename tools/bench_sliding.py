#!/usr/bin/env python3
"""BASELINE config 3: full-volume sliding-window inference, 96^3 windows, 50-step DDIM, patch-sharded over
the ranks of one node with one all-gather of per-window outputs (RCCL).  Single process = 1 GPU.

    python tools/bench_sliding.py [--volume 256 256 192] [--steps 50] [--overlap 0.25]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_sliding.py
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--volume", type=int, nargs=3, default=[256, 256, 192])
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--overlap", type=float, default=0.25)
    ap.add_argument("--classes", type=int, default=16)
    ap.add_argument("--gather-fp16", action="store_true")
    ap.add_argument("--sw-batch", type=int, default=1)
    args = ap.parse_args()
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    from diff_unet_amos_amd.diff_unet import DiffUNet
    from diff_unet_amos_amd import inference
    torch.manual_seed(0)
    net = DiffUNet(in_channels=1, out_channels=args.classes, sample_steps=args.steps).to(dev).eval()
    vol = torch.rand(1, 1, *args.volume, generator=torch.Generator().manual_seed(1)).to(dev)
    _, _, _, _, starts = inference._plan(vol, (96, 96, 96), args.overlap)
    with torch.no_grad():
        net(vol[:, :, :96, :96, :96].contiguous().repeat(args.sw_batch, 1, 1, 1, 1), pred_type="ddim_sample")   # warm-up: packs weights, captures the graph
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        if world > 1:
            out = inference.sharded_sliding_window_inference(vol, (96, 96, 96), args.sw_batch, net, args.overlap, pred_type="ddim_sample",
                                                            gather_dtype=torch.float16 if args.gather_fp16 else None)
        else:
            out = inference.sliding_window_inference(vol, (96, 96, 96), args.sw_batch, net, args.overlap, pred_type="ddim_sample")
        seg = inference.binarise(out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        nwin = len(starts)
        print(json.dumps({"workload": "sliding-window DDIM inference (BASELINE configs[2])", "volume": args.volume,
                          "windows": nwin, "ddim_steps": args.steps, "n_gpus": world, "sw_batch_size": args.sw_batch, "seconds_per_volume": dt,
                          "voxel_steps_per_s": nwin * 96 ** 3 * args.steps / dt, "windows_per_s": nwin / dt,
                          "foreground_fraction": float(seg.mean()), "finite": bool(torch.isfinite(out).all())}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
