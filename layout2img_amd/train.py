"""End-to-end training entry (the loop of reference train_context_app_v2.py:38-217 on the HIP path):

    python -m layout2img_amd.train --dataset coco --batch_size 32 --total_epoch 200 --out_path ./outputs/
    python -m torch.distributed.run --nproc-per-node 8 -m layout2img_amd.train ...      (one process per GPU, RCCL)

Same flags as the reference (:220-237). `--synthetic N` trains on N synthetic batches instead of a dataset on disk.
Checkpoints are written every 5 epochs in the reference's layout (G_<e>.pth / D_<e>.pth, `module.` prefix) plus the
optimizer state; `--checkpoint_epoch E` resumes from them (the reference's default of 55 would make a fresh run
return at once, :74-75 -- here 0 means "start fresh").
"""
import argparse
import os
import time

import torch


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", type=str, default="coco")
    ap.add_argument("--batch_size", type=int, default=16, help="per GPU")
    ap.add_argument("--total_epoch", type=int, default=200)
    ap.add_argument("--d_lr", type=float, default=0.0001)
    ap.add_argument("--g_lr", type=float, default=0.0001)
    ap.add_argument("--out_path", type=str, default="./outputs/")
    ap.add_argument("--checkpoint_epoch", type=int, default=0)
    ap.add_argument("--data_root", type=str, default=".")
    ap.add_argument("--synthetic", type=int, default=0, help="iterations per epoch on synthetic layouts (no dataset on disk; a pool of 16 device-resident batches is cycled)")
    ap.add_argument("--vgg_weights", type=str, default="", help="torchvision vgg19 state_dict (.pth) for the perceptual loss; empty = term omitted")
    ap.add_argument("--img_size", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--host_resize", action="store_true", help="resize images in the loader workers (PIL, as the reference) instead of on the GPU")
    ap.add_argument("--num_workers", type=int, default=2)
    ap.add_argument("--graph_iters", type=int, default=4,
                    help="one GPU, graph mode: iterations per graph replay (the launch of a replay costs ~0.3 ms; batches are collected in groups of this many)")
    ap.add_argument("--no_graph", action="store_true", help="run every iteration eagerly (default at one GPU: replay the captured HIP graph of the iteration)")
    args = ap.parse_args(argv)

    import layout2img_amd as L
    from layout2img_amd import data, generator, parallel
    from layout2img_amd.synthetic import make_batch
    rank, world, local = parallel.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    opd = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    num_classes = 184 if args.dataset == "coco" else 179                      # :44-45
    out_path = os.path.join(args.out_path, args.dataset, str(args.img_size))  # :48
    if args.dataset == "coco":
        netG = (L.ResnetGenerator128_context if args.img_size == 128 else L.ResnetGenerator64_context)(num_classes=num_classes)
    else:
        netG = generator.context_aware_generator(num_classes=num_classes)
    netD = (L.CombineDiscriminator128_app if args.img_size == 128 else L.CombineDiscriminator64)(num_classes=num_classes)
    netG.finalize(dev, opd), netD.finalize(dev, opd)
    vgg = None
    if args.vgg_weights:
        vgg = L.VGGLoss().finalize(dev, opd)
        vgg.load_torchvision_state_dict(torch.load(args.vgg_weights, map_location="cpu"))
    trainer = L.GanTrainer(netG, netD, g_lr=args.g_lr, d_lr=args.d_lr, vgg=vgg)
    start = 0
    if args.checkpoint_epoch > 0:
        start = L.load_checkpoint(out_path, args.checkpoint_epoch, netG, netD, trainer.g_opt, trainer.d_opt)
    if args.synthetic:
        # a pool of synthetic batches made ONCE, on the device, and cycled through: drawing 1.5 M uniform numbers on the host and a
        # blocking copy from pageable memory per iteration serialise host and GPU (measured: 48 ms per iteration against 19.5)
        pool = [make_batch(args.batch_size, args.img_size, args.dataset, seed=100003 + i * world + rank, device=dev)[:3]
                for i in range(min(args.synthetic, 16))]
        batches = lambda epoch: (pool[(epoch * args.synthetic + i) % len(pool)] for i in range(args.synthetic))
    else:
        ds = data.get_dataset(args.dataset, args.img_size, args.data_root, raw_images=not args.host_resize)
        loader = data.make_loader(ds, args.batch_size, num_workers=args.num_workers, shuffle=True, rank=rank, world=world)
        to_dev = data.DeviceBatcher(dev, (args.img_size, args.img_size))

        def batches(epoch):
            if hasattr(loader.sampler, "set_epoch"):
                loader.sampler.set_epoch(epoch)
            return (to_dev(b) for b in loader)
    netG.train(), netD.train()
    t0 = time.time()
    # One GPU: the iteration is ~900 launches and costs the host as long as the GPU, so it is captured ONCE (on the first
    # batch: shapes are fixed, drop_last) as a HIP graph and replayed with each batch copied into the graph's static inputs
    # (GanTrainer.capture / step_graphed -- what bench.py times). The latents z are drawn per iteration and passed in, as the
    # eager step would draw them. Data parallel runs stay eager (collectives are not captured).
    graphed = tried = False
    z_dim, n_obj = trainer.z_dim, None

    def run(real, label, bbox):
        nonlocal graphed, tried, n_obj
        b, o = label.shape[0], label.shape[1]
        z = torch.randn(b, o, z_dim, device=real.device)
        z_im = torch.randn(b, 128, device=real.device)
        if world == 1 and not args.no_graph:
            if not tried:   # ONE attempt: a failed capture falls back to the eager loop for the rest of the run
                from layout2img_amd.trainer import restore_state, snapshot_state
                tried = True
                st = snapshot_state(trainer)       # the capture runs warm-up iterations on this batch: undo them, the
                try:
                    graphed = trainer.capture(real, label, bbox, z, z_im)
                except Exception as e:
                    # (an invalidated capture leaves the HIP runtime in a sticky error state: nothing in this process can launch
                    #  any more -- tools/parity/capture_failure_probe.py; the eager loop needs a fresh process)
                    raise RuntimeError(f"HIP graph capture of the training iteration failed ({type(e).__name__}: {str(e)[:200]}); "
                                       "run again with --no_graph") from e
                restore_state(trainer, st)         # first REPLAY is the first training iteration
                n_obj = (b, o)
            if graphed and (b, o) == n_obj:
                return trainer.step_graphed(real, label, bbox, z, z_im)
        return trainer.step(real, label, bbox, z, z_im)

    # graph mode: batches are collected in groups of --graph_iters and run by ONE replay of a graph of that many consecutive iterations
    # (GanTrainer.capture_multi, captured on the first full group); what is left of an epoch runs on the one-iteration graph
    group, multi = [], None

    def run_group():
        nonlocal multi
        rs = None
        if graphed and multi is None and len(group) == args.graph_iters:
            try:
                multi = trainer.capture_multi(group)
            except Exception as e:
                raise RuntimeError(f"HIP graph capture of {args.graph_iters} iterations failed ({type(e).__name__}: {str(e)[:200]}); "
                                   "run again with --graph_iters 1") from e
        if multi and len(group) == args.graph_iters and all((b[1].shape[0], b[1].shape[1]) == n_obj for b in group):
            rs = trainer.step_graphed_multi(group)
        else:
            rs = [trainer.step_graphed(*b) if (graphed and (b[1].shape[0], b[1].shape[1]) == n_obj) else trainer.step(*b) for b in group]
        group.clear()
        return rs[-1]

    def log(epoch, idx, r):
        print(f"Time Elapsed: {time.time() - t0:.0f}s  Epoch[{epoch + 1}/{args.total_epoch}], Step[{idx + 1}], "
              f"d_loss: {float(r['d_loss']):.4f}, g_loss: {float(r['g_loss']):.4f}, pixel: {float(r['pixel']):.4f}", flush=True)

    for epoch in range(start, args.total_epoch):
        for idx, (real, label, bbox) in enumerate(batches(epoch)):
            if graphed and args.graph_iters > 1:
                b, o = label.shape[0], label.shape[1]
                group.append((real, label, bbox, torch.randn(b, o, z_dim, device=real.device), torch.randn(b, 128, device=real.device)))
                r = run_group() if (len(group) == args.graph_iters or (idx + 1) % 500 == 0) else None
            else:
                r = run(real, label, bbox)
            if rank == 0 and (idx + 1) % 500 == 0:   # (the only host synchronisation: logging, as :191-209)
                log(epoch, idx, r)
        if group:
            run_group()
        trainer.flush()   # (a deferred generator step of the last iteration, data parallel) before the weights are read
        if rank == 0 and (epoch + 1) % 5 == 0:       # :215-217
            L.save_checkpoint(out_path, epoch + 1, netG, netD, trainer.g_opt, trainer.d_opt)
    if rank == 0:
        L.save_checkpoint(out_path, args.total_epoch, netG, netD, trainer.g_opt, trainer.d_opt)
    return trainer


if __name__ == "__main__":
    main()
