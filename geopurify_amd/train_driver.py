"""Training driver -- the build's counterpart of run/train.py for the student's training step.

Reproduces from the reference driver: the CLI (`--config` + trailing `KEY VALUE` overrides), the optimizer
(AdamW over get_param_groups() at 0.1x / 1x / 5x lr_3d, run/train.py:188-198), the scheduler (LinearLR 1e-6 -> 1 over
warmup_epochs * len(loader) iterations, then CosineAnnealingLR(eta_min = lr_3d * 1e-3), stepped per iteration, :320-325),
resume (:215-263: `model_state_dict` / bare state_dict, `optimizer_state_dict`, epoch from the checkpoint or the file
name, scheduler fast-forwarded by start_epoch * len(loader) steps, :327-334), the loop
`zero_grad -> loss = model(batch) -> backward -> step -> scheduler.step` (:346-353), the log line (:357-358) and the
checkpoints `model/affinity_predictor_last.pth` every save_freq epochs and `..._epoch_{e}.pth` every 5 epochs and at the
end, holding {'epoch', 'model_state_dict', 'optimizer_state_dict', 'tensorboard_scalars'} (:371-391).

Differences, on purpose: scenes, 2D-VLM outputs and the teacher's per-point features are synthetic (no datasets, X-Decoder
or Sonata offline); with torch.distributed initialised every rank trains on its own scenes (the same number of steps per
rank) and the student gradients are averaged by bucketed all-reduces launched inside the backward pass (sharding.GradientBuckets; a
model whose forward does not do that gets ONE all-reduce after backward: sharding.allreduce_mean_gradients)
instead of DistributedDataParallel hooks.  As in the reference's multi-GPU recipe (run/train.py:212-213 converts the student
to MinkowskiSyncBatchNorm), BatchNorm statistics and the backward reductions are taken over the rows of ALL ranks: four small
fp64 all-reduces per BatchNorm layer and step (sharding.sync_batch_stats / sync_bwd_sums), so every rank holds the same
running statistics.  Single-GPU training (every shipped config) takes the local path.
"""
import argparse
import os
import random
import re

import numpy as np
import torch
from torch.optim.lr_scheduler import CosineAnnealingLR, LinearLR, SequentialLR

from . import config as gp_config
from . import sharding
from .validation import get_dataset_name, get_logger


class AverageMeter:
    """util/util.py:10-25."""

    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def get_parser(argv=None, make_dirs=True):
    parser = argparse.ArgumentParser(description="geopurify.")
    parser.add_argument("--config", type=str, default="config/geopurify_synthetic_scannet.yaml", help="config file")
    parser.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    a = parser.parse_args(argv)
    cfg = gp_config.load_cfg_from_cfg_file(a.config)
    if a.opts:
        cfg = gp_config.merge_cfg_from_list(cfg, a.opts)
    if make_dirs and cfg.get("save_path"):
        os.makedirs(os.path.join(cfg.save_path, "model"), exist_ok=True)
    return cfg


def build_optimizer(student, base_lr, weight_decay):
    """run/train.py:188-198."""
    groups = student.get_param_groups()
    return torch.optim.AdamW([{"params": groups["input"], "lr": base_lr * 0.1, "name": "input_group"},
                              {"params": groups["middle"], "lr": base_lr, "name": "middle_group"},
                              {"params": groups["output"], "lr": base_lr * 5.0, "name": "output_group"}], weight_decay=weight_decay)


def build_scheduler(optimizer, base_lr, warmup_epochs, epochs, iters_per_epoch):
    """run/train.py:318-325."""
    warmup_iters = warmup_epochs * iters_per_epoch
    main_iters = (epochs - warmup_epochs) * iters_per_epoch
    warm = LinearLR(optimizer, start_factor=1e-6, end_factor=1.0, total_iters=max(warmup_iters, 1))
    main = CosineAnnealingLR(optimizer, T_max=max(main_iters, 1), eta_min=base_lr * 1e-3)
    return SequentialLR(optimizer, schedulers=[warm, main], milestones=[max(warmup_iters, 1)])


def load_resume(student, optimizer, path, device, logger=None):
    """run/train.py:215-263.  Returns (start_epoch, tensorboard_scalars)."""
    ck = torch.load(path, map_location=device, weights_only=False)
    student.load_state_dict(ck["model_state_dict"] if "model_state_dict" in ck else ck)
    if "epoch" in ck:
        start = ck["epoch"] + 1
    else:
        m = re.search(r"epoch_(\d+)", path)
        start = int(m.group(1)) + 1 if m else 0
    if "optimizer_state_dict" in ck:
        optimizer.load_state_dict(ck["optimizer_state_dict"])
    if logger:
        logger.info("=> loaded checkpoint '{}' (will start from epoch {})".format(path, start))
    return start, dict(ck.get("tensorboard_scalars", {})) if isinstance(ck, dict) else {}


def save_checkpoint(path, epoch, student, optimizer, scalars):
    """run/train.py:371-391."""
    torch.save({"epoch": epoch, "model_state_dict": student.state_dict(), "optimizer_state_dict": optimizer.state_dict(),
                "tensorboard_scalars": scalars}, path)


def train(model, optimizer, scheduler, loader, args, start_epoch=0, scalars=None, logger=None, rank=0, world=1):
    """run/train.py:336-391.  loader: a sized iterable of batches (re-iterated every epoch)."""
    scalars = {} if scalars is None else scalars
    student = model.affinity_student
    for epoch in range(start_epoch, args.epochs):
        epoch_log = epoch + 1
        model.train()
        meter = AverageMeter()
        for i, batch in enumerate(loader):
            optimizer.zero_grad()
            loss = model(batch)
            loss.backward()
            if world > 1 and not getattr(loss, "gradients_averaged", False):     # (training_forward averages inside its backward pass)
                sharding.allreduce_mean_gradients({n: p.grad for n, p in student.named_parameters() if p.grad is not None})
            optimizer.step()
            scheduler.step()
            meter.update(float(loss.detach()))
            if rank == 0 and i % int(args.get("print_freq", 10)) == 0:
                lr = scheduler.get_last_lr()[1]
                if logger:
                    logger.info(f"Epoch: [{epoch}][{i}/{len(loader)}]\t Loss: {float(loss.detach()):.4f}\t LR: {lr:.7f}")
                scalars.setdefault("lr", {})[epoch_log] = lr
        if rank == 0:
            scalars.setdefault("loss_train", {})[epoch_log] = meter.avg
            if args.get("save_path"):
                if epoch_log % int(args.get("save_freq", 1)) == 0:
                    save_checkpoint(os.path.join(args.save_path, "model", "affinity_predictor_last.pth"), epoch, student, optimizer, scalars)
                if epoch_log % 5 == 0 or epoch == args.epochs - 1:
                    if logger:
                        logger.info(f"Saving checkpoint at epoch {epoch}...")
                    save_checkpoint(os.path.join(args.save_path, "model", f"affinity_predictor_epoch_{epoch}.pth"), epoch, student,
                                    optimizer, scalars)
    return scalars


def main(argv=None):
    from . import pipeline as pl
    from . import synthetic as syn
    from .affinity_module import SonataXAffinityTrainer
    args = get_parser(argv)
    logger = get_logger()
    dataset_name = get_dataset_name(args.data_root)
    if args.get("manual_seed") is not None:
        random.seed(args.manual_seed)
        np.random.seed(args.manual_seed)
        torch.manual_seed(args.manual_seed)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        torch.distributed.init_process_group("nccl")
    cfg_s = syn.CONFIGS[args.get("synthetic_config", "S")]
    seed0 = int(args.get("manual_seed") or 0)
    teacher_dim = int(args.get("teacher_dim", 1088))

    class Scenes:
        """one synthetic scene per step, this rank's share: the same number of steps on every rank (ids wrap around when
        num_scenes is not a multiple of the world size), so that every rank takes part in every gradient all-reduce and
        len(loader) -- hence the LR schedule and the resume fast-forward -- is rank-independent."""

        def __init__(self, n):
            self.ids = sharding.equal_steps_scene_ids(n, rank, world)

        def __len__(self):
            return len(self.ids)

        def __iter__(self):
            for i in self.ids:
                scene = syn.make_scene(cfg_s, seed0 + i)
                model.vlm = pl.SyntheticVLM(syn.make_vlm_outputs(cfg_s, cfg_s.num_views, seed0 + i), "cuda")
                g = torch.Generator(device="cuda").manual_seed(seed0 + i)
                feats = torch.randn(cfg_s.num_points, teacher_dim, device="cuda", generator=g)
                model.teacher = lambda b, f=feats: f
                yield pl.build_scene_batch(pl.upload_scene(scene, "cuda"), pl.scene_rigid_transform(cfg_s.voxel_size, seed0 + i), "cuda")

    model = SonataXAffinityTrainer(args, None, None, device="cuda", use_lseg=False, feature_dim=cfg_s.feat_dim,
                                   hidden_dim=int(args.get("hidden_dim", 512)), allow_deferred_vlm=True).to("cuda")
    loader = Scenes(int(args.get("num_scenes", 4)))
    base_lr = float(args.get("lr_3d", 1e-4))
    optimizer = build_optimizer(model.affinity_student, base_lr, float(args.get("weight_decay", 1e-5)))
    start_epoch, scalars = 0, {}
    if args.get("resume") and os.path.isfile(args.resume):
        start_epoch, scalars = load_resume(model.affinity_student, optimizer, args.resume, "cuda", logger if rank == 0 else None)
    args.epochs = int(args.get("epochs", 1))
    scheduler = build_scheduler(optimizer, base_lr, int(args.get("warmup_epochs", 0)), args.epochs, len(loader))
    for _ in range(start_epoch * len(loader)):                      # fast-forward (:327-334)
        scheduler.step()
    if rank == 0:
        logger.info(f"=> dataset {dataset_name} (synthetic {cfg_s.name}), {len(loader)} scenes/epoch on this rank, world {world}")
    train(model, optimizer, scheduler, loader, args, start_epoch, scalars, logger, rank, world)
    if rank == 0:
        logger.info("==> Train/Eval done!")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
