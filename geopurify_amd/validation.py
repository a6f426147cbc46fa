"""Evaluation driver -- the build's counterpart of run/validation.py for the per-scene hot path.

Reproduces from the reference driver: the CLI (`--config`, `--split_idx`, `--split_total`, trailing
`KEY VALUE` overrides, run/validation.py:64-94), dataset name from `data_root` (:99-107), seeding
(:111-116), accepted checkpoint formats (:209-229), the contiguous split rule (:269-280), one scene per
step, the per-scene tail (:413-439: normalise, classify, zero-row nearest fill using columns 1:4 of the
[N,3] coordinates, i.e. (y, z) only, then IoU counts), the running Base/Novel/All meters and the log
strings (:490-553), and the return value (mIoU_Base, mIoU_Novel).

Differences, on purpose: scenes come from the seeded synthetic generator (no datasets offline) and the 2D
VLM is the synthetic stand-in; with torch.distributed initialised each rank evaluates its own slice of
the scene list and ONE int64 all-reduce merges the counts (the reference's per-scene all_reduce calls are
dead code, :441-450); counts are exact int64 instead of fp32.
"""
import argparse
import logging
import os
import random

import numpy as np
import torch

from . import config as gp_config
from . import ops, sharding


def get_logger():
    logger = logging.getLogger("main-logger")
    if not logger.handlers:
        logger.setLevel(logging.INFO)
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter("[%(asctime)s %(levelname)s %(filename)s line %(lineno)d %(process)d] %(message)s"))
        logger.addHandler(h)
    return logger


def get_parser(argv=None, make_dirs=True):
    """run/validation.py:64-94."""
    parser = argparse.ArgumentParser(description="geopurify.")
    parser.add_argument("--config", type=str, default="config/geopurify_synthetic_scannet.yaml", help="config file")
    parser.add_argument("--split_idx", type=int, default=0)
    parser.add_argument("--split_total", type=int, default=1)
    parser.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    a = parser.parse_args(argv)
    cfg = gp_config.load_cfg_from_cfg_file(a.config)
    if a.opts:
        cfg = gp_config.merge_cfg_from_list(cfg, a.opts)
    cfg.split_idx, cfg.split_total = a.split_idx, a.split_total
    if make_dirs and cfg.get("save_path"):
        for sub in ("", "model", "result", "result/last", "result/best"):
            os.makedirs(os.path.join(cfg.save_path, sub), exist_ok=True)
    return cfg


def get_dataset_name(data_root: str) -> str:
    """run/validation.py:99-107."""
    dr = data_root.lower()
    if "matterport" in dr:
        return "matterport"
    if "scannet" in dr:
        return "scannet"
    raise ValueError(f"cannot identify the dataset from data_root: {data_root}")


def load_student_checkpoint(student, path, logger=None):
    """run/validation.py:209-229: {'model_state_dict': ...}, {'state_dict': ...} or a bare state_dict."""
    ckpt = torch.load(path, map_location="cpu")
    sd = ckpt.get("model_state_dict", ckpt.get("state_dict", ckpt)) if isinstance(ckpt, dict) else ckpt
    sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
    student.load_state_dict(sd)
    if logger:
        logger.info("=> loaded checkpoint '{}'".format(path))


def scene_tail(hot_path, eval_results, scene_coords, scene_label, test_classes, test_ignore_label, counts):
    """run/validation.py:413-439 for one scene; adds the scene's (I, O, T) histograms into `counts`."""
    feats = eval_results["scene_features"]
    text_norm = torch.nn.functional.normalize(eval_results["text_features"], dim=-1).contiguous()
    if text_norm.shape[0] > 32 and feats.shape[1] % 32 == 0:
        pred, zero = ops.classify_argmax_gemm(feats, text_norm)
    else:
        pred, zero = ops.classify_argmax(feats, text_norm, eval_results["logit_scale"])
    # zero-feature points take the prediction of the nearest non-zero point, measured on columns 1:4 of the
    # [N,3] coordinates = (y, z) only (the reference's slice quirk, run/validation.py:422-423)
    yz = torch.zeros_like(scene_coords)
    yz[:, 0], yz[:, 1] = scene_coords[:, 1], scene_coords[:, 2]
    nn = ops.nn1_masked(yz.contiguous(), 1 - zero, zero)
    pred = torch.where(nn >= 0, pred[nn.clamp(min=0)], pred)
    ops.iou_hist(pred, scene_label, test_classes, list(test_ignore_label), counts)
    return pred


def validate(scenes, evaluate_fn, args, hot_path=None, logger=None, rank=0, world_size=1, offer_fn=None):
    """scenes: list of (scene_id, batch) providers (callables returning a SceneBatch);
    evaluate_fn(batch, scene_id) -> the evaluate_scene dict.  Returns (mIoU_Base, mIoU_Novel)."""
    logger = logger or get_logger()
    device = torch.device("cuda", torch.cuda.current_device())
    C = args.test_classes
    counts = torch.zeros((3, C), dtype=torch.int64, device=device)
    split = args.get("category_split")
    summary = None
    with torch.no_grad():
        # offer_fn given: one scene of look-ahead -- the provider of scene i + 1 runs before scene i is evaluated and its batch is
        # offered to the evaluator (`offer_fn(batch, scene_id)`: SonataXAffinityTrainer.offer_next), which lifts it beside scene i's
        # student.  Without it the providers run scene by scene, each right before its evaluation (rounds 1-5).
        ahead = offer_fn is not None
        nxt = scenes[0][1]() if (scenes and ahead) else None
        for i, (scene_id, provider) in enumerate(scenes):
            if ahead:
                batch = nxt
                nxt = scenes[i + 1][1]() if i + 1 < len(scenes) else None
            else:
                batch = provider()
            if batch is None:
                print(f"Warning: batch_data is None at iteration {i}, skipping...")
                continue
            if ahead and nxt is not None:
                offer_fn(nxt, scenes[i + 1][0])
            res = evaluate_fn(batch, scene_id)
            scene_tail(hot_path, res, batch.scene_coords, batch.scene_label, C, args.test_ignore_label, counts)
            if rank == 0:
                logger.info("Process: [{}/{}]".format(i, len(scenes)))
                summary = sharding.summarize(counts, split)          # running (rank-local) metrics, as the reference
                for line in sharding.log_lines(summary):
                    logger.info(line)
    sharding.reduce_counts(counts)
    summary = sharding.summarize(counts, split)
    if rank == 0 and world_size > 1:
        logger.info("=> merged {} ranks".format(world_size))
        for line in sharding.log_lines(summary):
            logger.info(line)
    base = summary.get("Base", summary["All"])["mIoU"]
    novel = summary.get("Novel", summary["All"])["mIoU"]
    return (base, novel), counts


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
    if dataset_name == "matterport" and cfg_s.dataset != "matterport":
        cfg_s = syn.CONFIGS["M"]
    n_scenes = int(args.get("num_scenes", 4))
    scene_ids = [f"{dataset_name}_synthetic_{i:04d}" for i in range(n_scenes)]
    # scene sizes: the config's point count, or -- `synthetic_sizes scannet_val` -- the 312 ScanNet-val scene sizes in file
    # order (labelled-point counts of dataset/scannet_val_metrics.tsv; BASELINE configs[2])
    if args.get("synthetic_sizes") == "scannet_val":
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "scannet_val_point_counts.txt")
        table = [int(float(v)) for v in open(path).read().split()]
        sizes = {sid: table[i % len(table)] for i, sid in enumerate(scene_ids)}
    else:
        sizes = {sid: cfg_s.num_points for sid in scene_ids}
    scene_ids = sharding.get_batch_scenes(scene_ids, args.split_idx, args.split_total)      # the reference's manual split
    if rank == 0:
        logger.info(f"=> Validation split: {args.split_idx + 1}/{args.split_total}, num scenes: {len(scene_ids)}")
    if world > 1:
        policy = args.get("shard_policy", "contiguous")
        idx = list(range(len(scene_ids)))
        mine = sharding.assign_scenes_lpt([sizes[s] for s in scene_ids], world)[rank] if policy == "lpt" \
            else sharding.get_batch_scenes(idx, rank, world)
        scene_ids = [scene_ids[i] for i in mine]
    model = SonataXAffinityTrainer(args, None, None, device="cuda", use_lseg=False, feature_dim=cfg_s.feat_dim,
                                   allow_deferred_vlm=True)           # the synthetic VLM is attached per scene below
    model.num_pool_iters = int(args.get("pool_iters", 19))
    if args.get("resume"):
        load_student_checkpoint(model.affinity_student, args.resume, logger)
    model.eval()
    hp = model._hot_path()
    state = {}

    def provider_for(sid):
        def make():
            seed = int(args.get("manual_seed") or 0) + int(sid.rsplit("_", 1)[1])
            import dataclasses
            scene = syn.make_scene(dataclasses.replace(cfg_s, num_points=sizes[sid]) if sizes[sid] != cfg_s.num_points else cfg_s, seed)
            state["vlm"] = pl.SyntheticVLM(syn.make_vlm_outputs(cfg_s, cfg_s.num_views, seed), "cuda")
            rigid = pl.scene_rigid_transform(cfg_s.voxel_size, seed)
            return pl.build_scene_batch(pl.upload_scene(scene, "cuda"), rigid, "cuda", val_keep=int(args.get("val_keep", 10 ** 7)))
        return make

    vlm_of = {}

    def provider_with_vlm(sid):
        make = provider_for(sid)

        def run():
            b = make()
            vlm_of[sid] = state["vlm"]                  # (the synthetic 2D outputs are per scene)
            return b
        return run

    def evaluate(batch, sid):
        model.vlm = vlm_of.pop(sid)
        return model.evaluate_scene(batch, vis_prefix=sid)

    def offer(batch, sid):
        model.offer_next(batch, vlm=vlm_of[sid])

    result, counts = validate([(s, provider_with_vlm(s)) for s in scene_ids], evaluate, args, hp, logger, rank, world,
                              offer_fn=offer if args.get("look_ahead", True) else None)
    if rank == 0:
        logger.info("==> Train/Eval done!")
    if world > 1:
        torch.distributed.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
