"""Training-loop caller of the hot path: the counterpart of
train.py:12-64 / main.py:6-23 of the reference.

The reference's loop cannot start at HEAD (defect B1: train.py:13 calls the shadowed
GaussianSet constructor) and its update rule is wrong in several ways (defect B12: uint8
wrap-around in the loss, `lr * grad * loss`, log() of a gradient, an operator-precedence slip on
the opacity line; defect B4: no dL/dimage).  This module keeps its call signature, data flow and
outputs (a jpg every 50 iterations, an npz checkpoint every 100) and implements the evident
intent: explicit L1 dL/dimage, Adam on the five parameter groups with the reference's learning
rates (main.py:11-17), `views_per_step` views per iteration instead of one, sharded over the ranks
when torch.distributed is initialised.  method="reference" selects the reference's update rule
(param -= lr * grad * l1_loss) for comparison."""
import math
import os
import random
import time

import numpy as np
import torch

from . import io as gio
from .pipeline import FlatGrads, Optimizer, SplatPipeline, View, all_reduce_sum, train_step


def default_image_loader(image_dir, result_size):
    """train.py:46: PIL-loads ./input/images/<name>, resized to result_size, as float [W,H,3] in
    0..1 (the reference compares uint8 arrays in [H,W,3]; the pipeline's images are [W,H,3])."""
    from PIL import Image as PILImage

    def load(source_image):
        im = PILImage.open(os.path.join(image_dir, source_image.name)).convert("RGB")
        im = im.resize((int(result_size[0]), int(result_size[1])))
        return np.swapaxes(np.asarray(im, dtype=np.float32) / 255.0, 0, 1)
    return load


def train_model(cameras, images, point_cloud_data, learning_rates, ctx=None, queue=None, program=None,
                iters=2000, result_size=[979, 546], t0=None, *, views_per_step=8, image_loader=None,
                image_dir="./input/images", output_dir="output/test_run", method="adam", loss="l1",
                subset_fraction=.2, seed=None, gaussians=None, save_images_every=50, checkpoint_every=100,
                concurrent_views=4, log=print, densify_every=None, densify_from=500, densify_until=15000,
                densify_options=None):
    """train.py:12-64.  cameras / images / point_cloud_data: a COLMAP model (io.read_model or the
    reference's reader); learning_rates: the dict of main.py:11-17 (keys center, scaling, opacity,
    rotation, sh).  densify_every=k runs the Densifier (clone / split / prune; no reference
    counterpart) every k iterations between densify_from and densify_until.  Returns (gaussians,
    history of per-iteration mean losses)."""
    t0 = time.perf_counter() if t0 is None else t0
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if distributed:
        # every rank must draw the same subset and the same views per iteration ("view v goes to rank
        # v % world" shards ONE list): rank 0's seed is the seed, whatever the others were given
        box = [random.SystemRandom().randrange(2 ** 31) if seed is None else int(seed)]
        dist.broadcast_object_list(box, src=0)
        seed = box[0]
    rnd = random.Random(seed)
    if gaussians is None:
        gaussians = gio.gaussians_from_point_cloud(point_cloud_data)            # train.py:13
    items = list(images.items())
    subset = rnd.sample(items, max(1, math.floor(len(items) * subset_fraction)))  # train.py:15
    loader = image_loader or default_image_loader(image_dir, result_size)

    # the reference keeps raw opacities on the training path (train.py:37) and logits on the PLY
    # path (gaussian.py:50); point-cloud initialisation stores the raw value .1
    pipe = SplatPipeline(gaussians, result_size, opacity_is_logit=False, ctx=ctx)
    dev = pipe.ctx.device
    grads = FlatGrads(pipe.N, dev)
    opt = Optimizer(pipe, learning_rates, method=method)
    den = None
    if densify_every:
        from .densify import Densifier
        den = Densifier(pipe, opt, seed=0 if seed is None else seed, **(densify_options or {}))
    target_cache = {}
    history = []
    log("setup complete in {time}s".format(time=time.perf_counter() - t0))
    if output_dir:
        os.makedirs(output_dir, exist_ok=True)

    for i in range(iters + 1):
        t1 = time.perf_counter()
        picks = [subset[rnd.randint(0, len(subset) - 1)][1] for _ in range(views_per_step)]     # train.py:21-23
        views, targets = [], []
        for src in picks:
            cam = cameras[src.camera_id]
            views.append(View(cam, gio.quat_to_rotmat(src.qvec), np.asarray(src.tvec, dtype=np.float64)))  # :25-26
            if src.id not in target_cache:
                target_cache[src.id] = torch.from_numpy(np.ascontiguousarray(loader(src), dtype=np.float32)).to(dev)
            targets.append(target_cache[src.id])
        loss_sum, n_local = train_step(pipe, views, targets, grads, loss_scale=1.0 / views_per_step,
                                       concurrent_views=concurrent_views, loss_kind=loss,
                                       states_hook=den.accumulate if den else None)
        if distributed:
            # the replicated parameters stay in lock-step only if every rank applies the same update:
            # method="reference" scales it by the loss, so the loss is the mean over the WHOLE batch
            both = torch.stack([loss_sum.reshape(()).to(torch.float32),
                                torch.tensor(float(n_local), dtype=torch.float32, device=loss_sum.device)])
            all_reduce_sum(pipe.ctx, [both])      # behind a device synchronisation on host-staged back ends
            mean_loss = float(both[0].item()) / max(float(both[1].item()), 1.0)
        else:
            mean_loss = float(loss_sum.item()) / max(n_local, 1)
        opt.step(grads, loss=mean_loss)                                           # train.py:51-56
        history.append(mean_loss)
        if den and densify_from <= i <= densify_until and i > 0 and i % densify_every == 0:
            stats = den.step()
            grads = FlatGrads(pipe.N, dev)
            log("densify: {s}".format(s=stats))
        log("pass {iter} completed in {time}s\nloss = {loss}\n-----------------------".format(
            time=time.perf_counter() - t1, iter=i, loss=mean_loss))
        if output_dir and save_images_every and i % save_images_every == 0:       # train.py:61-62
            from PIL import Image as PILImage
            image, _ = pipe.forward(views[0])
            pic = np.swapaxes(np.uint8(np.clip(image.cpu().numpy(), 0, 1) * 255), 0, 1)
            PILImage.fromarray(pic).save(os.path.join(output_dir, "pass_{iter}.jpg".format(iter=i)))
        if output_dir and checkpoint_every and i % checkpoint_every == 0:         # train.py:63-64
            gio.save_gaussians(os.path.join(output_dir, "iter_{iter}_gaus".format(iter=i)), pipe.export())
    return pipe.export(), history
