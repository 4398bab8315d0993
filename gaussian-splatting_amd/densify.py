"""Densification and pruning between training steps.

The reference has no counterpart: BASELINE.json configs[3] ("growing gaussian counts") implies the
adaptive density control of the 3D Gaussian Splatting paper (Kerbl et al. 2023, section 5.2), which
this module follows: gaussians whose accumulated screen-space positional gradient is large are
CLONED when small and SPLIT in two when large; nearly transparent ones are PRUNED.  The statistic
comes straight from K7's gradient rows (d loss / d 2D centre, words 0-1 of every row), so no extra
kernel pass is needed.  Everything runs on the device with torch ops; it is a caller of the hot
path, not part of it.  With torch.distributed the statistics are all-reduced and the random draws
are seeded, so every rank applies identical changes."""
import torch

from . import api
from .pipeline import GROUP_WIDTHS, all_reduce_sum, flat_offsets


class Densifier:
    def __init__(self, pipe, optimizer=None, grad_threshold=0.0002, percent_dense=0.01, scene_extent=None,
                 min_opacity=0.005, split_factor=1.6, seed=0, max_gaussians=None):
        self.pipe, self.opt = pipe, optimizer
        self.grad_threshold, self.percent_dense = float(grad_threshold), float(percent_dense)
        self.min_opacity, self.split_factor = float(min_opacity), float(split_factor)
        self.max_gaussians = max_gaussians
        # host generator: the draws depend on the seed alone (not on the device or its RNG implementation), so every
        # rank applies identical changes and a CPU restatement can be given the very same samples
        self.gen = torch.Generator(device="cpu").manual_seed(int(seed))
        if scene_extent is None:
            c = pipe.means
            scene_extent = float((c - c.mean(dim=0)).norm(dim=1).max().item()) * 1.1
        self.scene_extent = scene_extent
        self.reset()

    def sample(self, n):
        """[n,3] standard-normal draws for the split children (one call per child generation)."""
        return torch.randn(n, 3, generator=self.gen, dtype=torch.float32)

    def reset(self):
        dev = self.pipe.ctx.device
        self.grad_accum = torch.zeros(self.pipe.N, dtype=torch.float32, device=dev)
        self.denom = torch.zeros(self.pipe.N, dtype=torch.float32, device=dev)

    def accumulate(self, states):
        """Adds ||dL/d(2D centre)|| of every view in `states` (after backward_rows) for the gaussians
        visible in it.  Pixel-unit gradients are scaled by max(W,H)/2 (gradient per NDC unit, the
        unit the paper's 0.0002 threshold refers to)."""
        W, H = self.pipe.result_size
        k = 0.5 * max(W, H)
        for st in states:
            vis = st["radii"] > 0
            self.grad_accum += torch.where(vis, st["rows"][:, :2].norm(dim=1) * k, torch.zeros_like(self.grad_accum))
            self.denom += vis.to(torch.float32)

    def _activated_scales(self):
        W, H = self.pipe.result_size
        return torch.exp(self.pipe.log_scales) * (3.0 / H)           # forward_pass.py:49-50 scale_mod

    def step(self):
        """Clone / split / prune.  Returns a dict with the counts and the new N.  The pipeline's
        parameter tensors and the optimizer's moments are rebuilt; lanes keep their workspaces."""
        import torch.distributed as dist
        p = self.pipe
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            all_reduce_sum(p.ctx, [self.grad_accum, self.denom])
        avg = self.grad_accum / self.denom.clamp(min=1.0)
        hot = avg >= self.grad_threshold
        big = self._activated_scales().max(dim=1).values > self.percent_dense * self.scene_extent
        # Kerbl et al. section 5.2 prune after clone / split, new points included: a clone has its parent's
        # opacity and a split's children too, so a prunable gaussian spawns nothing (and is dropped below)
        opac = torch.sigmoid(p.opacity) if p.opacity_is_logit else p.opacity
        prunable = opac < self.min_opacity
        clone_mask = hot & ~big & ~prunable
        split_mask = hot & big & ~prunable
        if self.max_gaussians is not None:
            # a clone adds one gaussian and a split a net one (two children replace the parent); the budget
            # is what is left AFTER pruning
            room = max(0, int(self.max_gaussians) - (p.N - int(prunable.sum().item())))
            for m in (split_mask, clone_mask):          # keep the first `room` candidates (deterministic)
                idx = torch.nonzero(m).flatten()
                if idx.numel() > room:
                    m[idx[room:]] = False
                room -= int(m.sum().item())
                room = max(room, 0)
        groups = [p.means, p.log_scales, p.quats, p.opacity, p.sh]

        def take(mask):
            return [g[mask] for g in groups]

        clones = take(clone_mask)
        # split: two children per parent, centres drawn from the parent's footprint, scales / split_factor
        sm, ss, sq, so, ssh = take(split_mask)
        n_split = sm.shape[0]
        children = []
        if n_split:
            q = sq / sq.norm(dim=1, keepdim=True)
            w, x, y, z = q.unbind(dim=1)
            R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                             2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                             2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=1).view(-1, 3, 3)
            std = self._activated_scales()[split_mask]
            for _ in range(2):
                eps = self.sample(n_split).to(sm.device)
                offs = torch.bmm(R, (eps * std).unsqueeze(-1)).squeeze(-1)
                children.append([sm + offs, ss - torch.log(torch.tensor(self.split_factor, device=sm.device)), sq, so, ssh])
        keep = ~split_mask & ~prunable
        kept = take(keep)
        parts = [kept, clones] + children
        new_groups = [torch.cat([part[k] for part in parts], dim=0).contiguous() for k in range(5)]
        n_old, n_new = p.N, new_groups[0].shape[0]
        p.means, p.log_scales, p.quats, p.opacity, p.sh = new_groups
        p.N = n_new
        if self.opt is not None and self.opt.m is not None:
            # surviving gaussians keep their Adam moments, new ones start at zero
            self.opt.m, self.opt.v = (self._remap_moments(buf, keep, n_old, n_new) for buf in (self.opt.m, self.opt.v))
        stats = dict(cloned=int(clone_mask.sum().item()), split=int(n_split),
                     pruned=int((~keep & ~split_mask).sum().item()), N=n_new)
        self.reset()
        return stats

    @staticmethod
    def _remap_moments(flat, keep, n_old, n_new):
        (old_offs, _), (new_offs, total) = flat_offsets(n_old), flat_offsets(n_new)
        out = torch.zeros(total, dtype=flat.dtype, device=flat.device)
        n_keep = int(keep.sum().item())
        for width, o_old, o_new in zip(GROUP_WIDTHS, old_offs, new_offs):
            src = flat[o_old:o_old + width * n_old].view(n_old, width)[keep]
            out[o_new:o_new + width * n_new].view(n_new, width)[:n_keep] = src
        return out
