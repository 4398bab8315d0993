"""ORACLE (test infrastructure only): NumPy float64 restatement of the adaptive density control of Kerbl et al.
2023, section 5.2, as gaussian-splatting_amd/densify.py applies it between training steps.  The reference has no
counterpart (BASELINE.json configs[3] implies it), so this pins the package's torch implementation to an
independent statement of the algorithm, element by element, instead of to invariants only.

    hot      = mean over the views a gaussian was visible in of ||dL/d(2D centre)|| * max(W,H)/2  >=  grad_threshold
    big      = max activated scale (exp(log_scale) * 3/H, forward_pass.py:49-50)  >  percent_dense * scene_extent
    prunable = opacity < min_opacity         (spawns nothing, is removed)
    clone    = hot & ~big & ~prunable        (copy appended)
    split    = hot &  big & ~prunable        (parent removed; two children at mean + R(q) (eps * scale), log-scale
                                              lowered by log(split_factor))
    order of the result: kept gaussians (old order), clones (old order), children of draw 0, children of draw 1."""
import numpy as np


def quat_rot(q):
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=1).reshape(-1, 3, 3)


def densify_ref(means, log_scales, quats, opacity, sh, grad_accum, denom, eps, *, H, grad_threshold, percent_dense,
                scene_extent, min_opacity=0.005, split_factor=1.6, opacity_is_logit=False, max_gaussians=None):
    """eps: two [n_split,3] arrays of standard-normal draws (the same samples the implementation used).
    Returns (dict of new parameter arrays, dict of index arrays keep / clone / split into the old set)."""
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    means, log_scales, quats, opacity, sh = f64(means), f64(log_scales), f64(quats), f64(opacity), f64(sh)
    n = len(means)
    avg = f64(grad_accum) / np.maximum(f64(denom), 1.0)
    hot = avg >= grad_threshold
    act = np.exp(log_scales) * (3.0 / H)
    big = act.max(axis=1) > percent_dense * scene_extent
    op = 1.0 / (1.0 + np.exp(-opacity)) if opacity_is_logit else opacity
    prunable = op < min_opacity
    clone = hot & ~big & ~prunable
    split = hot & big & ~prunable
    if max_gaussians is not None:       # splits first, then clones, each limited to the first candidates that fit
        room = max(0, int(max_gaussians) - (n - int(prunable.sum())))
        for m in (split, clone):
            idx = np.nonzero(m)[0]
            if len(idx) > room:
                m[idx[room:]] = False
            room = max(room - int(m.sum()), 0)
    keep = ~split & ~prunable
    ki, ci, si = np.nonzero(keep)[0], np.nonzero(clone)[0], np.nonzero(split)[0]
    R = quat_rot(quats[si]) if len(si) else np.zeros((0, 3, 3))
    kids = []
    for k in range(2):
        e = f64(eps[k])[:len(si)]
        offs = np.einsum("nij,nj->ni", R, e * act[si])
        kids.append(dict(means=means[si] + offs, log_scales=log_scales[si] - np.log(split_factor), quats=quats[si],
                         opacity=opacity[si], sh=sh[si]))
    src = dict(means=means, log_scales=log_scales, quats=quats, opacity=opacity, sh=sh)
    new = {name: np.concatenate([a[ki], a[ci], kids[0][name], kids[1][name]], axis=0) for name, a in src.items()}
    return new, dict(keep=ki, clone=ci, split=si, pruned=np.nonzero(prunable)[0])
