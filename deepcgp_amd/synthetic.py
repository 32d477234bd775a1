"""Synthetic minibatches and model specifications at the BASELINE.json configurations.

There is no dataset on the GPU box, so every measurement runs on seeded synthetic inputs that
mimic the reference's preprocessing and initialisation:

* images: N(0,1) noise, Gaussian-blurred (sigma = 1.5 px) per channel, globally standardised
  (stands in for StandardScaler, /root/reference/conv_gp/mnist.py:40-45, cifar.py:34-40);
* inducing patches Z: M patches cut at random from 64 further images + N(0, 0.01^2)
  (stands in for the k-means-of-patches init, conv_gp/kernels.py:147-164); deeper layers cut
  them from the images propagated through the identity convolution like
  conv_gp/models.py:29-33,104;
* RBF variance 5, lengthscale 5 (conv_gp/models.py:115-116); jitter 1e-3 (gpflowrc:11);
* q_mu ~ N(0,1) (or zeros), q_sqrt = scale * chol(Kuu) with scale 1e-5 (conv_gp/models.py:138)
  for conv layers and 1.0 for the head (SVGP_Layer default), patch_weights = 1.

A "model spec" is a plain dict -- the neutral description both the HIP-backed classes
(``deepcgp_amd.models.build_from_spec``) and the test oracle are built from:

    {"S": int, "num_data": int,
     "convs": [{"H","W","C","f","s","M","R","Z","Z0","variance","ls","q_mu","q_sqrt","white"}...],
     "head":  {"H","W","C","f","s","M","R","Z","variance","ls","w","q_mu","q_sqrt","white"}}
"""
import numpy as np

JITTER = 1e-3

# name -> (image HWC, [conv (f, s, R)...], head (f, s), M, batch, num_data)
CONFIGS = {
    # BASELINE.json configs[0]: the reference's own CPU-runnable case (head-only "1-layer").
    "cfg1_mnist_H_M32": dict(hwc=(28, 28, 1), convs=[], head=(5, 1), M=32, batch=32, num_data=1000),
    # configs[1], paper "1-layer" = head only, and the repo-default conv layer + head.
    "cfg2_mnist_H_M256": dict(hwc=(28, 28, 1), convs=[], head=(5, 1), M=256, batch=32, num_data=60000),
    "cfg2_mnist_CH_M256": dict(hwc=(28, 28, 1), convs=[(5, 2, 10)], head=(5, 1), M=256, batch=32,
                               num_data=60000),
    # configs[2]
    "cfg3_mnist_3layer_M256": dict(hwc=(28, 28, 1), convs=[(4, 2, 10), (5, 1, 10)], head=(5, 1), M=256,
                                   batch=64, num_data=60000),
    # configs[3]
    "cfg4_cifar_3layer_M384": dict(hwc=(32, 32, 3), convs=[(4, 2, 10), (5, 1, 10)], head=(5, 1), M=384,
                                   batch=32, num_data=50000),
    # configs[4]
    "cfg5_mnist_H_M1024": dict(hwc=(28, 28, 1), convs=[], head=(5, 1), M=1024, batch=128, num_data=60000),
    "cfg5_mnist_CH_M1024": dict(hwc=(28, 28, 1), convs=[(5, 2, 10)], head=(5, 1), M=1024, batch=128,
                                num_data=60000),
}


def _blur_images(rng, n, H, W, C, sigma=1.5):
    from scipy.ndimage import gaussian_filter
    x = rng.standard_normal((n, H, W, C))
    x = gaussian_filter(x, sigma=(0, sigma, sigma, 0), mode="nearest")
    return (x - x.mean()) / x.std()


def _rbf(A, B, variance, ls):
    A = A / ls
    B = B / ls
    d = np.sum(A * A, 1)[:, None] + np.sum(B * B, 1)[None, :] - 2.0 * A @ B.T
    return variance * np.exp(-0.5 * d)


def _cut_patches(rng, imgs, M, f):
    n, H, W, C = imgs.shape
    out = np.empty((M, f * f * C))
    for i in range(M):
        k = rng.integers(0, n)
        y = rng.integers(0, H - f + 1)
        x = rng.integers(0, W - f + 1)
        out[i] = imgs[k, y:y + f, x:x + f, :].reshape(-1)
    return out + 0.01 * rng.standard_normal(out.shape)


def _identity_conv(imgs, f, R, s):
    """IdentityConv2dMean (conv_gp/mean_functions.py:6-26): every output map is the sum over the
    input channels of the centre pixel of the window."""
    n, H, W, C = imgs.shape
    Ho, Wo = (H - f) // s + 1, (W - f) // s + 1
    c0 = f // 2
    centre = imgs[:, c0:c0 + (Ho - 1) * s + 1:s, c0:c0 + (Wo - 1) * s + 1:s, :].sum(-1)
    return np.repeat(centre[..., None], R, axis=-1)


def _acos0(A, B, variance, wv, bv):
    """ArcCosine(order=0) Gram matrix (host helper for synthetic q_sqrt initialisation only)."""
    num = wv * (A @ B.T) + bv
    da = np.sqrt(wv * np.sum(A * A, 1) + bv)
    db = np.sqrt(wv * np.sum(B * B, 1) + bv)
    theta = np.arccos(1e-15 + (1.0 - 2e-15) * num / da[:, None] / db[None, :])
    return variance * (np.pi - theta) / np.pi


def make_spec(hwc, convs, head, M, S=10, num_data=60000, seed=0, white=False, q_mu_random=True,
              conv_q_sqrt_scale=1e-5, head_q_sqrt_scale=1.0, head_outputs=10, variance=5.0, ls=5.0, base_kernel="rbf", head_kernel="conv"):
    """Build a model spec (see module docstring) with seeded synthetic parameters."""
    rng = np.random.default_rng(seed)
    H, W, C = hwc
    init_imgs = _blur_images(rng, 64, H, W, C)
    spec = {"S": int(S), "num_data": int(num_data), "convs": []}
    h, w, c = H, W, C
    for (f, s, R) in convs:
        Z = _cut_patches(rng, init_imgs, M, f)
        if base_kernel == "acos":   # conv layers only (conv_gp/models.py:113-121); gpflow defaults 1, 1, 1
            Kuu = _acos0(Z, Z, 1.0, 1.0, 1.0) + JITTER * np.eye(M)
        else:
            Kuu = _rbf(Z, Z, variance, ls) + JITTER * np.eye(M)
        Lu = np.linalg.cholesky(Kuu)
        layer = dict(H=h, W=w, C=c, f=f, s=s, M=M, R=R, Z=Z, Z0=Z.copy(), variance=variance, ls=ls, base=base_kernel,
                     q_mu=(rng.standard_normal((M, R)) if q_mu_random else np.zeros((M, R))),
                     q_sqrt=(np.tile(np.eye(M)[None], [R, 1, 1]) if white
                             else np.tile(Lu[None], [R, 1, 1]) * conv_q_sqrt_scale),
                     white=bool(white))
        spec["convs"].append(layer)
        init_imgs = _identity_conv(init_imgs, f, R, s)
        h, w, c = init_imgs.shape[1:]
    f, s = head
    if head_kernel == "rbf":   # dense RBF-ARD head on the flattened features (--last-kernel rbf, conv_gp/models.py:160-168)
        flat = init_imgs.reshape(init_imgs.shape[0], -1)
        D = flat.shape[1]
        Z = flat[rng.integers(0, flat.shape[0], M)] + 0.05 * rng.standard_normal((M, D))
        ls_ard = ls * (0.8 + 0.4 * rng.random(D))
        Zs = Z / ls_ard
        Ku = _rbf(Zs, Zs, variance, 1.0) + JITTER * np.eye(M)
        spec["head"] = dict(H=h, W=w, C=c, f=f, s=s, M=M, R=head_outputs, Z=Z, variance=variance, ls=ls, ls_ard=ls_ard,
                            kernel="rbf", w=np.ones(1),
                            q_mu=(rng.standard_normal((M, head_outputs)) if q_mu_random else np.zeros((M, head_outputs))),
                            q_sqrt=(np.tile(np.eye(M)[None], [head_outputs, 1, 1]) if white
                                    else np.tile(np.linalg.cholesky(Ku)[None], [head_outputs, 1, 1]) * head_q_sqrt_scale),
                            white=bool(white))
        return spec
    f, s = head
    Z = _cut_patches(rng, init_imgs, M, f)
    Ku = _rbf(Z, Z, variance, ls) + JITTER * np.eye(M)
    P = ((h - f) // s + 1) * ((w - f) // s + 1)
    spec["head"] = dict(H=h, W=w, C=c, f=f, s=s, M=M, R=head_outputs, Z=Z, variance=variance, ls=ls,
                        w=np.ones(P),
                        q_mu=(rng.standard_normal((M, head_outputs)) if q_mu_random
                              else np.zeros((M, head_outputs))),
                        q_sqrt=(np.tile(np.eye(M)[None], [head_outputs, 1, 1]) if white
                                else np.tile(np.linalg.cholesky(Ku)[None], [head_outputs, 1, 1]) * head_q_sqrt_scale),
                        white=bool(white))
    return spec


def make_batch(hwc, batch, seed=0):
    """Synthetic minibatch: X [batch, H*W*C] float64 (standardised blurred noise), Y [batch] int32."""
    rng = np.random.default_rng(10_000 + seed)
    H, W, C = hwc
    X = _blur_images(rng, batch, H, W, C).reshape(batch, H * W * C)
    Y = rng.integers(0, 10, size=batch).astype(np.int32)
    return np.ascontiguousarray(X), Y


def layer_output_dims(spec):
    """[(rows-per-image output width D_l)] for every layer: conv layers P*R, head R."""
    dims = []
    for c in spec["convs"]:
        P = ((c["H"] - c["f"]) // c["s"] + 1) * ((c["W"] - c["f"]) // c["s"] + 1)
        dims.append(P * c["R"])
    dims.append(spec["head"]["R"])
    return dims


def make_noise(spec, batch, seed=0):
    """z ~ N(0,1) per layer, shape [S, batch, D_l], indexed by (sample, image) so that a shard of
    the images sees exactly the rows it would see in the full batch."""
    rng = np.random.default_rng(20_000 + seed)
    return [rng.standard_normal((spec["S"], batch, d)) for d in layer_output_dims(spec)]


def make_config(name, seed=None, S=10, **overrides):
    """(spec, X, Y) for one of CONFIGS; seeds follow BASELINE.md (1234 + index)."""
    cfg = dict(CONFIGS[name])
    cfg.update(overrides)
    if seed is None:
        seed = 1234 + list(CONFIGS).index(name)
    spec = make_spec(cfg["hwc"], cfg["convs"], cfg["head"], cfg["M"], S=S, num_data=cfg["num_data"], seed=seed)
    X, Y = make_batch(cfg["hwc"], cfg["batch"], seed=seed)
    return spec, X, Y
