"""Model assembly -- the counterpart of /root/reference/conv_gp/models.py (ModelBuilder) plus a builder
from the neutral model spec used by the benchmarks and parity tests (deepcgp_amd.synthetic)."""
import numpy as np

from .dgp import DGP_Base
from .kernels import RBF, ArcCosine, ConvKernel, AdditivePatchKernel, PatchInducingFeatures, InducingPoints
from .layers import ConvLayer, SVGP_Layer
from .likelihoods import MultiClass
from .mean_functions import Conv2dMean, IdentityConv2dMean  # noqa: F401  (the names conv_gp/models.py:11 imports)
from .views import FullView


def parse_ints(int_string):
    """conv_gp/models.py:14-18."""
    if int_string == '':
        return []
    return [int(i) for i in int_string.split(',')]


def build_layers_from_spec(spec):
    layers = []
    for c in spec["convs"]:
        view = FullView((c["H"], c["W"]), c["f"], c["C"], c["s"])
        base = ArcCosine(view.patch_length, order=0) if c.get("base", "rbf") == "acos" else RBF(view.patch_length, c["variance"], c["ls"])
        mf = c.get("mean_function")
        if mf == "conv2d":      # --identity-mean: Conv2dMean(filter_size, NHWC[3], feature_map, stride=stride), conv_gp/models.py:95-97
            mf = Conv2dMean(c["f"], c["C"], c["R"], stride=c["s"])
            mf.set_trainable(False)                                                            # models.py:100
        layer = ConvLayer(base, mf,
                          feature=PatchInducingFeatures(c["Z"]), view=view, white=c["white"], gp_count=c["R"],
                          q_mu=c["q_mu"], q_sqrt=c["q_sqrt"])
        layer.Z_prior = np.array(c.get("Z0", c["Z"]), np.float64)
        layer._build_prior_cholesky()
        layers.append(layer)
    h = spec["head"]
    view = FullView((h["H"], h["W"], h["C"]), h["f"], h["C"], h["s"])
    if h.get("kernel", "conv") == "rbf":   # dense RBF-ARD head (--last-kernel rbf)
        layers.append(SVGP_Layer(kern=RBF(h["Z"].shape[1], h["variance"], h["ls_ard"], ARD=True), num_outputs=h["R"],
                                 feature=InducingPoints(h["Z"]), mean_function=None, white=h["white"], q_mu=h["q_mu"],
                                 q_sqrt=h["q_sqrt"]))
        return layers
    cls = AdditivePatchKernel if h.get("kernel", "conv") == "add" else ConvKernel
    kern = cls(RBF(view.patch_length, h["variance"], h["ls"]), view, patch_weights=h.get("w"))
    layers.append(SVGP_Layer(kern=kern, num_outputs=h["R"], feature=PatchInducingFeatures(h["Z"]),
                             mean_function=None, white=h["white"], q_mu=h["q_mu"], q_sqrt=h["q_sqrt"]))
    return layers


def build_from_spec(spec, X, Y):
    """DGP_Base on the HIP path from a model spec (see deepcgp_amd.synthetic)."""
    return DGP_Base(X, Y, likelihood=MultiClass(10), layers=build_layers_from_spec(spec),
                    num_samples=spec["S"], minibatch_size=None, num_data=spec["num_data"], name='DGP')


def save_model_parameters(model, path, global_step=0):
    """Write the reference's checkpoint: ``np.save(path, {param.pathname: value, 'global_step': int})``
    (Experiment._save_model_parameters, conv_gp/experiment.py:56-64) -- readable by ``--load-model`` on either side
    (ModelBuilder._load_layer_parameters, conv_gp/models.py:200-240)."""
    params = {p.pathname: np.array(p.value) for p in model.parameters}
    params['global_step'] = int(global_step)
    np.save(path, params)
    return params


def learning_rate(lr, global_step, lr_decay_steps, decay_rate=0.1):
    """tf.train.exponential_decay(..., staircase=True) of conv_gp/experiment.py:71-73."""
    return float(lr) * decay_rate ** (int(global_step) // int(lr_decay_steps))


def natgrad_gamma(global_step, gamma0=0.001, steps_back=0, gamma_step=1e-3, back_step=0.2, gamma_max=1.0):
    """The NatGrad step-size schedule of conv_gp/experiment.py:74-81."""
    t = float(global_step) / 100.0
    return min((t * gamma_step + gamma0) * back_step ** steps_back, gamma_max)


def train(model, steps, lr=0.01, lr_decay_steps=50000, global_step=0, seed=0, callback=None, optimizer="Adam", gamma=0.001,
          max_retries=5, dedup_layer0=True):
    """The reference's optimisation loop (conv_gp/experiment.py:84-108 + gpflow.actions.Loop at :44).  Every step draws a
    minibatch and evaluates the ELBO and its gradient on the device (``compute_gradients``), then
      "Adam":    one device Adam step on every parameter (value, gradient and update in ONE call, ``train_step``);
      "SGD":     one plain gradient step;
      "NatGrad": a natural-gradient step on every layer's (q_mu, q_sqrt) (``DGP_Base.natgrad_step``, step size from
                 ``natgrad_gamma``), then -- as the reference's loop does, with the variational parameters switched to
                 non-trainable -- a fresh gradient and an Adam step on everything else.
    ``dedup_layer0`` (default on): propagate() tiles the minibatch S times, so the first layer sees S identical copies;
    its conditional and reverse pass are evaluated on the distinct images only -- same ELBO, same gradients, about half the
    step time at the headline configuration.
    Returns the list of ELBO values; the Python-side parameter objects are refreshed at the end (``pull_parameters``)."""
    if optimizer not in ("Adam", "NatGrad", "SGD"):
        raise ValueError("Not a supported optimizer. Try Adam or NatGrad.")     # experiment.py:109-110
    rng = np.random.default_rng(seed)
    n = model.X.shape[0]
    bs = min(model.minibatch_size or n, n)
    history = []
    steps_back = 0
    model._build()
    if getattr(model._ctx, "nranks", 1) > 1:
        # every rank would draw the same minibatch and the all-reduced gradients would count it once per rank
        raise NotImplementedError("train() drives one GPU; shard the minibatch per rank and call compute_gradients / adam_step yourself")
    dedup_before, model.dedup_layer0 = model.dedup_layer0, bool(dedup_layer0)
    nl = len(model.layers)
    for li in range(nl):
        for which in ("q_mu", "q_sqrt"):
            model.set_trainable(li, which, optimizer != "NatGrad")
    for i in range(int(steps)):
        idx = rng.choice(n, size=bs, replace=False)
        step = global_step + i
        if optimizer == "Adam":      # value, gradient and update in one device call (dcgp_model_train_step_adam)
            elbo = model.train_step(model.X[idx], model.Y[idx], learning_rate(lr, step, lr_decay_steps), seed=seed + step)
            history.append(elbo)
            if callback is not None:
                callback(step + 1, elbo)
            continue
        elbo, _ = model.compute_gradients(model.X[idx], model.Y[idx], seed=seed + step, fetch=False)
        if optimizer == "NatGrad":
            # a step that leaves the positive-definite cone is retried with gamma scaled by 0.2, at most max_retries
            # times over the run -- the InvalidArgumentError / step_back_gamma handling of experiment.py:36-49
            while True:
                try:
                    model.natgrad_step(natgrad_gamma(step, gamma, steps_back))
                    break
                except np.linalg.LinAlgError:
                    steps_back += 1
                    if steps_back > max_retries:
                        raise
            model.compute_gradients(model.X[idx], model.Y[idx], seed=seed + step, fetch=False)
        if optimizer == "SGD":
            model.sgd_step(learning_rate(lr, step, lr_decay_steps))
        else:
            model.adam_step(learning_rate(lr, step, lr_decay_steps))   # bias correction: the model's own step count
        history.append(elbo)
        if callback is not None:
            callback(step + 1, elbo)
    model.dedup_layer0 = dedup_before
    model.pull_parameters()
    return history


class AccuracyLogger(object):
    """Test accuracy the way the reference's training log computes it (conv_gp/utils/log.py:50-67): batches of 32,
    five samples per image, arg-max of the sample-mean class probabilities.  Each batch is one device call
    (``predict_proba``); only N x num_classes probabilities cross the bus."""
    title = 'test_accuracy'

    def __init__(self, X_test, Y_test, batch_size=32, num_samples=5):
        self.X_test, self.Y_test = X_test, np.reshape(Y_test, (-1,))
        self.batch_size, self.num_samples = int(batch_size), int(num_samples)

    def __call__(self, model, seed=0):
        correct = 0
        for i, lo in enumerate(range(0, len(self.Y_test), self.batch_size)):
            sl = slice(lo, lo + self.batch_size)
            p = model.predict_proba(self.X_test[sl], self.num_samples, seed=seed + i)
            correct += int((p.argmax(axis=1) == self.Y_test[sl]).sum())
        return correct / max(self.Y_test.size, 1)


def identity_conv(NHWC_X, filter_size, feature_maps_in, feature_maps_out, stride, count=1000):
    """Propagate random images through IdentityConv2dMean to initialise the next layer
    (conv_gp/models.py:29-33, conv_gp/mean_functions.py:6-26)."""
    X = NHWC_X[np.random.choice(np.arange(NHWC_X.shape[0]), size=min(count, NHWC_X.shape[0]))]
    n, H, W, C = X.shape
    Ho, Wo = (H - filter_size) // stride + 1, (W - filter_size) // stride + 1
    c0 = filter_size // 2
    centre = X[:, c0:c0 + (Ho - 1) * stride + 1:stride, c0:c0 + (Wo - 1) * stride + 1:stride, :].sum(-1)
    return np.repeat(centre[..., None], feature_maps_out, axis=-1)


# ---------------------------------------------------------------------------------------------------
# flags -> architecture -> neutral model spec -> layers  (the job of conv_gp/models.py:35-247)
# ---------------------------------------------------------------------------------------------------
# Checkpoint keys are gpflow path names, "DGP/layers/<i>/<suffix>" (conv_gp/experiment.py:56-64, notebooks/Inspect.ipynb cell 6).
# suffix -> field of the per-layer record the spec is filled from; the first matching row wins.
CHECKPOINT_FIELDS = (
    ("feature/Z", "Z"),
    ("q_mu", "q_mu"),
    ("q_sqrt", "q_sqrt"),
    ("base_kernel/variance", "variance"),          # conv_kernel/base_kernel/... (conv layers), kern/base_kernel/... (patch heads)
    ("base_kernel/lengthscales", "ls"),
    ("kern/patch_weights", "w"),
    ("kern/variance", "variance"),                 # dense RBF head (--last-kernel rbf): the kernel sits directly under kern/
    ("kern/lengthscales", "ls"),
)


def read_checkpoint(path, n_layers):
    """``{'global_step': int, layer index: {field: array}}`` from a ``np.save``d ``{pathname: value}`` dict.  A checkpoint with
    fewer layers than the model being built keeps its conv layers in place and hands its LAST stored layer (the head it was
    trained with) to the model's last layer (conv_gp/models.py:231-238)."""
    raw = np.load(path, allow_pickle=True).item()
    records = {}
    for key, value in raw.items():
        parts = key.split("/")
        if len(parts) < 4 or parts[1] != "layers" or not parts[2].isdigit():
            continue
        suffix = "/".join(parts[3:])
        field = next((f for tail, f in CHECKPOINT_FIELDS if suffix.endswith(tail)), None)
        if field is not None:
            records.setdefault(int(parts[2]), {})[field] = np.asarray(value)
    stored = max(records) + 1 if records else 0
    if stored > n_layers:
        raise AssertionError("Can't load model if it has more layers than the one being built")
    if records and stored != n_layers:
        records[n_layers - 1] = records.pop(stored - 1)
    return int(raw.get("global_step", 0)), records


def draw_patches(NHWC_X, count, f, rng=np.random):
    """``count`` f x f patches, each cut at a random position of a random image: [count, f*f*C] in the (kh, kw, c) element order of
    FullView -- the sample PatchInducingFeatures.from_images clusters (conv_gp/kernels.py:139-164), drawn in one gather."""
    n, H, W, C = NHWC_X.shape
    img = rng.randint(0, n, size=count)
    top, left = rng.randint(0, H - f, size=count), rng.randint(0, W - f, size=count)    # upper bound exclusive, as the reference draws them
    dy, dx = np.arange(f)[None, :, None], np.arange(f)[None, None, :]
    return NHWC_X[img[:, None, None], top[:, None, None] + dy, left[:, None, None] + dx, :].reshape(count, f * f * C)


class ModelBuilder(object):
    """``ModelBuilder(flags, NHWC_X_train, Y_train, model_path).build() -> DGP_Base`` (the interface of conv_gp/models.py:35-70).
    The flags are turned into a stage list, the stage list -- walking the initialisation images through the identity convolution,
    clustering patches for the inducing inputs, filling in what a checkpoint holds -- into the neutral model spec of
    ``deepcgp_amd.synthetic``, and the spec into layers by ``build_layers_from_spec``."""

    def __init__(self, flags, NHWC_X_train, Y_train, model_path=None):
        self.flags = flags
        self.X_train = NHWC_X_train
        self.Y_train = Y_train
        self.model_path = model_path
        self.global_step = None

    # ---- flags -> stages ---------------------------------------------------------------------------
    def stages(self):
        """[(M, filter, stride, feature maps)] per conv layer and (M, filter, stride) of the head.  The comma lists follow
        conv_gp/arguments.py:27-31: one M / filter size / stride per GP layer (head included), one feature-map count per conv layer."""
        fl = self.flags
        M, fmaps = parse_ints(fl.M), parse_ints(fl.feature_maps)
        filt, strd = parse_ints(fl.filter_sizes), parse_ints(fl.strides)
        assert len(strd) == len(filt)
        assert len(fmaps) == len(M) - 1
        convs = [(M[i], filt[i], strd[i], fmaps[i]) for i in range(len(fmaps))]
        return convs, (M[-1], filt[-1], strd[-1])

    # ---- stages -> spec ----------------------------------------------------------------------------
    def spec(self):
        fl = self.flags
        convs, (head_M, head_f, head_s) = self.stages()
        n_layers = len(convs) + 1
        stored = {}
        if getattr(fl, "load_model", None) is not None:
            self.global_step, stored = read_checkpoint(self.model_path, n_layers)
        if fl.base_kernel not in ("rbf", "acos"):
            raise ValueError("Not a valid base-kernel value")
        if fl.last_kernel not in ("conv", "add", "rbf"):
            raise ValueError("Invalid last layer kernel")
        white = bool(fl.white)
        spec = {"S": int(fl.num_samples), "num_data": int(self.X_train.shape[0]), "convs": []}
        images = self.X_train                              # what the next layer is initialised on
        for li, (M, f, s, R) in enumerate(convs):
            have = stored.get(li, {})
            _, H, W, C = images.shape
            Z = have["Z"] if "Z" in have else PatchInducingFeatures.from_images(images, M, f).Z
            spec["convs"].append(dict(
                H=H, W=W, C=C, f=f, s=s, M=M, R=R, Z=Z, Z0=Z, white=white, base=fl.base_kernel,
                variance=float(have.get("variance", 5.0)), ls=float(have.get("ls", 5.0)),     # models.py:114-117
                q_mu=have.get("q_mu"), q_sqrt=have.get("q_sqrt"),
                q_sqrt_scale=None if "q_sqrt" in have else 1e-5,                               # start with low variance (models.py:136-138)
                mean_function="conv2d" if getattr(fl, "identity_mean", False) else None))
            images = identity_conv(images, f, C, R, s)                                          # models.py:29-33,104
        have = stored.get(n_layers - 1, {})
        _, H, W, C = images.shape
        if "Z" in have and fl.last_kernel != "rbf":
            stored_f = int(round(np.sqrt(have["Z"].shape[1] / C)))
            if stored_f != head_f:        # a head trained with another filter size starts afresh (models.py:152-158)
                print("filter_size {} != {} for last layer. Resetting parameters.".format(head_f, stored_f))
                have = {k: v for k, v in have.items() if k not in ("Z", "q_mu", "q_sqrt")}
        head = dict(H=H, W=W, C=C, f=head_f, s=head_s, M=head_M, R=10, white=white, kernel=fl.last_kernel,
                    variance=float(have.get("variance", 5.0)), q_mu=have.get("q_mu"), q_sqrt=have.get("q_sqrt"))
        if fl.last_kernel == "rbf":
            # dense head on the flattened features: one lengthscale per dimension, k-means++ inducing points (models.py:24-27,160-168)
            flat = images.reshape(images.shape[0], -1)
            head["ls_ard"] = np.broadcast_to(np.asarray(have.get("ls", 5.0), np.float64), (flat.shape[1],)).copy()
            head["ls"] = 1.0
            if "Z" in have:
                head["Z"] = have["Z"]
            else:
                from sklearn import cluster
                head["Z"] = cluster.KMeans(n_clusters=head_M, init="k-means++", n_init=1).fit(flat).cluster_centers_
            head["w"] = np.ones(1)
        else:
            head["ls"] = float(have.get("ls", 5.0))
            head["Z"] = have["Z"] if "Z" in have else PatchInducingFeatures.from_images(images, head_M, head_f).Z
            head["w"] = have.get("w")
        spec["head"] = head
        return spec

    # ---- spec -> model -----------------------------------------------------------------------------
    def build(self):
        spec = self.spec()
        layers = build_layers_from_spec(spec)
        for layer, c in zip(layers, spec["convs"]):
            if c["q_sqrt_scale"] is not None:
                layer.q_sqrt = layer.q_sqrt * c["q_sqrt_scale"]
        X = self.X_train.reshape(-1, int(np.prod(self.X_train.shape[1:])))
        return DGP_Base(X, self.Y_train, likelihood=MultiClass(10), num_samples=self.flags.num_samples,
                        layers=layers, minibatch_size=self.flags.batch_size, name='DGP')
