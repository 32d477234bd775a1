"""Model assembly -- the counterpart of /root/reference/conv_gp/models.py (ModelBuilder) plus a builder
from the neutral model spec used by the benchmarks and parity tests (deepcgp_amd.synthetic)."""
import numpy as np

from .dgp import DGP_Base
from .kernels import RBF, ArcCosine, ConvKernel, AdditivePatchKernel, PatchInducingFeatures, InducingPoints
from .layers import ConvLayer, SVGP_Layer
from .likelihoods import MultiClass
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
        layer = ConvLayer(base, c.get("mean_function"),
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
    kern = cls(RBF(view.patch_length, h["variance"], h["ls"]), view, patch_weights=h["w"])
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
      "Adam":    one device Adam step on every parameter;
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


class ModelBuilder(object):
    """Same flags and construction order as conv_gp/models.py:35-198 (checkpoint loading: :200-240)."""

    def __init__(self, flags, NHWC_X_train, Y_train, model_path=None):
        self.flags = flags
        self.X_train = NHWC_X_train
        self.Y_train = Y_train
        self.model_path = model_path
        self.global_step = None

    def build(self):
        Ms = parse_ints(self.flags.M)
        feature_maps = parse_ints(self.flags.feature_maps)
        strides = parse_ints(self.flags.strides)
        filter_sizes = parse_ints(self.flags.filter_sizes)
        loaded = {}
        if getattr(self.flags, "load_model", None) is not None:
            self.global_step, loaded = self._load_layer_parameters(Ms)
        assert len(strides) == len(filter_sizes)
        assert len(feature_maps) == (len(Ms) - 1)
        conv_layers, H_X = self._conv_layers(Ms[0:-1], feature_maps, strides, filter_sizes, loaded)
        last = self._last_layer(H_X, Ms[-1], filter_sizes[-1], strides[-1], self._last_layer_parameters(loaded))
        X = self.X_train.reshape(-1, int(np.prod(self.X_train.shape[1:])))
        return DGP_Base(X, self.Y_train, likelihood=MultiClass(10), num_samples=self.flags.num_samples,
                        layers=conv_layers + [last], minibatch_size=self.flags.batch_size, name='DGP')

    def _conv_layers(self, Ms, feature_maps, strides, filter_sizes, loaded):
        H_X, layers = self.X_train, []
        for i in range(len(feature_maps)):
            layer, H_X = self._conv_layer(H_X, Ms[i], feature_maps[i], filter_sizes[i], strides[i], loaded.get(i))
            layers.append(layer)
        return layers, H_X

    def _conv_layer(self, NHWC_X, M, feature_map, filter_size, stride, layer_params=None):
        layer_params = layer_params or {}
        NHWC = NHWC_X.shape
        view = FullView(input_size=NHWC[1:3], filter_size=filter_size, feature_maps=NHWC[3], stride=stride)
        conv_mean = 'conv2d' if getattr(self.flags, "identity_mean", False) else None
        H_X = identity_conv(NHWC_X, filter_size, NHWC[3], feature_map, stride)
        if len(layer_params) == 0:
            conv_features = PatchInducingFeatures.from_images(NHWC_X, M, filter_size)
        else:
            conv_features = PatchInducingFeatures(layer_params.get('Z'))
        patch_length = filter_size ** 2 * NHWC[3]
        if self.flags.base_kernel == 'rbf':
            base_kernel = RBF(patch_length, variance=float(layer_params.get('base_kernel/variance', 5.0)),
                              lengthscales=float(layer_params.get('base_kernel/lengthscales', 5.0)))
        elif self.flags.base_kernel == 'acos':
            base_kernel = ArcCosine(patch_length, order=0)   # gpflow defaults: variance = weight = bias = 1 (models.py:119)
        else:
            raise ValueError("Not a valid base-kernel value")
        q_mu, q_sqrt = layer_params.get('q_mu'), layer_params.get('q_sqrt')
        conv_layer = ConvLayer(base_kernel=base_kernel, mean_function=conv_mean, feature=conv_features, view=view,
                               white=self.flags.white, gp_count=feature_map, q_mu=q_mu, q_sqrt=q_sqrt)
        if q_sqrt is None:
            conv_layer.q_sqrt = conv_layer.q_sqrt * 1e-5      # start with low variance (models.py:136-138)
        return conv_layer, H_X

    def _last_layer(self, H_X, M, filter_size, stride, layer_params=None):
        layer_params = layer_params or {}
        NHWC = H_X.shape
        Z, q_mu, q_sqrt = layer_params.get('Z'), layer_params.get('q_mu'), layer_params.get('q_sqrt')
        if Z is not None:
            saved = int(np.sqrt(Z.shape[1] / NHWC[3]))
            if filter_size != saved:
                print("filter_size {} != {} for last layer. Resetting parameters.".format(filter_size, saved))
                Z = q_mu = q_sqrt = None
        if self.flags.last_kernel == 'rbf':
            # dense head on the flattened features: RBF with one lengthscale per dimension, k-means inducing points
            # (conv_gp/models.py:160-168, select_initial_inducing_points :24-27)
            flat = H_X.reshape(H_X.shape[0], -1)
            kernel = RBF(flat.shape[1], variance=float(layer_params.get('variance', 5.0)),
                         lengthscales=layer_params.get('lengthscales', 5.0), ARD=True)
            Z = layer_params.get('Z')
            if Z is None:
                from sklearn import cluster
                Z = cluster.KMeans(n_clusters=M, init='k-means++', n_init=1).fit(flat).cluster_centers_
            return SVGP_Layer(kern=kernel, num_outputs=10, feature=InducingPoints(Z), mean_function=None,
                              white=self.flags.white, q_mu=q_mu, q_sqrt=q_sqrt)
        variance = float(layer_params.get('base_kernel/variance', 5.0))
        lengthscales = float(layer_params.get('base_kernel/lengthscales', 5.0))
        input_dim = filter_size ** 2 * NHWC[3]
        view = FullView(input_size=NHWC[1:], filter_size=filter_size, feature_maps=NHWC[3], stride=stride)
        inducing = PatchInducingFeatures.from_images(H_X, M, filter_size) if Z is None else PatchInducingFeatures(Z)
        patch_weights = layer_params.get('patch_weights')
        if self.flags.last_kernel == 'conv':
            kernel = ConvKernel(RBF(input_dim, variance=variance, lengthscales=lengthscales), view, patch_weights)
        elif self.flags.last_kernel == 'add':
            kernel = AdditivePatchKernel(RBF(input_dim, variance=variance, lengthscales=lengthscales), view, patch_weights)
        else:
            raise ValueError("Invalid last layer kernel")
        return SVGP_Layer(kern=kernel, num_outputs=10, feature=inducing, mean_function=None,
                          white=self.flags.white, q_mu=q_mu, q_sqrt=q_sqrt)

    def _load_layer_parameters(self, Ms):
        parameters = np.load(self.model_path, allow_pickle=True).item()
        global_step = parameters.pop('global_step')
        layer_params = {}
        for key, value in parameters.items():
            if 'layers' not in key:
                continue
            parts = key.split('/')
            layer, path = int(parts[2]), "/".join(parts[3:])
            vals = layer_params.setdefault(layer, {})
            for tag in ('q_mu', 'q_sqrt', 'Z', 'base_kernel/variance', 'base_kernel/lengthscales', 'patch_weights'):
                if tag in path:
                    vals[tag] = value
                    break
        stored, model_layers = max(layer_params.keys()) + 1, len(Ms)
        assert stored <= model_layers, "Can't load model if it has more layers than the one being built"
        if stored != model_layers:
            layer_params[model_layers - 1] = layer_params.pop(stored - 1)
        return global_step, layer_params

    def _last_layer_parameters(self, layer_params):
        keys = list(layer_params.keys())
        return layer_params[max(keys)] if keys else None
