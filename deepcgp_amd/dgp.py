"""DGP_Base -- the model-level counterpart of doubly_stochastic_dgp.dgp.DGP_Base as the reference uses
it (/root/reference/conv_gp/models.py:65-70, conv_gp/utils/tensorboard.py:22-35, conv_gp/utils/log.py:62):
``propagate``, ``predict_y``, ``compute_log_likelihood`` and ``parameters``.  The whole forward ELBO of a
minibatch is ONE call into the HIP library (``dcgp_elbo_forward``); parameters live on the device and
are pushed when changed (``sync_parameters``)."""
import ctypes as C

import numpy as np

from . import device as dev
from .kernels import JITTER
from .layers import ConvLayer, SVGP_Layer


class Parameter:
    """Minimal stand-in for a gpflow Param: ``pathname`` + value access (conv_gp/experiment.py:56-62)."""

    def __init__(self, pathname, getter, setter):
        self.pathname = pathname
        self._get, self._set = getter, setter

    @property
    def value(self):
        return self._get()

    def assign(self, v):
        self._set(v)


class DGP_Base:
    def __init__(self, X, Y, likelihood, layers, minibatch_size=None, num_samples=1, name='DGP', num_data=None):
        self.X = np.ascontiguousarray(X, np.float64)
        self.Y = np.ascontiguousarray(np.reshape(Y, (-1,)), np.int32)
        self.likelihood = likelihood
        self.layers = list(layers)
        self.num_samples = int(num_samples)
        self.minibatch_size = minibatch_size
        self.name = name
        self.num_data = int(num_data if num_data is not None else self.X.shape[0])
        self.global_batch = None      # multi-rank runs: the minibatch size summed over the ranks (see _default_scale)
        self.dedup_layer0 = False     # exact optimisation: layer 0 sees S identical copies of the batch
        self._ctx = None
        self._model = None
        self._batch_rng = np.random.RandomState(0)    # Minibatch(seed=0)
        if not self.layers or not isinstance(self.layers[-1], SVGP_Layer):
            raise ValueError("the last layer must be an SVGP_Layer")
        for l in self.layers[:-1]:
            if not isinstance(l, ConvLayer):
                raise ValueError("hidden layers must be ConvLayer instances")

    def _default_scale(self, n_local):
        """num_data / minibatch size.  With an RCCL communicator on the ctx the data term is summed over the ranks inside
        the device call, so the minibatch is the GLOBAL one: set ``global_batch`` (or pass ``scale``) -- the local shard
        size would make the ELBO too large by the rank count."""
        if getattr(self._ctx, "nranks", 1) > 1:
            if getattr(self, "global_batch", None) is None:
                raise ValueError("an RCCL communicator is attached: pass scale=num_data/global_batch or set model.global_batch")
            return float(self.num_data) / float(self.global_batch)
        return float(self.num_data) / float(n_local)

    # ---- device model -----------------------------------------------------------------------------
    def _ptr(self, a):
        return np.ascontiguousarray(a, np.float64).ctypes.data

    def _check_means(self):
        """Every model-level call: the mean functions the device model was built with must still be the ones on the layers (a
        Conv2dMean whose conv_filter was changed afterwards would run as the centre-pixel mean at the model level and through its
        generic __call__ at the layer level -- a silent disagreement)."""
        for li, l in enumerate(self.layers[:-1]):
            now = (bool(l.identity_mean), getattr(l, "generic_mean", None) is not None)
            if now != self._built_means[li]:
                raise ValueError("layer %d: mean function changed after the device model was built (identity_mean %r -> %r, generic "
                                 "%r -> %r); build a new model" % (li, self._built_means[li][0], now[0], self._built_means[li][1], now[1]))

    def _build(self):
        if self._model is not None:
            self._check_means()
            return
        self._built_means = [(bool(l.identity_mean), getattr(l, "generic_mean", None) is not None) for l in self.layers[:-1]]
        ctx = self._ctx = dev.get_context()
        L = dev.lib()
        h = C.c_void_p()
        ctx._check(L.dcgp_model_create(ctx.handle, self.num_samples, JITTER, C.byref(h)))
        self._model = h.value
        for li, l in enumerate(self.layers[:-1]):
            v = l.view
            if getattr(l, "generic_mean", None) is not None:
                # the one-call ELBO adds Conv2dMean's centre pixel inside the layer launch (the only mean the reference builds,
                # conv_gp/models.py:95-99); an arbitrary callable exists at the layer level only
                raise ValueError("layer %d: the model path takes mean_function None / Zero() / Conv2dMean with its initial filter; "
                                 "got %r" % (li, l.mean_function))
            keep = [np.ascontiguousarray(a, np.float64) for a in (l.feature.Z, l.Z_prior, l.q_mu, l.q_sqrt)]
            ctx._check(L.dcgp_model_add_conv_layer(
                self._model, v.input_size[0], v.input_size[1], l.feature_maps_in, v.filter_size, v.stride,
                l.num_inducing, l.gp_count, int(l.white), int(l.identity_mean), l.base_kernel.variance,
                getattr(l.base_kernel, "lengthscales", 1.0), *[a.ctypes.data for a in keep]))
            desc = np.ascontiguousarray(l.base_kernel._describe(), np.float64)
            if desc[0] != 0.0:   # not the RBF the constructor call describes (ArcCosine, conv_gp/models.py:118-119)
                ctx._check(L.dcgp_model_set_param(self._model, li, b"base_kernel",
                                                  desc.ctypes.data, desc.size))
        h_ = self.layers[-1]
        if hasattr(h_.kern, "base_kernel"):      # ConvKernel / AdditivePatchKernel head (--last-kernel conv | add)
            v = h_.kern.view
            geom = (v.input_size[0], v.input_size[1], v.feature_maps, v.filter_size, v.stride)
            ktype, base, weights = int(h_.kern.kernel_type), h_.kern.base_kernel, h_.kern.patch_weights
        else:                                    # dense RBF head on the flattened features (--last-kernel rbf): one patch = the input
            geom = (1, 1, h_.kern.input_dim, 1, 1)
            ktype, base, weights = 0, h_.kern, np.ones(1)
        keep = [np.ascontiguousarray(a, np.float64) for a in (h_.feature.Z, weights, h_.q_mu, h_.q_sqrt)]
        ctx._check(L.dcgp_model_set_head(
            self._model, *geom, h_.num_inducing, h_.num_outputs, int(h_.white), ktype, base.variance,
            1.0 if getattr(base, "ARD", False) else base.lengthscales, *[a.ctypes.data for a in keep]))
        if getattr(base, "ARD", False):
            ls = np.ascontiguousarray(base.lengthscales, np.float64)
            ctx._check(L.dcgp_model_set_param(self._model, len(self.layers) - 1, b"ard_lengthscales", ls.ctypes.data, ls.size))
        eps = np.array([float(getattr(self.likelihood, "epsilon", 1e-3))])
        ctx._check(L.dcgp_model_set_param(self._model, 0, b"likelihood_epsilon", eps.ctypes.data, 1))

    def sync_parameters(self):
        """Push the current Python-side parameter values to the device copy."""
        self._build()
        L, ctx = dev.lib(), self._ctx

        def push(li, which, val):
            a = np.ascontiguousarray(np.atleast_1d(val), np.float64)
            ctx._check(L.dcgp_model_set_param(self._model, li, which.encode(), a.ctypes.data, a.size))
        for li, l in enumerate(self.layers):
            head = li == len(self.layers) - 1
            kern = (l.kern.base_kernel if hasattr(l.kern, "base_kernel") else l.kern) if head else l.base_kernel
            push(li, "Z", l.feature.Z)
            push(li, "q_mu", l.q_mu)
            push(li, "q_sqrt", l.q_sqrt)
            push(li, "base_kernel", kern._describe())
            if getattr(kern, "ARD", False):
                push(li, "ard_lengthscales", kern.lengthscales)
            if head:
                if hasattr(l.kern, "patch_weights"):
                    push(li, "w", l.kern.patch_weights)
            else:
                push(li, "Z0", l.Z_prior)
        push(0, "likelihood_epsilon", float(getattr(self.likelihood, "epsilon", 1e-3)))

    @property
    def parameters(self):
        """Objects with ``pathname`` + ``value`` in the reference's checkpoint naming
        (DGP/layers/<i>/..., notebooks/Inspect.ipynb cell 6)."""
        out = []
        if hasattr(self.likelihood, "epsilon"):     # RobustMax epsilon under the BroadcastingLikelihood wrapper's doubled path
            out.append(Parameter("%s/likelihood/likelihood/invlink/epsilon" % self.name, lambda: np.array(self.likelihood.epsilon),
                                 lambda v: setattr(self.likelihood, "epsilon", float(v))))
        for i, l in enumerate(self.layers):
            head = i == len(self.layers) - 1
            base = "%s/layers/%d" % (self.name, i)
            dense = head and not hasattr(l.kern, "base_kernel")
            kern = (l.kern if dense else l.kern.base_kernel) if head else l.base_kernel
            kpath = base + (("/kern" if dense else "/kern/base_kernel") if head else "/conv_kernel/base_kernel")
            out.append(Parameter(kpath + "/variance", lambda k=kern: np.array(k.variance), lambda v, k=kern: setattr(k, "variance", float(v))))
            for pname in (("lengthscales",) if hasattr(kern, "lengthscales") else ("weight_variances", "bias_variance")):
                out.append(Parameter(kpath + "/" + pname, lambda k=kern, n=pname: np.array(getattr(k, n)),
                                     lambda v, k=kern, n=pname: setattr(k, n, np.array(v, np.float64) if np.ndim(v) else float(v))))
            out.append(Parameter(base + "/feature/Z", lambda l=l: l.feature.Z, lambda v, l=l: setattr(l.feature, "Z", np.array(v, np.float64))))
            out.append(Parameter(base + "/q_mu", lambda l=l: l.q_mu, lambda v, l=l: setattr(l, "q_mu", np.array(v, np.float64))))
            out.append(Parameter(base + "/q_sqrt", lambda l=l: l.q_sqrt, lambda v, l=l: setattr(l, "q_sqrt", np.array(v, np.float64))))
            if head and not dense:
                out.append(Parameter(base + "/kern/patch_weights", lambda l=l: l.kern.patch_weights,
                                     lambda v, l=l: setattr(l.kern, "patch_weights", np.array(v, np.float64))))
        return out

    # ---- forward ----------------------------------------------------------------------------------
    def _z_table(self, zs, N, S):
        if zs is None:
            return None, []
        ctx, keep = self._ctx, []
        arr = (C.c_void_p * len(self.layers))()
        for i, z in enumerate(zs):
            if z is None:
                arr[i] = None
                continue
            D = self.layers[i].num_outputs
            dz = ctx.to_device(np.reshape(z, (S, N, D)))
            keep.append(dz)
            arr[i] = dz.ptr
        return arr, keep

    def compute_log_likelihood(self, X=None, Y=None, zs=None, seed=0, scale=None, return_parts=False):
        """ELBO of an explicit minibatch: sum_n E_q log p(y_n | f_n) * num_data / batch - sum_l KL_l
        (doubly_stochastic_dgp DGP_Base._build_likelihood; explicit feeds as at
        conv_gp/utils/tensorboard.py:32-35).  Without arguments a minibatch of ``minibatch_size`` is drawn."""
        self._build()
        if X is None:
            idx = self._batch_rng.choice(self.X.shape[0], size=min(self.minibatch_size or self.X.shape[0], self.X.shape[0]), replace=False)
            X, Y = self.X[idx], self.Y[idx]
        ctx, L = self._ctx, dev.lib()
        dX = ctx.as_device(np.reshape(X, (np.shape(X)[0], -1)) if not isinstance(X, dev.DeviceArray) else X)
        dY = ctx.as_device(np.reshape(Y, (-1,)) if not isinstance(Y, dev.DeviceArray) else Y, np.int32)
        N = dX.shape[0]
        if scale is None:
            scale = self._default_scale(N)
        arr, keep = self._z_table(zs, N, self.num_samples)
        out = (C.c_double * 3)()
        info = C.c_int(0)
        rc = L.dcgp_elbo_forward(self._model, dX.ptr, dY.ptr, N, float(scale), arr, int(seed), int(self.dedup_layer0), out, C.byref(info))
        ctx._check(rc, info)
        if return_parts:
            return out[0], out[1], out[2]
        return out[0]

    def enqueue_log_likelihood(self, X, Y, zs=None, seed=0, scale=None):
        """Throughput mode of ``compute_log_likelihood``: queue one ELBO step and return a ticket without waiting for
        the device (``dcgp_elbo_forward_enqueue``); ``collect_log_likelihood(ticket)`` returns its value.  At most 4
        tickets may be outstanding and they are collected in order.  The device buffers of the step are kept alive
        with the ticket."""
        self._build()
        ctx, L = self._ctx, dev.lib()
        dX = ctx.as_device(np.reshape(X, (np.shape(X)[0], -1)) if not isinstance(X, dev.DeviceArray) else X)
        dY = ctx.as_device(np.reshape(Y, (-1,)) if not isinstance(Y, dev.DeviceArray) else Y, np.int32)
        N = dX.shape[0]
        if scale is None:
            scale = self._default_scale(N)
        arr, keep = self._z_table(zs, N, self.num_samples)
        ticket = C.c_uint64(0)
        ctx._check(L.dcgp_elbo_forward_enqueue(self._model, dX.ptr, dY.ptr, N, float(scale), arr, int(seed), int(self.dedup_layer0),
                                               C.byref(ticket)))
        if not hasattr(self, "_inflight"):
            self._inflight = {}
        self._inflight[ticket.value] = (dX, dY, arr, keep)
        return ticket.value

    def collect_log_likelihood(self, ticket, return_parts=False):
        """Wait for an enqueued step and return its ELBO (same value and errors as ``compute_log_likelihood``)."""
        ctx, L = self._ctx, dev.lib()
        out = (C.c_double * 3)()
        info = C.c_int(0)
        rc = L.dcgp_elbo_forward_collect(self._model, int(ticket), out, C.byref(info))
        if rc != dev.ERR_ARG:
            getattr(self, "_inflight", {}).pop(int(ticket), None)
        ctx._check(rc, info)
        if return_parts:
            return out[0], out[1], out[2]
        return out[0]

    def compute_gradients(self, X, Y, zs=None, seed=0, scale=None, fetch=True, shards=None):
        """(ELBO, [per-layer dict]) -- the value and gradient TensorFlow hands the optimiser at
        conv_gp/experiment.py:84-108, from the hand-written reverse pass (csrc/grad.hip).  Keys: ``Z``,
        ``q_mu``, ``q_sqrt`` (lower triangle), ``variance``, ``lengthscales`` and, for the head,
        ``patch_weights``; all with respect to the constrained values."""
        self._build()
        ctx, L = self._ctx, dev.lib()
        dX = ctx.as_device(np.reshape(X, (np.shape(X)[0], -1)) if not isinstance(X, dev.DeviceArray) else X)
        dY = ctx.as_device(np.reshape(Y, (-1,)) if not isinstance(Y, dev.DeviceArray) else Y, np.int32)
        N = dX.shape[0]
        if scale is None:
            scale = self._default_scale(N)
        arr, keep = self._z_table(zs, N, self.num_samples)
        out = (C.c_double * 3)()
        info = C.c_int(0)
        # this call handles one of `shards` batch shards: the replicated KL term is weighted 1 / shards; None = the rank
        # count of the ctx's communicator (1 without one).  Always passed, so that it never sticks from an earlier call.
        ctx._check(L.dcgp_model_set_grad_shards(self._model, int(shards or 0)))
        ctx._check(L.dcgp_elbo_grad(self._model, dX.ptr, dY.ptr, N, float(scale), arr, int(seed), int(self.dedup_layer0), out,
                                    C.byref(info)), info)
        if not fetch:            # the gradients stay on the device (dcgp_model_get_grad / the optimiser step read them there)
            return out[0], None
        grads = []
        for li, l in enumerate(self.layers):
            head = li == len(self.layers) - 1
            M, R = l.num_inducing, (l.num_outputs if head else l.gp_count)
            shapes = {"Z": (M, np.shape(l.feature.Z)[1]), "q_mu": (M, R), "q_sqrt": (R, M, M), "variance": (), "lengthscale": ()}
            if head and hasattr(l.kern, "patch_weights"):
                shapes["w"] = (np.size(l.kern.patch_weights),)
            if head and getattr(l.kern, "ARD", False):       # dense RBF(ARD) head: one lengthscale per input dimension
                del shapes["lengthscale"]
                shapes["ard_lengthscales"] = (np.size(l.kern.lengthscales),)
            if not head and not hasattr(l.base_kernel, "lengthscales"):   # ArcCosine(order 0) base kernel
                del shapes["lengthscale"]
                shapes["weight_variances"], shapes["bias_variance"] = (), ()
            g = {}
            for which, shp in shapes.items():
                buf = np.empty(shp, np.float64)
                ctx._check(L.dcgp_model_get_grad(self._model, li, which.encode(), buf.ctypes.data, buf.size))
                g[{"lengthscale": "lengthscales", "ard_lengthscales": "lengthscales", "w": "patch_weights"}.get(which, which)] = buf
            grads.append(g)
        return out[0], grads

    def adam_step(self, lr, t=None, beta1=0.9, beta2=0.999, epsilon=1e-8):
        """One Adam step (tf.train.AdamOptimizer defaults, gpflow.train.AdamOptimizer at
        conv_gp/experiment.py:104-107) on the gradients the last ``compute_gradients`` left on the device,
        in gpflow's unconstrained space.  ``t`` is the 1-based step count of the bias correction; None (default) = the
        device model's own count of steps since its moment buffers were created (independent of any global_step a
        checkpoint carried, like a freshly built tf optimiser).  The device copy of the parameters
        moves; ``pull_parameters`` refreshes the Python-side values."""
        self._ctx._check(dev.lib().dcgp_model_adam_step(self._model, float(lr), float(beta1), float(beta2), float(epsilon), int(t or 0)))

    def train_step(self, X, Y, lr, zs=None, seed=0, scale=None, t=None, beta1=0.9, beta2=0.999, epsilon=1e-8, shards=None):
        """One training step in one call -- ``compute_gradients`` and ``adam_step`` enqueued back to back with a single wait
        at the end (dcgp_model_train_step_adam): the optimiser's ``minimize`` step of conv_gp/experiment.py:84-108.  Returns the
        step's ELBO (evaluated before the update, as TensorFlow's fetch of the objective beside the train op would be).  A step whose
        K_uu is not positive definite raises and leaves every parameter as it was."""
        self._build()
        ctx, L = self._ctx, dev.lib()
        dX = ctx.as_device(np.reshape(X, (np.shape(X)[0], -1)) if not isinstance(X, dev.DeviceArray) else X)
        dY = ctx.as_device(np.reshape(Y, (-1,)) if not isinstance(Y, dev.DeviceArray) else Y, np.int32)
        N = dX.shape[0]
        if scale is None:
            scale = self._default_scale(N)
        arr, keep = self._z_table(zs, N, self.num_samples)
        out = (C.c_double * 3)()
        info = C.c_int(0)
        ctx._check(L.dcgp_model_set_grad_shards(self._model, int(shards or 0)))
        ctx._check(L.dcgp_model_train_step_adam(self._model, dX.ptr, dY.ptr, N, float(scale), arr, int(seed), int(self.dedup_layer0),
                                                float(lr), float(beta1), float(beta2), float(epsilon), int(t or 0), out, C.byref(info)), info)
        return out[0]

    def set_grad_exchange(self, mode):
        """Multi-rank ``train_step``: 0 = all-reduce of the gradient blocks, every rank updates everything; 1 = reduce-scatter, Adam on this
        rank's shard of every layer's parameter block, all-gather of the parameters (dcgp_model_set_grad_exchange; dist.sharded_adam_step is
        the same three steps on host arrays)."""
        self._build()
        self._ctx._check(dev.lib().dcgp_model_set_grad_exchange(self._model, int(mode)))

    def set_factor_reuse(self, mode):
        """Parameter-only state across steps (dcgp_model_set_factor_reuse): 0 = every step runs the factorisation chain; 1 (default) = ``propagate`` /
        ``predict_y`` skip it while no parameter was pushed or stepped since the chain last ran (the reference's AccuracyLogger sweeps a test set at
        one parameter state, conv_gp/utils/log.py:55-68); 2 = ``compute_log_likelihood`` as well (LogLikelihoodLogger-style sweeps).  Bit-identical
        results; a training step never reuses it."""
        self._build()
        self._ctx._check(dev.lib().dcgp_model_set_factor_reuse(self._model, int(mode)))

    @property
    def chain_skips(self):
        """Steps of this model that reused the parameter-only chain of an earlier step (``set_factor_reuse``)."""
        self._build()
        out = C.c_uint64(0)
        self._ctx._check(dev.lib().dcgp_model_chain_skips(self._model, C.byref(out)))
        return int(out.value)

    def debug_sharded_adam(self, ranks, lr, t=None, beta1=0.9, beta2=0.999, epsilon=1e-8):
        """Debugging aid: ``adam_step`` taken the way ``ranks`` ranks take it in exchange mode 1, played on this one GPU (bit-identical).  Needs the
        complete gradient of a ``compute_gradients`` call."""
        self._build()
        self._ctx._check(dev.lib().dcgp_model_debug_sharded_adam(self._model, int(ranks), float(lr), float(beta1), float(beta2), float(epsilon), int(t or 0)))

    def sgd_step(self, lr):
        """Plain gradient ascent step in the unconstrained space (the "SGD" branch, conv_gp/experiment.py:100-103)."""
        self._ctx._check(dev.lib().dcgp_model_sgd_step(self._model, float(lr)))

    def set_shard(self, first_image, global_batch):
        """Multi-GPU: this rank holds images [first_image, first_image + N) of a minibatch of ``global_batch`` images (dist.shard_range).
        Also sets ``global_batch`` for the default ELBO scale.  The device RNG then draws every element at its position in the un-sharded
        batch: a step's ELBO is the same whatever the number of ranks."""
        self._build()
        dev.get_context()._check(dev.lib().dcgp_model_set_shard(self._model, int(first_image), int(global_batch)))
        self.global_batch = int(global_batch) if global_batch else None

    def set_trainable(self, layer, which, on):
        """param.set_trainable(on) for the device optimiser steps: which in Z, q_mu, q_sqrt, w, hyper."""
        self._build()
        self._ctx._check(dev.lib().dcgp_model_set_trainable(self._model, int(layer), which.encode(), int(bool(on))))

    def natgrad_step(self, gamma):
        """One natural-gradient step of size ``gamma`` on every layer's (q_mu, q_sqrt) -- gpflow.train.NatGradOptimizer
        on var_list=[(l.q_mu, l.q_sqrt)] as set up at conv_gp/experiment.py:90-99 -- from the gradients the last
        ``compute_gradients`` left on the device (dcgp_model_natgrad_step, csrc/natgrad.hip: batched Cholesky chains and
        GEMMs, nothing crosses the bus).  Raises ``numpy.linalg.LinAlgError`` and leaves the parameters untouched when
        the step leaves the positive-definite cone (the reference's loop then scales gamma back, experiment.py:36-49)."""
        info = C.c_int(0)
        rc = dev.lib().dcgp_model_natgrad_step(self._model, float(gamma), C.byref(info))
        if rc == dev.ERR_NOT_PD:
            raise np.linalg.LinAlgError("natural-gradient step not positive definite (column %d); reduce gamma" % info.value)
        self._ctx._check(rc, info)

    def pull_parameters(self):
        """Read the device copy of every trainable value back into the layer / kernel objects."""
        self._build()
        L, ctx = dev.lib(), self._ctx

        def pull(li, which, shape):
            buf = np.empty(shape, np.float64)
            ctx._check(L.dcgp_model_get_param(self._model, li, which.encode(), buf.ctypes.data, buf.size))
            return buf
        for li, l in enumerate(self.layers):
            head = li == len(self.layers) - 1
            kern = (l.kern.base_kernel if hasattr(l.kern, "base_kernel") else l.kern) if head else l.base_kernel
            l.feature.Z = pull(li, "Z", np.shape(l.feature.Z))
            l.q_mu = pull(li, "q_mu", np.shape(l.q_mu))
            l.q_sqrt = pull(li, "q_sqrt", np.shape(l.q_sqrt))
            kern.variance = float(pull(li, "variance", ()))
            if getattr(kern, "ARD", False):
                kern.lengthscales = pull(li, "ard_lengthscales", (np.size(kern.lengthscales),))
            elif hasattr(kern, "lengthscales"):
                kern.lengthscales = float(pull(li, "lengthscale", ()))
            else:                                             # ArcCosine(order 0)
                kern.weight_variances = float(pull(li, "weight_variances", ()))
                kern.bias_variance = float(pull(li, "bias_variance", ()))
            if head and hasattr(l.kern, "patch_weights"):
                l.kern.patch_weights = pull(li, "w", np.shape(l.kern.patch_weights))

    def propagate(self, X, full_cov=False, S=1, zs=None, seed=0):
        """(Fs, Fmeans, Fvars): per layer S x N x D_l arrays (doubly_stochastic_dgp DGP_Base.propagate)."""
        if full_cov:
            raise NotImplementedError("full_cov=True is outside the accelerated hot path")
        self._build()
        ctx, L = self._ctx, dev.lib()
        X = np.ascontiguousarray(np.reshape(X, (np.shape(X)[0], -1)), np.float64)
        N = X.shape[0]
        dX = ctx.to_device(X)
        # the device model was created with num_samples; propagate takes S explicitly
        arr, keep = self._z_table(zs, N, S)
        ctx._check(L.dcgp_model_set_keep_outputs(self._model, 1))
        info = C.c_int(0)
        try:
            rc = L.dcgp_model_propagate(self._model, dX.ptr, N, int(S), arr, int(seed), None, None, C.byref(info))
            ctx._check(rc, info)
            Fs, Fm, Fv = [], [], []
            for i, l in enumerate(self.layers):
                D = l.num_outputs
                bufs = [ctx.empty((S, N, D)) for _ in range(3)]
                rows, width = C.c_int(0), C.c_int(0)
                ctx._check(L.dcgp_model_layer_output(self._model, i, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, C.byref(rows), C.byref(width)))
                assert rows.value == S * N and width.value == D, (rows.value, width.value, S, N, D)
                Fs.append(bufs[0].numpy()), Fm.append(bufs[1].numpy()), Fv.append(bufs[2].numpy())
        finally:
            L.dcgp_model_set_keep_outputs(self._model, 0)
        return Fs, Fm, Fv

    def _predict(self, X, S, zs, seed, want_samples, want_mean):
        self._build()
        ctx, L = self._ctx, dev.lib()
        X = np.ascontiguousarray(np.reshape(X, (np.shape(X)[0], -1)), np.float64)
        N, K = X.shape[0], self.layers[-1].num_outputs
        dX = ctx.to_device(X)
        arr, keep = self._z_table(zs, N, S)
        p = ctx.empty((S * N, K)) if want_samples else None
        pm = ctx.empty((N, K)) if want_mean else None
        info = C.c_int(0)
        rc = L.dcgp_model_predict_y(self._model, dX.ptr, N, int(S), arr, int(seed), p.ptr if p else None,
                                    pm.ptr if pm else None, C.byref(info))
        ctx._check(rc, info)
        return (p.numpy().reshape(S, N, K) if p else None), (pm.numpy() if pm else None)

    def predict_y(self, X, S, zs=None, seed=0):
        """(mean, var) of p(y*) per sample: S x N x num_classes (used at conv_gp/utils/log.py:62-66).
        One device call: forward pass and RobustMax quadrature, only the probabilities come back."""
        if np.shape(X)[0] == 0:
            K = self.layers[-1].num_outputs
            return np.zeros((S, 0, K)), np.zeros((S, 0, K))
        ps, _ = self._predict(X, S, zs, seed, True, False)
        return ps, ps - np.square(ps)

    def predict_proba(self, X, S, zs=None, seed=0):
        """Class probabilities averaged over the S samples, N x num_classes (the quantity AccuracyLogger
        arg-maxes, conv_gp/utils/log.py:62-67); the sample mean is taken on the device."""
        if np.shape(X)[0] == 0:
            return np.zeros((0, self.layers[-1].num_outputs))
        return self._predict(X, S, zs, seed, False, True)[1]

    def KL(self):
        return float(sum(l.KL() for l in self.layers))

    def close(self):
        if self._model is not None:
            dev.lib().dcgp_model_destroy(self._model)
            self._model = None
