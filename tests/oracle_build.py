"""Test helper: build the oracle's DGP_Base from a neutral model spec (deepcgp_amd.synthetic)."""
import numpy as np
import oracle
from oracle.gpflow_ref import RBF, ArcCosine, MultiClass
from oracle.views import FullView
from oracle.layers import ConvLayer
from oracle.kernels import ConvKernel
from oracle.dgp import SVGP_Layer, DGP_Base


def oracle_layers(spec):
    layers = []
    for c in spec["convs"]:
        view = FullView((c["H"], c["W"]), c["f"], c["C"], c["s"])
        rbf = ArcCosine(view.patch_length, order=0) if c.get("base", "rbf") == "acos" else RBF(view.patch_length, c["variance"], c["ls"])
        mean = None
        if c.get("mean_function") == "conv2d":   # Conv2dMean, conv_gp/models.py:97-100
            from oracle.mean_functions import Conv2dMean
            mean = Conv2dMean(c["f"], c["C"], c["R"], c["s"])
        layer = ConvLayer(rbf, mean, c["Z"], view, white=c["white"], gp_count=c["R"],
                          q_mu=c["q_mu"], q_sqrt=c["q_sqrt"])
        layer.Z0 = np.array(c["Z0"], np.float64)
        layers.append(layer)
    h = spec["head"]
    if h.get("kernel", "conv") == "rbf":   # dense RBF-ARD head: gpflow InducingPoints + RBF(ARD=True)
        layers.append(SVGP_Layer(RBF(h["Z"].shape[1], h["variance"], h["ls_ard"], ARD=True), h["R"], h["Z"], None,
                                 white=h["white"], q_mu=h["q_mu"], q_sqrt=h["q_sqrt"]))
        return layers
    view = FullView((h["H"], h["W"], h["C"]), h["f"], h["C"], h["s"])
    kern = ConvKernel(RBF(view.patch_length, h["variance"], h["ls"]), view, patch_weights=h["w"])
    layers.append(SVGP_Layer(kern, h["R"], h["Z"], None, white=h["white"], q_mu=h["q_mu"], q_sqrt=h["q_sqrt"]))
    return layers


def oracle_model(spec, X, Y):
    return DGP_Base(X, Y, MultiClass(10), oracle_layers(spec), num_samples=spec["S"], num_data=spec["num_data"])


def oracle_param_handles(model):
    """[(layer index, gradient name, getter, setter)] of every trainable value of an oracle model with RBF base
    kernels -- the names oracle.grad.elbo_and_grad and dcgp_model_get_grad use."""
    out, nl = [], len(model.layers)
    for li, l in enumerate(model.layers):
        head = li == nl - 1
        kern = (l.kern.base_kernel if hasattr(l.kern, "base_kernel") else l.kern) if head else l.base_kernel
        out.append((li, "Z", lambda l=l: l.Z, lambda v, l=l: setattr(l, "Z", v)))
        out.append((li, "q_mu", lambda l=l: l.q_mu, lambda v, l=l: setattr(l, "q_mu", v)))
        out.append((li, "q_sqrt", lambda l=l: l.q_sqrt, lambda v, l=l: setattr(l, "q_sqrt", v)))
        out.append((li, "variance", lambda k=kern: np.array(k.variance), lambda v, k=kern: setattr(k, "variance", float(v))))
        if hasattr(kern, "lengthscales"):
            out.append((li, "lengthscales", lambda k=kern: np.array(k.lengthscales),
                        lambda v, k=kern: setattr(k, "lengthscales", np.array(v, np.float64) if np.ndim(v) else float(v))))
        else:                                         # ArcCosine(order 0)
            for pname in ("weight_variances", "bias_variance"):
                out.append((li, pname, lambda k=kern, n=pname: np.array(getattr(k, n)), lambda v, k=kern, n=pname: setattr(k, n, float(v))))
        if head and hasattr(l.kern, "patch_weights"):
            out.append((li, "patch_weights", lambda l=l: l.kern.patch_weights, lambda v, l=l: setattr(l.kern, "patch_weights", v)))
    return out
