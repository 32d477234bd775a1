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
        layer = ConvLayer(rbf, None, c["Z"], view, white=c["white"], gp_count=c["R"],
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
