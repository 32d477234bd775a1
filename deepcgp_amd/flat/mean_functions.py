"""Flat-import shim: ``from mean_functions import Conv2dMean, IdentityConv2dMean`` (conv_gp/models.py:11) resolves to the MI355X
path when ``deepcgp_amd/flat`` stands where ``conv_gp/`` stood on sys.path."""
from deepcgp_amd.mean_functions import Conv2dMean, IdentityConv2dMean, MeanFunction, Zero  # noqa: F401
