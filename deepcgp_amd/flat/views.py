"""Flat-import shim: the reference's modules import each other as top-level names (``from views import ...`` at
conv_gp/models.py:8-11, tests/context.py:3-4 puts conv_gp/ on sys.path).  Put ``deepcgp_amd/flat`` on sys.path in its place and
those imports resolve to the MI355X path."""
from deepcgp_amd.views import FullView, RandomPartialView, View  # noqa: F401
