"""CPU oracle for the DeepCGP conv-GP forward / ELBO hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``deepcgp_amd/`` imports it; the product path fails loudly
when the HIP library is missing.

It is a float64 NumPy/SciPy restatement of the reference's algorithm, written
from the cited reference lines (``/root/reference/conv_gp/*.py``) and, for the
two un-vendored dependencies (GPflow 1.2.0 -- ``requirements.txt:2`` -- and the
``doubly_stochastic_dgp`` git submodule -- ``.gitmodules:1-3``, directory empty
in the mount), from the published algorithms of those projects (SURVEY.md
Appendix A).

PARITY UNPINNED: the reference cannot be imported in the build container
(TensorFlow 1.x / GPflow 1.2.0 / doubly_stochastic_dgp are absent, no network)
and its own tests (``tests/test_views.py``, ``tests/test_mean_functions.py``,
``tests/test_conv_kernel.py``) contain shape assertions only -- no golden
numbers.  The oracle is therefore pinned by (a) the shape facts those tests
hold (576 patches for 28/5/1, Kuf is P x M x N, ...), (b) analytic
known-answer tests that any correct implementation must satisfy and (c) a
second, independently written restatement (``oracle/alt.py``) that must agree
to 1e-10.  See ``tests/test_oracle_*.py``.
"""
from . import gpflow_ref, views, conditionals, layers, kernels, dgp  # noqa: F401
