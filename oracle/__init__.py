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

Round 4 -- checked against INDEPENDENT third-party code that ships in the image
(``tests/test_oracle_thirdparty.py``; none of it shares code or authorship with
this package).  This does not pin parity with the reference's binaries -- that
needs the reference -- but it does pin the FORMULAS recalled in SURVEY App. A:
  gauss_kl                      torch.distributions.kl_divergence(MultivariateNormal, MultivariateNormal), both whitenings
  RBF.K / Kdiag (scalar, ARD)   sklearn.gaussian_process.kernels.RBF
  ConvKernel.Kzx / Kdiag / Kzz  explicit loops over hand-cut patches with the sklearn base kernel
  ArcCosine(order 0)            Monte-Carlo 2 E_w[step(w.x~) step(w.z~)] (Cho & Saul 2009), 2e6 draws
  RobustMax prob_is_largest,    Monte-Carlo orthant probabilities / E_q[log p(y|f)] / class
    variational_expectations,     probabilities, 1e6 draws
    predict_mean_and_var
  conditional(), SVGP_Layer     torch.linalg dense closed form k - k^T K^-1 k + k^T K^-1 S K^-1 k, both whitenings, full_cov=True (per-patch N x N
                                covariance) included; KL against K_uu
  reparameterize, ELBO assembly torch algebra; sum_n mean_s E * num_data / N - sum_l KL_l from those pieces
  NatGrad step (natgrad_ref)    theta <- theta + gamma dL/d eta with dL/d eta by torch autograd through eta -> (mu, S) -> chol(S)
  the hand-written reverse pass   torch autograd (CPU, float64) of an independently written textbook forward -- unfold patches,
    (oracle/grad.py), and the       cholesky_solve, Gaussian closed-form KL, RobustMax quadrature -- ELBO to 1e-10, every gradient entry of
    whole forward value             every layer to 1e-9: conv / additive / dense RBF(ARD) heads, Conv2dMean, three layers with a stride-2 first one, both
    whitenings; the ArcCosine(order 0) conv layers' ELBO value as well (``tests/test_oracle_autograd.py``; finite differences reach 1e-4).  The DEVICE value and gradients are compared with
    the same autograd directly in ``tests/test_gpu_model.py::test_device_gradient_matches_torch_autograd`` /
    ``test_device_elbo_matches_torch_forward_mnist_geometry``, and at the FULL size of BASELINE configs[0..3] in
    ``test_full_size_cfg1_vs_torch_forward`` / ``test_full_size_baseline_configs_vs_torch_forward`` (1e-9), configs[4] (M = 1024) on a
    reduced batch in ``test_cfg5_reduced_batch_vs_torch_forward``; the gradient of the
    headline configuration at full size against autograd in ``test_full_size_cfg2_gradient_vs_torch_autograd`` (1e-7)
and the stack as a whole learns real images (sklearn load_digits: 0.97 / 0.99 test
accuracy after 500 Adam steps; ``tests/test_gpu_model.py::test_learns_real_digits``).
"""
from . import gpflow_ref, views, conditionals, layers, kernels, dgp  # noqa: F401
