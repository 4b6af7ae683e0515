"""CPU oracle for the CT-WGAN adversarial-step hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (PyTorch-CPU, fp64 = truth, fp32 = tolerance twin) of the
arithmetic that biuyq/CT-GAN delegates to TensorFlow 1.2.1 for the path
``CT_gan_{mnist,cifar,cifar_resnet}.py``.  Every function cites the
reference file:line it follows (TF/ = CT-GANs/tensorflow_generative_model/).

PARITY UNPINNED.  The reference is Python-2 / TensorFlow-1.2.1 source with no tests, no golden
vectors and no seeds; neither TF nor a Python-2 interpreter exists in the build container, so the
reference can be neither imported nor compiled.  The oracle is therefore
pinned only by
  * the known-answer anchors the reference *does* contain (shape closure of the hard-coded
    flatten sizes, the parameter-count printout, init statistics, algebraic identities:
    tests/test_oracle_anchors.py), and
  * an independent second restatement of the TF conv / conv-transpose SAME semantics written as
    plain numpy tap loops (oracle/np_conv.py) that must agree with the torch formulation.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything from here; the product package ``ctgan_amd`` never does (tests/test_no_oracle_in_product.py
enforces it).
"""
