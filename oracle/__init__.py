"""CPU oracle for the rfi_toolbox hot path -- TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement of the reference algorithm
(preshanth/rfi_toolbox v0.2.0) used *only* as the checker by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.
The product (``rfi_toolbox_amd``) never imports it; the product path fails
loudly when the HIP library is missing instead of falling back to this code.

Parity pinning: every function here is checked against golden vectors that
were captured by importing the reference itself in the build container
(``tests/golden/make_golden.py`` is the generating script, the ``.npz`` /
``.json`` files next to it are the vectors).  See ``tests/test_oracle_golden.py``.

Modules
-------
unet_ref        torch-CPU fp32 restatement of ``rfi_toolbox/models/unet.py`` and
                of the optimisation step in ``rfi_toolbox/scripts/train_model.py``.
preprocess_ref  NumPy restatement of ``rfi_toolbox/preprocessing/preprocessor.py``.
metrics_ref     NumPy restatement of ``rfi_toolbox/evaluation/metrics.py``.
"""
