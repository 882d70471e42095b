"""CPU oracle for the CARMA Kalman-filter log-likelihood path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  ``carma_pack_amd`` (the product) never does.
"""
from . import post  # noqa: F401  (CarmaSample post-processing restated in numpy)
from .oracle import (  # noqa: F401
    NativeComparator,
    OracleModel,
    ar_roots,
    build,
    chol_update_r1,
    kfilter_car1,
    kfilter_carma,
    lib,
    ma_coefs,
    max_threads,
    predict_car1,
    predict_carma,
    sort_dedup,
    truth_filter,
    truth_logdensity,
    truth_variance,
    variance,
)
