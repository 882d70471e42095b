"""ctypes binding of libcarma_mi355.so (the C ABI declared in include/carma_mi355.h).

This is the ONLY route from Python to the compute path.  There is no CPU fallback: if the
shared library is missing the import fails, and without a gfx950 device every compute call
raises ``CarmaDeviceError``.
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CARMA_LIB_PATH: another build of the same C ABI (A/B measurements of kernel variants); default: the in-tree build
LIB_PATH = os.environ.get("CARMA_LIB_PATH") or os.path.join(_HERE, "libcarma_mi355.so")

CARMA_OK, CARMA_EINVAL, CARMA_ENODEV, CARMA_ENOMEM, CARMA_EHIP = 0, -22, -19, -12, -5
PMAX = 7

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class CarmaError(RuntimeError):
    """Boost.Python turned C++ exceptions into RuntimeError; so do we."""


class CarmaDeviceError(CarmaError):
    pass


def _share_hip_runtime_with_torch():
    """One process, ONE HIP runtime.  The PyTorch wheel bundles its own libamdhip64.so (soname libamdhip64.so.7, like
    the system's); whichever copy is mapped first serves every later request for that soname, but `import torch` asks
    for it by file name and would map a SECOND runtime if this library had already pulled in /opt/rocm's.  Two runtimes
    in one process cannot share streams or device pointers (the `_dev` entry points take torch's) and cooperative
    launches fail outright (measured: hipErrorUnknown).  So when PyTorch is installed and not yet imported, its copy is
    mapped first -- without importing torch.  CARMA_HIP_RUNTIME=system keeps the system runtime."""
    if "torch" in sys.modules or os.environ.get("CARMA_HIP_RUNTIME") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def _share_rccl_with_torch():
    """Same for RCCL (bound by the library at run time as librccl.so.1): map PyTorch's copy first when there is one, so
    that it -- and the HIP runtime it was built against -- is the one carma_comm_* finds."""
    if os.environ.get("CARMA_HIP_RUNTIME") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def build_ids():
    """What is running: {"build_id": sha256 of the shared library that is loaded, "source_id": sha256 over the sources it is
    built from (csrc/, include/, build.sh; None when the sources are not beside the library)}.  Profile records carry
    these so that counters measured on one build are never attributed to another (bench.py, tools/summarize_prof.py)."""
    import hashlib
    out = {"build_id": None, "source_id": None}
    try:
        with open(LIB_PATH, "rb") as f:
            out["build_id"] = hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        pass
    root = os.path.dirname(_HERE)
    files = []
    for sub in ("carma_pack_amd/csrc", "include"):
        dd = os.path.join(root, sub)
        if os.path.isdir(dd):
            files += [os.path.join(dd, f) for f in sorted(os.listdir(dd)) if f.endswith((".h", ".hip", ".hpp", ".cpp"))]
    bs = os.path.join(root, "build.sh")
    if files and os.path.exists(bs):
        h = hashlib.sha256()
        for f in files + [bs]:
            h.update(os.path.relpath(f, root).encode())
            with open(f, "rb") as fh:
                h.update(fh.read())
        out["source_id"] = h.hexdigest()[:16]
    return out


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "carma_pack_amd: %s not found -- build it with ./build.sh (or __graft_entry__.build()); "
            "there is no CPU fallback" % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    L.carma_version.restype = C.c_char_p
    L.carma_last_error.restype = C.c_char_p
    L.carma_device_count.restype = C.c_int
    L.carma_ctx_create.restype = C.c_void_p
    L.carma_ctx_create.argtypes = [_dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
    L.carma_ctx_destroy.argtypes = [C.c_void_p]
    L.carma_ctx_destroy.restype = None
    L.carma_ctx_n.argtypes = [C.c_void_p]
    L.carma_ctx_dim.argtypes = [C.c_void_p]
    L.carma_ctx_get_data.argtypes = [C.c_void_p, _dp, _dp, _dp]
    L.carma_ctx_get_prior.argtypes = [C.c_void_p, _dp]
    L.carma_ctx_set_prior.argtypes = [C.c_void_p, C.c_double]
    L.carma_logdensity_batch.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, _dp]
    L.carma_logdensity_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.carma_logdensity_kernel_name.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    L.carma_tune_set.argtypes = [C.c_char_p, C.c_long]
    L.carma_tune_set.restype = C.c_int
    L.carma_mle_batched.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double,
                                    C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.carma_mle_batched.restype = C.c_int
    L.carma_logprior.argtypes = [C.c_void_p, _dp]
    L.carma_logprior.restype = C.c_double
    L.carma_kfilter_batch_carma.argtypes = [_dp, _dp, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp,
                                            C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.carma_kfilter_batch_carma.restype = C.c_int
    L.carma_kfilter_carma.argtypes = [_dp, _dp, _dp, C.c_int, C.c_int, C.c_double, _dp, _dp, C.c_int, _dp, _dp,
                                      _ip, C.c_int]
    L.carma_kfilter_car1.argtypes = [_dp, _dp, _dp, C.c_int, C.c_double, C.c_double, _dp, _dp, _ip, C.c_int]
    L.carma_predict_carma.argtypes = [_dp, _dp, _dp, C.c_int, C.c_int, C.c_double, _dp, _dp, C.c_int, _dp, C.c_int,
                                      _dp, _dp, C.c_int]
    L.carma_predict_car1.argtypes = [_dp, _dp, _dp, C.c_int, C.c_double, C.c_double, _dp, C.c_int, _dp, _dp, C.c_int]
    L.carma_normalize_roots.argtypes = [C.c_int, _dp, _dp]
    L.carma_kf_create_carma.restype = C.c_void_p
    L.carma_kf_create_carma.argtypes = [_dp, _dp, _dp, C.c_int, C.c_int, C.c_double, _dp, _dp, C.c_int, C.c_int]
    L.carma_kf_create_car1.restype = C.c_void_p
    L.carma_kf_create_car1.argtypes = [_dp, _dp, _dp, C.c_int, C.c_double, C.c_double, C.c_int]
    L.carma_kf_destroy.argtypes = [C.c_void_p]
    L.carma_kf_destroy.restype = None
    L.carma_kf_n.argtypes = [C.c_void_p]
    L.carma_kf_filter.argtypes = [C.c_void_p, _dp, _dp]
    L.carma_kf_predict.argtypes = [C.c_void_p, _dp, C.c_int, _dp, _dp]
    L.carma_simulate_carma.argtypes = [_dp, C.c_int, C.c_int, C.c_double, _dp, _dp, C.c_int, C.c_int, C.c_uint64, _dp, C.c_int]
    L.carma_simulate_car1.argtypes = [_dp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_uint64, _dp, C.c_int]
    L.carma_sigma_noise_batch.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, _dp, C.c_int]
    L.carma_psd_band.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp, _dp, C.c_int]
    L.carma_pt_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_uint64,
                               _dp, _dp]
    L.carma_pt_create.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, C.c_int, C.c_uint64]
    L.carma_pt_shard.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.carma_pt_bind_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.carma_pt_start.argtypes = [C.c_void_p, _dp, C.c_int]
    L.carma_pt_set_chains.argtypes = [C.c_void_p, _dp, _dp]
    L.carma_pt_get_chains.argtypes = [C.c_void_p, _dp, _dp]
    L.carma_pt_iterate.argtypes = [C.c_void_p, C.c_long, C.c_int]
    L.carma_pt_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp]
    L.carma_pt_stats.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
    L.carma_pt_iterations_done.argtypes = [C.c_void_p]
    L.carma_pt_iterations_done.restype = C.c_long
    L.carma_comm_unique_id.argtypes = [C.c_void_p]
    L.carma_comm_create.restype = C.c_void_p
    L.carma_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.carma_comm_destroy.argtypes = [C.c_void_p]
    L.carma_comm_destroy.restype = None
    L.carma_comm_rank.argtypes = [C.c_void_p]
    L.carma_comm_size.argtypes = [C.c_void_p]
    L.carma_pt_iterate_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_long, C.c_void_p]
    L.carma_pt_sample_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p, _dp, _dp]
    L.carma_pt_boundary_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.carma_pt_boundary_check.argtypes = [C.c_void_p]
    L.carma_pt_kernel_in_use.argtypes = [C.c_void_p]
    L.carma_pt_row_pipeline.argtypes = []
    L.carma_pt_sweep.argtypes = [C.c_void_p]
    L.carma_pt_debug_draws.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_ulonglong, _dp, _dp, _dp]
    L.carma_pt_get_factor.argtypes = [C.c_void_p, _dp]
    L.carma_pt_set_factor.argtypes = [C.c_void_p, _dp]
    return L


lib = _load()

# every symbol include/carma_mi355.h declares (kept in sync by tests/test_capi_symbols.py)
EXPORTS = [
    "carma_version", "carma_last_error", "carma_device_count", "carma_ctx_create", "carma_ctx_destroy",
    "carma_ctx_n", "carma_ctx_dim", "carma_ctx_get_data", "carma_ctx_get_prior", "carma_ctx_set_prior",
    "carma_logdensity_batch", "carma_logdensity_batch_dev", "carma_logdensity_kernel_name", "carma_logprior", "carma_mle_batched", "carma_kfilter_carma", "carma_kfilter_batch_carma",
    "carma_kfilter_car1", "carma_predict_carma", "carma_predict_car1", "carma_normalize_roots", "carma_kf_create_carma", "carma_kf_create_car1",
    "carma_kf_destroy", "carma_kf_n", "carma_kf_filter", "carma_kf_predict", "carma_simulate_carma", "carma_simulate_car1", "carma_sigma_noise_batch", "carma_psd_band", "carma_pt_run", "carma_pt_create", "carma_pt_shard", "carma_pt_bind_state",
    "carma_pt_start", "carma_pt_set_chains", "carma_pt_get_chains", "carma_pt_iterate", "carma_pt_sample",
    "carma_pt_stats", "carma_pt_iterations_done", "carma_comm_unique_id", "carma_comm_create", "carma_comm_destroy",
    "carma_comm_rank", "carma_comm_size", "carma_pt_iterate_sharded", "carma_pt_sample_sharded", "carma_pt_boundary_stats",
    "carma_pt_boundary_check", "carma_pt_sweep", "carma_pt_kernel_in_use", "carma_pt_row_pipeline", "carma_pt_debug_draws", "carma_pt_get_factor",
    "carma_pt_set_factor", "carma_tune_set",
]


def tune_set(name, value):
    """Move a launch-shape switch ("WIN_ROWS", "WIN2_EVALS", "PT_ROW_WIN": carma_tune_set; measurements and parity tests).
    value None: back to the library's default."""
    check(lib.carma_tune_set(str(name).encode(), -2 ** 63 if value is None else int(value)), "carma_tune_set")


def tune_reset():
    """Every switch back to what the environment said when the library read it (CARMA_TUNE_<name>), or to the default."""
    for name in ("WIN_ROWS", "WIN2_EVALS", "PT_ROW_WIN"):
        e = os.environ.get("CARMA_TUNE_" + name)
        tune_set(name, None if e is None else int(e))


def last_error():
    return lib.carma_last_error().decode()


def check(rc, what):
    if rc == CARMA_OK:
        return
    msg = "%s failed (%d): %s" % (what, rc, last_error())
    if rc == CARMA_ENODEV:
        raise CarmaDeviceError(msg)
    if rc == CARMA_EINVAL:
        raise ValueError(msg)
    raise CarmaError(msg)


def as_f64(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def ptr(a):
    return a.ctypes.data_as(_dp)


def default_device():
    """One process per GPU: LOCAL_RANK picks the device when launched by torch.distributed.run."""
    n = lib.carma_device_count()
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    return lr % n if n > 0 else 0


class Context:
    """Owns one carma_ctx (series resident in HBM + prior bounds)."""

    def __init__(self, time, y, yerr, p, q=0, max_stdev=None, device=None):
        time, y, yerr = as_f64(time), as_f64(y), as_f64(yerr)
        if not (time.size == y.size == yerr.size):
            raise ValueError("time, y, yerr must have the same length")
        if max_stdev is None:
            # default of the CARMA_Base ctor: 10*sqrt(arma::var(y)) (sample variance, carpack.hpp:71)
            max_stdev = 10.0 * np.sqrt(np.var(y, ddof=1)) if y.size > 1 else 0.0
        self.device = default_device() if device is None else int(device)
        self._h = lib.carma_ctx_create(ptr(time), ptr(y), ptr(yerr), time.size, int(p), int(q), float(max_stdev),
                                       self.device)
        if not self._h:
            msg = "carma_ctx_create failed: " + last_error()
            if "no HIP device" in msg:
                raise CarmaDeviceError(msg)
            raise ValueError(msg)
        self.p, self.q = int(p), int(q)
        self.n = lib.carma_ctx_n(self._h)
        self.d = lib.carma_ctx_dim(self._h)

    def close(self):
        if getattr(self, "_h", None):
            lib.carma_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def data(self):
        t, y, e = np.empty(self.n), np.empty(self.n), np.empty(self.n)
        check(lib.carma_ctx_get_data(self._h, ptr(t), ptr(y), ptr(e)), "carma_ctx_get_data")
        return t, y, e

    def prior(self):
        out = np.empty(3)
        check(lib.carma_ctx_get_prior(self._h, ptr(out)), "carma_ctx_get_prior")
        return tuple(out)

    def set_prior(self, max_stdev):
        check(lib.carma_ctx_set_prior(self._h, float(max_stdev)), "carma_ctx_set_prior")

    def logdensity(self, thetas, ignore_prior=False):
        thetas = as_f64(thetas)
        one = thetas.ndim == 1
        thetas = thetas.reshape(-1, self.d)
        out = np.empty(thetas.shape[0])
        check(lib.carma_logdensity_batch(self._h, ptr(thetas), thetas.shape[0], int(bool(ignore_prior)), ptr(out)),
              "carma_logdensity_batch")
        return float(out[0]) if one else out

    def logdensity_dev(self, d_theta_ptr, B, d_out_ptr, ignore_prior=False, stream=0):
        """Enqueue on `stream` (a hipStream_t as int); pointers are device addresses."""
        check(lib.carma_logdensity_batch_dev(self._h, C.c_void_p(d_theta_ptr), int(B), int(bool(ignore_prior)),
                                             C.c_void_p(d_out_ptr), C.c_void_p(stream)),
              "carma_logdensity_batch_dev")

    def kernel_name(self, B):
        """Name of the kernel a launch of B evaluations takes (carma_logdensity_kernel_name)."""
        buf = C.create_string_buffer(128)
        check(lib.carma_logdensity_kernel_name(self._h, int(B), buf, 128), "carma_logdensity_kernel_name")
        return buf.value.decode()

    def mle_batched(self, x0, bounds, maxiter=2000, mem=8, ftol=2.220446049250313e-09, gtol=1e-5, fd_step=1e-6,
                    ignore_prior=True):
        """carma_mle_batched: lock-step bounded L-BFGS from every row of x0 on -LogDensity, host loop in the library.
        bounds = [(lo, hi)] with None for unbounded.  Returns (x [B, d], fun [B], nit [B], nfev [B], status [B])."""
        x0 = np.ascontiguousarray(np.atleast_2d(np.asarray(x0, dtype=np.float64)))
        B, d = x0.shape
        if d != self.d:
            raise ValueError("x0 must be [B, %d]" % self.d)
        lo = np.array([-np.inf if b[0] is None else b[0] for b in bounds], dtype=np.float64)
        hi = np.array([np.inf if b[1] is None else b[1] for b in bounds], dtype=np.float64)
        x, fun = np.empty((B, d)), np.empty(B)
        nit, nfev, status = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        check(lib.carma_mle_batched(self._h, ptr(x0), B, ptr(lo), ptr(hi), int(maxiter), int(mem), float(ftol), float(gtol),
                                    float(fd_step), 1 if ignore_prior else 0, ptr(x), ptr(fun),
                                    nit.ctypes.data_as(C.c_void_p), nfev.ctypes.data_as(C.c_void_p),
                                    status.ctypes.data_as(C.c_void_p)), "carma_mle_batched")
        return x, fun, nit, nfev, status

    def logprior(self, theta):
        theta = as_f64(theta)
        return lib.carma_logprior(self._h, ptr(theta))

    # ---- parallel-tempering sampler (RunCarmaSampler / RunCar1Sampler on the GPU) ----------------
    def pt_run(self, ntemps, nreplicas, sample_size, burnin, thin=1, init=None, seed=0):
        """Whole Sampler::Run; returns (samples[R][S][d], logposts[R][S]) of the coldest chains."""
        init_a = as_f64(init) if init is not None and len(init) else None
        samples = np.empty((nreplicas, sample_size, self.d))
        logposts = np.empty((nreplicas, sample_size))
        check(lib.carma_pt_run(self._h, int(ntemps), int(nreplicas), int(sample_size), int(burnin), int(thin),
                               ptr(init_a) if init_a is not None else None,
                               init_a.size if init_a is not None else 0, C.c_uint64(int(seed) & (2 ** 64 - 1)),
                               ptr(samples), ptr(logposts)), "carma_pt_run")
        self._pt_shape = (int(nreplicas), int(ntemps))
        return samples, logposts

    def pt_create(self, ntemps, nreplicas, adapt_iters, seed=0, temperatures=None):
        tt = as_f64(temperatures) if temperatures is not None else None
        check(lib.carma_pt_create(self._h, int(ntemps), int(nreplicas), ptr(tt) if tt is not None else None,
                                  int(adapt_iters), C.c_uint64(int(seed) & (2 ** 64 - 1))), "carma_pt_create")
        self._pt_shape = (int(nreplicas), int(ntemps))

    def pt_shard(self, ntemps_global, slot0, replica0):
        check(lib.carma_pt_shard(self._h, int(ntemps_global), int(slot0), int(replica0)), "carma_pt_shard")
        self._pt_slot0 = int(slot0)

    def pt_bind_state(self, d_theta_ptr, d_logpost_ptr):
        check(lib.carma_pt_bind_state(self._h, C.c_void_p(d_theta_ptr), C.c_void_p(d_logpost_ptr)),
              "carma_pt_bind_state")

    def pt_start(self, init=None):
        init_a = as_f64(init) if init is not None and len(init) else None
        check(lib.carma_pt_start(self._h, ptr(init_a) if init_a is not None else None,
                                 init_a.size if init_a is not None else 0), "carma_pt_start")

    def pt_set_chains(self, theta, logpost=None):
        R, T = self._pt_shape
        theta = as_f64(theta).reshape(R, T, self.d)
        lp = as_f64(logpost).reshape(R, T) if logpost is not None else None
        check(lib.carma_pt_set_chains(self._h, ptr(theta), ptr(lp) if lp is not None else None), "carma_pt_set_chains")

    def pt_get_chains(self):
        R, T = self._pt_shape
        theta, lp = np.empty((R, T, self.d)), np.empty((R, T))
        check(lib.carma_pt_get_chains(self._h, ptr(theta), ptr(lp)), "carma_pt_get_chains")
        return theta, lp

    def pt_iterate(self, niter, do_exchange=True):
        check(lib.carma_pt_iterate(self._h, int(niter), int(bool(do_exchange))), "carma_pt_iterate")

    def pt_sample(self, nsamples, thin=1):
        R, T = self._pt_shape
        samples, logposts = np.empty((R, nsamples, self.d)), np.empty((R, nsamples))
        check(lib.carma_pt_sample(self._h, int(nsamples), int(thin), ptr(samples), ptr(logposts)), "carma_pt_sample")
        return samples, logposts

    def pt_stats(self, reset=False):
        R, T = self._pt_shape
        acc, swp = np.empty((R, T)), np.empty((R, T))
        check(lib.carma_pt_stats(self._h, ptr(acc), ptr(swp), int(bool(reset))), "carma_pt_stats")
        return acc, swp

    def pt_get_factor(self):
        """chol_factor_ of every chain, [R][T][d][d] (upper triangular; src/steps.cpp:32)."""
        R, T = self._pt_shape
        chol = np.empty((R, T, self.d, self.d))
        check(lib.carma_pt_get_factor(self._h, ptr(chol)), "carma_pt_get_factor")
        return chol

    def pt_set_factor(self, chol):
        R, T = self._pt_shape
        chol = np.ascontiguousarray(chol, dtype=np.float64).reshape(R, T, self.d, self.d)
        check(lib.carma_pt_set_factor(self._h, ptr(chol)), "carma_pt_set_factor")

    def pt_debug_draws(self, replica, temperature, iteration):
        """(z[d], u_accept, u_swap): the variates chain (replica, temperature) uses at `iteration`, from the device's own
        generator (carma_pt_debug_draws)."""
        z, ua, us = np.empty(self.d), np.empty(1), np.empty(1)
        check(lib.carma_pt_debug_draws(self._h, int(replica), int(temperature), int(iteration), ptr(z), ptr(ua), ptr(us)),
              "carma_pt_debug_draws")
        return z, float(ua[0]), float(us[0])

    def pt_kernel(self):
        """"row" (k_pt_row), "ladder" (k_pt) or "lane" (k_pt_lane, large ensembles): the sampler kernel this context is on
        (carma_pt_kernel_in_use)."""
        return {1: "row", 2: "lane"}.get(lib.carma_pt_kernel_in_use(self._h), "ladder")

    @staticmethod
    def pt_row_pipeline():
        """Recursion of the process's last k_pt_row launch: "one-datum", "window", "two-sided" (carma_pt_row_pipeline); None before it."""
        return {0: "one-datum", 1: "window", 2: "two-sided"}.get(lib.carma_pt_row_pipeline())

    def pt_iterations_done(self):
        return lib.carma_pt_iterations_done(self._h)

    def pt_boundary_stats(self):
        """(proposed, accepted) swaps across this block's boundaries (carma_pt_iterate_sharded)."""
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        check(lib.carma_pt_boundary_stats(self._h, C.byref(a), C.byref(b)), "carma_pt_boundary_stats")
        return a.value, b.value

    def pt_boundary_check(self):
        """1: the last sharded call's boundary self-check agreed; -1: it differed; 0: none yet."""
        return int(lib.carma_pt_boundary_check(self._h))

    def pt_sweep(self):
        """The swap sweep inside this block for the iteration just run with do_exchange=False (carma_pt_sweep)."""
        check(lib.carma_pt_sweep(self._h), "carma_pt_sweep")


class Comm:
    """RCCL communicator owned by libcarma_mi355.so (carma_comm_*): one rank per process / GPU.

    ``Comm.unique_id()`` on rank 0 -> 128 bytes to broadcast with the host program's own bootstrap -> every rank
    ``Comm(id, nranks, rank, device)`` (collective)."""

    @staticmethod
    def unique_id():
        _share_rccl_with_torch()
        buf = C.create_string_buffer(128)
        check(lib.carma_comm_unique_id(buf), "carma_comm_unique_id")
        return buf.raw

    def __init__(self, unique_id, nranks, rank, device=None):
        _share_rccl_with_torch()
        self.device = default_device() if device is None else int(device)
        self._id = C.create_string_buffer(bytes(unique_id), 128)
        self._h = lib.carma_comm_create(self._id, int(nranks), int(rank), self.device)
        if not self._h:
            raise CarmaError("carma_comm_create failed: " + last_error())
        self.rank, self.size = int(rank), int(nranks)

    @classmethod
    def from_torch(cls, dist, device=None):
        """Bootstrap over an initialised torch.distributed process group (any backend)."""
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(box[0], world, rank, device)

    def close(self):
        if getattr(self, "_h", None):
            lib.carma_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pt_iterate_sharded(contexts, niter, comm=None):
    """carma_pt_iterate_sharded: `contexts` = this process's consecutive ladder blocks (normally one)."""
    arr = (C.c_void_p * len(contexts))(*[c.handle for c in contexts])
    check(lib.carma_pt_iterate_sharded(arr, len(contexts), int(niter), comm._h if comm is not None else None),
          "carma_pt_iterate_sharded")


class KalmanHandle:
    """carma_kf: a KalmanFilter1 / KalmanFilterp object whose series and model stay resident in HBM; Filter and any
    number of (batched) Predict calls only launch and copy results back."""

    def __init__(self, time, y, yerr, sigsqr, omega, ma=None, device=None):
        time, y, yerr = as_f64(time), as_f64(y), as_f64(yerr)
        dev = default_device() if device is None else int(device)
        if ma is None:                                   # CAR(1): omega is the real rate 1 / tau
            self._h = lib.carma_kf_create_car1(ptr(time), ptr(y), ptr(yerr), time.size, float(sigsqr), float(omega), dev)
        else:
            omega = np.asarray(omega, dtype=complex)
            om, ma = as_f64(np.c_[omega.real, omega.imag]), as_f64(ma)
            self._h = lib.carma_kf_create_carma(ptr(time), ptr(y), ptr(yerr), time.size, omega.size, float(sigsqr), ptr(om),
                                                ptr(ma), ma.size, dev)
        if not self._h:
            msg = "carma_kf_create failed: " + last_error()
            raise CarmaDeviceError(msg) if "no HIP device" in msg else ValueError(msg)
        self.n = lib.carma_kf_n(self._h)

    def filter(self):
        mean, var = np.empty(self.n), np.empty(self.n)
        rc = lib.carma_kf_filter(self._h, ptr(mean), ptr(var))
        if rc == 1:
            raise CarmaError("KalmanFilterp: singular eigenvector matrix (solve failed)")
        check(rc, "carma_kf_filter")
        return mean, var

    def predict(self, tpred):
        tp = as_f64(np.atleast_1d(tpred))
        pm, pv = np.empty(tp.size), np.empty(tp.size)
        rc = lib.carma_kf_predict(self._h, ptr(tp), tp.size, ptr(pm), ptr(pv))
        if rc == 1:
            raise CarmaError("KalmanFilterp: singular eigenvector matrix (solve failed)")
        check(rc, "carma_kf_predict")
        return pm, pv

    def close(self):
        if getattr(self, "_h", None):
            lib.carma_kf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def kfilter_carma(time, y, yerr, sigsqr, omega, ma, device=None):
    time, y, yerr = as_f64(time), as_f64(y), as_f64(yerr)
    omega = np.asarray(omega, dtype=complex)
    om = as_f64(np.c_[omega.real, omega.imag])
    ma = as_f64(ma)
    mean, var = np.empty(time.size), np.empty(time.size)
    nout = C.c_int(0)
    rc = lib.carma_kfilter_carma(ptr(time), ptr(y), ptr(yerr), time.size, omega.size, float(sigsqr), ptr(om), ptr(ma),
                                 ma.size, ptr(mean), ptr(var), C.byref(nout),
                                 default_device() if device is None else device)
    if rc == 1:
        raise CarmaError("KalmanFilterp: singular eigenvector matrix (solve failed)")
    check(rc, "carma_kfilter_carma")
    return mean[:nout.value], var[:nout.value]


def kfilter_carma_batch(time, y, yerr, sigsqr, omega, ma, mu=None, device=None):
    """Filter() of B models on one series in one launch: sigsqr [B], omega [B][p] complex, ma [B][nma], mu [B] or None
    -> (mean [B][n], var [B][n], singular [B] bool).  mu is subtracted from y inside and added back to mean."""
    time, y, yerr = as_f64(time), as_f64(y), as_f64(yerr)
    omega = np.atleast_2d(np.asarray(omega, dtype=complex))
    B, p = omega.shape
    om = as_f64(np.stack([omega.real, omega.imag], axis=-1))
    ma = as_f64(np.atleast_2d(ma))
    sig = as_f64(np.atleast_1d(sigsqr))
    if ma.shape[0] != B or sig.size != B:
        raise ValueError("kfilter_carma_batch: sigsqr, omega and ma must describe the same number of models")
    mu_ = None if mu is None else as_f64(np.atleast_1d(mu))
    if mu_ is not None and mu_.size != B:
        raise ValueError("kfilter_carma_batch: mu must have one entry per model")
    mean, var = np.empty((B, time.size)), np.empty((B, time.size))
    sing = np.zeros(B, dtype=np.int32)
    nout = C.c_int(0)
    rc = lib.carma_kfilter_batch_carma(ptr(time), ptr(y), ptr(yerr), time.size, p, B, ptr(sig), ptr(om), ptr(ma), ma.shape[1],
                                       ptr(mu_) if mu_ is not None else None, ptr(mean), ptr(var),
                                       sing.ctypes.data_as(C.POINTER(C.c_int)), C.byref(nout),
                                       default_device() if device is None else device)
    check(rc, "carma_kfilter_batch_carma")
    m = nout.value
    # (the library writes rows of n_out values back to back)
    mean = mean.reshape(-1)[:B * m].reshape(B, m)
    var = var.reshape(-1)[:B * m].reshape(B, m)
    return mean, var, sing.astype(bool)


def kfilter_car1(time, y, yerr, sigsqr, omega, device=None):
    time, y, yerr = as_f64(time), as_f64(y), as_f64(yerr)
    mean, var = np.empty(time.size), np.empty(time.size)
    nout = C.c_int(0)
    rc = lib.carma_kfilter_car1(ptr(time), ptr(y), ptr(yerr), time.size, float(sigsqr), float(omega), ptr(mean),
                                ptr(var), C.byref(nout), default_device() if device is None else device)
    check(rc, "carma_kfilter_car1")
    return mean[:nout.value], var[:nout.value]


def predict_carma(time, y, yerr, sigsqr, omega, ma, tpred, device=None):
    """KalmanFilterp::Predict for all `tpred` in one launch -> (mean[M], var[M])."""
    time, y, yerr = as_f64(time), as_f64(y), as_f64(yerr)
    omega = np.asarray(omega, dtype=complex)
    om = as_f64(np.c_[omega.real, omega.imag])
    ma = as_f64(ma)
    tp = as_f64(np.atleast_1d(tpred))
    pm, pv = np.empty(tp.size), np.empty(tp.size)
    rc = lib.carma_predict_carma(ptr(time), ptr(y), ptr(yerr), time.size, omega.size, float(sigsqr), ptr(om), ptr(ma),
                                 ma.size, ptr(tp), tp.size, ptr(pm), ptr(pv),
                                 default_device() if device is None else device)
    if rc == 1:
        raise CarmaError("KalmanFilterp: singular eigenvector matrix (solve failed)")
    check(rc, "carma_predict_carma")
    return pm, pv


def predict_car1(time, y, yerr, sigsqr, omega, tpred, device=None):
    time, y, yerr = as_f64(time), as_f64(y), as_f64(yerr)
    tp = as_f64(np.atleast_1d(tpred))
    pm, pv = np.empty(tp.size), np.empty(tp.size)
    check(lib.carma_predict_car1(ptr(time), ptr(y), ptr(yerr), time.size, float(sigsqr), float(omega), ptr(tp), tp.size,
                                 ptr(pm), ptr(pv), default_device() if device is None else device),
          "carma_predict_car1")
    return pm, pv


def simulate_carma(time, sigsqr, omega, ma, npaths=1, seed=0, device=None):
    """carma_process for `npaths` paths in one launch -> [npaths][n] at the sorted times (carma_simulate_carma)."""
    time = as_f64(np.sort(np.asarray(time, dtype=float)))
    omega = np.asarray(omega, dtype=complex)
    om = as_f64(np.c_[omega.real, omega.imag])
    ma = as_f64(ma)
    out = np.empty((int(npaths), time.size))
    rc = lib.carma_simulate_carma(ptr(time), time.size, omega.size, float(sigsqr), ptr(om), ptr(ma), ma.size, int(npaths),
                                  C.c_uint64(int(seed) & (2 ** 64 - 1)), ptr(out), default_device() if device is None else device)
    if rc == 1:
        raise CarmaError("carma_process: repeated AR root (singular eigenvector matrix)")
    check(rc, "carma_simulate_carma")
    return out


def simulate_car1(time, sigsqr, omega, npaths=1, seed=0, device=None):
    time = as_f64(np.sort(np.asarray(time, dtype=float)))
    out = np.empty((int(npaths), time.size))
    check(lib.carma_simulate_car1(ptr(time), time.size, float(sigsqr), float(omega), int(npaths),
                                  C.c_uint64(int(seed) & (2 ** 64 - 1)), ptr(out), default_device() if device is None else device),
          "carma_simulate_car1")
    return out


def sigma_noise_batch(ar_roots, ma_coefs, var, device=None):
    """CarmaSample._sigma_noise for all samples in one launch (carma_sigma_noise_batch): ar_roots [ns, p] complex,
    ma_coefs [ns, nma] lowest order first, var [ns] -> sigma [ns]."""
    roots = np.atleast_2d(np.asarray(ar_roots, dtype=complex))
    ns, p = roots.shape
    om = as_f64(np.stack([roots.real, roots.imag], axis=-1))
    ma = as_f64(np.atleast_2d(np.asarray(ma_coefs, dtype=float)))
    v = as_f64(np.ravel(var))
    if ma.shape[0] != ns or v.size != ns:
        raise ValueError("sigma_noise_batch: one row of roots, MA coefficients and one variance per sample")
    out = np.empty(ns)
    check(lib.carma_sigma_noise_batch(p, ma.shape[1], ptr(om), ptr(ma), ptr(v), ns, ptr(out),
                                      default_device() if device is None else device), "carma_sigma_noise_batch")
    return out


def psd_band(ar_coefs, ma_coefs, sigma, freq, percentiles, return_samples=False, device=None):
    """The power spectrum of every sample on `freq` and np.percentile(..., percentiles) over the samples, on the device
    (carma_psd_band).  ar_coefs [ns, p + 1] highest order first, ma_coefs [ns, nma] lowest first, sigma [ns].
    Returns band [nf, nperc] (and the grid [nf, ns] with return_samples)."""
    ar = as_f64(np.atleast_2d(np.asarray(ar_coefs, dtype=float)))
    ma = as_f64(np.atleast_2d(np.asarray(ma_coefs, dtype=float)))
    sg = as_f64(np.ravel(sigma))
    fr = as_f64(np.ravel(freq))
    pc = as_f64(np.ravel(percentiles))
    ns = ar.shape[0]
    if ma.shape[0] != ns or sg.size != ns:
        raise ValueError("psd_band: one row of AR coefficients, MA coefficients and one sigma per sample")
    band = np.empty((fr.size, pc.size))
    grid = np.empty((fr.size, ns)) if return_samples else None
    check(lib.carma_psd_band(ar.shape[1], ma.shape[1], ptr(ar), ptr(ma), ptr(sg), ns, ptr(fr), fr.size, ptr(pc), pc.size,
                             ptr(band), ptr(grid) if return_samples else None, default_device() if device is None else device),
          "carma_psd_band")
    return (band, grid) if return_samples else band


def pt_sample_sharded(contexts, nsamples, thin=1, comm=None):
    """carma_pt_sample_sharded: returns (samples[R][nsamples][d], logposts[R][nsamples]) on the process that owns
    temperature 0, (None, None) elsewhere."""
    arr = (C.c_void_p * len(contexts))(*[c.handle for c in contexts])
    c0 = contexts[0]
    owner = getattr(c0, "_pt_slot0", 0) == 0
    R = c0._pt_shape[0]
    samples = np.empty((R, int(nsamples), c0.d)) if owner else None
    logposts = np.empty((R, int(nsamples))) if owner else None
    check(lib.carma_pt_sample_sharded(arr, len(contexts), int(nsamples), int(thin), comm._h if comm is not None else None,
                                      ptr(samples) if owner else None, ptr(logposts) if owner else None),
          "carma_pt_sample_sharded")
    return samples, logposts


def normalize_roots(omega):
    """AR roots in any order -> conjugate pairs adjacent (negative imaginary part first), real roots last
    (carma_normalize_roots; host arithmetic, no device needed).  ValueError when the set is not closed under conjugation."""
    omega = np.asarray(omega, dtype=complex)
    om = as_f64(np.c_[omega.real, omega.imag])
    out = np.empty_like(om)
    rc = lib.carma_normalize_roots(omega.size, ptr(om), ptr(out))
    if rc != CARMA_OK:
        raise ValueError("the AR roots must be real or come in complex-conjugate pairs")
    return out[:, 0] + 1j * out[:, 1]
