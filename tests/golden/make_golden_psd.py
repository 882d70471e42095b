"""Golden vectors for the post-processing rows: the reference's own
``CarmaSample.plot_power_spectrum`` / ``Car1Sample.plot_power_spectrum`` (carma_pack.py:548-648, 950-1035) and
``Car1Sample.makeKalmanFilter`` inputs, run HERE (build container only) on a stub sampler.

    python tests/golden/make_golden_psd.py      ->  tests/golden/psd.npz

The reference is imported from /root/reference with its compiled extension stubbed (as make_golden.py does);
nothing of it travels: only the numbers below are committed."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference's carma_pack as mg.cp)

cp = mg.cp


class StubSampler(object):
    """What CarmaSample / Car1Sample need from the C++ object: samples, stored log-posteriors, log-densities."""

    def __init__(self, samples):
        self.s = np.asarray(samples)

    def getSamples(self):
        return self.s.tolist()

    def GetLogLikes(self):
        return np.linspace(-100.0, -90.0, self.s.shape[0]).tolist()

    def SetMLE(self, flag):
        pass

    def getLogDensity(self, theta):
        return -95.0

    def getLogPrior(self, theta):
        return -1.0


def main():
    g = np.load(os.path.join(HERE, "carma53_readme.npz"))
    t, y, yerr, th = g["t"], g["y"], g["yerr"], g["theta"]
    s = cp.CarmaSample(t, y, yerr, StubSampler(th), q=3)
    lo, hi, med, f = s.plot_power_spectrum(percentile=68.0, doShow=False)[:4]
    lo9, hi9, med9, _ = s.plot_power_spectrum(percentile=95.0, nsamples=9, doShow=False)[:4]
    # CAR(1): theta = (sigma_y, measerr scale, mu, log omega)
    rng = np.random.default_rng(11)
    th1 = np.c_[2.3 + 0.1 * rng.standard_normal(40), 1.0 + 0.02 * rng.standard_normal(40),
                17.0 + 0.1 * rng.standard_normal(40), np.log(0.01) + 0.2 * rng.standard_normal(40)]
    s1 = cp.Car1Sample(t, y, yerr, StubSampler(th1))
    lo1, hi1, med1, f1 = s1.plot_power_spectrum(percentile=68.0, doShow=False)[:4]
    np.savez_compressed(os.path.join(HERE, "psd.npz"), freq=f, lo68=lo, hi68=hi, med68=med, lo95_n9=lo9, hi95_n9=hi9,
                        med95_n9=med9, car1_theta=th1, car1_freq=f1, car1_lo68=lo1, car1_hi68=hi1, car1_med68=med1,
                        car1_sigma=np.ravel(s1._samples["sigma"]))
    print("wrote psd.npz:", f.size, "frequencies;", th.shape[0], "CARMA samples,", th1.shape[0], "CAR(1) samples")


if __name__ == "__main__":
    main()
