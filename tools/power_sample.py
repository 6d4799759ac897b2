"""Background sampler of the GPUs' hwmon power / shader-clock files (what `rocm-smi --showpower --showclocks` reads), for the dev tools that
price a kernel against the part's power cap (tools/conv_clock.py, tools/micro/mfma_power.hip has its own copy in C++).
    with PowerSampler() as ps: ...run kernels...;  print(ps.summary())"""
import glob
import threading
import time


class PowerSampler:
    def __init__(self, period=0.002):
        self.pw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")) or \
            sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
        self.fq = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
        self.period = period
        self.rows = []

    @staticmethod
    def _rd(p):
        try:
            with open(p) as f:
                return float(f.read().split()[0])
        except (OSError, ValueError, IndexError):
            return float("nan")

    def __enter__(self):
        self._stop = False
        self.rows = []

        def loop():
            while not self._stop:
                self.rows.append([self._rd(p) for p in self.pw + self.fq])
                time.sleep(self.period)
        self._t = threading.Thread(target=loop, daemon=True)
        self._t.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        self._t.join()

    def summary(self):
        if not self.rows or not self.pw:
            return "power: n/a"
        n = len(self.rows)
        mean = [sum(r[i] for r in self.rows) / n for i in range(len(self.pw) + len(self.fq))]
        return "power W: %s  sclk MHz: %s (%d samples)" % (" ".join("%.0f" % (v * 1e-6) for v in mean[:len(self.pw)]),
                                                            " ".join("%.0f" % (v * 1e-6) for v in mean[len(self.pw):]), n)
