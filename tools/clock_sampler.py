#!/usr/bin/env python3
"""tools/clock_sampler.py out.txt [period_s] -- sample the GPU's shader clock and socket power from sysfs (hwmon) until killed.
Cheap enough for 10 ms periods (rocm-smi takes ~0.2 s per sample)."""
import glob
import sys
import time

out = open(sys.argv[1], "w")
period = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
freq = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
power = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")) or \
    sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average"))
out.write("# t_s sclk_MHz power_W   (%s, %s)\n" % (freq[:1], power[:1]))
t0 = time.time()
try:
    while True:
        f = p = -1.0
        try:
            if freq:
                f = int(open(freq[0]).read()) / 1e6
            if power:
                p = int(open(power[0]).read()) / 1e6
        except Exception:
            pass
        out.write("%.3f %.0f %.0f\n" % (time.time() - t0, f, p))
        out.flush()
        time.sleep(period)
except KeyboardInterrupt:
    pass
