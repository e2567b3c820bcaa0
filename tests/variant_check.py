"""Child process of test_conv_kernel_variants: conv parity with SGMSE_CONV_VARIANT taken from the environment.
Usage: python variant_check.py <library path> <device>"""
import sys

import torch  # noqa: F401

from sgmse_amd import _lib

_lib.load_library(sys.argv[1])
import parity as P

dev = sys.argv[2]
for shp in [(1, 64, 128, 9, 33, 3), (1, 128, 128, 8, 32, 3), (2, 32, 128, 20, 40, 3), (1, 96, 128, 8, 32, 1), (1, 8, 128, 8, 8, 3),
            (1, 24, 128, 8, 8, 3), (1, 12, 128, 8, 32, 3), (1, 160, 128, 10, 36, 1), (1, 64, 256, 8, 32, 3)]:
    P.check_conv(dev, *shp)
P.check_conv(dev, 2, 96, 128, 12, 36, 3, dual=64, xform=True)
P.check_conv(dev, 2, 128, 128, 16, 32, 3, dual=64, xform=True)
P.check_conv(dev, 1, 64, 128, 8, 32, 1, dual=32, xform=True)
P.check_conv(dev, 2, 96, 256, 5, 40, 1, xform=True)
P.check_conv(dev, 1, 160, 128, 16, 20, 1, dual=64, xform=True)
print("VARIANT-OK")
