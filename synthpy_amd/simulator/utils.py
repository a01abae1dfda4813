"""Seeded RNG wrappers of the reference (src/simulator/utils.py:8-24): with seed=True each call
re-seeds the global NumPy stream with 0 before drawing — that is what defines a "seeded" Beam."""
import numpy as np


def random_array(length, seed=False):
    if seed:
        np.random.seed(0)
    return np.random.rand(length)


def random_array_n(length, seed=False):
    if seed:
        np.random.seed(0)
    return np.random.randn(length)


def random_inv_pow_array(power, length, seed=False):
    if seed:
        np.random.seed(0)
    return np.random.power(power, length)


def mem_conversion(mem_size):
    """Bytes -> human readable string (src/simulator/utils.py:40-57)."""
    count = 0
    while mem_size > 1024:
        mem_size /= 1024
        count += 1
    unit = ["B", "KB", "MB", "GB"][count] if count < 4 else "TB+"
    return str(mem_size) + " " + unit


def domain_estimate(dim):
    return dim[0] * dim[1] * dim[2] * 4
