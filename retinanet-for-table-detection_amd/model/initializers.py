"""model/initializers.py:9-22."""
import math

import numpy as np


class PriorProbability:
    """ Apply a prior probability to the weights (bias = -log((1 - p) / p))."""

    def __init__(self, probability=0.01):
        self.probability = probability

    def get_config(self):
        return {'probability': self.probability}

    def __call__(self, shape, dtype=None):
        return np.ones(shape, dtype=dtype) * -math.log((1 - self.probability) / self.probability)
