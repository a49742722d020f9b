"""Initialisers of the head biases (the reference's model/initializers.py:9-22 surface: PriorProbability).

The classification head's output bias starts at the logit of a prior foreground probability p, logit(p) = ln p - ln(1 - p),
so that sigmoid(bias) = p on every anchor at step 0 (focal-loss paper, section 4.1).  NumPy only: weights.py consumes the
array when it builds a state dict."""
import numpy as np


def prior_logit(probability):
    """ln(p / (1 - p)) in float64."""
    p = float(probability)
    if not 0.0 < p < 1.0:
        raise ValueError("probability must lie strictly between 0 and 1, got %r" % (probability,))
    return float(np.log(p) - np.log1p(-p))


class PriorProbability:
    """Callable initialiser: PriorProbability(p)(shape, dtype) -> array filled with logit(p).  Keras-style get_config()."""

    def __init__(self, probability=0.01):
        self.probability = probability
        self._logit = prior_logit(probability)

    def __call__(self, shape, dtype=None):
        return np.full(shape, self._logit, dtype=dtype if dtype is not None else np.float32)

    def get_config(self):
        return dict(probability=self.probability)

    @classmethod
    def from_config(cls, config):
        return cls(**config)
