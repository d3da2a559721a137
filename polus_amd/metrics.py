"""polus/metrics.py drop-in: IMetric protocol, int32 confusion-matrix accumulation, macro-F1
in float64 with divide_no_nan.  Eval-only bookkeeping on a C x C matrix (SURVEY.md §2 row 12:
out of scope as compute), kept on the host in NumPy; predictions arrive as device or host
arrays."""
import numpy as np


def _np(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


class IMetric:
    def __init__(self, reduce_f=None):
        super().__init__()
        if self.__class__.__name__ == "IMetric":
            raise Exception("This is an interface that cannot be instantiated")
        self.name = self.__class__.__name__
        self.reduce_f = reduce_f

    def samples_from_batch(self, samples):
        if self.reduce_f is not None:
            samples = self.reduce_f(samples)
        self._samples_from_batch(samples)

    def _samples_from_batch(self, samples):
        raise Exception("_samples_from_batch was internally called, but is not implemented")

    def reset(self):
        raise Exception("clear was called, but is not implemented")

    def _evaluate(self):
        raise Exception("_evaluate was internally called, but is not implemented")

    def evaluate(self):
        measure = self._evaluate()
        self.reset()
        return measure


class IConfusionMatrixTF(IMetric):
    def __init__(self, num_classes, reduce_f=None):
        super().__init__(reduce_f=reduce_f)
        if self.__class__.__name__ == "IConfusionMatrixTF":
            raise Exception("This is an interface that cannot be instantiated")
        self.num_classes = num_classes
        self.reset()

    def _samples_from_batch(self, samples):
        self.confusion_matrix += self._build_confusion_matrix(*samples)

    def _build_confusion_matrix(self, y_true, y_pred):
        """tf.math.confusion_matrix: rows = first argument (polus/metrics.py:51-59)."""
        cm = np.zeros((self.num_classes, self.num_classes), np.int32)
        np.add.at(cm, (_np(y_true).reshape(-1).astype(np.int64), _np(y_pred).reshape(-1).astype(np.int64)), 1)
        return cm

    def reset(self):
        self.confusion_matrix = np.zeros((self.num_classes, self.num_classes), np.int32)


def _divide_no_nan(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    out = np.zeros(np.broadcast(a, b).shape, np.float64)
    np.divide(a, b, out=out, where=(b != 0))
    return out


class MacroF1Score(IConfusionMatrixTF):
    def _evaluate(self):
        m = self.confusion_matrix
        tp = np.diag(m).astype(np.float64)
        precision = _divide_no_nan(tp, m.sum(-1))
        recall = _divide_no_nan(tp, m.sum(-2))
        inv_p, inv_r = _divide_no_nan(1.0, precision), _divide_no_nan(1.0, recall)
        return float(np.mean(_divide_no_nan(2.0, inv_p + inv_r)))


class Accuracy(IConfusionMatrixTF):
    def _evaluate(self):
        m = self.confusion_matrix.astype(np.float64)
        return float(_divide_no_nan(np.trace(m), m.sum()))
