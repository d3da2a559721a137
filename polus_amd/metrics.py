"""polus/metrics.py drop-in: IMetric protocol, int32 confusion-matrix accumulation, macro-F1
in float64 with divide_no_nan.  Predictions that arrive as device tensors (the inference kernels'
int32 output) are counted where they are -- one `polus_confusion_matrix` launch per batch into a
C x C int32 matrix in HBM, no device-to-host copy per validation step -- and only the C x C counts
come to the host in `evaluate()`; host arrays (the gathered tuples of other ranks) are counted in NumPy."""
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
        y_true, y_pred = samples
        if getattr(y_true, "is_cuda", False) and getattr(y_pred, "is_cuda", False) and self.num_classes <= 128:
            import torch
            from . import ops
            if self._dev_cm is None:
                self._dev_cm = torch.zeros((self.num_classes, self.num_classes), dtype=torch.int32, device=y_true.device)
                self._dev_rejected = torch.zeros(1, dtype=torch.int32, device=y_true.device)
            a = y_true.reshape(-1).to(torch.int32).contiguous()
            b = y_pred.reshape(-1).to(torch.int32).contiguous()
            ops.confusion_matrix(a, b, self._dev_cm, self._dev_rejected)
            return
        # host arrays, and device tensors with more than 128 classes (the kernel keeps its C x C histogram in LDS)
        self._host_cm += self._build_confusion_matrix(y_true, y_pred)

    def _build_confusion_matrix(self, y_true, y_pred):
        """tf.math.confusion_matrix: rows = first argument (polus/metrics.py:51-59)."""
        cm = np.zeros((self.num_classes, self.num_classes), np.int32)
        r, c = _np(y_true).reshape(-1).astype(np.int64), _np(y_pred).reshape(-1).astype(np.int64)
        bad = int(((r < 0) | (r >= self.num_classes) | (c < 0) | (c >= self.num_classes)).sum())
        if bad:
            raise ValueError(f"{self.name}: {bad} label / prediction value(s) outside [0, {self.num_classes}) "
                             "(tf.math.confusion_matrix rejects them too)")
        np.add.at(cm, (r, c), 1)
        return cm

    @property
    def confusion_matrix(self):
        """The counts so far (host int32 [C, C]); reading it synchronises with the device side."""
        if self._dev_cm is None:
            return self._host_cm
        bad = int(self._dev_rejected.item())
        if bad:
            # same verdict as the host path (and as tf.math.confusion_matrix), raised where the counts are read
            raise ValueError(f"{self.name}: {bad} label / prediction value(s) outside [0, {self.num_classes}) "
                             "(tf.math.confusion_matrix rejects them too)")
        return self._host_cm + self._dev_cm.cpu().numpy()

    def reset(self):
        self._host_cm = np.zeros((self.num_classes, self.num_classes), np.int32)
        self._dev_cm = self._dev_rejected = None


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
