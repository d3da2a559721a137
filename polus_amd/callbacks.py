"""Callback protocol of polus/callbacks.py, re-stated over the MI355X engine.

Pure host-side control flow (out of scope as compute) but part of the
drop-in surface: six hooks + add_coordinator (:47-73,:134-140), the coordinator's shared
blackboard and output-streamer bus (:77-132), and the stock callbacks.  Reference quirks
that user code can observe are kept and marked "quirk".
"""
import os
import sys
from collections import OrderedDict, defaultdict
from functools import wraps
from timeit import default_timer as timer

import numpy as np

from . import comm as hvd
from .context import logger


def runs_if_root(method):
    """polus/callbacks.py:24-29 — gated on local_rank() == 0 (single-node semantics)."""
    @wraps(method)
    def _impl(self, *a, **kw):
        if hvd.local_rank() == 0:
            return method(self, *a, **kw)
    return _impl


class IOutput:
    def __init__(self):
        super().__init__()
        self.data = OrderedDict()

    def write(self, key, value):
        self.data[key] = value

    def flush(self):
        out, self.data = self.data, OrderedDict()
        return out


class ICallback:
    def __init__(self):
        super().__init__()
        if self.__class__.__name__ == "ICallback":
            raise Exception("This is an interface that cannot be instantiated")

    def on_train_begin(self): pass
    def on_epoch_begin(self, epoch): pass
    def on_train_batch_begin(self, epoch, step): pass
    def on_train_batch_end(self, epoch, step, loss): pass
    def on_epoch_end(self, epoch): pass
    def on_train_end(self): pass


class CallbackCoordinator(ICallback):
    def __init__(self, callbacks, trainer, epochs, steps):
        super().__init__()
        self.callbacks, self.trainer, self.epochs, self.steps = callbacks, trainer, epochs, steps
        self.shared_dict = {}
        self.output_streamers = []
        for c in self.callbacks:
            c.add_coordinator(self)
            if isinstance(c, IOutput):
                self.output_streamers.append(c)

    def has_callback(self, callback_class):
        return any(isinstance(c, callback_class) for c in self.callbacks)

    def on_train_begin(self):
        for c in self.callbacks: c.on_train_begin()

    def on_epoch_begin(self, epoch):
        for c in self.callbacks: c.on_epoch_begin(epoch)

    def on_train_batch_begin(self, epoch, step):
        for c in self.callbacks: c.on_train_batch_begin(epoch, step)

    def on_train_batch_end(self, epoch, step, loss):
        for c in self.callbacks: c.on_train_batch_end(epoch, step, loss)

    def on_epoch_end(self, epoch):
        for c in self.callbacks: c.on_epoch_end(epoch)

    def on_train_end(self):
        for c in self.callbacks: c.on_train_end()


class Callback(ICallback):
    def __init__(self):
        super().__init__()
        self.coordinator = None

    def add_coordinator(self, coordinator):
        self.coordinator = coordinator


class TimerCallback(Callback):
    """Wall time from on_train_batch_begin to on_train_batch_end: data fetch + step
    (on_train_batch_begin fires before the fetch, polus/training.py:308-313)."""

    def __init__(self):
        super().__init__()
        self.start = None

    def on_train_batch_begin(self, epoch, step):
        self.start = timer()

    def on_train_batch_end(self, epoch, step, loss):
        for output in self.coordinator.output_streamers:
            output.write("time", timer() - self.start)


class LossSmoothCallback(Callback):
    def __init__(self, beta=0.97, output=False):
        super().__init__()
        self.beta, self.output = beta, output
        self.mov_avg, self.n, self.smooth_loss = 0, 0, 0

    def _maybe_output(self):
        if self.output:
            for output in self.coordinator.output_streamers:
                output.write("smooth loss", self.smooth_loss)

    @runs_if_root
    def on_train_batch_end(self, epoch, step, loss):
        self.n += 1
        self.mov_avg = self.beta * self.mov_avg + (1 - self.beta) * float(loss)
        self.smooth_loss = self.mov_avg / (1 - self.beta ** self.n)
        self.coordinator.shared_dict["smooth_loss"] = self.smooth_loss
        self._maybe_output()

    @runs_if_root
    def on_epoch_end(self, epoch):
        self._maybe_output()


class ValidationDataCallback(Callback):
    """polus/callbacks.py:190-261.  Quirk kept: the (prediction, label) tuple is handed to
    metrics whose signature is (y_true, y_pred) (:233 vs polus/metrics.py:51-56)."""

    def __init__(self, tf_validation, custom_inference_f=None, name=None, show_progress=False, validation_interval=1):
        super().__init__()
        self.tf_validation, self.custom_inference_f = tf_validation, custom_inference_f
        self.name, self.show_progress, self.validation_interval = name, show_progress, validation_interval

    @runs_if_root
    def on_train_begin(self):
        sd = self.coordinator.shared_dict
        sd.setdefault("validation", {})
        if self.name is None:
            self.name = len(sd["validation"])
        sd["validation"][self.name] = {m.name: [] for m in self.coordinator.trainer.metrics}

    def get_metrics(self):
        return self.coordinator.shared_dict["validation"][self.name]

    def on_epoch_end(self, epoch):
        if epoch % self.validation_interval:
            return
        from .models import PolusClassifier
        trainer = self.coordinator.trainer
        for step, sample in enumerate(self.tf_validation):
            if self.show_progress:
                print(f"{step}", end="\r")
            if self.custom_inference_f is not None:
                y = self.custom_inference_f(trainer.model, sample)
            elif isinstance(sample, (list, tuple)) and len(sample) == 2:
                if isinstance(trainer.model, PolusClassifier) or hasattr(trainer.model, "inference"):
                    y = trainer.model.inference(sample[0]), sample[1]
                else:
                    logger.warning("model has no inference(); running it directly over the validation data")
                    y = trainer.model(sample[0]), sample[1]
            else:
                raise ValueError("Sample format outputed by the validator dataset is not supported, "
                                 "change to a dict or a two length tuple")
            all_predictions = hvd.allgather_object(y)
            if hvd.local_rank() == 0:
                for pred in all_predictions:
                    for metric in trainer.metrics:
                        metric.samples_from_batch(pred)
        if hvd.local_rank() == 0:
            res = self.coordinator.shared_dict["validation"][self.name]
            for metric in trainer.metrics:
                res[metric.name].append(metric.evaluate())
            for output in self.coordinator.output_streamers:
                output.write(f"Validation {self.name}", res)


class SaveModelCallback(Callback):
    def __init__(self, strategy, validation_name=None, metric_name=None, cache_folder=None, selection_dict_key=None):
        super().__init__()
        self.strategy, self.validation_name, self.metric_name = strategy, validation_name, metric_name
        self.cache_folder, self.selection_dict_key = cache_folder, selection_dict_key
        if strategy not in ("every", "best", "end"):
            logger.warning(f"The selected strategy ({strategy}) is not supported, so this callback will be ignored")
        if strategy == "best":
            self.best = 0  # quirk: starts at 0 and maximises

    def _save(self, **kw):
        if self.cache_folder is not None:
            kw["base_path"] = self.cache_folder
        self.coordinator.trainer.model.save(**kw)

    @runs_if_root
    def on_epoch_end(self, epoch):
        if self.strategy == "best":
            last = self.coordinator.shared_dict["validation"][self.validation_name][self.metric_name][-1]
            metric = self.selection_dict_key(last) if isinstance(last, dict) else last
            if metric > self.best:
                self.best = metric
                self._save(extension=f"_{self.validation_name}_{self.metric_name}_best")
        elif self.strategy == "every":
            self._save(extension=f"_epoch_{epoch}")

    @runs_if_root
    def on_train_end(self):
        if self.strategy == "end":
            self._save()


class EarlyStop(Callback):
    """polus/callbacks.py:315-363.  Quirk kept: last_loss is never updated from 1000, so the
    patience rule only fires for losses above 1000; the NaN stop is the live part."""

    def __init__(self, patience=3, use_smooth_loss=True):
        super().__init__()
        self.current_patience, self.patience = 0, patience
        self.last_loss, self.use_smooth_loss = 1000, use_smooth_loss

    @runs_if_root
    def on_train_begin(self):
        if self.use_smooth_loss and not self.coordinator.has_callback(LossSmoothCallback):
            logger.warning("LossSmoothCallback was not found on the coordinator; EarlyStop will use the normal loss")
            self.use_smooth_loss = False
        self.loss = []

    @runs_if_root
    def on_train_batch_end(self, epoch, step, loss):
        if not self.use_smooth_loss:
            self.loss.append(float(loss))

    @runs_if_root
    def on_epoch_end(self, epoch):
        if self.use_smooth_loss:
            loss = self.coordinator.shared_dict["smooth_loss"]
        else:
            loss = sum(self.loss) / len(self.loss)
            self.loss = []
        if np.isnan(loss):
            logger.info("The training will stop early since the loss became nan")
            self.coordinator.trainer.early_stop = True
            from .hpo import HPOContext
            ctx = HPOContext()
            if ctx.is_hpo_enable():
                ctx.prune("loss became nan")
        if self.last_loss < loss:
            self.current_patience += 1
        if self.current_patience > self.patience:
            self.coordinator.trainer.early_stop = True
            logger.info(f"The training will stop early since the loss did not improve in {self.patience} consecutive epochs")


class HPOPruneCallback(Callback):
    def __init__(self, validator_name, metric_name):
        super().__init__()
        from .hpo import HPOContext
        self.hpo_backend = HPOContext().hpo_backend
        if self.hpo_backend is None:
            logger.warning("HPOPruneCallback was initialized however, there is no hpo context at the moment")
        self.validator_name, self.metric_name = validator_name, metric_name

    @runs_if_root
    def on_epoch_end(self, epoch):
        if self.hpo_backend is None:
            return
        score = self.coordinator.shared_dict["validation"][self.validator_name][self.metric_name][-1]
        if hasattr(self.hpo_backend, "report"):
            self.hpo_backend.report(score, step=epoch)
            if self.hpo_backend.should_prune():
                from .hpo import HPOContext
                HPOContext().prune(f"Trial was pruned at epoch {epoch} with a score of {score}.")


class Profiler(Callback):
    """polus/callbacks.py:408-470 traced TF ops over a step window.  Here the window
    [lo, hi) of global steps is bracketed with roctx ranges (visible to
    `rocprofv3 --marker-trace`) and per-step device time from HIP events is written to
    <logs_dir>/step_times.csv.  As in the reference the run stops once the window is done.
    Quirk kept: no super().__init__() (add_coordinator sets the attribute)."""

    def __init__(self, write_graph=True, steps_interval=[10, 20], logs_dir="logs/profiler_logs"):
        self.write_graph, self.steps_interval, self.logs_dir = write_graph, steps_interval, logs_dir
        self.trace_started = False
        self.rows = []

    @runs_if_root
    def on_train_begin(self):
        os.makedirs(self.logs_dir, exist_ok=True)

    @runs_if_root
    def on_train_batch_begin(self, epoch, step):
        import torch
        sc = self.coordinator.trainer.step_counter
        if self.steps_interval[0] <= sc < self.steps_interval[1]:
            self.trace_started = True
            self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            torch.cuda.nvtx.range_push(f"step{sc}")
            self._ev[0].record()

    @runs_if_root
    def on_train_batch_end(self, epoch, step, loss):
        import torch
        sc = self.coordinator.trainer.step_counter
        if self.steps_interval[0] <= sc < self.steps_interval[1] and self.trace_started:
            self._ev[1].record()
            torch.cuda.nvtx.range_pop()
            self._ev[1].synchronize()
            self.rows.append((sc, self._ev[0].elapsed_time(self._ev[1])))
        if sc >= self.steps_interval[1] - 1 and self.trace_started:
            self.trace_started = False
            self.coordinator.trainer.early_stop = True

    @runs_if_root
    def on_train_end(self):
        if self.rows:
            with open(os.path.join(self.logs_dir, "step_times.csv"), "w") as f:
                f.write("global_step,device_ms\n")
                for s, ms in self.rows:
                    f.write(f"{s},{ms:.4f}\n")


def _last_leaf(d):
    if isinstance(d, dict):
        out = {}
        for k, e in d.items():
            o = _last_leaf(e)
            if isinstance(o, dict):
                for k2, v in o.items():
                    out[f"{k} {k2}"] = v
            else:
                out[k] = o
        return out
    if isinstance(d, list):
        return _last_leaf(d[-1])
    return d


class WandBLogCallback(Callback, IOutput):
    """polus/callbacks.py:473-558; wandb is optional (absent from this image)."""

    def __init__(self, project, init_args, entity=None, additional_info=None, model_config=None, model_name_prefix=""):
        super().__init__()
        self.project, self.entity, self.init_args = project, entity, init_args
        self.additional_info, self.model_config, self.model_name_prefix = additional_info, model_config, model_name_prefix
        try:
            import wandb
            self.wandb = wandb
        except ImportError:
            self.wandb = None
            logger.warning("wandb is not installed: WandBLogCallback will only buffer values")

    @runs_if_root
    def on_train_begin(self):
        if self.wandb is None:
            return
        trainer = self.coordinator.trainer
        cfg = self.model_config if self.model_config is not None else getattr(trainer.model, "savable_config", {})
        kw = {"project": self.project, "config": cfg}
        if self.entity is not None:
            kw["entity"] = self.entity
        self.wandb.init(**kw)
        self.wandb.config.update(self.init_args)
        if self.additional_info:
            self.wandb.config.update(self.additional_info)
        opt = dict(getattr(trainer.optimizer, "get_config", lambda: {})())
        opt["name"] = trainer.optimizer.__class__.__name__
        self.wandb.config.update({"loss": {"name": getattr(trainer.loss, "__name__", trainer.loss.__class__.__name__)},
                                  "optimizer": opt})
        trainer.model.set_name(self.wandb.run.name)

    @runs_if_root
    def on_train_batch_end(self, epoch, step, loss):
        data = _last_leaf(self.flush())
        data["loss"] = float(loss)
        if self.wandb is not None:
            self.wandb.log(data)

    @runs_if_root
    def on_epoch_end(self, epoch):
        data = _last_leaf(self.flush())
        data["epoch"] = epoch
        if self.wandb is not None:
            self.wandb.log(data)


class ConsoleLogCallback(Callback, IOutput):
    def __init__(self, log_on_train_step=False):
        super().__init__()
        self.log_on_train_step = log_on_train_step
        self.loss_per_epoch = defaultdict(list)

    def _fmt(self, d, sep=" - "):
        if isinstance(d, dict):
            parts = []
            for key, e in d.items():
                o = self._fmt(e, ", ")
                if isinstance(e, dict) or (isinstance(e, list) and e and isinstance(e[0], dict)):
                    o = f"[{o}]"
                parts.append(f"{key}: {o}")
            return sep.join(parts)
        if isinstance(d, list):
            return self._fmt(d[-1], ", ")
        return f"{d:.3f}"

    @runs_if_root
    def on_train_begin(self):
        logger.info(f"Begin training of the model \"{self.coordinator.trainer.model.name}\" for {self.coordinator.epochs} epochs")

    @runs_if_root
    def on_epoch_begin(self, epoch):
        logger.info(f"Begin epoch {epoch}")

    @runs_if_root
    def on_train_batch_end(self, epoch, step, loss):
        self.loss_per_epoch[epoch].append(float(loss))
        msg = f"{step}/{self.coordinator.steps} - loss: {loss:.3f} - " + self._fmt(self.flush())
        if self.log_on_train_step:
            logger.info(msg)
        else:
            print(msg, end="\r")

    @runs_if_root
    def on_epoch_end(self, epoch):
        n = len(self.loss_per_epoch[epoch])
        avg = sum(self.loss_per_epoch[epoch]) / n if n else 0
        logger.info(f"Average loss: {avg:.3f} - " + self._fmt(self.flush()))

    @runs_if_root
    def on_train_end(self):
        logger.info("End of training")
