"""Checkpoint import / export for the MI355X path.

Host-side mirror of ``MyCheckpointer`` (core/utils/my_checkpoint.py:9-54) and of the way the reference trainer drives it
(core/gdrn_modeling/engine.py:190-210,328-333; main_gdrn.py:121): same constructor, ``save`` / ``load`` /
``resume_or_load`` / ``has_checkpoint`` / ``get_checkpoint_file`` and the same file format -- ``{"model": state_dict,
"optimizer": ..., "scheduler": ..., "iteration": ...}`` in ``<save_dir>/<name>.pth`` plus the ``last_checkpoint`` tag file -- so
files written by the reference (released GDR-Net weights included) load here and files written here load in the reference.
``MyCheckpointer`` derives from detectron2's ``DetectionCheckpointer`` / fvcore's ``Checkpointer`` (third-party, not in the
reference tree); the behaviour restated from their published definition: bare state dicts are wrapped as ``{"model": ...}``, a
``module.`` prefix common to all keys is stripped, tensors whose shape differs from the model's are skipped and reported, the
rest is loaded non-strictly and the incompatible keys are returned.

Nothing here touches kernels: the parameters are ordinary ``nn.Parameter``s holding the reference's shapes in fp32, the HIP
engine re-packs its operands from them at the next step (``engine.repack``) and refreshes its folded eval-mode BatchNorm
(``engine.eval_refresh``) when they change.  ``torchvision://`` / ``http(s)://`` sources need a network and the Caffe2 /
Detectron ``.pkl`` zoo formats need detectron2's name-matching heuristics; both raise instead of guessing.
"""
import logging
import os
import pickle

import numpy as np
import torch

logger = logging.getLogger(__name__)


class IncompatibleKeys:
    def __init__(self, missing_keys, unexpected_keys, incorrect_shapes):
        self.missing_keys, self.unexpected_keys, self.incorrect_shapes = missing_keys, unexpected_keys, incorrect_shapes

    def __repr__(self):
        return f"IncompatibleKeys(missing={self.missing_keys}, unexpected={self.unexpected_keys}, incorrect_shapes={self.incorrect_shapes})"


class MyCheckpointer:
    def __init__(self, model, save_dir="", *, save_to_disk=None, **checkpointables):
        while hasattr(model, "module") and isinstance(getattr(model, "module"), torch.nn.Module):  # DDP / DataParallel / Lite wrappers
            model = model.module
        self.model = model
        self.checkpointables = dict(checkpointables)
        self.save_dir = save_dir
        if save_to_disk is None:  # detectron2: only the main process writes
            save_to_disk = (not torch.distributed.is_available()) or (not torch.distributed.is_initialized()) or torch.distributed.get_rank() == 0
        self.save_to_disk = bool(save_to_disk)

    # ---------------------------------------------------------------- export
    def save(self, name, **kwargs):
        if not self.save_dir or not self.save_to_disk:
            return None
        data = {"model": self.model.state_dict()}
        for key, obj in self.checkpointables.items():
            data[key] = obj.state_dict()
        data.update(kwargs)
        basename = f"{name}.pth"
        save_file = os.path.join(self.save_dir, basename)
        os.makedirs(self.save_dir, exist_ok=True)
        logger.info("Saving checkpoint to %s", save_file)
        tmp = save_file + ".tmp"
        torch.save(data, tmp)
        os.replace(tmp, save_file)
        self.tag_last_checkpoint(basename)
        return save_file

    def tag_last_checkpoint(self, last_filename_basename):
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(last_filename_basename)

    # ---------------------------------------------------------------- import
    def has_checkpoint(self):
        return bool(self.save_dir) and os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def get_checkpoint_file(self):
        try:
            with open(os.path.join(self.save_dir, "last_checkpoint")) as f:
                last_saved = f.read().strip()
        except OSError:
            return ""
        return os.path.join(self.save_dir, last_saved)

    def get_all_checkpoint_files(self):
        if not self.save_dir or not os.path.isdir(self.save_dir):
            return []
        return [os.path.join(self.save_dir, f) for f in os.listdir(self.save_dir)
                if os.path.isfile(os.path.join(self.save_dir, f)) and f.endswith(".pth")]

    def resume_or_load(self, path, *, resume=True):
        if resume and self.has_checkpoint():
            return self.load(self.get_checkpoint_file())
        return self.load(path, checkpointables=[])

    def load(self, path, checkpointables=None):
        if not path:
            logger.info("No checkpoint found. Initializing model from scratch")
            return {}
        if not os.path.isfile(path) and not path.startswith(("torchvision://", "http://", "https://")):
            raise FileNotFoundError(f"Checkpoint {path} not found!")
        checkpoint = self._load_file(path)
        incompatible = self._load_model(checkpoint)
        if incompatible.missing_keys or incompatible.unexpected_keys or incompatible.incorrect_shapes:
            logger.warning("%s: %r", path, incompatible)
        for key in self.checkpointables if checkpointables is None else checkpointables:
            if key in checkpoint:
                self.checkpointables[key].load_state_dict(checkpoint.pop(key))
        checkpoint["__incompatible__"] = incompatible
        return checkpoint

    def _load_file(self, filename):
        if filename.startswith(("torchvision://", "http://", "https://")):
            raise NotImplementedError(f"{filename}: remote weights need a network; download the file and pass its path")
        if filename.endswith(".pkl"):
            with open(filename, "rb") as f:
                data = pickle.load(f, encoding="latin1")
            if "model" in data and "__author__" in data and not data.get("matching_heuristics", False):
                return data
            raise NotImplementedError(f"{filename}: Caffe2 / Detectron zoo pickles need detectron2's name-matching heuristics")
        loaded = torch.load(filename, map_location="cpu", weights_only=False)
        if "model" not in loaded:
            loaded = {"model": loaded}
        return loaded

    def _load_model(self, checkpoint):
        state = checkpoint.pop("model")
        state = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in state.items()}
        if state and all(k.startswith("module.") for k in state):
            state = {k[len("module."):]: v for k, v in state.items()}
        own = self.model.state_dict()
        incorrect = []
        for k in list(state):
            if k in own and isinstance(state[k], torch.Tensor) and tuple(own[k].shape) != tuple(state[k].shape):
                if own[k].dim() == 0 and tuple(state[k].shape) == (1,):
                    continue  # nn.Module.load_state_dict accepts the old one-element form of scalar buffers
                incorrect.append((k, tuple(state[k].shape), tuple(own[k].shape)))
                state.pop(k)
        res = self.model.load_state_dict(state, strict=False)
        missing = [k for k in res.missing_keys if k not in {n for n, _, _ in incorrect}]
        return IncompatibleKeys(missing, list(res.unexpected_keys), incorrect)


class PeriodicCheckpointer:
    """fvcore ``PeriodicCheckpointer`` as engine.py:209-211,333 uses it: a checkpoint every ``period`` iterations named
    ``model_{iteration:07d}``, at most ``max_to_keep`` recent ones kept, and ``model_final`` at ``max_iter - 1``."""

    def __init__(self, checkpointer, period, max_iter=None, max_to_keep=None, file_prefix="model"):
        self.checkpointer, self.period, self.max_iter = checkpointer, int(period), max_iter
        if max_to_keep is not None:
            assert max_to_keep > 0
        self.max_to_keep, self.file_prefix = max_to_keep, file_prefix
        self.recent_checkpoints = []

    def step(self, iteration, **kwargs):
        iteration = int(iteration)
        additional_state = {"iteration": iteration}
        additional_state.update(kwargs)
        if (iteration + 1) % self.period == 0:
            self.checkpointer.save(f"{self.file_prefix}_{iteration:07d}", **additional_state)
            if self.max_to_keep is not None and self.checkpointer.save_to_disk and self.checkpointer.save_dir:
                self.recent_checkpoints.append(self.checkpointer.get_checkpoint_file())
                if len(self.recent_checkpoints) > self.max_to_keep:
                    old = self.recent_checkpoints.pop(0)
                    if os.path.exists(old) and not old.endswith(f"{self.file_prefix}_final.pth"):
                        os.remove(old)
        if self.max_iter is not None and iteration >= self.max_iter - 1:
            self.checkpointer.save(f"{self.file_prefix}_final", **additional_state)

    def save(self, name, **kwargs):
        self.checkpointer.save(name, **kwargs)
