"""TEST INFRASTRUCTURE ONLY — never imported by the product path (`rectools_amd/`).

Import shims that let the *unmodified* reference (`/root/reference`, RecTools v0.17.0) be imported in the
build container, where `typeguard`, `implicit` and `pytorch_lightning` are absent and cannot be installed.
None of the three is *called* on the transformer fit/recommend path except the Lightning `Trainer`, whose
loop is restated below (order of RNG-consuming events matters).

Used only by `tests/golden/make_golden.py` (fixture generation, this container only) and by
`oracle/cpu_reference.py` when `/root/reference` is present.  `/root/reference` does not exist on the GPU
box, so nothing that runs there may depend on this module succeeding.
"""
from __future__ import annotations

import os
import random
import sys
import types
import typing as tp

def _find_reference_root() -> str:
    """`/root/reference` in the build container; on the GPU box the staged, git-ignored copy `oracle/_ref` (oracle/make_ref.py)."""
    env = os.environ.get("RECTOOLS_REFERENCE_ROOT")
    if env:
        return env
    if os.path.isdir("/root/reference/rectools"):
        return "/root/reference"
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


REFERENCE_ROOT = _find_reference_root()


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "rectools"))


def _mod(name: str) -> types.ModuleType:
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = []  # type: ignore[attr-defined]  # behave as a package
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(_mod(parent), child, m)
    return m


def _install_typeguard() -> None:
    if "typeguard" in sys.modules:
        return
    m = _mod("typeguard")

    class TypeCheckError(Exception):
        pass

    def check_type(value: tp.Any, expected_type: tp.Any, *a: tp.Any, **k: tp.Any) -> tp.Any:
        return value

    m.TypeCheckError = TypeCheckError
    m.check_type = check_type


def _install_implicit() -> None:
    if "implicit" in sys.modules:
        return
    for name in (
        "implicit", "implicit.cpu", "implicit.gpu", "implicit.als", "implicit.bpr", "implicit.utils",
        "implicit.nearest_neighbours", "implicit.cpu.als", "implicit.gpu.als", "implicit.cpu.bpr",
        "implicit.gpu.bpr", "implicit.cpu.topk", "implicit.cpu.matrix_factorization_base",
    ):
        _mod(name)
    sys.modules["implicit.gpu"].HAS_CUDA = False

    def _dummy(name: str) -> type:
        return type(name, (), {"__init__": lambda self, *a, **k: None})

    for modname in ("implicit.als", "implicit.cpu.als", "implicit.gpu.als"):
        sys.modules[modname].AlternatingLeastSquares = _dummy("AlternatingLeastSquares")
    for modname in ("implicit.bpr", "implicit.cpu.bpr", "implicit.gpu.bpr"):
        sys.modules[modname].BayesianPersonalizedRanking = _dummy("BayesianPersonalizedRanking")
    nn_mod = sys.modules["implicit.nearest_neighbours"]
    item_item = _dummy("ItemItemRecommender")
    nn_mod.ItemItemRecommender = item_item
    for n in ("BM25Recommender", "CosineRecommender", "TFIDFRecommender"):
        setattr(nn_mod, n, type(n, (item_item,), {}))
    utils = sys.modules["implicit.utils"]
    utils.ParameterWarning = type("ParameterWarning", (Warning,), {})
    utils.check_random_state = lambda rs: rs
    sys.modules["implicit.cpu.matrix_factorization_base"]._filter_items_from_sparse_matrix = (
        lambda *a, **k: (_ for _ in ()).throw(RuntimeError("implicit is shimmed"))
    )
    sys.modules["implicit.cpu.topk"].topk = (
        lambda *a, **k: (_ for _ in ()).throw(RuntimeError("implicit is shimmed"))
    )


def _install_lightning() -> None:
    if "pytorch_lightning" in sys.modules:
        return
    import torch

    pl = _mod("pytorch_lightning")
    loggers = _mod("pytorch_lightning.loggers")
    callbacks = _mod("pytorch_lightning.callbacks")

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a: tp.Any, **k: tp.Any) -> None:
            return None

        def log(self, name: str, value: tp.Any, *a: tp.Any, **k: tp.Any) -> None:
            logged = self.__dict__.setdefault("_shim_logged", {})
            try:
                logged.setdefault(name, []).append(float(value))
            except Exception:  # pragma: no cover
                pass

    class Trainer:
        """Minimal stand-in for `pytorch_lightning.Trainer.fit`.

        The epoch-0 dataloader iterator is created BEFORE `on_train_start` (xavier init), which is what
        reproduces Lightning's RNG order and hence the reference's `expected_cpu_1` golden frames.
        """

        def __init__(self, max_epochs: tp.Optional[int] = None, min_epochs: tp.Optional[int] = None,
                     **kwargs: tp.Any) -> None:
            self.max_epochs = max_epochs
            self.min_epochs = min_epochs
            self.kwargs = kwargs
            self.lightning_module: tp.Optional[LightningModule] = None
            self.fit_loop = types.SimpleNamespace(
                max_epochs=max_epochs, min_epochs=min_epochs,
                epoch_progress=types.SimpleNamespace(current=types.SimpleNamespace(ready=0)),
            )

        def fit(self, model: LightningModule, train_dataloaders: tp.Any = None, val_dataloaders: tp.Any = None,
                ckpt_path: tp.Any = None) -> None:
            self.lightning_module = model
            # accelerator="gpu" (explicitly: the default stays on the host, where the golden fixtures are generated): the module and
            # every batch move to the device, as Lightning's accelerator connector / transfer_batch_to_device do
            on_gpu = str(self.kwargs.get("accelerator", "cpu")) in ("gpu", "cuda") and torch.cuda.is_available()
            if on_gpu:
                model.to("cuda")
            model.train()
            opt = model.configure_optimizers()
            if ckpt_path is not None:
                # Lightning restores a checkpoint BEFORE the fit loop starts (`_checkpoint_connector.restore_*` in `Trainer._run`):
                # module weights (strict), optimizer states, the loops' progress — what `load_from_checkpoint`
                # (transformers/base.py:591-654) relies on when it calls `fit(ckpt_path=...)` with an EMPTY stub dataloader; with no
                # training batches the fit loop is skipped (`_FitLoop.skip`), so `on_train_start` (xavier) never runs
                ck = torch.load(ckpt_path, map_location="cpu", weights_only=False)
                model.load_state_dict(ck["state_dict"], strict=True)
                if ck.get("optimizer_states"):
                    opt.load_state_dict(ck["optimizer_states"][0])
                self.fit_loop.epoch_progress.current.ready = int(ck.get("epoch", 0))
                self.restored = {"epoch": ck.get("epoch"), "global_step": ck.get("global_step")}
                try:
                    if len(train_dataloaders) == 0:
                        return
                except TypeError:
                    pass
            n_epochs = self.fit_loop.max_epochs if self.fit_loop.max_epochs is not None else 1
            it = iter(train_dataloaders)
            model.on_train_start()
            start = self.fit_loop.epoch_progress.current.ready
            for epoch in range(start, n_epochs):
                if epoch > start:
                    it = iter(train_dataloaders)
                for i, batch in enumerate(it):
                    if on_gpu:
                        batch = {k: v.to("cuda") for k, v in batch.items()}
                    opt.zero_grad()
                    loss = model.training_step(batch, i)
                    loss.backward()
                    opt.step()
                self.fit_loop.epoch_progress.current.ready = epoch + 1
            model.on_train_end()

        def save_checkpoint(self, *a: tp.Any, **k: tp.Any) -> None:  # pragma: no cover
            raise RuntimeError("pytorch_lightning is shimmed: checkpoints unavailable")

    def seed_everything(seed: int, workers: bool = False) -> int:
        import numpy as np

        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        return seed

    pl.LightningModule = LightningModule
    pl.Trainer = Trainer
    pl.Callback = object
    pl.seed_everything = seed_everything
    loggers.Logger = object
    loggers.CSVLogger = object
    callbacks.Callback = object
    callbacks.ModelCheckpoint = object
    callbacks.EarlyStopping = object


def install(add_reference_to_path: bool = True) -> None:
    """Install the three shims and (optionally) put the read-only reference on `sys.path`."""
    sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference
    _install_typeguard()
    _install_implicit()
    _install_lightning()
    if add_reference_to_path:
        if not reference_available():
            raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
        if REFERENCE_ROOT not in sys.path:
            sys.path.insert(0, REFERENCE_ROOT)


def seed_all(seed: int = 32) -> None:
    """Stand-in for `seed_everything(seed, workers=True)` + deterministic algorithms (reference tests)."""
    import numpy as np
    import torch

    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.use_deterministic_algorithms(True)
