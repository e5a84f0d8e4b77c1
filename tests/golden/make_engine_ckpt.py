"""Checkpoints WRITTEN BY THE ENGINE, for the direction SURVEY.md §8f-4 calls "vice-versa": the reference's own
`load_from_checkpoint` (transformers/base.py:591-654) must read what `rectools_amd` saves.

Runs on the GPU box (the engine has no CPU path):  gpurun -- 'python tests/golden/make_engine_ckpt.py gpurun_out/engine_ckpt'
then copy gpurun_out/engine_ckpt/engine_ckpt_*.ckpt into tests/golden/.  For every model kind: the reference-trained fixture
`ckpt_<name>.ckpt` is loaded into the engine, trained one more epoch BY THE ENGINE (so weights, Adam moments and counters are its own),
saved with `model.save_to_checkpoint`, and the engine's recommend() / recommend_to_items() frames for the saved weights are stored
beside the checkpoint's own keys under "expected_engine" (the reference ignores unknown keys).  tests/test_reference_live.py
loads the files with the unmodified reference (build container only) and compares weights, optimizer state and frames."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from test_checkpoint import CKPTS, _context, _dataset, _model_class   # noqa: E402  (the fixtures' datasets / classes)

GOLDEN = os.path.dirname(os.path.abspath(__file__))


def main(out_dir: str) -> None:
    os.makedirs(out_dir, exist_ok=True)
    users = [10, 30, 40]
    for name in CKPTS:
        klass = _model_class(name)
        model = klass.load_from_checkpoint(os.path.join(GOLDEN, f"ckpt_{name}.ckpt"))
        ds = _dataset(name)
        model.fit_partial(ds, max_epochs=1)
        path = os.path.join(out_dir, f"engine_ckpt_{name}.ckpt")
        model.save_to_checkpoint(path)
        ctx = _context(model)
        expected = {}
        for tag, rk in (("filter", dict(k=3, filter_viewed=True)), ("nofilter", dict(k=4, filter_viewed=False)),
                        ("whitelist", dict(k=2, filter_viewed=False, items_to_recommend=[11, 13, 17]))):
            r = model.recommend(users=users, dataset=ds, context=ctx, **rk)
            expected[tag] = {c: r[c].tolist() for c in r.columns}
        i2i = model.recommend_to_items(target_items=[11, 12], dataset=ds, k=2)
        expected["i2i"] = {c: i2i[c].tolist() for c in i2i.columns}
        ck = torch.load(path, map_location="cpu", weights_only=False)
        ck["expected_engine"] = expected
        torch.save(ck, path)
        print(f"{name}: epoch {ck['epoch']} global_step {ck['global_step']} tensors {len(ck['state_dict'])} "
              f"reco {expected['filter']['item_id']}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/engine_ckpt")
