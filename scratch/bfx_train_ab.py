"""A/B of a real (small) training run in the two arithmetics: experiments/run_experiment.py's synthetic task, same seeds, same
data order, exact fp32 against ops.set_conv_precision("bf16x3").  Per epoch: train loss, validation loss; at the end the test
measures of the reference's test flow (threshold 0.4).  Writes gpurun_out/r04_bf16x3_train_ab.json.
usage: python scratch/bfx_train_ab.py [config ...]   (default: tiny:SAUnet tiny:DRCNN)"""
import json, logging, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "experiments"))
import torch
from multipitch_architectures_amd import experiment, ops
from run_experiment import synthetic_recording

EPOCHS = int(os.environ.get("AB_EPOCHS", "12")); NREC = int(os.environ.get("AB_RECORDINGS", "6")); FRAMES = int(os.environ.get("AB_FRAMES", "1500"))


def run(config, precision):
    ops.set_conv_precision(precision)
    ops.manual_seed(1234)
    torch.manual_seed(0)
    model, criterion, cfg = experiment.build(config)
    train_files = [synthetic_recording(FRAMES, 100 + k) for k in range(NREC)]
    val_files, test_files = [synthetic_recording(FRAMES, 7)], [synthetic_recording(FRAMES // 2, 8)]
    t0 = time.time()
    hist = experiment.train(model, criterion, train_files, val_files, lr=cfg["lr"], max_epochs=EPOCHS, log=lambda *_: None)
    mean, _ = experiment.test(model, test_files, ["synthetic-test"], log=lambda *_: None)
    ops.set_conv_precision("f32")
    return {"precision": precision, "epochs": len(hist), "train_loss": [h[0] for h in hist], "val_loss": [h[1] for h in hist],
            "test": {k: float(v) for k, v in mean.items()}, "seconds": time.time() - t0}


if __name__ == "__main__":
    logging.basicConfig(level=logging.WARNING)
    out = {"task": f"experiments/run_experiment.py --synthetic {NREC} --frames {FRAMES} --epochs {EPOCHS}", "runs": {}}
    for config in (sys.argv[1:] or ["tiny:SAUnet", "tiny:DRCNN"]):
        a, b = run(config, "f32"), run(config, "bf16x3")
        out["runs"][config] = {"f32": a, "bf16x3": b,
                               "final_val_loss_diff": abs(a["val_loss"][-1] - b["val_loss"][-1]),
                               "f_measure_diff_pp": 100.0 * abs(a["test"]["f_measure"] - b["test"]["f_measure"])}
        print(config, "f32 val", [round(v, 4) for v in a["val_loss"]], "F", round(a["test"]["f_measure"], 4))
        print(config, "bfx val", [round(v, 4) for v in b["val_loss"]], "F", round(b["test"]["f_measure"], 4), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_bf16x3_train_ab.json"), "w"), indent=1)
